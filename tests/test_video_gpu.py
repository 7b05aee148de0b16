"""GPU: df_frames_to_tensor (PIL-exact antialiased resize + ToTensor on device) and the ExtractCAVPFeatures front end
(mirror of Extract_CAVP_Features, inference/demo_util.py:80-173) against Pillow's golden outputs and the oracle."""
import hashlib

import numpy as np
import pytest
import torch

from helpers import gold
from test_video_cpu import frames

pytestmark = pytest.mark.gpu


def test_frames_to_tensor_bit_exact_vs_pillow_golden():
    import diff_foley_amd as P
    g = gold("g9_video_frames.npz")
    t = P.frames_to_tensor(frames(900, 3, 90, 160), (64, 64))
    assert t.shape == (3, 3, 64, 64) and t.dtype == torch.float32 and t.is_cuda
    ref = torch.from_numpy(g["small_90x160_to_64x64"].numpy()).permute(0, 3, 1, 2).float() / 255.0
    assert torch.equal(t.cpu(), ref)
    for tag in ("d360", "d1080", "up", "same", "tall"):
        seed, T, H, W, oh, ow = (int(v) for v in g[f"spec_{tag}"])
        t = P.frames_to_tensor(frames(seed, T, H, W), (oh, ow))
        u8 = (t.cpu() * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()
        assert torch.equal(t.cpu(), torch.from_numpy(u8).permute(0, 3, 1, 2).float() / 255.0)   # exactly k/255 values
        assert hashlib.sha256(u8.tobytes()).digest() == bytes(g[f"sha_{tag}"].numpy().tolist()), tag


def test_extract_cavp_features_front_end_batches_like_the_reference():
    """45 frames -> a batch of 40 and a batch of 5 (demo_util.py:152-166); features equal the encoder run on the
    oracle-preprocessed frames batch by batch (the pre-processing is bit-exact, so equality is exact)."""
    import diff_foley_amd as P
    from diff_foley_amd import synth
    from oracle import video as ov
    enc = P.CAVPInference(embed_dim=synth.CAVP_TINY["embed_dim"], stage_blocks=synth.CAVP_TINY["stage_blocks"])
    enc.load_state_dict(synth.make_state_dict(synth.cavp_spec(synth.CAVP_TINY)))
    enc.cuda()
    ex = P.ExtractCAVPFeatures(fps=4, batch_size=40, video_shape=(64, 64), stage1_model=enc)
    f = frames(960, 45, 48, 80)
    feats = ex.forward_frames(f)
    assert feats.shape == (45, synth.CAVP_TINY["embed_dim"]) and np.isfinite(feats).all()
    assert np.allclose(np.linalg.norm(feats, axis=-1), 1.0, atol=1e-4)            # normalize=True
    ref = []
    for lo, hi in ov.batches(45, 40):
        x = torch.from_numpy(ov.frames_to_tensor(f[lo:hi], (64, 64))).cuda()
        ref.extend(enc.encode_video(x.unsqueeze(0), normalize=True, pool=False).cpu().numpy())
    assert np.array_equal(feats, np.concatenate(ref))
    with pytest.raises(RuntimeError):
        P.frames_to_tensor(np.zeros((2, 8, 8, 4), np.uint8))
