"""Randomised geometries through the frame front end (df_frames_to_tensor: Pillow's 8-bit antialiased BILINEAR resize + ToTensor,
inference/demo_util.py:100-103) against the oracle's restatement (oracle/video.py, itself bit-exact against Pillow -- tests/
test_video_cpu.py) AND against Pillow itself on the box: 1 x 1 sources, prime sizes, extreme aspect ratios, up- and down-scaling
by factors up to 30, non-square targets, 1 .. 7 frames.  Integer / byte work: the bar is equality."""
import numpy as np
import pytest
import torch

from helpers import fuzz_seeds

pytestmark = pytest.mark.gpu


def _draw(seed):
    r = np.random.default_rng(8100 + seed)
    pick = lambda: int(r.choice([1, 2, 3, 7, 13, 31, 64, 97, 224, 360, 641, 1080]))
    H, W = pick(), pick()
    while H * W > 720 * 1280:
        H, W = pick(), pick()
    oh, ow = (int(r.choice([1, 5, 32, 64, 111, 224, 256])) for _ in range(2))
    return int(r.integers(1, 8)), H, W, oh, ow


@pytest.mark.parametrize("seed", fuzz_seeds(24))
def test_frames_to_tensor_equals_oracle_and_pillow(seed):
    import diff_foley_amd as P
    from oracle import video as ov
    T, H, W, oh, ow = _draw(seed)
    f = np.random.default_rng(seed).integers(0, 256, (T, H, W, 3), dtype=np.uint8)
    if seed % 3 == 0:                       # smooth content as well as noise (rounding ties differ)
        yy, xx = np.mgrid[0:H, 0:W]
        f = ((np.sin(xx[None, :, :, None] * 0.07 + np.arange(3)[None, None, None, :]) * 0.5 + 0.5) * 255).astype(np.uint8) \
            + np.zeros((T, H, W, 3), np.uint8) + (yy[None, :, :, None] % 7).astype(np.uint8)
    t = P.frames_to_tensor(f, (oh, ow)).cpu()
    assert t.shape == (T, 3, oh, ow) and t.dtype == torch.float32
    ref = torch.from_numpy(ov.frames_to_tensor(f, (oh, ow)))
    assert torch.equal(t, ref), (T, H, W, oh, ow, float((t - ref).abs().max()))
    try:
        from PIL import Image
    except ImportError:
        return
    bil = getattr(getattr(Image, "Resampling", Image), "BILINEAR")
    pil = np.stack([np.asarray(Image.fromarray(fr).resize((ow, oh), bil)) for fr in f])
    assert torch.equal(t, torch.from_numpy(pil).permute(0, 3, 1, 2).float() / 255.0), (T, H, W, oh, ow)


@pytest.mark.parametrize("H,W,oh,ow", [(224, 2, 64, 111), (1080, 7, 5, 256), (641, 2, 64, 64), (200, 2, 64, 16), (201, 2, 64, 16),
                                        (501, 5, 5, 16), (500, 5, 5, 16), (1081, 1, 9, 9), (301, 3, 301, 8), (301, 3, 17, 3),
                                        (224, 2, 256, 5), (224, 2, 223, 5), (224, 2, 225, 5), (801, 2, 1600, 3), (501, 5, 500, 8)])
def test_tall_frames_take_pillows_pass_order(H, W, oh, ow):
    """Frames more than 100 times taller than wide: Pillow runs the vertical pass first and the product follows (resize_v_kernel /
    resize_h_totensor_kernel; the rule and how it was found: tests/test_video_cpu.py).  Both sides of the boundary, one-pass cases."""
    import diff_foley_amd as P
    from oracle import video as ov
    f = np.random.default_rng(H * 7 + W).integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    t = P.frames_to_tensor(f, (oh, ow)).cpu()
    assert torch.equal(t, torch.from_numpy(ov.frames_to_tensor(f, (oh, ow)))), (H, W, oh, ow)
    try:
        from PIL import Image
    except ImportError:
        return
    bil = getattr(getattr(Image, "Resampling", Image), "BILINEAR")
    pil = np.stack([np.asarray(Image.fromarray(fr).resize((ow, oh), bil)) for fr in f])
    assert torch.equal(t, torch.from_numpy(pil).permute(0, 3, 1, 2).float() / 255.0), (H, W, oh, ow)


def test_frames_to_tensor_rejects_what_it_cannot_read():
    import diff_foley_amd as P
    for bad in (np.zeros((2, 8, 8, 4), np.uint8), np.zeros((8, 8, 3), np.uint8), np.zeros((2, 8, 8, 3), np.float32)):
        with pytest.raises((RuntimeError, ValueError, TypeError)):
            P.frames_to_tensor(bad)
    with pytest.raises((RuntimeError, ValueError)):
        P.frames_to_tensor(np.zeros((2, 8, 8, 3), np.uint8), (0, 16))
    assert P.frames_to_tensor(np.zeros((0, 8, 8, 3), np.uint8), (16, 16)).shape == (0, 3, 16, 16)
