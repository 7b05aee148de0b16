"""Probe: df_frames_to_tensor on 4K / 8K frames, the identity size, a x10 up-scale, extreme aspect ratios in both directions and a
64-frame batch, against PIL on the box (bit equality).  usage: python tools/big_frames_probe.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import diff_foley_amd as P
from PIL import Image
for (T, H, W, oh, ow) in [(2, 2160, 3840, 224, 224), (1, 4320, 7680, 224, 224), (3, 1080, 1920, 1080, 1920), (1, 224, 224, 2160, 3840), (1, 3000, 20, 100, 700), (1, 20, 3000, 700, 100), (64, 360, 640, 224, 224)]:
    f = np.random.default_rng(T * H + W).integers(0, 256, (T, H, W, 3), dtype=np.uint8)
    t0 = time.time(); t = P.frames_to_tensor(f, (oh, ow)); torch.cuda.synchronize(); dt = time.time() - t0
    pil = np.stack([np.asarray(Image.fromarray(fr).resize((ow, oh), Image.BILINEAR)) for fr in f[:2]])
    ok = torch.equal(t[:2].cpu(), torch.from_numpy(pil).permute(0, 3, 1, 2).float() / 255.0)
    print(f"{T} x {H} x {W} -> {oh} x {ow}: equal to PIL: {ok}  ({dt * 1e3:.1f} ms incl. upload)", flush=True)
