"""CPU: the frame pre-processing oracle (oracle/video.py, a restatement of Pillow's 8-bit BILINEAR resampling that
torchvision's Resize uses in demo_util.py:100-104) against Pillow itself, against the golden fixture made from Pillow's
outputs, and the product's coefficient tables (diff_foley_amd/video.py) against the oracle's."""
import hashlib

import numpy as np
import pytest

import diff_foley_amd  # noqa: F401
from diff_foley_amd import video as V
from helpers import gold
from oracle import video as ov


def frames(seed, T, H, W):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    f[0] = np.stack([(xx * 255 // max(W - 1, 1)), (yy * 255 // max(H - 1, 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
    return f


def test_oracle_matches_pillow_bit_exactly():
    Image = pytest.importorskip("PIL.Image")
    for seed, (H, W, oh, ow) in enumerate([(90, 160, 64, 64), (100, 120, 224, 224), (224, 224, 224, 224), (37, 501, 224, 224)]):
        f = frames(950 + seed, 2, H, W)
        for x in f:
            ref = np.asarray(Image.fromarray(x).resize((ow, oh), Image.BILINEAR))
            assert np.array_equal(ov.resize_bilinear_u8(x, oh, ow), ref)


def test_oracle_pass_order_matches_pillow_on_tall_frames():
    """Pillow runs the vertical pass first on frames more than 100 times taller than wide whose height shrinks (found by the round-6 seed sweep of
    tests/test_video_fuzz_gpu.py: 224 x 2 -> 64 x 111 and 1080 x 7 -> 5 x 256 differed from PIL by one count in 10 % / 3 % of the
    values with the horizontal pass first).  The boundary H = 100 W vs 100 W + 1 for several widths and output sizes, and random tall
    and wide geometries, against PIL itself; and the two orders really differ there (the test can tell them apart)."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(77)
    geo = [(100 * W + d, W, oh, ow) for W in (2, 3, 5) for d in (0, 1) for oh, ow in ((5, 16), (64, 111))]
    geo += [(224, 2, 64, 111), (1080, 7, 5, 256), (641, 2, 64, 64), (64, 2, 32, 64), (1080, 3, 5, 64), (2, 224, 111, 64), (7, 1080, 256, 5)]
    # ... and only while the height shrinks (a later sweep: 224 x 2 -> 256 x 5 runs the horizontal pass first): out_h = H - 1 / H + 1 / 2 H
    geo += [(224, 2, 256, 5), (224, 2, 223, 5), (224, 2, 225, 5), (501, 5, 500, 8), (501, 5, 502, 8), (301, 3, 602, 40), (801, 2, 1600, 1)]
    for _ in range(24):
        W = int(rng.integers(1, 9))
        geo.append((int(rng.integers(2, 1300)), W, int(rng.integers(1, 200)), int(rng.integers(1, 200))))
    differ = 0
    for H, W, oh, ow in geo:
        x = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(x).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(ov.resize_bilinear_u8(x, oh, ow), ref), (H, W, oh, ow)
        if H > 100 * W and oh < H and W != ow:
            bw, kw = ov.resample_coeffs(W, ow)
            bh, kh = ov.resample_coeffs(H, oh)
            differ += int(not np.array_equal(ov._pass(ov._pass(x, bw, kw, 1), bh, kh, 0), ref))
    assert differ >= 5


def test_oracle_matches_golden_fixture():
    g = gold("g9_video_frames.npz")
    f = frames(900, 3, 90, 160)
    got = np.stack([ov.resize_bilinear_u8(x, 64, 64) for x in f])
    assert np.array_equal(got, g["small_90x160_to_64x64"].numpy())
    for tag in ("up", "same"):                          # the large down-scales are checked on the GPU (oracle loops are slow)
        seed, T, H, W, oh, ow = (int(v) for v in g[f"spec_{tag}"])
        r = np.stack([ov.resize_bilinear_u8(x, oh, ow) for x in frames(seed, T, H, W)])
        assert hashlib.sha256(r.tobytes()).digest() == bytes(g[f"sha_{tag}"].numpy().tolist())
    t = ov.frames_to_tensor(f, (64, 64))
    assert t.shape == (3, 3, 64, 64) and t.dtype == np.float32 and t.max() <= 1.0
    assert np.array_equal(t[1, 2], got[1, :, :, 2].astype(np.float32) / np.float32(255.0))


@pytest.mark.parametrize("a,b", [(640, 224), (360, 224), (1920, 224), (1080, 224), (224, 224), (120, 224), (17, 224), (100, 64)])
def test_product_coefficient_tables_equal_the_oracles(a, b):
    b1, k1 = V.resample_coeffs(a, b)
    b2, k2 = ov.resample_coeffs(a, b)
    assert np.array_equal(b1, b2) and np.array_equal(k1, k2)
    assert np.all(np.abs(k1.sum(1) - (1 << 22)) <= k1.shape[1])        # rows are normalised to 1.0 in 22-bit fixed point


def test_batches_follow_the_reference_loop():
    assert ov.batches(45, 40) == [(0, 40), (40, 45)]
    assert ov.batches(80, 40) == [(0, 40), (40, 80)]
    assert ov.batches(33, 40) == [(0, 33)]
