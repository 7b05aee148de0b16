"""TEST INFRASTRUCTURE ONLY (never imported by diff_foley_amd/): CPU restatement of the frame pre-processing in front of
the CAVP encoder -- ``Extract_CAVP_Features.forward`` (inference/demo_util.py:124-173): every decoded RGB frame goes
through ``transforms.Compose([Resize((224, 224)), ToTensor()])`` (demo_util.py:100-104) on a PIL image, i.e.
``Image.resize((224, 224), BILINEAR)`` followed by uint8 HWC -> float CHW / 255, and frames are handed to
``encode_video`` in batches of ``batch_size`` (40 in the notebook, ipynb cell 3).

The resize itself lives in a third-party dependency, Pillow (the reference pins no version; torchvision calls
``PIL.Image.resize``).  Its published algorithm (src/libImaging/Resample.c, 8 bits per channel path) is restated here:
separable convolution with a triangle filter whose support scales with the down-scaling factor (antialiasing),
horizontal pass first into a uint8 intermediate, then the vertical pass (vertical first on shrinking frames with H > 100 W); coefficients are normalised in double precision
and rounded to 22-bit fixed point; every output value is (2^21 + sum pixel * coeff) >> 22 clipped to 0..255.
Pinned: bit-exact against Pillow 12.2.0 itself in this container (tests/test_oracle_golden.py) and through the golden
fixture tests/golden/g9_video_frames.npz (made by tests/golden/make_golden.py --video from PIL's output)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1.0) over the full axis.
    Returns (bounds [out][2] = (first input index, tap count), coeffs int32 [out][ksize])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, np.float64)
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = w[:xmax].sum()
        if ww != 0.0:
            w[:xmax] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One separable pass over `axis` of a uint8 array [H][W][C]."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + img.shape[1:], np.uint8)
    for o in range(bounds.shape[0]):
        x0, n = bounds[o]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += img[x0 + x] * int(kk[o, x])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(frame, out_h, out_w):
    """uint8 [H][W][C] -> uint8 [out_h][out_w][C], identical to PIL.Image.resize((out_w, out_h), BILINEAR)."""
    H, W = frame.shape[:2]
    bw, kw = resample_coeffs(W, out_w)
    bh, kh = resample_coeffs(H, out_h)
    if H > 100 * W and out_h < H:
        # frames more than 100 times taller than wide whose height shrinks: Pillow runs the VERTICAL pass first.  Observed, not read:
        # Pillow 12.2.0 in this container switches at exactly H = 100 W + 1 and at out_h = H - 1, whatever the output width
        # (tests/test_video_cpu.py walks both boundaries and random geometries against PIL); the uint8 intermediate makes the orders
        # differ by one count in up to 10 % of the values.
        tmp = _pass(frame, bh, kh, 0) if H != out_h else frame
        return _pass(tmp, bw, kw, 1) if W != out_w else tmp
    tmp = _pass(frame, bw, kw, 1) if W != out_w else frame          # horizontal first (ImagingResampleInner)
    return _pass(tmp, bh, kh, 0) if H != out_h else tmp


def frames_to_tensor(frames, size=(224, 224)):
    """[T][H][W][3] uint8 RGB -> float32 [T][3][h][w] in [0, 1]: Resize(size) + ToTensor() per frame."""
    out = np.stack([resize_bilinear_u8(f, size[0], size[1]) for f in frames])
    return np.ascontiguousarray(out.transpose(0, 3, 1, 2)).astype(np.float32) / np.float32(255.0)


def batches(n_frames, batch_size):
    """Frame ranges handed to encode_video (demo_util.py:152-166): full batches, then the remainder."""
    return [(i, min(i + batch_size, n_frames)) for i in range(0, n_frames, batch_size)]
