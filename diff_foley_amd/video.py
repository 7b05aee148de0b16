"""Host side of the video front end: mirror of ``Extract_CAVP_Features`` (inference/demo_util.py:80-173).

The reference re-encodes the clip at 4 fps with ffmpeg, reads the frames with cv2, pushes every frame through
``transforms.Compose([Resize((224, 224)), ToTensor()])`` on a PIL image and hands batches of ``batch_size`` frames to
``CAVP_Inference.encode_video(normalize=True, pool=False)``.  Here the per-frame transform and the encoder both run in
libdfengine.so: ``df_frames_to_tensor`` (csrc/cavp.hip; bit-identical to Pillow's antialiased BILINEAR resize) and
``df_cavp_encode``.  Container decoding (ffmpeg / cv2) is host-side glue and not part of the compute path:
``forward(video_path, ...)`` uses cv2 when it is importable, ``forward_frames(frames)`` takes decoded RGB frames.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import engine as E

PRECISION_BITS = 32 - 8 - 2          # Pillow, Resample.c: 8 bits per channel -> 22-bit fixed-point coefficients


def resample_coeffs(in_size, out_size):
    """Coefficient tables of Pillow's BILINEAR resampling along one axis (precompute_coeffs + normalize_coeffs_8bpc):
    bounds int32 [out][2] = (first input index, tap count), coeffs int32 [out][ksize].  Double precision like Pillow."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = filterscale                      # triangle filter, support 1.0, stretched when down-scaling (antialias)
    ksize = int(math.ceil(support)) * 2 + 1
    centers = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((centers - support + 0.5).astype(np.int64), 0)           # C truncation of non-negative values
    xmax = np.minimum((centers + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    w = 1.0 - np.abs((x + xmin[:, None] - centers[:, None] + 0.5) / filterscale)
    w = np.where((w > 0.0) & (x < n[:, None]), w, 0.0)
    ww = w.sum(1, keepdims=True)
    w = np.where(ww != 0.0, w / np.where(ww == 0.0, 1.0, ww), w)
    kk = np.floor(0.5 + w * float(1 << PRECISION_BITS)).astype(np.int32)        # all weights are >= 0 for this filter
    bounds = np.stack([xmin, n], 1).astype(np.int32)
    return bounds, kk


class _Tables:
    def __init__(self, H, W, oh, ow, device):
        bw, kw = resample_coeffs(W, ow)
        bh, kh = resample_coeffs(H, oh)
        self.ksw, self.ksh = kw.shape[1], kh.shape[1]
        self.bw = torch.from_numpy(bw).to(device)
        self.kw = torch.from_numpy(kw).to(device)
        self.bh = torch.from_numpy(bh).to(device)
        self.kh = torch.from_numpy(kh).to(device)


_tables = {}


def frames_to_tensor(frames, size=(224, 224)):
    """uint8 RGB frames (T, H, W, 3) (numpy or torch, host or device) -> float32 (T, 3, h, w) in [0, 1] on the GPU:
    torchvision ``Resize(size)`` + ``ToTensor()`` per frame, computed by libdfengine (no CPU / torch fallback)."""
    if not torch.cuda.is_available():
        raise RuntimeError("diff_foley_amd.video needs a ROCm GPU (MI355X); no CPU fallback exists")
    if isinstance(frames, np.ndarray):
        frames = torch.from_numpy(np.ascontiguousarray(frames))
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
        raise RuntimeError("frames must be uint8 RGB of shape (T, H, W, 3)")
    dev = frames.device if frames.is_cuda else torch.device("cuda", torch.cuda.current_device())
    frames = frames.to(dev).contiguous()
    T, H, W, _ = frames.shape
    oh, ow = int(size[0]), int(size[1])
    if oh <= 0 or ow <= 0 or H <= 0 or W <= 0:
        raise ValueError(f"frames_to_tensor: frame size {H} x {W} -> {oh} x {ow} (every size must be positive)")
    if T == 0:
        return torch.empty(0, 3, oh, ow, dtype=torch.float32, device=dev)
    key = (H, W, oh, ow, dev)
    if key not in _tables:
        _tables[key] = _Tables(H, W, oh, ow, dev)
    tb = _tables[key]
    out = torch.empty(T, 3, oh, ow, dtype=torch.float32, device=dev)
    # first pass's result: horizontal pass first, [T][H][ow][3] -- except on shrinking frames with H > 100 W, where Pillow (and the
    # library) run the vertical pass first: [T][oh][W][3]
    tmp = torch.empty(T * (oh * W if (H > 100 * W and oh < H) else H * ow) * 3, dtype=torch.uint8, device=dev)
    L = E.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    E._chk(L.df_frames_to_tensor(p(frames), p(out), p(tmp), T, H, W, oh, ow, p(tb.bw), p(tb.kw), tb.ksw, p(tb.bh), p(tb.kh),
                                 tb.ksh, C.c_void_p(torch.cuda.current_stream().cuda_stream)), L)
    return out


class ExtractCAVPFeatures:
    """``Extract_CAVP_Features`` (demo_util.py:80-173) on the engine.  ``stage1_model`` is a
    :class:`diff_foley_amd.CAVPInference` (already loaded and on the GPU), or is built from ``config`` (the ``model`` node of
    inference/config/Stage1_CAVP.yaml as a dict) + ``ckpt_path`` like the reference does."""

    def __init__(self, fps=4, batch_size=2, device=None, tmp_path="./", video_shape=(224, 224), config=None,
                 ckpt_path=None, stage1_model=None):
        self.fps = fps
        self.batch_size = batch_size
        self.device = device
        self.tmp_path = tmp_path
        self.video_shape = tuple(video_shape)
        if stage1_model is None:
            from .ldm import instantiate_from_config
            if config is None or ckpt_path is None:
                raise RuntimeError("ExtractCAVPFeatures needs stage1_model= or config= and ckpt_path=")
            stage1_model = instantiate_from_config(config)
            sd = torch.load(ckpt_path, map_location="cpu")
            sd = sd.get("state_dict", sd)
            stage1_model.load_state_dict({k.replace("module.", ""): v for k, v in sd.items()}, strict=False)
            stage1_model.to(device if device is not None else "cuda")
        self.stage1_model = stage1_model.eval()

    @torch.no_grad()
    def forward_frames(self, frames):
        """Decoded RGB frames (T, H, W, 3) uint8 at ``fps`` -> (T, embed_dim) numpy features, exactly the loop of
        demo_util.py:141-170: batches of ``batch_size`` frames, ``encode_video(normalize=True, pool=False)``."""
        T = frames.shape[0]
        if T == 0:                            # np.concatenate([]) of the reference loop would raise: an empty clip has no features
            raise ValueError("forward_frames: no frames")
        feats = []
        for i in range(0, T, self.batch_size):
            x = frames_to_tensor(frames[i:i + self.batch_size], self.video_shape)          # (t, 3, 224, 224)
            f = self.stage1_model.encode_video(x.unsqueeze(0), normalize=True, pool=False)  # (1, t, embed)
            feats.extend(f.detach().cpu().numpy())
        return np.concatenate(feats)

    @torch.no_grad()
    def forward(self, video_path, start_second=None, truncate_second=None, tmp_path="./tmp_folder"):
        """Reads ``video_path`` at ``fps`` with cv2 (host-side container decoding, as the reference) and returns
        (features, path of the clip).  Raises when cv2 is not installed: use forward_frames() with decoded frames."""
        try:
            import cv2
        except ImportError as ex:
            raise RuntimeError("decoding a video file needs cv2 (+ ffmpeg for re-encoding), which this environment does not "
                               "have; decode the clip at 4 fps yourself and call forward_frames(frames)") from ex
        cap = cv2.VideoCapture(video_path)
        src_fps = cap.get(cv2.CAP_PROP_FPS) or self.fps
        first = int((start_second or 0) * src_fps)
        last = int(((start_second or 0) + truncate_second) * src_fps) if truncate_second else None
        step = src_fps / float(self.fps)
        frames, idx, nxt = [], 0, float(first)
        while cap.isOpened():
            ok, bgr = cap.read()
            if not ok or (last is not None and idx >= last):
                break
            if idx >= nxt - 1e-6:
                frames.append(cv2.cvtColor(bgr, cv2.COLOR_BGR2RGB))
                nxt += step
            idx += 1
        cap.release()
        if not frames:
            raise RuntimeError(f"no frames decoded from {video_path}")
        return self.forward_frames(np.stack(frames)), video_path

    __call__ = forward
