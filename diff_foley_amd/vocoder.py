"""Mel-spectrogram -> waveform on the GPU: mirror of ``inverse_op`` (inference/demo_util.py:196-211).

    wav = inverse_op(mel[k])            # notebook cell 13, one call per generated sample, 24 of its 30 seconds on CPU

``inverse_op(spec)`` here has the same signature and return value (float32 numpy waveform of (T-1)*256 samples) but runs
in libdfengine.so: ``df_mel_to_stft`` (NNLS inversion of the Slaney mel filterbank) + ``df_griffinlim`` (32 fast
Griffin-Lim iterations).  ``mel_to_wave`` is the batched form for a whole ``decode_first_stage`` output.  The constants
librosa 0.8.0 builds on the fly -- mel filterbank, its pseudo-inverse (the clipped least-squares start of
``librosa.util.nnls``), hann window, window sum-square, FFT twiddles -- are computed once per shape on the host in double
precision and kept on the device.  There is no CPU fallback.

Algorithmic difference to librosa, stated: the NNLS objective is minimised with FISTA (fixed 200 iterations) instead of
scipy's L-BFGS-B; the problem is under-determined (513 unknowns, 128 equations per frame), so the minimisers agree in
their residual |A x - mel|, not element by element (tests/test_vocoder_gpu.py, oracle/vocoder.py: parity unpinned --
librosa itself is not available offline)."""
import ctypes as C

import numpy as np
import torch

from . import engine as E

SR, N_FFT, HOP, FMIN, FMAX = 22050, 1024, 256, 125.0, 7600.0
N_BIN = N_FFT // 2 + 1


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    return np.where(f >= 1000.0, 1000.0 / f_sp + np.log(np.maximum(f, 1e-30) / 1000.0) / (np.log(6.4) / 27.0), f / f_sp)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    return np.where(m >= 1000.0 / f_sp, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 1000.0 / f_sp)), f_sp * m)


def mel_filterbank(n_mels, sr=SR, n_fft=N_FFT, fmin=FMIN, fmax=FMAX):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax): Slaney scale, triangular filters, area-normalised, float32."""
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    return (w * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]).astype(np.float32)


class _Consts:
    def __init__(self, n_mels, T, dev):
        A = mel_filterbank(n_mels)
        A64 = A.astype(np.float64)
        self.A = torch.from_numpy(A).to(dev)
        self.At = torch.from_numpy(np.ascontiguousarray(A.T)).to(dev)
        self.Pt = torch.from_numpy(np.ascontiguousarray(np.linalg.pinv(A64).T).astype(np.float32)).to(dev)
        self.inv_L = float(1.0 / np.linalg.norm(A64, 2) ** 2)
        win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N_FFT) / N_FFT)                  # periodic hann
        self.window = torch.from_numpy(win.astype(np.float32)).to(dev)
        wss = np.zeros(N_FFT + HOP * (T - 1), dtype=np.float32)
        wsq = (win ** 2).astype(np.float32)
        for i in range(T):
            wss[i * HOP:i * HOP + N_FFT] += wsq
        self.wss = torch.from_numpy(wss).to(dev)
        k = np.arange(N_FFT // 2)
        tw = np.stack([np.cos(2 * np.pi * k / N_FFT), -np.sin(2 * np.pi * k / N_FFT)], -1)
        self.tw = torch.from_numpy(tw.astype(np.float32)).to(dev)


_consts = {}


def _get_consts(n_mels, T, dev):
    key = (n_mels, T, dev)
    if key not in _consts:
        _consts[key] = _Consts(n_mels, T, dev)
    return _consts[key]


def _p(t):
    return C.c_void_p(t.data_ptr())


@torch.no_grad()
def mel_to_stft(mel, nnls_iters=200):
    """(B, n_mels, T) normalised log-mel (decode_first_stage(z)[:, 0]) on the GPU -> (B, T, 513) linear STFT magnitude."""
    if not mel.is_cuda:
        raise RuntimeError("diff_foley_amd.vocoder runs on a ROCm GPU only (no CPU path)")
    mel = mel.to(torch.float32).contiguous()
    B, NM, T = mel.shape
    c = _get_consts(NM, T, mel.device)
    S = torch.empty(B, T, N_BIN, dtype=torch.float32, device=mel.device)
    if S.numel() == 0:                    # empty batch / no frames: an empty result
        return S
    L = E.lib()
    E._chk(L.df_mel_to_stft(_p(mel), B, NM, T, _p(c.A), _p(c.At), _p(c.Pt), c.inv_L, int(nnls_iters), _p(S), E._stream()), L)
    return S


@torch.no_grad()
def griffinlim(S, phase0=None, n_iter=32, momentum=0.99, generator=None):
    """(B, T, 513) magnitudes -> (B, (T-1)*256) waveform.  ``phase0`` (B, 513, T) uniform in [0, 1) plays the role of
    librosa's ``rng.rand(*S.shape)``; drawn with ``generator`` when omitted (librosa's default is unseeded too)."""
    B, T, F = S.shape
    if F != N_BIN:
        raise RuntimeError(f"griffinlim expects {N_BIN} frequency bins (n_fft 1024)")
    dev = S.device
    if phase0 is None:
        phase0 = torch.rand(B, N_BIN, T, device=dev, generator=generator)
    phase0 = phase0.to(dev, torch.float32).contiguous()
    if B == 0 or T <= 1:                  # no clips, or a single frame (zero hops of audio): an empty waveform
        return torch.empty(B, HOP * max(T - 1, 0), dtype=torch.float32, device=dev)
    c = _get_consts(128, T, dev)          # window / sum-square / twiddles do not depend on the mel size
    ang = torch.empty(B, T, N_BIN, 2, dtype=torch.float32, device=dev)
    r0, r1 = torch.empty_like(ang), torch.empty_like(ang)
    frames = torch.empty(B, T, N_FFT, dtype=torch.float32, device=dev)
    wav = torch.empty(B, HOP * (T - 1), dtype=torch.float32, device=dev)
    L = E.lib()
    E._chk(L.df_griffinlim(_p(S.contiguous()), _p(phase0), B, T, int(n_iter), float(momentum), _p(c.tw), _p(c.window),
                           _p(c.wss), _p(ang), _p(r0), _p(r1), _p(frames), _p(wav), E._stream()), L)
    return wav


@torch.no_grad()
def mel_to_wave(mel, phase0=None, generator=None):
    """Batched inverse_op: (B, n_mels, T) -> (B, (T-1)*256) float32 on the GPU."""
    return griffinlim(mel_to_stft(mel), phase0=phase0, generator=generator)


def inverse_op(spec, phase0=None):
    """Drop-in for demo_util.inverse_op(spec): spec (n_mels, T) numpy / tensor -> float32 numpy waveform."""
    t = torch.as_tensor(np.asarray(spec) if not torch.is_tensor(spec) else spec, dtype=torch.float32)
    dev = torch.device("cuda", torch.cuda.current_device()) if not t.is_cuda else t.device
    ph = None if phase0 is None else torch.as_tensor(np.asarray(phase0), dtype=torch.float32)[None].to(dev)
    return mel_to_wave(t[None].to(dev), phase0=ph)[0].cpu().numpy()
