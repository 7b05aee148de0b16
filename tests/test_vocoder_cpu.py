"""CPU: the mel -> waveform oracle (oracle/vocoder.py, restating librosa 0.8.0's mel_to_stft + griffinlim for
demo_util.inverse_op).  PARITY UNPINNED: librosa is not installed and the reference ships no vector for this path, so
the restatement is checked against the DEFINING PROPERTIES of each piece, not against librosa outputs."""
import numpy as np

import diff_foley_amd  # noqa: F401
from diff_foley_amd import vocoder as V
from oracle import vocoder as ov


def test_slaney_mel_scale_and_filterbank_properties():
    assert abs(float(ov.hz_to_mel(1000.0)) - 15.0) < 1e-12                      # linear below 1 kHz: 200/3 Hz per mel
    f = np.array([60.0, 440.0, 1000.0, 4000.0, 7600.0])
    assert np.allclose(ov.mel_to_hz(ov.hz_to_mel(f)), f, rtol=1e-12)
    A = ov.mel_filterbank(128)
    assert A.shape == (128, 513) and A.dtype == np.float32 and (A >= 0).all()
    freqs = np.linspace(0, ov.SR / 2, 513)
    centre = (A * freqs[None]).sum(1) / A.sum(1)
    assert np.all(np.diff(centre) > 0) and centre[0] > ov.FMIN and centre[-1] < ov.FMAX
    assert not A[:, freqs < ov.FMIN].any() and not A[:, freqs > ov.FMAX].any()
    wide = (A > 0).sum(1) >= 8                                                 # Slaney normalisation: unit area in Hz
    assert np.allclose((A[wide] * (ov.SR / ov.N_FFT)).sum(1), 1.0, atol=0.05)
    assert np.array_equal(V.mel_filterbank(128), A) and np.array_equal(V.mel_filterbank(80), ov.mel_filterbank(80))


def test_stft_istft_are_an_exact_inverse_pair():
    rng = np.random.default_rng(0)
    y = rng.standard_normal(256 * 40).astype(np.float32)
    S = ov.stft(y)
    assert S.shape == (513, 41) and S.dtype == np.complex64
    assert np.abs(ov.istft(S) - y).max() < 5e-6                               # hann, hop n_fft/4: COLA
    w = ov.window_sumsquare(41)
    assert np.allclose(w[1024:-1024], 1.5, atol=1e-5)                          # sum of 4 shifted hann^2 = 1.5


def test_nnls_reaches_the_exact_solution_set():
    rng = np.random.default_rng(1)
    A = ov.mel_filterbank(128)
    St = np.abs(rng.standard_normal((513, 6))).astype(np.float32)
    B = A @ St
    X = ov.nnls_lbfgs(A, B)
    assert (X >= 0).all()
    assert np.linalg.norm(A @ X - B) / np.linalg.norm(B) < 5e-3


def test_griffinlim_improves_consistency_and_undo_normalisation():
    rng = np.random.default_rng(2)
    y = np.sin(2 * np.pi * 440 * np.arange(256 * 20) / ov.SR).astype(np.float32) + 0.1 * rng.standard_normal(256 * 20).astype(np.float32)
    S = np.abs(ov.stft(y))
    ph = rng.random(S.shape)
    err = []
    for n in (1, 32):
        w = ov.griffinlim(S, ph, n_iter=n)
        assert w.shape == (256 * 20,)
        err.append(np.linalg.norm(np.abs(ov.stft(w)) - S) / np.linalg.norm(S))
    assert err[1] < 0.6 * err[0] and err[1] < 0.2          # measured 0.357 -> 0.183 (noisy tone, 20 frames)
    assert np.allclose(ov.undo_mel_normalisation(np.array([0.8, 1.0])), [1.0, 10.0])     # 10 ** ((v*100 - 80) / 20)
