"""mel -> waveform (csrc/vocoder.hip) over clip lengths and batch sizes other than the decoded 512-frame mel: the NNLS kernel works on
tiles of 16 frames (ragged last tile), Griffin-Lim's overlap-add gathers a data-dependent number of frames per sample near both
ends.  Product against the CPU oracle (oracle/vocoder.py; PARITY UNPINNED, see its header) on the same seeded magnitudes and phases."""
import numpy as np
import pytest
import torch

from helpers import fuzz_seeds

pytestmark = pytest.mark.gpu


def _norm_logmel(amp):
    return ((20.0 * np.log10(amp) - 20.0) + 100.0) / 100.0


@pytest.mark.parametrize("T,B", [(2, 1), (3, 2), (5, 1), (15, 3), (16, 1), (17, 2), (33, 1), (100, 2)])
def test_griffinlim_and_nnls_over_clip_lengths(T, B):
    from diff_foley_amd import vocoder as V
    from oracle import vocoder as ov
    rng = np.random.default_rng(100 + T)
    S = (np.abs(rng.standard_normal((B, 513, T))) * np.exp(-np.arange(513) / 120.0)[None, :, None]).astype(np.float32)
    ph = rng.random((B, 513, T)).astype(np.float32)
    St = torch.from_numpy(np.ascontiguousarray(S.transpose(0, 2, 1))).cuda()          # [B][T][513]
    w = V.griffinlim(St, torch.from_numpy(ph).cuda(), n_iter=4).cpu().numpy()
    assert w.shape == (B, 256 * (T - 1)) and np.isfinite(w).all()
    for b in range(B):
        ref = ov.griffinlim(S[b], ph[b], n_iter=4)
        err = np.linalg.norm(w[b] - ref) / (np.linalg.norm(ref) + 1e-30)
        assert err < 2e-3, (T, b, err)               # 4 iterations: 1 iteration measures 4e-7, 32 measure 6e-5 .. 2e-2
    A = ov.mel_filterbank(128)
    amp = np.einsum("mf,bft->bmt", A, S)
    mel = torch.from_numpy(_norm_logmel(np.maximum(amp, 1e-7)).astype(np.float32)).cuda()
    X = V.mel_to_stft(mel).cpu().numpy()                                              # [B][T][513]
    assert X.shape == (B, T, 513) and (X >= 0).all() and np.isfinite(X).all()
    for b in range(B):
        res = np.linalg.norm(A @ X[b].T - amp[b]) / np.linalg.norm(amp[b])
        assert res < 1e-2, (T, b, res)
    # rows of a batch are independent: clip 0 alone gives the same bits
    if B > 1:
        w0 = V.griffinlim(St[:1].contiguous(), torch.from_numpy(ph[:1]).cuda(), n_iter=4).cpu().numpy()
        assert np.array_equal(w0[0], w[0])
        X0 = V.mel_to_stft(mel[:1].contiguous()).cpu().numpy()
        assert np.array_equal(X0[0], X[0])


@pytest.mark.parametrize("seed", fuzz_seeds(6))
def test_vocoder_random_clip_lengths_batches_iterations(seed):
    """Seeded draws over what the shape list above fixes: clip length 2 .. 300 frames, batch 1 .. 6, 1 .. 6 Griffin-Lim iterations,
    momentum 0 / 0.5 / 0.99, 64 / 80 / 128 mel bands for the NNLS inversion (DF_FUZZ_SEED0 / DF_FUZZ_CASES sweep further seeds)."""
    from diff_foley_amd import vocoder as V
    from oracle import vocoder as ov
    r = np.random.default_rng(8800 + seed)
    T, B = int(r.integers(2, 301)), int(r.integers(1, 7))
    n_iter, mom, n_mels = int(r.integers(1, 7)), float(r.choice([0.0, 0.5, 0.99])), int(r.choice([64, 80, 128]))
    S = (np.abs(r.standard_normal((B, 513, T))) * np.exp(-np.arange(513) / float(r.choice([40.0, 120.0, 400.0])))[None, :, None]).astype(np.float32)
    ph = r.random((B, 513, T)).astype(np.float32)
    St = torch.from_numpy(np.ascontiguousarray(S.transpose(0, 2, 1))).cuda()
    w = V.griffinlim(St, torch.from_numpy(ph).cuda(), n_iter=n_iter, momentum=mom).cpu().numpy()
    assert w.shape == (B, 256 * (T - 1)) and np.isfinite(w).all(), (T, B, n_iter, mom)
    for b in range(B):
        ref = ov.griffinlim(S[b], ph[b], n_iter=n_iter, momentum=mom)
        err = np.linalg.norm(w[b] - ref) / (np.linalg.norm(ref) + 1e-30)
        assert err < 5e-3, (T, B, b, n_iter, mom, err)
    A = ov.mel_filterbank(n_mels)
    amp = np.einsum("mf,bft->bmt", A, S)
    mel = torch.from_numpy(_norm_logmel(np.maximum(amp, 1e-7)).astype(np.float32)).cuda()
    X = V.mel_to_stft(mel).cpu().numpy()
    assert X.shape == (B, T, 513) and (X >= 0).all() and np.isfinite(X).all()
    for b in range(B):
        res = np.linalg.norm(A @ X[b].T - amp[b]) / np.linalg.norm(amp[b])
        assert res < 2e-2, (T, B, b, n_mels, res)


def test_a_single_frame_and_an_empty_batch():
    """One STFT frame is zero hops of audio (librosa: length = hop * (frames - 1)); an empty batch is an empty result."""
    from diff_foley_amd import vocoder as V
    S = torch.rand(2, 1, 513).cuda()
    w = V.griffinlim(S, torch.rand(2, 513, 1).cuda(), n_iter=2)
    assert w.shape == (2, 0)
    assert V.griffinlim(torch.rand(0, 8, 513).cuda(), n_iter=2).shape == (0, 256 * 7)
    assert V.mel_to_stft(torch.rand(0, 128, 8).cuda()).shape == (0, 8, 513)
