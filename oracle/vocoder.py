"""TEST INFRASTRUCTURE ONLY (never imported by diff_foley_amd/).  **PARITY UNPINNED.**

CPU restatement of the mel -> waveform tail of the notebook, ``inverse_op`` (inference/demo_util.py:196-211):

    spec = spec * 100 - 100;  spec = (spec + 20) / 20;  spec = 10 ** spec
    S    = librosa.feature.inverse.mel_to_stft(spec, sr=22050, n_fft=1024, fmin=125, fmax=7600, power=1)
    wav  = librosa.griffinlim(S, hop_length=256)

The algorithm lives in a third-party dependency, **librosa 0.8.0** (pinned in /root/reference/requirements.txt:56), which
is not installed here and whose source is not under /root/reference; the reference holds no test or golden vector for this
path.  What follows restates librosa 0.8.0's published algorithms from its documentation / source layout:
  * ``filters.mel`` -- Slaney mel scale (htk=False), triangular filters, Slaney area normalisation, float32;
  * ``util.nnls`` -- non-negative least squares per block of columns with scipy's L-BFGS-B (bounds (0, None)) started
    from the clipped least-squares solution; block width MAX_MEM_BLOCK // (n_freq * itemsize) = 127 columns;
  * ``griffinlim`` -- "fast" Griffin-Lim (Perraudin et al. 2013), momentum 0.99, 32 iterations, random initial phase,
    ``stft`` (hann, centre, reflect padding) / ``istft`` (window-sum-square normalised overlap-add, centre trimmed).
With neither librosa nor a reference fixture available the restatement cannot be checked against the real thing:
the oracle and every test that uses it carry the label "parity unpinned" (DESIGN.md section 4).  What IS checked without
librosa: the filterbank against its defining properties, stft/istft as an exact inverse pair, and the HIP path against
this restatement on the same seeded phases."""
import numpy as np

SR, N_FFT, HOP, FMIN, FMAX = 22050, 1024, 256, 125.0, 7600.0


# ---------------------------------------------------------------------------------------- librosa.filters.mel (slaney)
def hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(n_mels, sr=SR, n_fft=N_FFT, fmin=FMIN, fmax=FMAX):
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


# ---------------------------------------------------------------------------------------- librosa.util.nnls
def nnls_lbfgs(A, B):
    """librosa 0.8.0 util.nnls for 2-D B: blocks of MAX_MEM_BLOCK // (A.shape[-1] * itemsize) columns, each solved with
    scipy.optimize.fmin_l_bfgs_b from the clipped least-squares solution."""
    from scipy.optimize import fmin_l_bfgs_b
    x = np.linalg.lstsq(A, B, rcond=None)[0].astype(A.dtype)
    np.clip(x, 0, None, out=x)
    n_columns = int((2 ** 8 * 2 ** 10) // (A.shape[-1] * A.itemsize))

    def obj(xf, shape, A_, B_):
        xx = xf.reshape(shape)
        diff = np.dot(A_, xx) - B_
        return 0.5 * np.sum(diff ** 2), np.dot(A_.T, diff).flatten()
    for s in range(0, x.shape[-1], n_columns):
        t = min(s + n_columns, B.shape[-1])
        x0 = x[:, s:t]
        sol, _, _ = fmin_l_bfgs_b(obj, x0.ravel().astype(np.float64), args=(x0.shape, A.astype(np.float64), B[:, s:t].astype(np.float64)),
                                  bounds=[(0, None)] * x0.size)
        x[:, s:t] = sol.reshape(x0.shape)
    return x


def undo_mel_normalisation(spec):
    """First three lines of inverse_op (demo_util.py:205-207): [0,1]-normalised log-mel -> linear amplitude."""
    s = spec * 100 - 100
    s = (s + 20) / 20
    return 10 ** s


# ---------------------------------------------------------------------------------------- librosa stft / istft / griffinlim
def hann(n):                                     # scipy.signal.get_window('hann', n, fftbins=True)
    return (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n)).astype(np.float64)


def stft(y, n_fft=N_FFT, hop=HOP):
    w = hann(n_fft)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    return np.fft.rfft(yp[idx] * w[:, None], axis=0).astype(np.complex64)          # [1 + n_fft/2][frames]


def window_sumsquare(n_frames, n_fft=N_FFT, hop=HOP):
    n = n_fft + hop * (n_frames - 1)
    x = np.zeros(n, dtype=np.float32)
    wsq = (hann(n_fft) ** 2).astype(np.float32)          # norm=None
    for i in range(n_frames):
        x[i * hop:i * hop + n_fft] += wsq
    return x


def istft(S, n_fft=N_FFT, hop=HOP):
    n_frames = S.shape[1]
    w = hann(n_fft)
    ytmp = w[:, None] * np.fft.irfft(S, n=n_fft, axis=0)
    y = np.zeros(n_fft + hop * (n_frames - 1), dtype=np.float32)
    for i in range(n_frames):
        y[i * hop:i * hop + n_fft] += ytmp[:, i].astype(np.float32)
    wss = window_sumsquare(n_frames, n_fft, hop)
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2:-(n_fft // 2)]


def griffinlim(S, phase0, n_iter=32, hop=HOP, momentum=0.99):
    """librosa.griffinlim(S, hop_length=256): n_fft inferred as 2 * (S.shape[0] - 1); ``phase0`` in [0, 1) stands for
    the ``rng.rand(*S.shape)`` draw of init='random' (random_state is None in the notebook: not reproducible there)."""
    n_fft = 2 * (S.shape[0] - 1)
    angles = np.exp(2j * np.pi * phase0).astype(np.complex64)
    rebuilt = 0.0
    for _ in range(n_iter):
        tprev = rebuilt
        inverse = istft(S * angles, n_fft, hop)
        rebuilt = stft(inverse, n_fft, hop)
        angles[:] = rebuilt - (momentum / (1 + momentum)) * tprev
        angles[:] /= np.abs(angles) + 1e-16
    return istft(S * angles, n_fft, hop)


def inverse_op(spec, phase0):
    """demo_util.inverse_op with the random phase made explicit: spec [n_mels][T] -> wav [(T-1)*256]."""
    amp = undo_mel_normalisation(spec.astype(np.float32))
    A = mel_filterbank(spec.shape[0])
    S = nnls_lbfgs(A, amp.astype(np.float32))          # power = 1: no root
    return griffinlim(S, phase0)
