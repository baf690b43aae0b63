"""CPU: independent cross-checks of the oracle pieces that no reference-held vector pins ("parity unpinned").

The reference hands these steps to third-party packages that are absent here (librosa 0.7.1: audio_util.py:57,64; libsndfile:
train_nele.py:313; pystoi: intel.py:126,133; pysiib: intel.py:77,100).  The oracle restates them from their published
behaviour; the tests below compare each restatement with a SECOND, independently written implementation that IS installed
(torch.stft / torch.istft, scipy.signal.stft, scipy.signal.resample_poly, scipy.linalg.eigh + numpy.cov, exact rational
arithmetic).  They remove single-author risk; they do not pin parity with the reference's own packages.
"""
from fractions import Fraction

import numpy as np
import pytest
import scipy.linalg
import scipy.signal
import torch

from oracle import estoi as O_estoi
from oracle import features as F
from oracle import siib as O_siib
from oracle import step as O_step


def _speechlike(L, seed):
    rs = np.random.RandomState(seed)
    x = rs.randn(L)
    x = scipy.signal.lfilter([1.0], [1.0, -0.9], x)                      # low-pass tilt
    env = 0.5 * (1 - np.cos(2 * np.pi * 4.0 * np.arange(L) / 16000.0))    # 4 Hz syllabic envelope
    x = x * env
    return (x / np.sqrt(np.mean(x ** 2)) * 0.03).astype(np.float32)


# ------------------------------------------------------------------ STFT / iSTFT (audio_util.py:53-65 -> librosa 0.7.1)
@pytest.mark.parametrize('L', [4096, 33536, 34048, 64000])
def test_stft_vs_torch_stft(L):
    """librosa.stft(center=True, reflect pad, periodic Hann 512, hop 256) == torch.stft with the same conventions."""
    x = _speechlike(L, 1)
    X = F.stft(x)                                                          # [257, T] complex64
    ref = torch.stft(torch.from_numpy(x).double(), n_fft=512, hop_length=256, win_length=512,
                     window=torch.hann_window(512, periodic=True, dtype=torch.float64), center=True, pad_mode='reflect',
                     normalized=False, onesided=True, return_complex=True).numpy()
    assert ref.shape == X.shape == (257, 1 + L // 256)
    scale = np.abs(ref).max()
    assert np.abs(X - ref).max() <= 1e-6 * scale                          # complex64 rounding of a float64 transform


@pytest.mark.parametrize('L', [4096, 33536])
def test_stft_vs_scipy_stft(L):
    """scipy.signal.stft on the reflect-padded signal, un-normalised (scipy divides by sum(window))."""
    x = _speechlike(L, 2)
    X = F.stft(x)
    w = scipy.signal.get_window('hann', 512, fftbins=True)
    xp = np.pad(x.astype(np.float64), 256, mode='reflect')
    _, _, Z = scipy.signal.stft(xp, window=w, nperseg=512, noverlap=256, nfft=512, boundary=None, padded=False, detrend=False,
                                return_onesided=True)
    Z = Z * w.sum()
    assert Z.shape[1] >= X.shape[1]
    Z = Z[:, :X.shape[1]]
    assert np.abs(X - Z).max() <= 1e-6 * np.abs(Z).max()


@pytest.mark.parametrize('L', [4096, 33536, 64000])
def test_istft_vs_torch_istft(L):
    """librosa.istft (window, overlap-add, divide by the window-sum-square, trim 256) == torch.istft(center=True)."""
    x = _speechlike(L, 3)
    X = F.stft(x)
    # a modified spectrum (what Resyn inverts, audio_util.py:76-90): per-bin gains that vary over frames
    rs = np.random.RandomState(4)
    g = np.exp(0.3 * rs.randn(*X.shape)).astype(np.float32)
    Y = (X * g).astype(np.complex64)
    y = F.istft(Y)
    T = X.shape[1]
    ref = torch.istft(torch.from_numpy(Y).to(torch.complex128), n_fft=512, hop_length=256, win_length=512,
                      window=torch.hann_window(512, periodic=True, dtype=torch.float64), center=True, normalized=False,
                      onesided=True, length=256 * (T - 1)).numpy()
    assert y.shape == ref.shape == (256 * (T - 1),)
    # interior: both divide by the same window-sum-square (== 1.5 for a periodic Hann at 50 % overlap)
    assert np.abs(y[256:-256] - ref[256:-256]).max() <= 2e-7 * max(1e-3, np.abs(ref).max()) + 1e-9
    # first / last half frame: only one window covers them; librosa divides by the float32 window-sum-square where it exceeds `tiny`
    assert np.abs(y - ref).max() <= 1e-5 * np.abs(ref).max()


def test_istft_vs_scipy_istft():
    x = _speechlike(8192, 5)
    X = F.stft(x)
    y = F.istft(X)
    w = scipy.signal.get_window('hann', 512, fftbins=True)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')                                    # NOLA warning: boundary=False leaves the first half frame uncovered (trimmed below)
        _, z = scipy.signal.istft(X.astype(np.complex128) / w.sum(), window=w, nperseg=512, noverlap=256, nfft=512, input_onesided=True,
                                  boundary=False)
    z = z[256:256 + y.shape[0]]
    assert np.abs(y[256:-256] - z[256:-256]).max() <= 2e-7 * np.abs(z).max() + 1e-9


# ------------------------------------------------------------------ PCM_16 round trip (train_nele.py:313 + dataloader.py:58)
def test_pcm16_roundtrip_ties_and_clipping():
    """libsndfile's float -> PCM_16 path: lrintf(x * 32767) (round half to even, the FPU default), read back as q / 32768.
    Exact rational arithmetic on values that sit exactly on ties in float32."""
    ties = [(k + 0.5) / 32767.0 for k in (-3, -2, -1, 0, 1, 2, 3, 100, 101, 12344, 12345)]
    vals = np.array(ties + [0.0, 1e-9, -1e-9, 0.03, -0.03, 0.999, 1.0, -1.0, 1.5, -1.5], dtype=np.float32)
    got = O_step.pcm16_roundtrip(vals)
    exp = []
    for v in vals:
        p = Fraction(float(np.float32(v) * np.float32(32767.0)))          # the float32 product the writer rounds
        q = round(p)                                                       # python: round half to even on exact rationals
        q = max(-32768, min(32767, q))
        exp.append(np.float32(q / 32768.0))
    assert np.array_equal(got, np.array(exp, dtype=np.float32))
    assert got.dtype == np.float32
    # a second write of a decoded file (dataloader.py:58 reads what train_nele.py:313 wrote; D's samples are written again next epoch):
    # exact rational check on a grid of decoded PCM_16 values
    q = np.arange(-32768, 32768, 97)
    v32 = (q / 32768.0).astype(np.float32)
    back = O_step.pcm16_roundtrip(v32)
    exp = [np.float32(max(-32768, min(32767, round(Fraction(float(np.float32(v) * np.float32(32767.0)))))) / 32768.0) for v in v32]
    assert np.array_equal(back, np.array(exp, dtype=np.float32))
    assert np.abs(back - v32).max() <= 1.0 / 32768.0                       # at most one LSB per re-encoding (the 32767 / 32768 asymmetry)


# ------------------------------------------------------------------ ESTOI's 16 -> 10 kHz resampler (pystoi.utils.resample_oct)
def test_estoi_resampler_vs_scipy_default_window():
    """The oracle resamples with pystoi's Octave-compatible Kaiser filter (60 dB, roll-off 1/10 of the stop band).  scipy's default
    polyphase filter (Kaiser beta 5) has another transition band, so the two differ - by a BOUNDED amount for band-limited input:
    in the pass band shared by both filters (below 3.5 kHz) the outputs agree to 1e-3 of the signal level."""
    fs = 16000
    t = np.arange(fs) / fs
    x = sum(np.sin(2 * np.pi * f * t + p) for f, p in ((200.0, 0.1), (1000.0, 0.7), (2500.0, 1.3), (3400.0, 2.1)))
    a = O_estoi.resample_16k_to_10k(x)
    b = scipy.signal.resample_poly(x, 5, 8)
    assert a.shape == b.shape == (10000,)
    core = slice(400, -400)                                                # away from the filters' start-up
    assert np.abs(a[core] - b[core]).max() <= 1e-3 * np.abs(b).max()
    # and against the closed form (the resampled signal is the same sinusoids at 10 kHz)
    t10 = np.arange(10000) / 10000.0
    y = sum(np.sin(2 * np.pi * f * t10 + p) for f, p in ((200.0, 0.1), (1000.0, 0.7), (2500.0, 1.3), (3400.0, 2.1)))
    assert np.abs(a[core] - y[core]).max() <= 2e-3 * np.abs(y).max()


def test_estoi_resampler_filter_design():
    """Octave's resample() recipe that pystoi copies: half-length L = ceil((60 - 8) / (28.714 * rw)), Kaiser beta 0.1102 (60 - 8.7),
    unit DC gain after normalisation, cut-off at the output Nyquist."""
    h = O_estoi.resample_window_oct(10000, 16000)
    L = int(np.ceil((60 - 8) / (28.714 * (1.0 / 16) / 10)))
    assert h.shape == (2 * L + 1,) and np.allclose(h, h[::-1])
    hn = h / h.sum()
    H = np.abs(np.fft.rfft(hn, 1 << 16))
    f = np.fft.rfftfreq(1 << 16, 1.0 / 80000.0)                            # the filter runs at 16 kHz * 5
    assert abs(H[0] - 1) < 1e-12
    assert H[f <= 4000.0].min() > 0.97                                     # flat over the band ESTOI uses (< 4.3 kHz third-octave edge)
    assert H[f >= 5600.0].max() < 2e-3                                     # about 60 dB down beyond the roll-off


# ------------------------------------------------------------------ SIIB's KLT step (pysiib: eigh of the stacked covariance)
def test_siib_klt_vs_numpy_cov_and_scipy_eigh():
    L = 33871                                                              # no multiple of 100: nothing repeats (full-rank covariance)
    x = _speechlike(L, 7).astype(np.float64)
    rs = np.random.RandomState(8)
    y = 0.8 * x + 0.01 * rs.randn(L)
    M = 6
    xs, ys = np.tile(x, M), np.tile(y, M)
    val, parts = O_siib.siib_gauss(xs, ys, return_parts=True)
    # rebuild the stacked, mean-removed log-spectra exactly as the oracle does, then take the independent route
    from oracle import intel
    xh = intel.stft(xs).T
    yh = intel.stft(ys).T
    xh = xh.real ** 2 + xh.imag ** 2
    yh = yh.real ** 2 + yh.imag ** 2
    vad = intel.get_vad(xs)
    G2 = O_siib.gammatone_matrix() ** 2
    X = np.log(G2 @ xh[:, vad] + O_siib.EPS)
    Y = np.log(G2 @ yh[:, vad] + O_siib.EPS)
    Tf = int(np.floor(0.2 * O_siib.R))
    X = O_siib.forward_masking(X, Tf); Y = O_siib.forward_masking(Y, Tf)
    X -= X.mean(axis=1, keepdims=True); Y -= Y.mean(axis=1, keepdims=True)
    Xs, Ys = O_siib.stack(X, O_siib.K_STACK), O_siib.stack(Y, O_siib.K_STACK)
    C = np.cov(Xs)                                                         # numpy's own centring and 1 / (n - 1)
    lam, U = scipy.linalg.eigh(C, driver='evd')                            # LAPACK dsyevd through scipy (the oracle: numpy's eigh)
    assert np.allclose(lam, parts['lam'], rtol=1e-9, atol=1e-12 * lam.max())
    Xp, Yp = U.T @ (Xs - Xs.mean(1, keepdims=True)), U.T @ (Ys - Ys.mean(1, keepdims=True))
    rho = np.array([np.corrcoef(Xp[i], Yp[i])[0, 1] for i in range(Xp.shape[0])])
    I = -0.5 * np.log2(1 - (O_siib.RHO_P * rho) ** 2)
    val2 = max(0.0, float(O_siib.R / O_siib.K_STACK * I.sum()))
    assert lam.min() > 1e-8 * lam.max()                                    # full rank: the documented eigenvalue cut is inactive here
    assert abs(val - val2) <= 1e-9 * abs(val2)


def test_siib_gammatone_matrix_shape_and_peaks():
    """28 ERB-spaced 4th-order gammatone magnitude responses, 100 .. 6500 Hz, each normalised to a peak of 1 at the bin nearest
    its centre frequency."""
    A = O_siib.gammatone_matrix()
    assert A.shape == (28, 201) and np.allclose(A.max(axis=1), 1.0)
    f = np.linspace(0, 16000, 401)[:201]
    erb = 21.4 * np.log10(4.37 * np.array([0.1, 6.5]) + 1)
    cf = (10 ** (np.linspace(erb[0], erb[1], 28) / 21.4) - 1) / 4.37 * 1000.0
    assert np.all(np.abs(f[A.argmax(axis=1)] - cf) <= 20.0 + 1e-9)         # bin spacing 40 Hz


# ------------------------------------------------------------------ HASPI's 16 -> 24 kHz resampler (pyhaspi2.py:810-821 -> resampy kaiser_best)
def test_haspi_resampler_vs_scipy_polyphase_and_closed_form():
    """resampy's 'kaiser_best' interpolates a tabulated Kaiser-windowed sinc (64 zero crossings, roll-off 0.9476); scipy's polyphase
    resampler designs its own Kaiser low-pass.  For band-limited input both reproduce the same sinusoids at 24 kHz: bounded difference
    (the oracle then rescales to the input's RMS, pyhaspi2.py:817-818 - undone here through the ratio of the RMS values)."""
    from oracle import haspi as O_haspi
    fs = 16000
    n = 16000
    t = np.arange(n) / fs
    comps = ((150.0, 0.2), (900.0, 1.1), (3100.0, 0.4), (6200.0, 2.0))
    x = sum(np.sin(2 * np.pi * f * t + p) for f, p in comps).astype(np.float32)
    a = np.asarray(O_haspi.resample_24k(x, fs), dtype=np.float64)
    assert a.shape == (24000,)
    b = scipy.signal.resample_poly(x.astype(np.float64), 3, 2)
    t24 = np.arange(24000) / 24000.0
    y = sum(np.sin(2 * np.pi * f * t24 + p) for f, p in comps)
    core = slice(300, -300)
    k = np.sqrt(np.mean(y[core] ** 2) / np.mean(a[core] ** 2))               # the RMS re-scaling of pyhaspi2.py:817-818
    assert abs(k - 1) < 2e-3
    assert np.abs(k * a[core] - y[core]).max() <= 2e-3 * np.abs(y).max()
    assert np.abs(k * a[core] - b[core]).max() <= 5e-3 * np.abs(b).max()
