"""Oracle: signal features / resynthesis (numpy).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates reference ``audio_util.py`` (STFT, ISTFT, compute_band_E, interp_band_gain, Resyn,
NoisePSD, Sp_and_phase_Speech, Sp_and_phase_Noise, SP_to_wav, rms) and
``noise_est/imcra.py`` (imcra_est.estimate + imcra.update).

dtype notes: the goldens are produced by importing the reference under numpy 2.2 (NEP-50
"weak" python scalars), so this file reproduces numpy-2 promotion where the reference mixes
python floats, float32 scalars and float64 arrays; DESIGN.md lists the places where numpy 1.17
(the reference's declared version) would have promoted differently.

STFT / iSTFT follow librosa 0.7.1 (not installed: PARITY UNPINNED, algorithm restated):
  stft : reflect-pad n_fft//2, periodic Hann(512) float64, frames of 512 at hop 256,
         rfft in float64, cast to complex64; T = 1 + L//256.
  istft: irfft float64, * window, overlap-add into a float32 buffer, divide by the float32
         window-sum-square where > tiny, trim 256 each side; length 256*(T-1).
"""
import numpy as np

# audio_util.py:23
GMTBAND = [0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 28, 30, 32,
           34, 36, 38, 41, 43, 46, 49, 52, 55, 58, 62, 66, 70, 74, 79, 83, 88, 93, 99, 105, 111, 117, 124, 131, 139,
           147, 156, 165, 174, 184, 195, 206, 218, 230, 243, 257]
NB_BANDS = 64
N_FFT = 512
HOP = 256
N_BINS = 257


def hann_periodic(n=N_FFT):
    """scipy.signal.get_window('hann', n, fftbins=True), float64."""
    from scipy.signal import get_window
    return get_window('hann', n, fftbins=True)


def n_frames(L):
    return 1 + L // HOP


def stft(x):
    """audio_util.py:53-58 -> librosa.stft(x, n_fft=512, hop_length=256, win_length=512).
    x: [L] float32 -> [257, T] complex64."""
    x = np.asarray(x)
    assert x.ndim == 1 and x.shape[0] > N_FFT // 2
    xp = np.pad(x, N_FFT // 2, mode='reflect')
    T = n_frames(x.shape[0])
    idx = np.arange(N_FFT)[:, None] + HOP * np.arange(T)[None, :]
    frames = xp[idx]                                   # [512, T] float32
    w = hann_periodic()[:, None]
    return np.fft.rfft(w * frames, axis=0).astype(np.complex64)


def window_sumsquare(T):
    """librosa.filters.window_sumsquare(hann, T, hop 256, n_fft 512, dtype float32): float64
    squares accumulated into a float32 buffer frame by frame."""
    wsq = hann_periodic() ** 2
    n = N_FFT + HOP * (T - 1)
    out = np.zeros(n, dtype=np.float32)
    for i in range(T):
        s = i * HOP
        out[s:s + N_FFT] += wsq[:max(0, min(N_FFT, n - s))]
    return out


def istft(X):
    """audio_util.py:60-65 -> librosa.istft(X, hop_length=256, win_length=512).
    X: [257, T] complex -> [256*(T-1)] float32."""
    X = np.asarray(X)
    T = X.shape[1]
    w = hann_periodic()[:, None]
    ytmp = w * np.fft.irfft(X, n=N_FFT, axis=0)        # float64 [512, T]
    n = N_FFT + HOP * (T - 1)
    y = np.zeros(n, dtype=np.float32)
    for i in range(T):
        s = i * HOP
        y[s:s + N_FFT] = y[s:s + N_FFT] + ytmp[:, i]   # float64 add, stored float32
    wss = window_sumsquare(T)
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[N_FFT // 2:-(N_FFT // 2)]


def band_weights():
    """The (bin -> band) triangular weights of audio_util.py:30-50 as two float32 arrays
    [(band_lo, w_lo, w_hi)] per bin: bin k in band i contributes w_lo to band i, w_hi to i+1."""
    lo = np.zeros(N_BINS, np.int32)
    wl = np.zeros(N_BINS, np.float32)
    wh = np.zeros(N_BINS, np.float32)
    for i in range(NB_BANDS - 1):
        size = GMTBAND[i + 1] - GMTBAND[i]
        for j in range(size):
            frac = float(j) / size
            lo[GMTBAND[i] + j] = i
            wl[GMTBAND[i] + j] = np.float32(1 - frac)
            wh[GMTBAND[i] + j] = np.float32(frac)
    return lo, wl, wh


def compute_band_E(X):
    """audio_util.py:30-50.  X: [T, 257] magnitudes (float32) -> [T, 64] float32.
    Per bin: tmp = X**2 in float32; (1-frac)*tmp and frac*tmp rounded to float32 (numpy-2 weak
    scalars), accumulated in float64 in the reference's loop order, stored float32."""
    X = np.asarray(X, dtype=np.float32)
    T = X.shape[0]
    sumE = np.zeros((T, NB_BANDS), dtype=np.float64)
    tmp = X * X                                          # float32
    for i in range(NB_BANDS - 1):
        size = GMTBAND[i + 1] - GMTBAND[i]
        for j in range(size):
            frac = float(j) / size
            t = tmp[:, GMTBAND[i] + j]
            sumE[:, i] += np.float32(1 - frac) * t       # float32 product, float64 accumulate
            sumE[:, i + 1] += np.float32(frac) * t
    return sumE.astype(np.float32)


def interp_band_gain(bandE):
    """audio_util.py:93-110.  bandE: [64] float32 -> g [257] float64 (values formed in float32)."""
    bandE = np.asarray(bandE, dtype=np.float32)
    g = np.ones(N_BINS, dtype=np.float64)
    for i in range(NB_BANDS - 1):
        size = GMTBAND[i + 1] - GMTBAND[i]
        for j in range(size):
            frac = float(j) / size
            g[GMTBAND[i] + j] = np.float32(1 - frac) * bandE[i] + np.float32(frac) * bandE[i + 1]
    g[0] = 1e-4
    g[1] = 1e-4
    g[256] = 1e-2
    return g


def interp_band_gain_batch(alpha):
    """Vectorised interp_band_gain over frames: alpha [T,64] float32 -> g [257,T] float64."""
    alpha = np.asarray(alpha, dtype=np.float32)
    lo, wl, wh = band_weights()
    g = (wl[None, :] * alpha[:, lo] + wh[None, :] * alpha[:, np.minimum(lo + 1, NB_BANDS - 1)]).astype(np.float64)
    g[:, 0] = 1e-4
    g[:, 1] = 1e-4
    g[:, 256] = 1e-2
    return g.T


def resyn(X, alpha):
    """audio_util.py:76-90.  X [257,T] complex, alpha [T,64] (alpha^2 energy gains) -> wav."""
    gain = np.sqrt(interp_band_gain_batch(alpha))
    return istft(gain * X)


def sp_to_wav(alpha2, mag, phase):
    """audio_util.py:458-461."""
    cm = np.multiply(mag, np.exp(1j * phase))
    return resyn(cm, alpha2)


def rms(x):
    """audio_util.py:463-464."""
    return np.sqrt(np.mean(x ** 2))


# --------------------------------------------------------------------------- IMCRA
def _fsmooth(P):
    """imcra.py:335-336 with w=1: rows [1/4,1/2,1/4], edges [2/3,1/3] / [1/3,2/3]; float64,
    summed left to right."""
    P = P.astype(np.float64)
    K = P.shape[0]
    w = np.tile(np.array([0.5, 1.0, 0.5]), (K, 1))
    w[0, 0] = 0.0
    w[K - 1, 2] = 0.0
    w = w / np.sum(w, 1, keepdims=True)
    Pm = np.concatenate((P[:1], P[:-1]))                 # clipped index k-1
    Pp = np.concatenate((P[1:], P[-1:]))                 # clipped index k+1
    return (w[:, 0] * Pm + w[:, 1] * P) + w[:, 2] * Pp


def imcra_noise_psd(Y):
    """noise_est/imcra.py:521-577 (imcra_est.estimate, Bmin=3.2, alpha=0.92, xi_min=10**(-25/20),
    IS=15) driving imcra.update (imcra.py:363-484) with init_params (338-361), fsmooth (335-336)
    and post_speech_prob (22-36).  Y: [257, T] complex64 -> noise PSD [257, T] float32.

    Working vectors are 1-D over the 257 bins; dtypes follow the reference under numpy 2.2:
    |Y|^2 float32; Lambda_D / Gamma float32 while l < 15, float64 afterwards."""
    Y = np.asarray(Y)
    assert Y.dtype == np.complex64
    K, L = Y.shape
    IS, U, V = 15, 8, 15
    alpha_s, alpha_d = 0.9, 0.85
    Gamma0, Gamma1, zeta0, beta, Bmin = 4.6, 3, 1.67, 1.47, 3.2
    alpha_dd, xi_min = 0.92, 10 ** (-25. / 20)
    p_upthr = 0.9

    out = np.zeros((K, L), dtype=np.float32)
    G = 1
    Gamma = 1
    Lambda_D = 1e-6 * np.ones(K)
    j = 0
    u = 0
    Storing = np.zeros((K, U))
    tStoring = np.zeros((K, U))
    for l in range(L):
        Y2 = np.abs(Y[:, l]) ** 2                        # float32
        xi_G = (G ** 2) * Gamma
        Gamma = Y2 / Lambda_D
        xi_ML = Gamma - 1
        xi_ML[xi_ML < 1e-6] = 1e-6
        xi = alpha_dd * xi_G + (1 - alpha_dd) * xi_ML
        xi = np.asarray(xi, dtype=np.float64)
        xi[xi < xi_min] = xi_min
        G = xi / (1 + xi)
        # ---- imcra.update
        if l == 0:
            S = _fsmooth(Y2)
            tS = S.copy(); Smin = S.copy(); tSmin = S.copy(); Smin_sw = S.copy(); tSmin_sw = S.copy()
            ov_Lambda_D = Y2
            Lambda_D = ov_Lambda_D
        if l < IS:
            Sf = _fsmooth(Y2)
            S = alpha_s * S + (1 - alpha_s) * Sf
            Smin = np.minimum(Smin, S)
            Smin_sw = np.minimum(Smin_sw, S)
            Lambda_D = alpha_d * Lambda_D + (1 - alpha_d) * Y2          # float32
        else:
            Sf = _fsmooth(Y2)
            S = alpha_s * S + (1 - alpha_s) * Sf
            Smin = np.minimum(Smin, S)
            Smin_sw = np.minimum(Smin_sw, S)
            Gamma_min = Y2 / (Bmin * Smin)
            zeta = S / (Bmin * Smin)
            I = np.zeros(K)
            I[(Gamma_min < Gamma0) & (zeta < zeta0)] = 1
            norm = _fsmooth(I)
            tSf = _fsmooth(I * Y2)
            pos = norm > 0
            tSf[pos] = tSf[pos] / norm[pos]
            tS = alpha_s * tS + (1 - alpha_s) * tSf
            tSmin = np.minimum(tSmin, tS)
            tSmin_sw = np.minimum(tSmin_sw, tS)
            tGamma_min = Y2 / (Bmin * tSmin)
            tzeta = S / (Bmin * tSmin)
            q = np.zeros(K)
            q[(tGamma_min <= 1) & (tzeta < zeta0)] = 1
            m = (1 < tGamma_min) & (tGamma_min < Gamma1) & (tzeta < zeta0)
            q[m] = (Gamma1 - tGamma_min[m]) / (Gamma1 - 1)
            nu = Gamma * xi / (1 + xi)
            p = np.zeros(K)
            a = q < 1
            p[a] = 1. / (1 + (q[a] / (1 - q[a])) * (1 + xi[a]) * np.exp(-nu[a]))
            p[p > p_upthr] = p_upthr
            tad = alpha_d + (1 - alpha_d) * p
            ov_Lambda_D = tad * ov_Lambda_D + (1 - tad) * Y2
            Lambda_D = beta * ov_Lambda_D
            j += 1
            if j == V:
                if u < U:
                    Storing[:, u] = Smin_sw
                    tStoring[:, u] = tSmin_sw
                else:
                    Storing = np.roll(Storing, -1, axis=1); Storing[:, -1] = Smin_sw
                    tStoring = np.roll(tStoring, -1, axis=1); tStoring[:, -1] = tSmin_sw
                Smin = np.min(Storing[:, :u + 1], 1)
                Smin_sw = S
                tSmin = np.min(tStoring[:, :u + 1], 1)
                tSmin_sw = tS
                j = 0
                u += 1
        out[:, l] = Lambda_D
    return out


def noise_psd(Y):
    """audio_util.py:113-117."""
    return imcra_noise_psd(Y)


def sp_and_phase_speech(signal, power):
    """audio_util.py:422-437 -> (bandE [T,64] f32, mag [257,T] f32, phase [257,T] f32)."""
    F = stft(signal)
    mag = np.abs(F)
    phase = np.angle(F)
    bandE = compute_band_E(mag.T) ** power
    return bandE, mag, phase


def sp_and_phase_noise(signal, power):
    """audio_util.py:439-456."""
    F = stft(signal)
    psd = noise_psd(F).T
    bandE = compute_band_E(np.sqrt(psd)) ** power
    return bandE, np.abs(F), np.angle(F)
