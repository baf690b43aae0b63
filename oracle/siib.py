"""Oracle: SIIB^Gauss (Van Kuyk, Kleijn, Hendriks 2018) as called by reference intel.py:57-100
(``pysiib.SIIB(x, y, fs, gauss=True)`` after the wrapper's VAD-driven replication).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The wrapper part (framing, VAD, replication factor M) is the reference's own code and is PINNED
(oracle/intel.py vs tests/golden/intel.npz).  The SIIB core is PARITY UNPINNED: pysiib
(github kamo-naoyuki/pySIIB, no version pin, README.md:13) is not vendored or installed; this
restates the published algorithm:
  400/200 Hann power spectra of x and y, frames kept where the clean VAD is active (40 dB),
  28 ERB-spaced gammatone magnitude responses (100-6500 Hz) applied to the power spectra, log,
  forward temporal masking (Rhebergen 2006: each frame masks the next Tf = 16 frames with a level
  decaying linearly in log-time to the band minimum), mean removal, stacking of K = 15 consecutive
  frames (420 dims), KLT with the eigenvectors of cov(X), per-component correlation rho,
  I = -1/2 log2(1 - rho_p^2 rho^2) with rho_p = 0.75, SIIB = R/K * sum(I) clamped at 0.
Deliberate, documented deviation: components whose KLT eigenvalue is <= 1e-10 * the largest carry
no information (I = 0).  They only arise when the replicated signal is exactly frame-periodic
(L a multiple of 200: 170-260 of the 420 components; L a multiple of 100: a handful) and the
covariance is rank deficient; there the reference scores rounding noise that no second
implementation can reproduce.  Size of the effect on the bench utterances (DESIGN.md section 2,
tests/test_oracle_metrics.py): raw SIIB 1 % .. 23 % below the uncut value at L = 64 000,
2e-4 .. 2e-3 at L = 63 900, exactly zero at lengths that are no multiple of 100 (any real file).
"""
import numpy as np

from . import intel

EPS = np.finfo(np.float64).eps
FS = 16000
WLEN, WSHIFT = 400, 200
R = FS / WSHIFT
J_BANDS = 28
K_STACK = 15
RHO_P = 0.75
EIG_TOL = 1e-10


def gammatone_matrix(fs=FS, n_fft=WLEN, num_bands=J_BANDS, cf_min=100.0, cf_max=6500.0):
    erb = 21.4 * np.log10(4.37 * (np.array([cf_min, cf_max]) / 1000.0) + 1)
    cf_erb = np.linspace(erb[0], erb[1], num_bands)
    cf = (10 ** (cf_erb / 21.4) - 1) / 4.37 * 1000.0
    order = 4
    from math import factorial, pi
    a = factorial(order - 1) ** 2 / (pi * factorial(2 * order - 2) * 2.0 ** (-(2 * order - 2)))
    b = a * 24.7 * (4.37 * cf / 1000.0 + 1)
    f = np.linspace(0, fs, n_fft + 1)[:n_fft // 2 + 1]
    A = np.zeros((num_bands, len(f)))
    for i in range(num_bands):
        t = 1.0 / (b[i] ** 2 + (f - cf[i]) ** 2) ** (order / 2)
        A[i] = t / np.max(t)
    return A


def forward_masking(X, Tf):
    """In place, sequential over frames: frame i raises frames i..i+Tf-1 to
    X[j,i] - (X[j,i] - min_j) * log(tau)/log(Tf), tau = 1..Tf."""
    Jb, n = X.shape
    eX = X.min(axis=1)
    lt = np.log(np.arange(1, Tf + 1)) / np.log(Tf)
    for j in range(Jb):
        row = X[j]
        for i in range(n):
            m = min(Tf, n - i)
            fm = row[i] - (row[i] - eX[j]) * lt[:m]
            row[i:i + m] = np.maximum(row[i:i + m], fm)
    return X


def stack(X, K):
    Jb, n = X.shape
    cols = n - K + 1
    return np.concatenate([X[:, k:k + cols] for k in range(K)], axis=0)   # row index = k*J + j


def siib_gauss(x, y, fs=FS, return_parts=False):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    assert fs == FS and x.shape == y.shape and x.ndim == 1
    xh = intel.stft(x).T
    yh = intel.stft(y).T
    xh = xh.real ** 2 + xh.imag ** 2
    yh = yh.real ** 2 + yh.imag ** 2
    vad = intel.get_vad(x)
    xh, yh = xh[:, vad], yh[:, vad]
    G2 = gammatone_matrix() ** 2
    X = np.log(G2 @ xh + EPS)
    Y = np.log(G2 @ yh + EPS)
    Tf = int(np.floor(0.2 * R))
    X = forward_masking(X, Tf)
    Y = forward_masking(Y, Tf)
    X = X - X.mean(axis=1, keepdims=True)
    Y = Y - Y.mean(axis=1, keepdims=True)
    if X.shape[1] < K_STACK + 1:
        raise ValueError('SIIB: not enough active frames')
    Xs = stack(X, K_STACK)
    Ys = stack(Y, K_STACK)
    Xs = Xs - Xs.mean(axis=1, keepdims=True)
    Ys = Ys - Ys.mean(axis=1, keepdims=True)
    n = Xs.shape[1]
    Cxx = Xs @ Xs.T / (n - 1)
    lam, U = np.linalg.eigh(Cxx)
    Xp = U.T @ Xs
    Yp = U.T @ Ys
    vx = np.sum(Xp * Xp, axis=1)
    vy = np.sum(Yp * Yp, axis=1)
    cxy = np.sum(Xp * Yp, axis=1)
    good = lam > EIG_TOL * lam.max()
    rho = np.zeros_like(lam)
    rho[good] = cxy[good] / np.sqrt(vx[good] * vy[good])
    I = -0.5 * np.log2(1 - (RHO_P ** 2) * rho ** 2)
    val = max(0.0, float(R / K_STACK * np.sum(I)))
    if return_parts:
        return val, dict(lam=lam, rho=rho, n_active=int(vad.sum()), n_cols=n)
    return val


def siib_wrapper(x, y, fs=FS, norm=True):
    """intel.py:57-100: truncate to the common length, replicate M times when the active speech is
    shorter than 20 s, SIIB^Gauss, optional logistic map (intel.py:102-106)."""
    L = min(len(x), len(y))
    x, y = np.asarray(x[:L]), np.asarray(y[:L])
    M, _ = intel.siib_replication(x, fs)
    if M > 1:
        x = np.hstack([x] * M)
        y = np.hstack([y] * M)
    s = siib_gauss(x, y, fs)
    return float(intel.mapping_SIIB_harvard(s)) if norm else s
