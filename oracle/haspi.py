"""Oracle: HASPI v2 as computed by reference ``pyHASPI/pyhaspi2.py:haspi_v2`` (the call tree of
intel.py:108-114), normal-hearing case HL = 0.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PINNED at fs = 24 kHz against tests/golden/haspi.npz (made by importing the reference's pyhaspi2 with
numba.jit as the identity): centre frequencies, control / signal bandwidths, group-delay shifts,
sub-sampled envelopes, cepstral sequences (with the captured dither), the 10 modulation-band
correlations and the final score.  The 16 -> 24 kHz step (``librosa.resample`` = resampy
``kaiser_best``, pyhaspi2.py:815) is PARITY UNPINNED: resampy and its filter table are not
installed; the filter is regenerated from resampy's published recipe (sinc_window(num_zeros=64,
precision=9, rolloff=0.9475937167399596, kaiser beta=14.769656459379492)) and its sample loop restated.

Facts kept from the reference: the control filter bank uses the SAME centre
frequencies as the signal bank (the 0.02 basal shift is never applied, pyhaspi2.py:762, :1170);
group-delay compensation of BOTH envelopes uses BWx (:1239-1240); ebm_EnvFilt sub-samples by
int(24000 // 2560) = 9 while ebm_ModFilt assumes 2560 Hz; the dither of ebm_CepCoef is an input here.
"""
import ctypes
import os

import numpy as np
from scipy.signal import lfilter

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        p = os.path.join(_HERE, '_build', 'liboracle_loops.so')
        if not os.path.exists(p):
            import subprocess
            subprocess.check_call(['make', '-C', os.path.join(_HERE, 'csrc')], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(p)
        dp = ctypes.POINTER(ctypes.c_double)
        fp = ctypes.POINTER(ctypes.c_float)
        L.eb_cos_sin_cf.argtypes = [ctypes.c_long, ctypes.c_double, ctypes.c_double, dp, dp]
        L.eb_ihc_adapt.argtypes = [dp, ctypes.c_long, ctypes.c_double, ctypes.c_double, dp]
        L.resample_f32.argtypes = [fp, ctypes.c_long, fp, ctypes.c_long, ctypes.c_double, dp, dp, ctypes.c_long, ctypes.c_long]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


NCHAN = 32
FSAMP = 24000
LEVEL1 = 65.0


def center_freq(nchan=NCHAN):
    """pyhaspi2.py:753-777 (shift never applied)."""
    lowFreq, highFreq = 80.0, 8000.0
    EarQ, minBW = 9.26449, 24.7
    cf = -(EarQ * minBW) + np.exp(np.arange(1, nchan) * (-np.log(highFreq + EarQ * minBW) + np.log(lowFreq + EarQ * minBW)) / (nchan - 1)) \
        * (highFreq + EarQ * minBW)
    cf = np.concatenate((np.array([highFreq]), cf))
    return np.flipud(cf)


def loss_parameters(HL, cfreq):
    """pyhaspi2.py:779-807."""
    aud = [250.0, 500.0, 1000.0, 2000.0, 4000.0, 6000.0]
    nfilt = len(cfreq)
    fv = [cfreq[0]] + aud + [cfreq[-1]]
    loss = np.interp(cfreq, fv, np.concatenate((np.array([HL[0]]), HL, np.array([HL[-1]]))))
    loss[loss < 0] = 0.0
    CR = 1.25 + 2.25 * np.arange(nfilt) / (nfilt - 1)
    maxOHC = 70 * (1 - (1 / CR))
    thrOHC = 1.25 * maxOHC
    attnOHC = np.where(loss < thrOHC, 0.8 * loss, 0.8 * thrOHC)
    attnIHC = np.where(loss < thrOHC, 0.2 * loss, 0.2 * thrOHC + (loss - thrOHC))
    BW = np.ones(nfilt) + (attnOHC / 50.0) + 2.0 * (attnOHC / 50.0) ** 6
    lowknee = attnOHC + 30
    upamp = 30 + 70 / CR
    CR = (100 - lowknee) / (upamp + attnOHC - lowknee)
    return attnOHC, BW, lowknee, CR, attnIHC


def middle_ear(x):
    """pyhaspi2.py:833-841."""
    y = lfilter(np.array([0.434173751206302, 0.434173751206302]), np.array([1.0, -0.131652497587396]), x)
    return lfilter(np.array([0.937260390269893, -1.874520780539785, 0.937260390269893]),
                   np.array([1.0, -1.870580640735279, 0.878460920344291]), y)


def gammatone_coeffs(BW, cf, fs=FSAMP):
    earQ, minBW = 9.26449, 24.7
    ERB = minBW + (cf / earQ)
    tpt = 2 * np.pi / fs
    tptBW = BW * tpt * ERB * 1.019
    a = np.exp(-tptBW)
    a1, a2, a3, a4, a5 = 4.0 * a, -6.0 * a * a, 4.0 * a * a * a, -a * a * a * a, 4.0 * a * a
    gain = 2.0 * (1 - a1 - a2 - a3 - a4) / (1 + a1 + a5)
    return a, a1, a2, a3, a4, a5, gain


def cos_sin_cf(npts, fs, cf):
    c = np.empty(npts)
    s = np.empty(npts)
    _lib().eb_cos_sin_cf(npts, float(fs), float(cf), _dp(c), _dp(s))
    return c, s


def gammatone_env(x, BW, coscf, sincf, cf):
    """pyhaspi2.py:917-968 (one signal)."""
    a, a1, a2, a3, a4, a5, gain = gammatone_coeffs(BW, cf)
    b = [1, a1, a5]
    aa = [1, -a1, -a2, -a3, -a4]
    ureal = lfilter(b, aa, x * coscf)
    uimag = lfilter(b, aa, x * sincf)
    return gain * np.sqrt(ureal * ureal + uimag * uimag)


def bw_adjust(control, BWmin, BWmax, Level1=LEVEL1):
    """pyhaspi2.py:971-980."""
    cRMS = np.sqrt(np.mean(control ** 2))
    cdB = 20 * np.log10(cRMS) + Level1
    if cdB < 50:
        return BWmin
    if cdB > 100:
        return BWmax
    return BWmin + ((cdB - 50) / 50) * (BWmax - BWmin)


def env_compress(envsig, control, attnOHC, thrLow, CR, Level1=LEVEL1):
    """pyhaspi2.py:982-999 (envelope only)."""
    logenv = np.clip(control, a_min=1.0e-30, a_max=None)
    logenv = Level1 + 20 * np.log10(logenv)
    logenv = np.clip(logenv, a_min=thrLow, a_max=100.0)
    gain = -attnOHC - (logenv - thrLow) * (1 - (1 / CR))
    gain = np.power(10, (gain / 20))
    gain = lfilter([0.095107983402496, 0.095107983402496], [1.0, -0.809784033195007], gain)
    return gain * envsig


def env_sl2(env, attnIHC, Level1=LEVEL1):
    """pyhaspi2.py:1080-1088 (envelope only)."""
    y = Level1 - attnIHC + 20 * np.log10(env + 1.0e-30)
    y[y < 0] = 0.0
    return y


def ihc_adapt(xdB, delta=2.0, fsamp=FSAMP):
    y = np.empty_like(xdB)
    _lib().eb_ihc_adapt(_dp(np.ascontiguousarray(xdB)), len(xdB), float(delta), float(fsamp), _dp(y))
    return y


def group_delay_shifts(BW, cfreq, fsamp=FSAMP):
    """pyhaspi2.py:1098-1131: group delay at omega = 0 of [1,a1,a5]/[1,-4a,6a^2,-4a^3,a^4], closed form
    (a1 + 2 a5)/(1 + a1 + a5) + 4a/(1 - a) (equals scipy.signal.group_delay(..., w=1) to 1e-7), rounded."""
    a, a1, a2, a3, a4, a5, _ = gammatone_coeffs(BW, cfreq, fsamp)
    gd = (a1 + 2 * a5) / (1 + a1 + a5) + 4 * a / (1 - a)
    gd = np.round(gd)
    gd = gd - np.min(gd)
    return (np.max(gd) - gd).astype(np.int64)


def resample_filter():
    """resampy 'kaiser_best': sinc_window(num_zeros=64, precision=9, rolloff=0.9475937167399596) with a
    Kaiser(beta=14.769656459379492) taper -> (half window [32769], num_table = 512)."""
    num_zeros, precision, rolloff, beta = 64, 9, 0.9475937167399596, 14.769656459379492
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def resample_24k(x, fsampx):
    """pyhaspi2.py:810-821."""
    if fsampx == FSAMP:
        return x
    assert fsampx == 16000
    x = np.ascontiguousarray(x, dtype=np.float32)           # x / rms_x is float32 in the reference
    ratio = float(FSAMP) / fsampx
    n_out = int(x.shape[0] * ratio)
    win, num_table = resample_filter()
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    y = np.zeros(n_out, dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    _lib().resample_f32(x.ctypes.data_as(fp), x.shape[0], y.ctypes.data_as(fp), n_out, ratio, _dp(win), _dp(delta), win.shape[0], num_table)
    xRMS = np.sqrt(np.mean(x ** 2))
    yRMS = np.sqrt(np.mean(y ** 2))
    return (xRMS / yRMS) * y


def ear_model(x, fx, y, fy):
    """pyhaspi2.py:1155-1248 for HL = 0, itype = 0: -> (xdB, ydB [32, nsamp], parts)."""
    HL = np.zeros(6)
    cfreq = center_freq()
    attnOHC, BWmin, lowknee, CR, attnIHC = loss_parameters(HL, cfreq)
    _, BW1, _, _, _ = loss_parameters(100 * np.ones(6), cfreq)
    x24 = resample_24k(x, fx)
    y24 = resample_24k(y, fy)
    nsamp = len(x24)
    xmid = middle_ear(x24)
    ymid = middle_ear(y24)
    xdB = np.zeros((NCHAN, nsamp))
    ydB = np.zeros((NCHAN, nsamp))
    BWx = np.zeros(NCHAN)
    BWy = np.zeros(NCHAN)
    for n in range(NCHAN):
        coscf, sincf = cos_sin_cf(nsamp, FSAMP, cfreq[n])
        xcontrol = gammatone_env(xmid, BW1[n], coscf, sincf, cfreq[n])
        ycontrol = gammatone_env(ymid, BW1[n], coscf, sincf, cfreq[n])
        BWx[n] = bw_adjust(xcontrol, BWmin[n], BW1[n])
        BWy[n] = bw_adjust(ycontrol, BWmin[n], BW1[n])
        xenv = gammatone_env(xmid, BWx[n], coscf, sincf, cfreq[n])
        yenv = gammatone_env(ymid, BWy[n], coscf, sincf, cfreq[n])
        xc = env_compress(xenv, xcontrol, attnOHC[n], lowknee[n], CR[n])
        yc = env_compress(yenv, ycontrol, attnOHC[n], lowknee[n], CR[n])
        xc = env_sl2(xc, attnIHC[n])
        yc = env_sl2(yc, attnIHC[n])
        xdB[n] = ihc_adapt(xc)
        ydB[n] = ihc_adapt(yc)
    shifts = group_delay_shifts(BWx, cfreq)
    for arr in (xdB, ydB):                                  # both use BWx (pyhaspi2.py:1239-1240)
        for n in range(NCHAN):
            s = int(shifts[n])
            if s > 0:
                arr[n] = np.concatenate((np.zeros(s), arr[n, :nsamp - s]))
    return xdB, ydB, dict(cfreq=cfreq, BW1=BW1, BWx=BWx, BWy=BWy, shifts=shifts)


def env_filt(xdB, ydB, fcut=320, fsub=2560, fsamp=FSAMP):
    """pyhaspi2.py:378-414: input [32, nsamp] -> [ceil(nsamp/9), 32]."""
    xdB, ydB = xdB.T, ydB.T
    nsamp = xdB.shape[0]
    tfilt = 0.7 * (1000 * (1 / fcut))
    nfilt = round(0.001 * tfilt * fsamp)                    # round(52.5) = 52 (banker's)
    nhalf = int(nfilt // 2)
    nfilt = 2 * nhalf
    benv = np.hanning(nfilt)
    benv = benv / np.sum(benv)
    space = int(fsamp // fsub)
    index = np.arange(0, nsamp, space)
    # only the sub-sampled outputs are needed: out[i] = sum_k benv[k] x[i + nhalf - k]
    xp = np.concatenate((np.zeros((nfilt, xdB.shape[1])), xdB, np.zeros((nfilt, xdB.shape[1]))))
    yp = np.concatenate((np.zeros((nfilt, ydB.shape[1])), ydB, np.zeros((nfilt, ydB.shape[1]))))
    xLP = np.zeros((len(index), xdB.shape[1]))
    yLP = np.zeros((len(index), ydB.shape[1]))
    for k in range(nfilt):
        src = index + nhalf - k + nfilt
        xLP += benv[k] * xp[src]
        yLP += benv[k] * yp[src]
    return xLP, yLP


def cep_coef(xdB, ydB, thrCep=2.5, thrNerve=0.1, nbasis=6, dither_x=None, dither_y=None):
    """pyhaspi2.py:342-375; dither arrays (standard normal, [n_active, 32]) are inputs."""
    nbands = xdB.shape[1]
    k = np.arange(0, nbands)
    cepm = np.zeros([nbands, nbasis])
    for nb in range(nbasis):
        basis = np.cos(nb * np.pi * k / (nbands - 1))
        cepm[:, nb] = basis / np.linalg.norm(basis)
    xLinear = np.power(10, (xdB / 20))
    xsum = np.sum(xLinear, axis=1) / nbands
    xsum = 20 * np.log10(xsum)
    index = np.where(xsum > thrCep)[0]
    if len(index) <= 1:
        raise Exception('Function ebm_CepCoef: Signal below threshold')
    xdB = xdB[index, :]
    ydB = ydB[index, :]
    if dither_x is not None:
        xdB = xdB + thrNerve * dither_x
        ydB = ydB + thrNerve * dither_y
    xcep = np.matmul(xdB, cepm)
    ycep = np.matmul(ydB, cepm)
    xcep = xcep - np.mean(xcep, axis=0, keepdims=True)
    ycep = ycep - np.mean(ycep, axis=0, keepdims=True)
    return xcep, ycep, index


MOD_CF = np.array([2, 6, 10, 16, 25, 40, 64, 100, 160, 256])


def mod_filters(fsub=2560):
    """pyhaspi2.py:275-305: FIR windows (np.hanning(nfir+1) normalised) and half lengths."""
    cf = MOD_CF
    nmod = len(cf)
    t0 = 0.24
    t = np.zeros(nmod)
    t[0] = t0
    t[1] = t0
    t[2:nmod] = t0 * cf[2] / cf[2:nmod]
    nfir = 2 * np.floor(t * fsub / 2)
    b = []
    for k in range(nmod):
        w = np.hanning(int(nfir[k]) + 1)
        b.append(w / np.sum(w))
    return b, (nfir / 2).astype(int)


def mod_filt(Xenv, Yenv, fsub=2560):
    """pyhaspi2.py:275-339 -> Xmod[basis][band] arrays."""
    nsamp, nchan = Xenv.shape
    b, nhalf = mod_filters(fsub)
    fNyq = 0.5 * fsub
    n = np.arange(1, nsamp + 1)
    out = []
    for E in (Xenv, Yenv):
        mod = [[None] * len(MOD_CF) for _ in range(nchan)]
        for k in range(len(MOD_CF)):
            if k == 0:
                c, s = 1.0, 0.0
            else:
                c = np.sqrt(2) * np.cos(np.pi * n * MOD_CF[k] / fNyq)
                s = np.sqrt(2) * np.sin(np.pi * n * MOD_CF[k] / fNyq)
            for m in range(nchan):
                x = E[:, m]
                u = np.convolve((x * c - 1j * x * s), b[k])
                u = u[nhalf[k]:nhalf[k] + nsamp]
                mod[m][k] = np.real(u) * c - np.imag(u) * s
        out.append(mod)
    return out[0], out[1]


def mod_corr(Xmod, Ymod):
    """pyhaspi2.py:254-273."""
    nchan, nmod = len(Xmod), len(Xmod[0])
    small = 1.0e-30
    CM = np.zeros([nchan, nmod])
    for m in range(nmod):
        for j in range(nchan):
            xj = Xmod[j][m] - np.mean(Xmod[j][m])
            yj = Ymod[j][m] - np.mean(Ymod[j][m])
            xsum, ysum = np.sum(xj ** 2), np.sum(yj ** 2)
            CM[j, m] = 0 if (xsum < small or ysum < small) else np.abs(np.sum(xj * yj)) / np.sqrt(xsum * ysum)
    return np.mean(CM[1:6], axis=0)


WEIGHTS = np.array([1.361, 1.521, 1.164, 0.492, 0.436, 0.690, 1.142, 0.816, 1.576, 2.269])


def haspi_v2(x, fx, y, fy, dither_x=None, dither_y=None, return_parts=False):
    """pyhaspi2.py:76-107.  dither_* = None -> no dither (deterministic); else standard-normal arrays
    [n_active, 32] as np.random.randn would have produced inside ebm_CepCoef."""
    L = min(len(x), len(y))
    x = x[:L]
    y = y[:L]
    rms_x = np.sqrt(np.sum(x ** 2) / L)
    rms_y = np.sqrt(np.sum(y ** 2) / L)
    x = x / rms_x
    y = y / rms_y
    xdB, ydB, parts = ear_model(x, fx, y, fy)
    xLP, yLP = env_filt(xdB, ydB)
    xcep, ycep, index = cep_coef(xLP, yLP, dither_x=dither_x, dither_y=dither_y)
    xmod, ymod = mod_filt(xcep, ycep)
    aveCM = mod_corr(xmod, ymod)
    intel = float(np.sum(WEIGHTS * aveCM))
    if return_parts:
        parts.update(xLP=xLP, yLP=yLP, index=index, xcep=xcep, ycep=ycep, aveCM=aveCM)
        return intel, parts
    return intel, aveCM


def haspi_wrapper(x, y, fs=16000, norm=True, dither_x=None, dither_y=None):
    """intel.py:108-120."""
    from .intel import mapping_HASPI_harvard
    s, _ = haspi_v2(x, fs, y, fs, dither_x=dither_x, dither_y=dither_y)
    return float(mapping_HASPI_harvard(s)) if norm else s
