"""Oracle: HASPI v2 as computed by reference ``pyHASPI/pyhaspi2.py:haspi_v2`` (the call tree of
intel.py:108-114), normal hearing (HL = 0, what the loop uses) and hearing-loss audiograms (HL, itype 0 / 2).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

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
    if fsampx > FSAMP:
        raise NotImplementedError                            # pyhaspi2.py:819-820
    x = np.ascontiguousarray(x, dtype=np.float32)           # x / rms_x is float32 in the reference
    ratio = float(FSAMP) / fsampx
    n_out = int(x.shape[0] * ratio)                          # resampy.resample: int(shape * sample_ratio) outputs ...
    n_fix = int(np.ceil(x.shape[0] * ratio))                 # ... librosa.resample(fix=True): zero-padded to ceil(n * ratio)
    win, num_table = resample_filter()
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    y = np.zeros(n_fix, dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    _lib().resample_f32(x.ctypes.data_as(fp), x.shape[0], y.ctypes.data_as(fp), n_out, ratio, _dp(win), _dp(delta), win.shape[0], num_table)
    xRMS = np.sqrt(np.mean(x ** 2))
    yRMS = np.sqrt(np.mean(y ** 2))
    return (xRMS / yRMS) * y


def gammatone_bm(x, BW, coscf, sincf, cf):
    """pyhaspi2.py:863-915 (one signal): -> (envelope, basilar-membrane motion)."""
    a, a1, a2, a3, a4, a5, gain = gammatone_coeffs(BW, cf)
    b = [1, a1, a5]
    aa = [1, -a1, -a2, -a3, -a4]
    ureal = lfilter(b, aa, x * coscf)
    uimag = lfilter(b, aa, x * sincf)
    return gain * np.sqrt(ureal * ureal + uimag * uimag), gain * (ureal * coscf + uimag * sincf)


def compress_gain(control, attnOHC, thrLow, CR, Level1=LEVEL1):
    """pyhaspi2.py:982-995: the low-passed compression gain both the envelope and the BM motion are multiplied by."""
    logenv = np.clip(control, a_min=1.0e-30, a_max=None)
    logenv = Level1 + 20 * np.log10(logenv)
    logenv = np.clip(logenv, a_min=thrLow, a_max=100.0)
    gain = -attnOHC - (logenv - thrLow) * (1 - (1 / CR))
    gain = np.power(10, (gain / 20))
    return lfilter([0.095107983402496, 0.095107983402496], [1.0, -0.809784033195007], gain)


def ave_sl(env, control, attnOHC, thrLow, CR, attnIHC, Level1=LEVEL1):
    """pyhaspi2.py:1135-1152: average band levels (RMS of the signal / control envelopes, [32]) -> dB SL."""
    small = 1.0e-30
    logenv = Level1 + 20 * np.log10(np.clip(control, a_min=small, a_max=None))
    logenv = np.clip(logenv, a_min=thrLow, a_max=100.0)
    gain = -attnOHC - (logenv - thrLow) * (1 - (1 / CR))
    logenv = Level1 + 20 * np.log10(np.clip(env, a_min=small, a_max=None))
    logenv = np.clip(logenv, a_min=0, a_max=None)
    return np.clip(logenv + gain - attnIHC, a_min=0.0, a_max=None)


def _loss_pair(HL, itype):
    """pyhaspi2.py:1160-1166: the processed signal is heard with the audiogram HL, the reference with HLx = 0 * HL for the
    intelligibility model (itype 0) and with HL otherwise (hasqi_v2 passes 2).  itype 1 would need eb_NALR, which raises in the
    reference.  -> (per-signal tuple of attnOHC, BWmin, lowknee, CR, attnIHC)."""
    if itype == 1:
        raise NotImplementedError('eb_NALR (pyhaspi2.py:830-831)')
    HL = np.zeros(6) if HL is None else np.asarray(HL, dtype=np.float64)
    cfreq = center_freq()
    return loss_parameters(0 * HL if itype == 0 else HL, cfreq), loss_parameters(HL, cfreq)


def ear_model_bm(x, fx, y, fy, noise_x=None, noise_y=None, HL=None, itype=0):
    """pyhaspi2.py:1155-1248 (for HL = 0 itype 0 and 2 coincide) with the basilar-membrane outputs:
    -> (xdB, xBM, ydB, yBM [32, nsamp], xSL, ySL [32], parts).  noise_* [32, nsamp]: the standard-normal draws of eb_BMaddnoise
    (pyhaspi2.py:1091-1095; the reference draws channel by channel, x before y), None = no noise."""
    small = 1.0e-30
    cfreq = center_freq()
    lossp = _loss_pair(HL, itype)
    _, BW1, _, _, _ = loss_parameters(100 * np.ones(6), cfreq)
    x24 = resample_24k(x, fx)
    y24 = resample_24k(y, fy)
    nsamp = len(x24)
    mid = (middle_ear(x24), middle_ear(y24))
    dB = np.zeros((2, NCHAN, nsamp))
    BM = np.zeros((2, NCHAN, nsamp))
    ave = np.zeros((2, NCHAN))
    cave = np.zeros((2, NCHAN))
    BW = np.zeros((2, NCHAN))
    gn = 10 ** ((-10.0 - LEVEL1) / 20.0)                    # IHCthr = -10 (pyhaspi2.py:1229)
    noise = (noise_x, noise_y)
    for n in range(NCHAN):
        coscf, sincf = cos_sin_cf(nsamp, FSAMP, cfreq[n])
        control = [gammatone_env(m, BW1[n], coscf, sincf, cfreq[n]) for m in mid]
        for s in range(2):
            attnOHC, BWmin, lowknee, CR, attnIHC = lossp[s]
            BW[s, n] = bw_adjust(control[s], BWmin[n], BW1[n])
            env, bm = gammatone_bm(mid[s], BW[s, n], coscf, sincf, cfreq[n])
            ave[s, n] = np.sqrt(np.mean(env ** 2))
            cave[s, n] = np.sqrt(np.mean(control[s] ** 2))
            g = compress_gain(control[s], attnOHC[n], lowknee[n], CR[n])
            c, b = g * env, g * bm
            sl = env_sl2(c, attnIHC[n])
            b = ((sl + small) / (c + small)) * b            # eb_EnvSL2: the gain that took the envelope to dB SL, on the BM motion
            out = ihc_adapt(sl)
            b = ((out + small) / (sl + small)) * b          # eb_IHCadapt: the adaptation gain on the BM motion
            if noise[s] is not None:
                b = b + gn * noise[s][n]
            dB[s, n], BM[s, n] = out, b
    shifts = group_delay_shifts(BW[0], cfreq)
    for arr in (dB[0], dB[1], BM[0], BM[1]):                # all four use BWx (pyhaspi2.py:1239-1242)
        for n in range(NCHAN):
            sft = int(shifts[n])
            if sft > 0:
                arr[n] = np.concatenate((np.zeros(sft), arr[n, :nsamp - sft]))
    SL = []
    for s in range(2):
        attnOHC, BWmin, lowknee, CR, attnIHC = lossp[s]
        SL.append(ave_sl(ave[s], cave[s], attnOHC, lowknee, CR, attnIHC))
    xSL, ySL = SL
    return dB[0], BM[0], dB[1], BM[1], xSL, ySL, dict(cfreq=cfreq, BW1=BW1, BWx=BW[0], BWy=BW[1], shifts=shifts)


def ear_model(x, fx, y, fy, HL=None, itype=0):
    """pyhaspi2.py:1155-1248 (envelope outputs only): -> (xdB, ydB [32, nsamp], parts)."""
    cfreq = center_freq()
    (aOx, BWminx, lkx, CRx, aIx), (aOy, BWminy, lky, CRy, aIy) = _loss_pair(HL, itype)
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
        BWx[n] = bw_adjust(xcontrol, BWminx[n], BW1[n])
        BWy[n] = bw_adjust(ycontrol, BWminy[n], BW1[n])
        xenv = gammatone_env(xmid, BWx[n], coscf, sincf, cfreq[n])
        yenv = gammatone_env(ymid, BWy[n], coscf, sincf, cfreq[n])
        xc = env_compress(xenv, xcontrol, aOx[n], lkx[n], CRx[n])
        yc = env_compress(yenv, ycontrol, aOy[n], lky[n], CRy[n])
        xc = env_sl2(xc, aIx[n])
        yc = env_sl2(yc, aIy[n])
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


def _mix64(z):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    z = (z + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def dither_rows(utt_id, seed, nsub):
    """The per-utterance dither draws of the HIP path (csrc/haspi.hip: haspi_dither_rows_kernel) restated in numpy: standard normals
    [2, nsub, 32] (x rows, y rows) as a pure function of (seed, utterance id, signal, frame, channel).  The REFERENCE draws
    np.random.randn(n_active, 32) from numpy's global generator on every call (pyhaspi2.py:362-365); which normals are drawn is not
    part of its contract, that they are i.i.d. N(0, 1) per (active frame, channel) is."""
    with np.errstate(over='ignore'):
        key = _mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ _mix64(np.array([utt_id & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)))[0]
        e = np.arange(nsub * NCHAN, dtype=np.uint64)
        out = np.empty((2, nsub, NCHAN))
        for sig in range(2):
            idx = (np.uint64(sig) << np.uint64(40)) | e
            r1 = _mix64(key ^ _mix64(np.uint64(2) * idx))
            r2 = _mix64(key ^ _mix64(np.uint64(2) * idx + np.uint64(1)))
            u1 = ((r1 >> np.uint64(11)).astype(np.float64) + 1.0) * 2.0 ** -53
            u2 = (r2 >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
            out[sig] = (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).reshape(nsub, NCHAN)
    return out


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


def haspi_v2(x, fx, y, fy, dither_x=None, dither_y=None, return_parts=False, HL=None):
    """pyhaspi2.py:76-107.  dither_* = None -> no dither (deterministic); else standard-normal arrays
    [n_active, 32] as np.random.randn would have produced inside ebm_CepCoef."""
    L = min(len(x), len(y))
    x = x[:L]
    y = y[:L]
    rms_x = np.sqrt(np.sum(x ** 2) / L)
    rms_y = np.sqrt(np.sum(y ** 2) / L)
    x = x / rms_x
    y = y / rms_y
    xdB, ydB, parts = ear_model(x, fx, y, fy, HL=HL, itype=0)
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


# ------------------------------------------------------------------------------------------------------------------------------
# HASPI (version 1) and HASQI v2: the remaining entry points of pyhaspi2.py.  Same ear model, plus the basilar-
# membrane outputs; 16 ms raised-cosine segments at 50 % overlap (125 Hz) instead of the 320 Hz envelope filter.
def _seg_window(segsize, fsamp=FSAMP):
    nwin = round(segsize * (0.001 * fsamp))
    if nwin % 2:
        nwin += 1
    return nwin, np.hanning(nwin)


def n_segments(npts, nwin):
    return int(1 + np.floor(npts / nwin) + np.floor((npts - nwin / 2) / nwin))


def env_smooth(env, segsize=16, fsamp=FSAMP):
    """pyhaspi2.py:674-703: [32, npts] -> [32, nseg]; half windows at both ends."""
    nwin, window = _seg_window(segsize, fsamp)
    nhalf = nwin // 2
    npts = env.shape[1]
    nseg = n_segments(npts, nwin)
    out = np.zeros((env.shape[0], nseg))
    out[:, 0] = env[:, :nhalf] @ window[nhalf:] / np.sum(window[nhalf:])
    for n in range(1, nseg - 1):
        out[:, n] = env[:, n * nhalf:n * nhalf + nwin] @ window / np.sum(window)
    st = (nseg - 1) * nhalf
    out[:, nseg - 1] = env[:, st:st + nhalf] @ window[:nhalf] / np.sum(window[nhalf:])
    return out


def _loud_index(xdB_like, thr):
    """Segments whose band-averaged linear level, back in dB, exceeds thr (pyhaspi2.py:717-721, :422-427, :172-176)."""
    xsum = 20 * np.log10(np.sum(np.power(10, xdB_like / 20), axis=0) / xdB_like.shape[0])
    return xsum, np.where(xsum > thr)[0]


def melcor(x, y, thr=2.5):
    """pyhaspi2.py:706-751 with addnoise = 0: -> (mean of |corr| of cepstral coefficients 2-6, all six)."""
    nbands, nbasis = x.shape[0], 6
    k = np.arange(nbands)
    cepm = np.stack([np.cos(nb * np.pi * k / (nbands - 1)) for nb in range(nbasis)], axis=1)
    cepm = cepm / np.linalg.norm(cepm, axis=0, keepdims=True)
    _, index = _loud_index(x, thr)
    if len(index) <= 1:
        raise Exception('Function eb_melcor: Signal below threshold, outputs set to 0.')
    xcep = cepm.T @ x[:, index]
    ycep = cepm.T @ y[:, index]
    xcep = xcep - xcep.mean(axis=1, keepdims=True)
    ycep = ycep - ycep.mean(axis=1, keepdims=True)
    xs, ys = np.sum(xcep ** 2, axis=1), np.sum(ycep ** 2, axis=1)
    xy = np.zeros(nbasis)
    ok = (xs >= 1.0e-30) & (ys >= 1.0e-30)
    xy[ok] = np.abs(np.sum(xcep * ycep, axis=1)[ok] / np.sqrt(xs[ok] * ys[ok]))
    return float(np.sum(xy[1:]) / (nbasis - 1)), xy


def window_corr(window, maxlag):
    """1 / xcorr(window, window, maxlag): the normalisation tables pyhaspi2.py:563 / :570 keep as literals (MATLAB output)."""
    n = len(window)
    full = np.correlate(window, window, 'full')
    return 1.0 / full[n - 1 - maxlag:n + maxlag]


def bm_covary(xBM, yBM, segsize=16, fsamp=FSAMP):
    """pyhaspi2.py:550-657: per band and segment, max over lags |l| <= 1 ms of the normalised cross-covariance of the windowed,
    mean-removed BM motion, and the mean-square levels.  -> sigcov, sigMSx, sigMSy [32, nseg]."""
    small = 1.0e-30
    maxlag = round(1.0 * (0.001 * fsamp))
    nwin, window = _seg_window(segsize, fsamp)
    nhalf = nwin // 2
    wincorr = window_corr(window, maxlag)
    halfcorr = window_corr(window[nhalf:], maxlag)
    winsum2 = 1.0 / np.sum(window ** 2)
    halfsum2 = 1.0 / np.sum(window[nhalf:] ** 2)
    nchan, npts = xBM.shape
    nseg = n_segments(npts, nwin)
    sigcov = np.zeros((nchan, nseg))
    sigMSx = np.zeros((nchan, nseg))
    sigMSy = np.zeros((nchan, nseg))
    lags = np.arange(-maxlag, maxlag + 1)

    def one(segx, segy, corr, norm2):
        segx = segx - segx.mean(axis=1, keepdims=True)
        segy = segy - segy.mean(axis=1, keepdims=True)
        MSx = np.sum(segx ** 2, axis=1) * norm2
        MSy = np.sum(segy ** 2, axis=1) * norm2
        N = segx.shape[1]
        c = np.zeros((nchan, len(lags)))
        for i, l in enumerate(lags):                        # np.correlate(segx, segy, 'full')[N - 1 + l] = sum_n segx[n + l] segy[n]
            if l >= 0:
                c[:, i] = np.sum(segx[:, l:] * segy[:, :N - l], axis=1)
            else:
                c[:, i] = np.sum(segx[:, :N + l] * segy[:, -l:], axis=1)
        Mxy = np.max(np.abs(c * corr), axis=1)
        ok = (MSx > small) & (MSy > small)
        cov = np.zeros(nchan)
        cov[ok] = Mxy[ok] / np.sqrt(MSx[ok] * MSy[ok])
        return cov, MSx, MSy

    sigcov[:, 0], sigMSx[:, 0], sigMSy[:, 0] = one(xBM[:, :nhalf] * window[nhalf:], yBM[:, :nhalf] * window[nhalf:], halfcorr, halfsum2)
    for n in range(1, nseg - 1):
        st = n * nhalf
        sigcov[:, n], sigMSx[:, n], sigMSy[:, n] = one(xBM[:, st:st + nwin] * window, yBM[:, st:st + nwin] * window, wincorr, winsum2)
    st = (nseg - 1) * nhalf
    sigcov[:, -1], sigMSx[:, -1], sigMSy[:, -1] = one(xBM[:, st:st + nhalf] * window[:nhalf], yBM[:, st:st + nhalf] * window[:nhalf], halfcorr, halfsum2)
    return np.clip(sigcov, 0, 1), 2.0 * sigMSx, 2.0 * sigMSy


SII_CF = [150, 250, 350, 450, 570, 700, 840, 1000, 1170, 1370, 1600, 1850, 2150, 2500, 2900, 3400, 4000, 4800, 5800, 7000, 8500]
SII_WGT = [.0103, .0261, .0419, .0577, .0577, .0577, .0577, .0577, .0577, .0577, .0577, .0577, .0577, .0577, .0577, .0577, .0577, .0460,
           .0343, .0226, .0110]


def three_level_covary(sigcov, sigMSx, thr=2.5):
    """pyhaspi2.py:416-547: the above-threshold segments are split into thirds of the cumulative 0.5 dB histogram of their
    loudness; per third, the band-average of the per-band mean covariance.  -> cov3, covSII [low, mid, high]."""
    from scipy.interpolate import interp1d
    nbands = sigcov.shape[0]
    sigRMS = np.sqrt(sigMSx)
    xsum, index = _loud_index(sigRMS, thr)
    if len(index) <= 1:
        raise Exception('Function eb_3LevelCovary: Signal below threshold, outputs set to 0.')
    cfreq = center_freq(nbands)
    wfreq = interp1d(np.array([0] + SII_CF + [FSAMP]), np.array([0] + SII_WGT + [0]), kind='cubic')(cfreq)
    wfreq[0] = wfreq[1] = 0.0
    wfreq = wfreq / np.sum(wfreq)
    sigcov, sigRMS, xsum = sigcov[:, index], sigRMS[:, index], xsum[index]
    bins = np.arange(np.min(xsum), np.max(xsum) + 0.5, 0.5)
    # np.histogram over the mid-points between bin centres = nearest-centre assignment, open at both ends
    edges = np.concatenate(([-1e8], (bins + np.concatenate((bins[1:], [1e8]))) / 2))
    xhist, _ = np.histogram(xsum, edges)
    xcum = np.cumsum(xhist.astype(np.float64))
    xcum = xcum / xcum[-1]
    edge = np.zeros(2)
    for n in range(len(bins)):
        if xcum[n] < 0.333:
            edge[0] = bins[n]
        if xcum[n] < 0.667:
            edge[1] = bins[n]
    groups = (np.where(xsum < edge[0])[0], np.where((xsum >= edge[0]) & (xsum < edge[1]))[0], np.where(xsum >= edge[1])[0])
    weight = (sigRMS > thr).astype(np.float64)
    sigcov = weight * sigcov
    cov3, covSII = np.zeros(3), np.zeros(3)
    with np.errstate(invalid='ignore', divide='ignore'):
        for g, idx in enumerate(groups):
            ssum, wsum = sigcov[:, idx].sum(axis=1), weight[:, idx].sum(axis=1)
            ok = wsum != 0
            ave = np.zeros(nbands)
            ave[ok] = ssum[ok] / wsum[ok]
            cov3[g] = np.float64(np.sum(ave)) / np.float64(np.count_nonzero(ok))
            covSII[g] = np.float64(np.sum(ave * wfreq * ok)) / np.float64(np.sum(wfreq[ok]))
    return cov3, covSII


def ave_covary2(sigcov, sigMSx, thr=2.5):
    """pyhaspi2.py:160-220: average covariance over the above-threshold time-frequency cells, plain and with six low-pass
    band weightings ("synchrony" up to 1.5 ... 4 kHz)."""
    nchan = sigcov.shape[0]
    cfreq = center_freq(nchan)
    p = np.array([1, 3, 5, 5, 5, 5])
    fcut = 1000 * np.array([1.5, 2.0, 2.5, 3.0, 3.5, 4.0])
    fsync = np.stack([np.sqrt(fcut[n] ** (2 * p[n]) / (fcut[n] ** (2 * p[n]) + cfreq ** (2 * p[n]))) for n in range(6)])
    sigRMS = np.sqrt(sigMSx)
    _, index = _loud_index(sigRMS, thr)
    if len(index) <= 1:
        return 0, 0
    sigcov, sigRMS = sigcov[:, index], sigRMS[:, index]
    weight = (sigRMS > thr).astype(np.float64)
    wsum = np.sum(weight)
    with np.errstate(invalid='ignore', divide='ignore'):
        syncov = np.array([np.sum(fsync[n][:, None] * weight * sigcov) / np.sum(fsync[n][:, None] * weight) for n in range(6)])
    return (0 if wsum < 1 else np.sum(weight * sigcov) / wsum), syncov


def spect_diff(xSL, ySL):
    """pyhaspi2.py:222-251: differences of the normalised long-term spectra and of their slopes: (sum |d|, nbands std d, max |d|)."""
    nbands = len(xSL)
    x = 10 ** (xSL / 20)
    y = 10 ** (ySL / 20)
    x = x / np.sum(x)
    y = y / np.sum(y)

    def three(d):
        return np.array([np.sum(np.abs(d)), nbands * np.std(d), np.max(np.abs(d))])
    return three(x - y), three((x - y) / (x + y)), three((x[1:] - x[:-1]) - (y[1:] - y[:-1]))


def _normalised_pair(x, y):
    L = min(len(x), len(y))
    x, y = x[:L], y[:L]
    return x / np.sqrt(np.sum(x ** 2) / L), y / np.sqrt(np.sum(y ** 2) / L)


def haspi_v1(x, fx, y, fy, alpha=-1.0, noise_x=None, noise_y=None, return_parts=False, HL=None):
    """pyhaspi2.py:109-157 (`haspi`): logistic of cepstral correlation + high-level BM covariance.  -> (Intel, [CepCorr, cov3])."""
    x, y = _normalised_pair(x, y)
    xenv, xBM, yenv, yBM, xSL, ySL, parts = ear_model_bm(x, fx, y, fy, noise_x, noise_y, HL=HL, itype=0)
    xdB, ydB = env_smooth(xenv), env_smooth(yenv)
    CepCorr, xy = melcor(xdB, ydB)
    sigcov, sigMSx, sigMSy = bm_covary(xBM, yBM)
    cov3, covSII = three_level_covary(sigcov, sigMSx)
    arg = -9.047 + 14.816 * CepCorr + np.sum(np.array([0, 0, 4.616]) * cov3)
    intel = 1.0 / (1.0 + np.exp(alpha * arg))
    raw = np.concatenate((np.array([CepCorr]), cov3))
    if return_parts:
        parts.update(xdB=xdB, ydB=ydB, xy=xy, sigcov=sigcov, sigMSx=sigMSx, sigMSy=sigMSy, covSII=covSII, xSL=xSL, ySL=ySL)
        return float(intel), raw, parts
    return float(intel), raw


def hasqi_v2(x, fx, y, fy, noise_x=None, noise_y=None, return_parts=False, HL=None):
    """pyhaspi2.py:32-74: -> (Combined, Nonlin, Linear, [CepCorr, BMsync5, Dloud, Dslope])."""
    x, y = _normalised_pair(x, y)
    xenv, xBM, yenv, yBM, xSL, ySL, parts = ear_model_bm(x, fx, y, fy, noise_x, noise_y, HL=HL, itype=2)      # eq = 2 (pyhaspi2.py:41-43)
    xdB, ydB = env_smooth(xenv), env_smooth(yenv)
    CepCorr, xy = melcor(xdB, ydB)
    dloud, dnorm, dslope = spect_diff(xSL, ySL)
    sigcov, sigMSx, sigMSy = bm_covary(xBM, yBM)
    avecov, syncov = ave_covary2(sigcov, sigMSx)
    BMsync5 = syncov[4]
    Dloud = float(np.clip(1.0 - dloud[1] / 2.5, 0, 1))
    Dslope = float(np.clip(1.0 - dslope[1], 0, 1))
    Nonlin = (CepCorr ** 2) * BMsync5
    Linear = 0.579 * Dloud + 0.421 * Dslope
    raw = [CepCorr, BMsync5, Dloud, Dslope]
    if return_parts:
        parts.update(xdB=xdB, ydB=ydB, xy=xy, sigcov=sigcov, sigMSx=sigMSx, avecov=avecov, syncov=syncov, dloud=dloud, dnorm=dnorm,
                     dslope=dslope, xSL=xSL, ySL=ySL)
        return Nonlin * Linear, Nonlin, Linear, raw, parts
    return Nonlin * Linear, Nonlin, Linear, raw
