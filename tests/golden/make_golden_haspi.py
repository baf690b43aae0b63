"""HASPI v2 goldens: runs the reference's pyHASPI/pyhaspi2.py (imported, numba.jit = identity) at
fs = 24 kHz on a seeded speech-like pair and records stage outputs + the RNG draws ebm_CepCoef used.
Called from make_golden.py (which installs the import stand-ins first)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def golden_dither(seed, n_samples, n_active):
    rs = np.random.RandomState(seed)
    for _ in range(64):
        rs.randn(n_samples)
    return rs.randn(n_active, 32), rs.randn(n_active, 32)


def gen_haspi(ref):
    import pyhaspi2 as P
    from nele_gan_amd import synth
    n = 30000                                               # 1.25 s at 24 kHz
    x = synth.clean_utterance(77, n).astype(np.float32)
    v = synth.noise_utterance(77, n, x)
    y = (x + np.float32(0.3) * v).astype(np.float32)
    rec = {}
    draws = []
    real_randn = np.random.randn

    def randn(*shape):
        a = real_randn(*shape)
        if len(shape) == 2:
            draws.append(a.copy())
        return a

    orig = {k: getattr(P, k) for k in ('eb_BWadjust', 'ebm_EnvFilt', 'ebm_CepCoef', 'ebm_ModCorr', 'eb_GroupDelayComp')}
    bw = []

    def BWadjust(*a, **k):
        r = orig['eb_BWadjust'](*a, **k)
        bw.append(r)
        return r

    def EnvFilt(*a, **k):
        r = orig['ebm_EnvFilt'](*a, **k)
        rec['xLP'], rec['yLP'] = r
        return r

    def CepCoef(*a, **k):
        r = orig['ebm_CepCoef'](*a, **k)
        rec['xcep'], rec['ycep'] = r
        return r

    def ModCorr(*a, **k):
        r = orig['ebm_ModCorr'](*a, **k)
        rec['aveCM'] = r
        return r

    def GroupDelayComp(xenv, BW, cfreq, fsamp):
        if 'shifts' not in rec:
            probe = orig['eb_GroupDelayComp'](np.ones_like(xenv), BW, cfreq, fsamp)
            rec['shifts'] = np.array([int(np.argmax(probe[c] > 0)) for c in range(probe.shape[0])])
            rec['cfreq'] = np.asarray(cfreq)
        return orig['eb_GroupDelayComp'](xenv, BW, cfreq, fsamp)

    P.eb_BWadjust, P.ebm_EnvFilt, P.ebm_CepCoef, P.ebm_ModCorr, P.eb_GroupDelayComp = BWadjust, EnvFilt, CepCoef, ModCorr, GroupDelayComp
    np.random.randn = randn
    try:
        np.random.seed(4321)
        intel, raw = P.haspi_v2(x, 24000, y, 24000)
    finally:
        np.random.randn = real_randn
        for k, f in orig.items():
            setattr(P, k, f)
    assert len(draws) == 2
    out = dict(x=x, y=y, intel=np.float64(intel), aveCM=np.asarray(raw), BWx=np.array(bw[0::2]), BWy=np.array(bw[1::2]),
               cfreq=rec['cfreq'], shifts=rec['shifts'], seed=np.int64(4321), n_active=np.int64(draws[0].shape[0]),
               dither_x_head=draws[0][:4], dither_y_sum=np.float64(draws[1].sum()),
               xLP_head=rec['xLP'][:40], yLP_head=rec['yLP'][:40], xLP_tail=rec['xLP'][-8:], n_sub=np.int64(rec['xLP'].shape[0]),
               xcep_head=rec['xcep'][:64], ycep_head=rec['ycep'][:64], xcep_sq=(rec['xcep'] ** 2).sum(axis=0), ycep_sq=(rec['ycep'] ** 2).sum(axis=0))
    # the dither is NOT stored: numpy's legacy RandomState stream is frozen, so the tests regenerate it with
    # golden_dither(): seed, skip the 64 eb_BMaddnoise draws of randn(n_samples), then two randn(n_active, 32)
    dx, dy = golden_dither(4321, len(x), draws[0].shape[0])
    assert np.array_equal(dx, draws[0]) and np.array_equal(dy, draws[1])
    # sums over all rows pin the rest of xLP / yLP without storing 2 x 3334 x 32 doubles
    out['xLP_colsum'] = rec['xLP'].sum(axis=0)
    out['yLP_colsum'] = rec['yLP'].sum(axis=0)
    np.savez_compressed(os.path.join(HERE, 'haspi.npz'), **out)
    # ---- check the oracle against it right away
    from oracle import haspi as H
    val, parts = H.haspi_v2(x, 24000, y, 24000, dither_x=dx, dither_y=dy, return_parts=True)
    err = abs(val - intel) / abs(intel)
    print('haspi.npz written: Intel %.6f (oracle %.6f, rel err %.2e), n_sub %d, n_active %d, shifts max %d' %
          (intel, val, err, out['n_sub'], rec['xcep'].shape[0], rec['shifts'].max()))
    assert err < 1e-9, err


def golden_bm_noise(seed, n_samples):
    """The 64 draws of eb_BMaddnoise (pyhaspi2.py:1091-1095) in the reference's order: per channel, x then y."""
    rs = np.random.RandomState(seed)
    nx = np.empty((32, n_samples))
    ny = np.empty((32, n_samples))
    for n in range(32):
        nx[n] = rs.randn(n_samples)
        ny[n] = rs.randn(n_samples)
    return nx, ny


def gen_haspi_quality(ref):
    """HASPI version 1 (`haspi`) and HASQI v2 of the reference on the same seeded pair: final values and stage outputs."""
    import pyhaspi2 as P
    from nele_gan_amd import synth
    n = 30000
    x = synth.clean_utterance(77, n).astype(np.float32)
    v = synth.noise_utterance(77, n, x)
    y = (x + np.float32(0.3) * v).astype(np.float32)
    rec = {}
    names = ('eb_EnvSmooth', 'eb_melcor', 'eb_BMcovary', 'eb_3LevelCovary', 'eb_AveCovary2', 'eb_SpectDiff', 'eb_EarModel')
    orig = {k: getattr(P, k) for k in names}

    def tap(name):
        def f(*a, **k):
            r = orig[name](*a, **k)
            rec.setdefault(name, []).append(r)
            return r
        return f
    for k in names:
        setattr(P, k, tap(k))
    try:
        np.random.seed(9001)
        intel, raw1 = P.haspi(x, 24000, y, 24000)
        np.random.seed(9001)
        comb, nonlin, lin, raw2 = P.hasqi_v2(x, 24000, y, 24000)
    finally:
        for k, f in orig.items():
            setattr(P, k, f)
    xdB, ydB = rec['eb_EnvSmooth'][0], rec['eb_EnvSmooth'][1]
    sigcov, sigMSx, sigMSy = rec['eb_BMcovary'][0]
    cov3, covSII = rec['eb_3LevelCovary'][0]
    avecov, syncov = rec['eb_AveCovary2'][0]
    dloud, dnorm, dslope = rec['eb_SpectDiff'][0]
    ear = rec['eb_EarModel'][0]
    out = dict(x=x, y=y, seed=np.int64(9001), intel=np.float64(intel), raw_v1=np.asarray(raw1, dtype=np.float64),
               hasqi=np.array([comb, nonlin, lin], dtype=np.float64), raw_q=np.asarray(raw2, dtype=np.float64),
               xdB_s=xdB[:, ::25].copy(), ydB_s=ydB[:, ::25].copy(), xdB_colsum=xdB.sum(axis=0), ydB_colsum=ydB.sum(axis=0),
               xy=rec['eb_melcor'][0][1], sigcov_s=sigcov[:, ::25].copy(), sigcov_colsum=sigcov.sum(axis=0), sigMSx_colsum=sigMSx.sum(axis=0),
               sigMSy_colsum=sigMSy.sum(axis=0), cov3=cov3, covSII=covSII, avecov=np.float64(avecov), syncov=np.asarray(syncov),
               dloud=dloud, dnorm=dnorm, dslope=dslope, xSL=ear[4], ySL=ear[5], nseg=np.int64(xdB.shape[1]))
    np.savez_compressed(os.path.join(HERE, 'haspi_quality.npz'), **out)
    from oracle import haspi as H
    nx, ny = golden_bm_noise(9001, len(x))
    v1, r1, parts = H.haspi_v1(x, 24000, y, 24000, noise_x=nx, noise_y=ny, return_parts=True)
    q = H.hasqi_v2(x, 24000, y, 24000, noise_x=nx, noise_y=ny)
    e1 = abs(v1 - intel) / abs(intel)
    e2 = abs(q[0] - comb) / abs(comb)
    print('haspi_quality.npz written: HASPI v1 %.6f (oracle rel err %.2e), HASQI %.6f / %.6f / %.6f (oracle rel err %.2e), nseg %d' %
          (intel, e1, comb, nonlin, lin, e2, xdB.shape[1]))
    print('  raw v1', raw1, 'oracle', r1)
    print('  raw q ', raw2, 'oracle', q[3])
    assert e1 < 1e-9 and e2 < 1e-9, (e1, e2)


# Audiograms (dB HL at 250, 500, 1000, 2000, 4000, 6000 Hz) of the hearing-loss goldens: a mild, nearly flat loss and a sloping
# moderate-to-severe one whose high-frequency channels pass the outer-hair-cell threshold (attnOHC saturates, attnIHC takes the rest).
HL_MILD = (20.0, 20.0, 25.0, 30.0, 40.0, 45.0)
HL_SLOPING = (10.0, 15.0, 30.0, 50.0, 70.0, 80.0)


def gen_haspi_hl(ref):
    """The reference's own haspi_v2 / haspi / hasqi_v2 for listeners with a hearing loss (pyhaspi2.py:779-807, 1155-1166): final values,
    the loss parameters and the adjusted bandwidths, on the seeded pair of gen_haspi.  Noise / dither draws are regenerated from the seed
    by the tests (golden_dither / golden_bm_noise), exactly as for the normal-hearing goldens."""
    import pyhaspi2 as P
    from nele_gan_amd import synth
    n = 30000
    x = synth.clean_utterance(77, n).astype(np.float32)
    v = synth.noise_utterance(77, n, x)
    y = (x + np.float32(0.3) * v).astype(np.float32)
    out = dict(x=x, y=y, seed=np.int64(4321))
    real_randn = np.random.randn
    orig_bw = P.eb_BWadjust
    for tag, HL in (('mild', HL_MILD), ('sloping', HL_SLOPING)):
        HLa = np.asarray(HL, dtype=np.float64)
        cf = P.eb_CenterFreq(32)
        lp = P.eb_LossParameters(HLa, cf)
        out['%s_HL' % tag] = HLa
        out['%s_loss' % tag] = np.stack([np.asarray(a, dtype=np.float64) for a in lp])          # attnOHC, BW, lowknee, CR, attnIHC
        bw, draws = [], []

        def BWadjust(*a, **k):
            r = orig_bw(*a, **k)
            bw.append(r)
            return r

        def randn(*shape):
            a = real_randn(*shape)
            if len(shape) == 2:
                draws.append(a.copy())
            return a
        P.eb_BWadjust = BWadjust
        np.random.randn = randn
        try:
            np.random.seed(4321)
            intel, raw = P.haspi_v2(x, 24000, y, 24000, HLa)
            out['%s_v2_intel' % tag], out['%s_v2_aveCM' % tag] = np.float64(intel), np.asarray(raw)
            out['%s_v2_BWx' % tag], out['%s_v2_BWy' % tag] = np.array(bw[0::2]), np.array(bw[1::2])
            out['%s_v2_n_active' % tag] = np.int64(draws[0].shape[0])
            np.random.seed(4321)
            i1, r1 = P.haspi(x, 24000, y, 24000, HLa)
            out['%s_v1' % tag] = np.concatenate(([i1], np.asarray(r1)))                         # Intel, CepCorr, cov3 low / mid / high
            np.random.seed(4321)
            comb, nonlin, lin, rq = P.hasqi_v2(x, 24000, y, 24000, HLa)
            out['%s_hasqi' % tag] = np.asarray([comb, nonlin, lin] + list(rq), dtype=np.float64)   # Combined, Nonlin, Linear, CepCorr, BMsync5, Dloud, Dslope
        finally:
            np.random.randn = real_randn
            P.eb_BWadjust = orig_bw
        print('haspi_hl %s: haspi_v2 %.6f, haspi %.6f, hasqi_v2 %.6f' % (tag, intel, i1, comb))
    np.savez_compressed(os.path.join(HERE, 'haspi_hl.npz'), **out)
    # ---- check the oracle against it right away
    from oracle import haspi as H
    for tag, HL in (('mild', HL_MILD), ('sloping', HL_SLOPING)):
        dx, dy = golden_dither(4321, n, int(out['%s_v2_n_active' % tag]))
        val, _ = H.haspi_v2(x, 24000, y, 24000, dither_x=dx, dither_y=dy, HL=HL)
        nx, ny = golden_bm_noise(4321, n)
        v1, r1 = H.haspi_v1(x, 24000, y, 24000, noise_x=nx, noise_y=ny, HL=HL)
        q = H.hasqi_v2(x, 24000, y, 24000, noise_x=nx, noise_y=ny, HL=HL)
        e0 = abs(val - out['%s_v2_intel' % tag]) / abs(out['%s_v2_intel' % tag])
        e1 = abs(v1 - out['%s_v1' % tag][0])
        e2 = abs(q[0] - out['%s_hasqi' % tag][0])
        print('  oracle %s: haspi_v2 rel err %.2e, haspi abs err %.2e, hasqi abs err %.2e' % (tag, e0, e1, e2))
        assert e0 < 1e-9 and e1 < 1e-9 and e2 < 1e-9, (e0, e1, e2)
