"""CPU: metric oracles -- the reference-owned parts against the goldens, the restated third-party
parts against published properties (ESTOI/SIIB cores are PARITY UNPINNED, see oracle docstrings)."""
import os
import wave

import numpy as np
import pytest

from oracle import estoi, intel, siib

HERE = os.path.dirname(__file__)
GI = np.load(os.path.join(HERE, 'golden', 'intel.npz'))


def toy(name):
    w = wave.open(os.path.join(HERE, 'golden', 'toy', name))
    return np.frombuffer(w.readframes(w.getnframes()), dtype='<i2').astype(np.float32) / 32768.0


def test_framing_vad_stft_match_reference():
    x = toy('Train_Clean.wav')
    fr = intel.framing(x)
    assert fr.shape[0] == int(GI['n_frames']) == intel.n_frames(len(x))
    assert np.array_equal(fr[:3], GI['frames_head']) and np.array_equal(fr[-2:], GI['frames_tail'])
    assert np.array_equal(intel.get_vad(x), GI['vad'])                 # integer decisions: bit-exact
    np.testing.assert_allclose(intel.stft(x)[:4], GI['stft_head'], rtol=1e-12, atol=1e-15)
    M, nact = intel.siib_replication(x)
    assert (M, nact) == (int(GI['M']), int(GI['n_active'])) == (14, 141)


def test_logistic_maps_match_reference():
    p = GI['map_pts']
    np.testing.assert_allclose(intel.mapping_SIIB_harvard(p), GI['map_siib'], rtol=1e-15)
    np.testing.assert_allclose(intel.mapping_HASPI_harvard(p), GI['map_haspi'], rtol=1e-15)
    np.testing.assert_allclose(intel.mapping_ESTOI_harvard(p), GI['map_estoi'], rtol=1e-15)


def test_estoi_properties():
    x, v = toy('Train_Clean.wav'), toy('Train_Noise.wav')
    assert estoi.estoi(x, x) == pytest.approx(1.0, abs=1e-9)
    s_noisy = estoi.estoi(x, x + v)
    s_less = estoi.estoi(x, x + 0.25 * v)
    assert -0.1 < s_noisy < s_less < 1.0          # ESTOI can dip slightly below 0 at -11 dB SNR
    # scale invariance of the degraded signal (normalised correlations)
    assert estoi.estoi(x, 3.0 * (x + v)) == pytest.approx(s_noisy, rel=1e-9)
    # too short -> pystoi's 1e-5
    assert estoi.estoi(x[:3000], x[:3000]) == 1e-5
    obm, edges = estoi.thirdoct()
    assert obm.shape == (15, 257) and edges[0] == (7, 9) and edges[-1][1] <= 257
    assert len(estoi.resample_window_oct(10000, 16000)) == 581
    assert len(estoi.resample_16k_to_10k(x)) == -(-len(x) * 5 // 8)


def test_siib_properties():
    x, v = toy('Train_Clean.wav'), toy('Train_Noise.wav')
    G = siib.gammatone_matrix()
    assert G.shape == (28, 201) and np.allclose(G.max(axis=1), 1.0)
    s_clean = siib.siib_wrapper(x, x, norm=False)
    s_noisy = siib.siib_wrapper(x, x + v, norm=False)
    s_less = siib.siib_wrapper(x, x + 0.25 * v, norm=False)
    # identical signals: every component has rho = 1 -> R/K * 420 * (-1/2 log2(1 - 0.75^2))
    assert s_clean == pytest.approx(80 / 15 * 420 * (-0.5 * np.log2(1 - 0.5625)), rel=1e-6)
    assert 0.0 <= s_noisy < s_less < s_clean
    assert 0.0 < siib.siib_wrapper(x, x + v) < 1.0


def test_siib_rank_deficient_case_is_deterministic():
    from nele_gan_amd import synth
    c, v = synth.batch(1, 16000)                                       # L multiple of 200: frame-periodic after tiling
    val, parts = siib.siib_gauss(np.hstack([c[0]] * 8), np.hstack([c[0] + v[0]] * 8), return_parts=True)
    assert np.isfinite(val) and val > 0
    assert (parts['lam'] <= siib.EIG_TOL * parts['lam'].max()).sum() > 0


def test_siib_rank_deficient_components_cut_size_of_the_documented_deviation():
    """DESIGN section 2, deviation (i): with L a multiple of 200 samples the replicated signal is exactly frame-periodic, the stacked
    clean covariance is rank deficient and the information of its null components is a ratio of rounding errors in the reference
    (pysiib has no cut).  This pins HOW MUCH the cut (EIG_TOL = 1e-10, oracle and kernels alike) changes the score on the bench's own
    utterances at the headline length 64 000: per-cent level, NOT below 1e-4 - so the 1e-4 SIIB claim rests on lengths that are not
    multiples of 100 (any real file; bench companion `nonperiodic`, L = 63 871), where the cut selects nothing."""
    from nele_gan_amd import synth
    from oracle import siib
    c, v = synth.batch(2, 64000, start=0)
    rel = []
    try:
        for k in range(2):
            x, y = c[k], 0.8 * c[k] + v[k]
            siib.EIG_TOL = 1e-10
            a = siib.siib_wrapper(x, y, norm=False)
            siib.EIG_TOL = -1.0                       # no cut: every component counts, as in pysiib
            b = siib.siib_wrapper(x, y, norm=False)
            rel.append(abs(a - b) / abs(b))
        # a non-periodic length (like any real file): the covariance has full rank (smallest eigenvalue ~ 1e-6 of the largest) and the
        # cut selects nothing; a multiple of 100 that is no multiple of 200 (period of 2 L / 200 frames): a handful of null components
        ab = {}
        for L in (63871, 63900):
            x, y = c[0][:L], (0.8 * c[0] + v[0])[:L]
            siib.EIG_TOL = 1e-10
            a = siib.siib_wrapper(x, y, norm=False)
            siib.EIG_TOL = -1.0
            ab[L] = (a, siib.siib_wrapper(x, y, norm=False))
    finally:
        siib.EIG_TOL = 1e-10
    assert ab[63871][0] == ab[63871][1]
    assert 1e-5 < abs(ab[63900][0] - ab[63900][1]) / ab[63900][1] < 1e-2
    assert all(1e-3 < r < 0.3 for r in rel), rel
