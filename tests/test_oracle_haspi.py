"""CPU: the HASPI v2 oracle against the golden made by running the reference's pyhaspi2 (fs = 24 kHz)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
from make_golden_haspi import golden_dither  # noqa: E402

from oracle import haspi as H  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'haspi.npz'))


def test_static_tables_match_reference():
    np.testing.assert_allclose(H.center_freq(), G['cfreq'], rtol=1e-14)
    attn, bw, knee, cr, ihc = H.loss_parameters(np.zeros(6), H.center_freq())
    assert np.all(attn == 0) and np.all(bw == 1) and np.all(knee == 30) and np.all(ihc == 0)
    np.testing.assert_allclose(cr, 1.25 + 2.25 * np.arange(32) / 31, rtol=1e-14)
    b, nh = H.mod_filters()
    assert [len(w) - 1 for w in b] == [614, 614, 614, 384, 244, 152, 96, 60, 38, 24]


def test_haspi_v2_matches_reference_with_captured_dither():
    x, y = G['x'], G['y']
    dx, dy = golden_dither(int(G['seed']), len(x), int(G['n_active']))
    np.testing.assert_array_equal(dx[:4], G['dither_x_head'])
    assert dy.sum() == pytest.approx(float(G['dither_y_sum']), rel=1e-14)
    val, p = H.haspi_v2(x, 24000, y, 24000, dither_x=dx, dither_y=dy, return_parts=True)
    np.testing.assert_allclose(p['BWx'], G['BWx'], rtol=1e-12)
    np.testing.assert_allclose(p['BWy'], G['BWy'], rtol=1e-12)
    assert np.array_equal(p['shifts'], G['shifts'])                                        # integers: bit-exact
    assert p['xLP'].shape[0] == int(G['n_sub']) and len(p['index']) == int(G['n_active'])
    np.testing.assert_allclose(p['xLP'][:40], G['xLP_head'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(p['yLP'][:40], G['yLP_head'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(p['xLP'][-8:], G['xLP_tail'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(p['xLP'].sum(axis=0), G['xLP_colsum'], rtol=1e-10)
    np.testing.assert_allclose(p['xcep'][:64], G['xcep_head'], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose((p['ycep'] ** 2).sum(axis=0), G['ycep_sq'], rtol=1e-10)
    np.testing.assert_allclose(p['aveCM'], G['aveCM'], rtol=1e-10)
    assert val == pytest.approx(float(G['intel']), rel=1e-10)


def test_haspi_properties_at_16k():
    from nele_gan_amd import synth
    c, v = synth.batch(1, 16000, start=5)
    s_clean, _ = H.haspi_v2(c[0], 16000, c[0], 16000)
    s_noisy, _ = H.haspi_v2(c[0], 16000, c[0] + v[0], 16000)
    assert s_clean == pytest.approx(float(np.sum(H.WEIGHTS)), rel=1e-9)                   # identical signals: every |rho| = 1
    assert 0 < s_noisy < s_clean
    y24 = H.resample_24k(c[0], 16000)
    assert len(y24) == 24000 and y24.dtype == np.float32
    assert np.sqrt(np.mean(y24.astype(np.float64) ** 2)) == pytest.approx(np.sqrt(np.mean(c[0].astype(np.float64) ** 2)), rel=1e-6)


def test_below_threshold_raises_like_reference():
    # pyhaspi2.py:357-358: fewer than two sub-sampled frames above 2.5 dB SL -> exception
    z = np.zeros((100, 32))
    with pytest.raises(Exception):
        H.cep_coef(z, z)


# ------------------------------------------------------------------ HASPI version 1 and HASQI v2
Q = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'haspi_quality.npz'))


def test_window_normalisation_tables_equal_the_reference_literals():
    """pyhaspi2.py:563 / :570 keep 1/xcorr(window, window, 24) as MATLAB literals; the oracle (and the kernel) compute them."""
    w = np.hanning(384)
    wc, hc = H.window_corr(w, 24), H.window_corr(w[192:], 24)
    assert wc[24] == pytest.approx(0.00696257615317668, rel=1e-13) and wc[0] == pytest.approx(0.00714486118736300, rel=1e-13)
    assert hc[24] == pytest.approx(0.0139251523063533, rel=1e-13) and hc[0] == pytest.approx(0.0171564012932667, rel=1e-13)
    assert np.allclose(wc, wc[::-1], rtol=1e-14) and np.allclose(hc, hc[::-1], rtol=1e-14)


def test_haspi_v1_and_hasqi_v2_match_the_reference_with_captured_noise():
    from make_golden_haspi import golden_bm_noise
    x, y = Q['x'], Q['y']
    nx, ny = golden_bm_noise(int(Q['seed']), len(x))
    intel, raw, p = H.haspi_v1(x, 24000, y, 24000, noise_x=nx, noise_y=ny, return_parts=True)
    assert p['xdB'].shape[1] == int(Q['nseg'])
    np.testing.assert_allclose(p['xdB'][:, ::25], Q['xdB_s'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(p['ydB'].sum(axis=0), Q['ydB_colsum'], rtol=1e-10)
    np.testing.assert_allclose(p['xy'], Q['xy'], rtol=1e-10)
    np.testing.assert_allclose(p['sigcov'][:, ::25], Q['sigcov_s'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(p['sigcov'].sum(axis=0), Q['sigcov_colsum'], rtol=1e-10)
    np.testing.assert_allclose(p['sigMSx'].sum(axis=0), Q['sigMSx_colsum'], rtol=1e-10)
    np.testing.assert_allclose(p['sigMSy'].sum(axis=0), Q['sigMSy_colsum'], rtol=1e-10)
    np.testing.assert_allclose(raw[1:], Q['cov3'], rtol=1e-10)
    np.testing.assert_allclose(p['covSII'], Q['covSII'], rtol=1e-10)
    np.testing.assert_allclose(p['xSL'], Q['xSL'], rtol=1e-11)
    np.testing.assert_allclose(raw, Q['raw_v1'], rtol=1e-10)
    assert intel == pytest.approx(float(Q['intel']), rel=1e-10)
    comb, nonlin, lin, rawq, pq = H.hasqi_v2(x, 24000, y, 24000, noise_x=nx, noise_y=ny, return_parts=True)
    np.testing.assert_allclose(pq['syncov'], Q['syncov'], rtol=1e-10)
    assert float(pq['avecov']) == pytest.approx(float(Q['avecov']), rel=1e-10)
    for k in ('dloud', 'dnorm', 'dslope'):
        np.testing.assert_allclose(pq[k], Q[k], rtol=1e-9)
    np.testing.assert_allclose(rawq, Q['raw_q'], rtol=1e-10)
    np.testing.assert_allclose([comb, nonlin, lin], Q['hasqi'], rtol=1e-10)


def test_bm_noise_changes_the_quality_scores_by_less_than_a_thousandth():
    """The reference adds N(0, 10^((-10 - 65)/20)) to the BM motion (pyhaspi2.py:1091-1095): what a different noise realisation (or
    none) does to the scores - the tolerance a device-side generator has to meet."""
    x, y = Q['x'], Q['y']
    v0, r0 = H.haspi_v1(x, 24000, y, 24000)
    q0 = H.hasqi_v2(x, 24000, y, 24000)
    assert v0 == pytest.approx(float(Q['intel']), abs=1e-3)
    assert q0[0] == pytest.approx(float(Q['hasqi'][0]), abs=1e-3)
    np.testing.assert_allclose(r0, Q['raw_v1'], atol=2e-3)


def test_hearing_loss_oracle_matches_the_reference(golden_dir):
    """eb_LossParameters and the HLx / HL split of eb_EarModel (pyhaspi2.py:779-807, 1155-1166): haspi_v2, haspi and hasqi_v2 of the
    reference itself for a mild and a sloping audiogram (tests/golden/haspi_hl.npz, make_golden_haspi.gen_haspi_hl)."""
    import os
    from make_golden_haspi import HL_MILD, HL_SLOPING, golden_bm_noise, golden_dither
    G = np.load(os.path.join(golden_dir, 'haspi_hl.npz'))
    x, y, n = G['x'], G['y'], len(G['x'])
    for tag, HL in (('mild', HL_MILD), ('sloping', HL_SLOPING)):
        np.testing.assert_array_equal(G[tag + '_HL'], HL)
        lp = H.loss_parameters(np.asarray(HL, dtype=np.float64), H.center_freq())
        np.testing.assert_allclose(np.stack(lp), G[tag + '_loss'], rtol=1e-14)
        dx, dy = golden_dither(int(G['seed']), n, int(G[tag + '_v2_n_active']))
        val, parts = H.haspi_v2(x, 24000, y, 24000, dither_x=dx, dither_y=dy, return_parts=True, HL=HL)
        assert val == pytest.approx(float(G[tag + '_v2_intel']), rel=1e-10)
        np.testing.assert_allclose(parts['aveCM'], G[tag + '_v2_aveCM'], rtol=1e-9)
        np.testing.assert_allclose(parts['BWx'], G[tag + '_v2_BWx'], rtol=1e-12)      # the reference signal: normal hearing (BWmin = 1)
        np.testing.assert_allclose(parts['BWy'], G[tag + '_v2_BWy'], rtol=1e-12)      # the processed signal: BWmin of the loss
        assert np.all(parts['BWy'] >= parts['BWx'] - 1e-12) and np.any(parts['BWy'] > parts['BWx'] + 1e-3)
        nx, ny = golden_bm_noise(int(G['seed']), n)
        v1, r1 = H.haspi_v1(x, 24000, y, 24000, noise_x=nx, noise_y=ny, HL=HL)
        np.testing.assert_allclose(np.concatenate(([v1], r1)), G[tag + '_v1'], rtol=1e-10)
        q = H.hasqi_v2(x, 24000, y, 24000, noise_x=nx, noise_y=ny, HL=HL)
        np.testing.assert_allclose([q[0], q[1], q[2]] + list(q[3]), G[tag + '_hasqi'], rtol=1e-10)
    with pytest.raises(NotImplementedError):
        H.ear_model(x, 24000, y, 24000, HL=HL_MILD, itype=1)             # NAL-R: eb_NALR raises in the reference too


def test_per_utterance_dither_rows_are_a_pure_function_of_seed_and_id():
    """oracle/haspi.py:dither_rows (the numpy statement of csrc/haspi.hip:haspi_dither_rows_kernel; the GPU test compares the two):
    standard normals per (signal, frame, channel), reproducible, independent of how many frames are asked for, different per id / seed."""
    a = H.dither_rows(123456789012, 11, 400)
    assert a.shape == (2, 400, 32) and a.dtype == np.float64
    np.testing.assert_array_equal(a, H.dither_rows(123456789012, 11, 400))
    np.testing.assert_array_equal(a[:, :150], H.dither_rows(123456789012, 11, 150))          # row k does not depend on nsub
    assert abs(a.mean()) < 0.02 and abs(a.std() - 1.0) < 0.02 and np.abs(a).max() < 6.0
    assert abs(np.corrcoef(a[0].ravel(), a[1].ravel())[0, 1]) < 0.03                           # x rows and y rows are different draws
    b, c = H.dither_rows(123456789013, 11, 400), H.dither_rows(123456789012, 12, 400)
    assert abs(np.corrcoef(a.ravel(), b.ravel())[0, 1]) < 0.03 and abs(np.corrcoef(a.ravel(), c.ravel())[0, 1]) < 0.03
