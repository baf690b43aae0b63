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
