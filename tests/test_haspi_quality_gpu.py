"""GPU: HASPI version 1 and HASQI v2 (pyhaspi2.py:109-157, :32-74) through the C ABI against the golden made by
running the reference, and against the oracle on other signals.  The reference adds N(0, 1.8e-4) noise to every basilar-membrane sample
from numpy's global generator: the parity comparisons run noise-free on both sides (tolerance 2e-5: float32 storage of the per-sample
envelopes and BM motion, float64 arithmetic); what a noise realisation does to the scores (< 1e-3, tests/test_oracle_haspi.py) bounds
the comparison with the reference's own noisy run."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
Q = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'haspi_quality.npz'))
RTOL = 2e-5


def _oracle_row(x, fx, y, fy):
    from oracle import haspi as H
    v1, raw = H.haspi_v1(x, fx, y, fy)
    q = H.hasqi_v2(x, fx, y, fy)
    return np.array([v1, raw[0], raw[1], raw[2], raw[3], q[0], q[1], q[2], q[3][1], q[3][2], q[3][3]])


def test_golden_pair_matches_the_reference_and_the_noise_free_oracle():
    from nele_gan_amd import metrics as mt
    x, y = Q['x'], Q['y']
    out, info = mt.batch_haspi_quality(x, y, 24000, noise=False, return_info=True)
    o = out[0].cpu().numpy()
    assert int(info[0, 1]) == 0 and int(info[0, 0]) > 100 and int(info[0, 2]) > 100
    np.testing.assert_allclose(o[:11], _oracle_row(x, 24000, y, 24000), rtol=RTOL)
    # the reference's own (noisy) run
    assert o[0] == pytest.approx(float(Q['intel']), abs=1e-3)
    np.testing.assert_allclose(o[1:5], Q['raw_v1'], atol=2e-3)
    np.testing.assert_allclose(o[5:8], Q['hasqi'], atol=1e-3)
    np.testing.assert_allclose([o[1], o[8], o[9], o[10]], Q['raw_q'], atol=2e-3)
    assert o[11] == pytest.approx(float(Q['avecov']), abs=2e-3)


def test_batch_at_16_khz_matches_the_oracle():
    from nele_gan_amd import metrics as mt
    from nele_gan_amd import synth
    c, v = synth.batch(3, 16000, start=4100)
    y = (c + np.array([0.2, 0.6, 1.5], dtype=np.float32)[:, None] * v).astype(np.float32)
    out = mt.batch_haspi_quality(c, y, 16000, noise=False).cpu().numpy()
    for b in range(3):
        np.testing.assert_allclose(out[b, :11], _oracle_row(c[b], 16000, y[b], 16000), rtol=1e-4, err_msg=str(b))
    assert np.all(np.diff(out[:, 5]) < 0)                               # more noise, lower quality


def test_mixed_lengths_in_one_padded_batch_equal_the_per_file_calls():
    from nele_gan_amd import dataio
    from nele_gan_amd import metrics as mt
    from nele_gan_amd import synth
    c, v = synth.batch(3, 24000, start=4200)
    lens = [24000, 17000, 20480]
    xs = [c[b, :n] for b, n in enumerate(lens)]
    ys = [(c[b, :n] + 0.5 * v[b, :n]).astype(np.float32) for b, n in enumerate(lens)]
    xp, L = dataio.pad_batch(xs)
    yp, _ = dataio.pad_batch(ys)
    yp[1, lens[1]:] = 7.0                                               # garbage behind a row's end must not matter
    got = mt.batch_haspi_quality(xp, yp, 16000, lengths=torch.from_numpy(L), noise=False).cpu().numpy()
    for b in range(3):
        one = mt.batch_haspi_quality(xs[b], ys[b], 16000, noise=False).cpu().numpy()[0]
        np.testing.assert_allclose(got[b], one, rtol=1e-9, err_msg=str(b))


def test_bm_noise_is_seeded_and_small():
    from nele_gan_amd import metrics as mt
    x, y = Q['x'], Q['y']
    quiet = mt.batch_haspi_quality(x, y, 24000, noise=False).cpu().numpy()[0]
    a = mt.batch_haspi_quality(x, y, 24000, noise=True, seed=11).cpu().numpy()[0]
    a2 = mt.batch_haspi_quality(x, y, 24000, noise=True, seed=11).cpu().numpy()[0]
    b = mt.batch_haspi_quality(x, y, 24000, noise=True, seed=12).cpu().numpy()[0]
    np.testing.assert_array_equal(a, a2)
    assert not np.array_equal(a, b) and not np.array_equal(a, quiet)
    np.testing.assert_allclose(a, quiet, atol=2e-3)
    np.testing.assert_allclose(a, b, atol=2e-3)
    # the same size of effect as the reference's generator has (golden = reference with noise, oracle without)
    assert abs(a[0] - float(Q['intel'])) < 1e-3 and abs(a[5] - float(Q['hasqi'][0])) < 1e-3


def test_reference_shaped_entry_points():
    from nele_gan_amd import metrics as mt
    x, y = Q['x'], Q['y']
    intel, raw = mt.haspi(x, 24000, y, 24000)
    comb, nonlin, lin, rawq = mt.hasqi_v2(x, 24000, y, 24000, HL=np.zeros(6))
    assert intel == pytest.approx(float(Q['intel']), abs=1e-3) and raw.shape == (4,)
    assert comb == pytest.approx(nonlin * lin, rel=1e-12) and len(rawq) == 4
    assert comb == pytest.approx(float(Q['hasqi'][0]), abs=1e-3)
    same = mt.hasqi_v2(x, 24000, x, 24000)
    assert same[0] == pytest.approx(1.0, abs=2e-3)                       # a signal against itself: perfect quality
    with pytest.raises(NotImplementedError):
        mt.haspi(x, 24000, y, 24000, HL=[10, 10, 20, 30, 40, 50])
