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


def test_hearing_loss_audiograms_match_the_reference_and_the_oracle():
    """pyhaspi2.py:779-807 (eb_LossParameters) + the HLx / HL split of eb_EarModel (:1155-1166): haspi_v2 with the reference's own
    dither draws against the reference (1e-4), haspi / hasqi_v2 against the noise-free oracle (tight) and the reference's noisy run."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    from make_golden_haspi import HL_MILD, HL_SLOPING, golden_dither
    from nele_gan_amd import metrics as mt
    from oracle import haspi as H
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'haspi_hl.npz'))
    x, y, n = G['x'], G['y'], len(G['x'])
    normal = float(mt.batch_haspi(x, y, 24000)[0][0])
    for tag, HL in (('mild', HL_MILD), ('sloping', HL_SLOPING)):
        # haspi_v2: the dither rows are inputs (row k perturbs the k-th active frame)
        na = int(G[tag + '_v2_n_active'])
        dx, dy = golden_dither(int(G['seed']), n, na)
        nsub = (n + 8) // 9
        d = np.zeros((1, 2, nsub, 32))
        d[0, 0, :na], d[0, 1, :na] = dx, dy
        raw, mapped, info = mt.batch_haspi(x, y, 24000, dither=torch.from_numpy(d).cuda(), return_info=True, HL=HL)
        assert int(info[0, 0]) == na and int(info[0, 1]) == 0
        assert float(raw[0]) == pytest.approx(float(G[tag + '_v2_intel']), rel=1e-4)
        assert abs(float(raw[0]) - normal) > 0.05                        # the loss really changes the score
        # haspi (itype 0) and hasqi_v2 (itype 2)
        o0 = mt.batch_haspi_quality(x, y, 24000, noise=False, HL=HL, itype=0).cpu().numpy()[0]
        o2 = mt.batch_haspi_quality(x, y, 24000, noise=False, HL=HL, itype=2).cpu().numpy()[0]
        v1, r1 = H.haspi_v1(x, 24000, y, 24000, HL=HL)
        q = H.hasqi_v2(x, 24000, y, 24000, HL=HL)
        np.testing.assert_allclose(o0[0:5], np.concatenate(([v1], r1)), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(o2[5:11], [q[0], q[1], q[2], q[3][1], q[3][2], q[3][3]], rtol=2e-5, atol=2e-6)
        # Against the reference's own (noisy) run the device generator's eb_BMaddnoise has to be ON: with a severe loss the attenuated BM
        # motion of the high bands sits near the -10 dB SL threshold noise, which then is part of the model, not a perturbation (sloping
        # audiogram: haspi 0.670 without noise, 0.7035 +- 0.0005 with it over noise seeds; oracle, tools-free check in the test history)
        n0 = mt.batch_haspi_quality(x, y, 24000, noise=True, seed=5, HL=HL, itype=0).cpu().numpy()[0]
        n2 = mt.batch_haspi_quality(x, y, 24000, noise=True, seed=5, HL=HL, itype=2).cpu().numpy()[0]
        assert abs(n0[0] - float(G[tag + '_v1'][0])) < 3e-3 and abs(n2[5] - float(G[tag + '_hasqi'][0])) < 1e-3
        assert abs(o0[5] - o2[5]) > 1e-3                                 # with a loss the two ear models differ: one call serves one model
    # reference-shaped wrappers take the audiogram; NAL-R (itype 1) is refused as in the reference (eb_NALR raises)
    intel, raw4 = mt.haspi(x, 24000, y, 24000, HL=HL_SLOPING)
    assert intel == pytest.approx(float(G['sloping_v1'][0]), abs=3e-3)
    comb = mt.hasqi_v2(x, 24000, y, 24000, HL=np.asarray(HL_MILD))[0]
    assert comb == pytest.approx(float(G['mild_hasqi'][0]), abs=1e-3)
    assert mt.haspi_v2(x, 24000, y, 24000, HL=HL_MILD) == pytest.approx(float(G['mild_v2_intel']), rel=2e-3)     # fresh dither draws
    with pytest.raises(Exception):
        mt.batch_haspi_quality(x, y, 24000, HL=HL_MILD, itype=1)
