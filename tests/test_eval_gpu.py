"""GPU: the validation conditions of eval_metrics.py (RIR filtering, level normalisation, clip) against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _rir(n=3000, seed=5, peak_at=37):
    rng = np.random.default_rng(seed)
    h = rng.standard_normal(n) * np.exp(-np.arange(n) / 400.0) * 0.2
    h[peak_at] = 1.0
    return h.astype(np.float32)


@pytest.fixture(scope='module')
def ev():
    assert torch.cuda.is_available()
    from nele_gan_amd import eval_metrics
    return eval_metrics


@pytest.mark.parametrize('L,Lh', [(5000, 1), (5000, 700), (2500, 4000), (24000, 3000)])
def test_fir_filter_is_scipy_lfilter(ev, L, Lh):
    from scipy.signal import lfilter
    rng = np.random.default_rng(L + Lh)
    x = rng.standard_normal((3, L)).astype(np.float32)
    h = rng.standard_normal(Lh) * np.exp(-np.arange(Lh) / 300.0)
    y = ev.lfilter_fir(h, x).cpu().numpy()
    ref = lfilter(h, [1], x.astype(np.float64), axis=1)
    assert y.dtype == np.float64 and np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()


def test_norm_clip_matches_the_reference_loop(ev):
    from oracle import evalpath
    rng = np.random.default_rng(9)
    a = rng.standard_normal((4, 9000))
    a[1] *= 0.01                              # quiet: no clipping
    a[2, 100] = 40.0                          # several clip rounds after normalisation
    a[3] *= 3.0
    out, steps = ev.norm_clip(torch.from_numpy(a).cuda(), target_rms=0.25, return_steps=True)
    out, steps = out.cpu().numpy(), steps.cpu().numpy()
    for b in range(4):
        ref = evalpath.clip(a[b] / evalpath.rms(a[b]) * 0.25)
        np.testing.assert_allclose(out[b], ref.astype(np.float32), rtol=1e-6, atol=1e-9)
        assert out[b].max() < 1 and out[b].min() >= -1
    assert steps[1] == 0 and steps[2] >= 2
    plain = ev.norm_clip(torch.from_numpy(a[:1].astype(np.float32)).cuda(), add=a[1:2].astype(np.float32))
    np.testing.assert_allclose(plain.cpu().numpy()[0], evalpath.clip(a[0].astype(np.float32).astype(np.float64) + a[1].astype(np.float32)),
                               rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('reverb', [False, True])
def test_listening_condition_and_raw_metrics_vs_oracle(ev, reverb):
    from nele_gan_amd import synth
    from oracle import evalpath, step
    c, v = synth.batch(2, 32000, start=70)
    enh = (c * np.float32(2.0))[:, :31744]
    rir = _rir() if reverb else None
    clean_a, mixed = ev.listening_condition(c, enh, v, rir)
    res = ev.evaluate(c, enh, v, rir, metrics=('siib', 'estoi'))
    for b in range(2):
        ca, mx = evalpath.listening_condition(c[b], enh[b], v[b], rir)
        assert clean_a.shape[1] == len(ca) and mixed.shape[1] == len(mx)
        np.testing.assert_allclose(clean_a[b].cpu().numpy(), ca, rtol=1e-5, atol=2e-8)
        np.testing.assert_allclose(mixed[b].cpu().numpy(), mx, rtol=1e-5, atol=2e-8)
        ref = step.metric_targets(ca.astype(np.float32), mx.astype(np.float32), np.zeros(len(mx), np.float32), ['siib', 'estoi'], norm=False)
        assert res['siib'][b] == pytest.approx(ref[0], rel=2e-4, abs=1e-3)
        assert res['estoi'][b] == pytest.approx(ref[1], rel=2e-4, abs=2e-4)
    assert res['summary'].startswith('SIIB is ')
