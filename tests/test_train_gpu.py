"""GPU: one canonical GAN_epoch step (features -> G-step -> generate -> metrics -> D-step) against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def test_canonical_step_matches_cpu_oracle():
    assert torch.cuda.is_available()
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    from oracle.step import CpuStep
    B, L = 2, 24000
    c, v = synth.batch(B, L, start=300)
    tr = GanTrainer('siib&estoi')
    g0 = {k: t.detach().cpu().clone() for k, t in tr.G.state_dict().items()}
    d0 = {k: t.detach().cpu().clone() for k, t in tr.D.state_dict().items()}
    cpu = CpuStep(g0, d0, metrics=('siib', 'estoi'))
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    # --- stage by stage so that a mismatch is attributable
    f = tr.features(cw, nw)
    cb, cm, cp, nb = cpu.features(c, v)
    np.testing.assert_allclose(f['clean_band'].cpu().numpy(), cb, rtol=1e-5)
    np.testing.assert_allclose(f['noise_band'].cpu().numpy(), nb, rtol=2e-5)
    lg = tr.g_step(f['clean_band'], f['noise_band'])
    lg_ref = cpu.g_step(cb, nb)
    assert float(lg) == pytest.approx(lg_ref, rel=1e-4)
    for k, t in tr.G.state_dict().items():                             # Adam-updated generator
        np.testing.assert_allclose(t.cpu().numpy(), cpu.g[k].detach().numpy(), rtol=1e-3, atol=2e-5, err_msg=k)
    enh = tr.generate(f['clean_band'], f['noise_band'], f['clean_spec'])
    enh_ref = cpu.generate(cb, nb, cm, cp)
    assert enh.shape == (B, 256 * (1 + L // 256 - 1))
    for b in range(B):
        d = np.abs(enh[b].cpu().numpy() - enh_ref[b])
        assert d.max() <= 1.5 / 32768 and np.mean(d > 1e-7) < 0.02     # PCM_16: rare one-LSB rounding flips
    tgt = tr.true_metrics(cw, enh, nw)
    tgt_ref = cpu.targets(c, [e for e in enh.cpu().numpy()], v)       # metrics on the SAME waveform
    np.testing.assert_allclose(tgt.cpu().numpy(), tgt_ref, rtol=1e-4)
    din = tr.d_inputs(enh, f['noise_band'], f['clean_band'])
    ld = tr.d_step(din, tgt)
    ld_ref = cpu.d_step([e for e in enh.cpu().numpy()], nb, cb, tgt_ref)
    assert float(ld) == pytest.approx(ld_ref, rel=2e-4)
    for k, t in tr.D.state_dict().items():
        np.testing.assert_allclose(t.cpu().numpy(), cpu.d[k].detach().numpy(), rtol=2e-3, atol=2e-5, err_msg=k)


def test_inference_path_rms_and_length():
    from nele_gan_amd import synth
    from nele_gan_amd.inference import Enhancer
    c, v = synth.batch(2, 32000, start=500)
    torch.manual_seed(1)
    e = Enhancer()
    out = e.enhance(torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda(), pcm16=False)
    assert out.shape == (2, 256 * (32000 // 256))
    rms = out.pow(2).mean(dim=1).sqrt().cpu().numpy()
    np.testing.assert_allclose(rms, 0.03, rtol=1e-5)                   # inference.py:109


def test_steps_do_not_accumulate_device_memory():
    """A custom autograd Function that keeps its own output on ctx forms an uncollectable cycle (2 MB per G-step once)."""
    import gc
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    c, v = synth.batch(4, 24000, start=5)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    tr = GanTrainer('estoi')
    for _ in range(3):
        tr.canonical_step(cw, nw)
    torch.cuda.synchronize(); gc.collect()
    m0 = torch.cuda.memory_allocated()
    for _ in range(40):
        tr.canonical_step(cw, nw)
    torch.cuda.synchronize(); gc.collect()
    growth = torch.cuda.memory_allocated() - m0
    assert growth < 40 * 4096, "device memory grows by %d bytes per step" % (growth // 40)


def test_d_epoch_three_passes_and_history_replay():
    """train_nele.py:342-426: pass A on the current list, pass B on 1/30 of the (shuffled) history + current, history += current,
    pass C on the current list again."""
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    c, v = synth.batch(2, 16000, start=9)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    tr = GanTrainer('estoi')
    f = tr.features(cw, nw)
    din = tr.d_inputs(cw, f['noise_band'], f['clean_band'])
    item = lambda i: (din[i % 2].clone(), torch.tensor([0.25 + 0.01 * i], device='cuda'))
    seen = []
    orig = tr.d_step
    tr.d_step = lambda d, t, *a, **k: (seen.append(d.shape[0]), orig(d, t, *a, **k))[1]   # equal lengths here: plain batches
    tr.history = [item(i) for i in range(60)]                       # 60 // 30 = 2 replayed items
    cur = [item(100 + i) for i in range(5)]
    w0 = tr.D.layers[4].weight_orig.detach().clone()
    tr.d_epoch(cur, batch=4)
    assert sum(seen) == 5 + (2 + 5) + 5                             # items seen by D in passes A, B, C
    assert seen == [4, 1, 4, 3, 4, 1]                               # batches of at most 4
    assert len(tr.history) == 65
    assert not torch.equal(w0, tr.D.layers[4].weight_orig.detach())
