"""GPU: parity of the path bench.py times - the multi-stream ``GanTrainer.canonical_step`` at the BASELINE shapes - against the
CPU oracle, and of G / D / HASPI at the BASELINE frame counts (T = 251: 4 s, T = 501: 8 s).

What is compared with what:
  * multi-stream step  vs  the stage-by-stage sequence on one stream: BIT-identical losses, targets and updated weights (f32 and bf16)
  * B = 32, L = 64 000 step (f32 and bf16 operand modes)  vs  oracle.step.CpuStep run at the same batch
  * metric rows of the B = 32 / B = 256 steps  vs  small-batch launches on the same waveforms: bit-identical (batch invariance)
  * G / D forward + backward at (B, T) = (32, 251) and (4, 501)  vs  oracle/nets.py (f32: summation-order level; bf16: the operand
    rounding tolerances documented in DESIGN 4.1, written out below)
  * HASPI at 16 kHz, L = 64 000 (the chunk-parallel recurrences with their 8192-sample warm-up)  vs  oracle/haspi.py
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

from weights_recipe import seeded_state_arrays  # noqa: E402


def _state(module):
    return {k: t.detach().cpu().clone() for k, t in module.state_dict().items()}


def _trainer(metrics, precision, seed=666):
    from nele_gan_amd.train_nele import GanTrainer
    tr = GanTrainer(metrics, seed=seed)
    tr.G.precision = precision
    tr.D.precision = precision
    return tr


def _staged_step(tr, cw, nw):
    f = tr.features(cw, nw)
    lg = tr.g_step(f['clean_band'], f['noise_band'])
    enh = tr.generate(f['clean_band'], f['noise_band'], f['clean_spec'])
    tgt = tr.true_metrics(cw, enh, nw)
    ld = tr.d_step(tr.d_inputs(enh, f['noise_band'], f['clean_band']), tgt)
    return lg, ld, tgt, enh


@pytest.mark.parametrize('precision', ['f32', 'bf16'])
def test_multistream_step_is_bit_identical_to_the_staged_sequence(precision):
    """canonical_step() spreads one step over seven streams ordered by events only.  A missing wait would change targets or
    gradients: from identical state, two steps of it must reproduce the single-stream stage-by-stage sequence bit for bit."""
    from nele_gan_amd import synth
    c, v = synth.batch(3, 24000, start=40)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    a = _trainer('siib&haspi&estoi', precision)
    b = _trainer('siib&haspi&estoi', precision)
    for k, t in a.G.state_dict().items():
        assert torch.equal(t, b.G.state_dict()[k])
    for step in range(2):
        lg_a, ld_a, tgt_a = a.canonical_step(cw, nw)
        lg_b, ld_b, tgt_b, enh_b = _staged_step(b, cw, nw)
        torch.cuda.synchronize()
        assert torch.equal(a._last_enh, enh_b), 'enhanced batch, step %d' % step
        assert torch.equal(tgt_a, tgt_b), 'targets, step %d' % step
        assert float(lg_a) == float(lg_b) and float(ld_a) == float(ld_b), (step, float(lg_a), float(lg_b), float(ld_a), float(ld_b))
        assert torch.equal(a.G.flat_parameters().flat, b.G.flat_parameters().flat), 'G parameters, step %d' % step
        assert torch.equal(a.D.flat_parameters().flat, b.D.flat_parameters().flat), 'D parameters, step %d' % step
        for (k, ta), (_, tb) in zip(a.D.state_dict().items(), b.D.state_dict().items()):
            assert torch.equal(ta, tb), 'D buffer %s, step %d' % (k, step)
    assert all(x == 0 for x in a.check_status().values())


def test_delayed_side_streams_do_not_change_the_step():
    """Every cross-stream hand-over of canonical_step is ordered by events only.  Holding a side stream back (a 25 ms spin kernel in front
    of D.prepare / of the clean-feature branch) must not change anything; and the flat parameter buffers must stay where they are
    (round 2: GanTrainer('cuda') handed an index-less device to D.prepare, torch.device('cuda') != torch.device('cuda:0') made every
    cache look stale, and D's parameters were re-homed into a fresh flat buffer on the side stream twice per step)."""
    from nele_gan_amd import audio_util as au
    from nele_gan_amd import synth
    c, v = synth.batch(3, 24000, start=40)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    ref = _trainer('siib&estoi', 'bf16')
    r0 = ref.canonical_step(cw, nw)
    torch.cuda.synchronize()
    for which in ('prepare', 'features'):
        tr = _trainer('siib&estoi', 'bf16')
        orig_prepare, orig_stft = tr.D.prepare, au.stft_band
        if which == 'prepare':
            tr.D.prepare = lambda *a, **k: (torch.cuda._sleep(60_000_000), orig_prepare(*a, **k))[1]
        else:
            au.stft_band = lambda *a, **k: ((torch.cuda._sleep(60_000_000) if torch.cuda.current_stream() != torch.cuda.default_stream() else None),
                                            orig_stft(*a, **k))[1]
        try:
            r1 = tr.canonical_step(cw, nw)
            torch.cuda.synchronize()
        finally:
            au.stft_band = orig_stft
        assert float(r1[0]) == float(r0[0]) and float(r1[1]) == float(r0[1]) and torch.equal(r1[2], r0[2]), which
        tr.D.prepare = orig_prepare
        p0 = (tr.D.flat_parameters().flat.data_ptr(), tr.G.flat_parameters().flat.data_ptr(), tr.D._w['sigma'].data_ptr())
        tr.canonical_step(cw, nw)
        torch.cuda.synchronize()
        assert p0 == (tr.D.flat_parameters().flat.data_ptr(), tr.G.flat_parameters().flat.data_ptr(), tr.D._w['sigma'].data_ptr())
        assert len(tr.D._bufs) == 1 and len(tr.G._bufs) == 1


@pytest.mark.parametrize('what', ['all', 'features'])
def test_prefetched_pipeline_is_bit_identical_to_plain_steps(what):
    """canonical_step(pre=..., next_batch=..., early=False) moves the input-only work of batch k + 1 (late_prefetch = 'all': features, SIIB /
    HASPI clean halves; 'features': the features only, the default for large batches) behind the targets of batch k.  Three steps over two
    alternating batches must reproduce the plain step-by-step sequence bit for bit."""
    from nele_gan_amd import synth
    batches = []
    for st in (40, 140):
        c, v = synth.batch(3, 24000, start=st)
        batches.append((torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()))
    a = _trainer('siib&haspi&estoi', 'bf16')
    b = _trainer('siib&haspi&estoi', 'bf16')
    a.late_prefetch = what
    pre = a.prefetch(*batches[0], with_metrics=(what == 'all'))
    for k in range(3):
        cur, nxt = batches[k % 2], batches[(k + 1) % 2]
        ra = a.canonical_step(cur[0], cur[1], pre=pre, next_batch=nxt, early=False)
        pre = a.prefetched
        rb = b.canonical_step(cur[0], cur[1])
        torch.cuda.synchronize()
        assert torch.equal(ra[2], rb[2]) and float(ra[0]) == float(rb[0]) and float(ra[1]) == float(rb[1]), k
        assert torch.equal(a._last_enh, b._last_enh)
        assert torch.equal(a.G.flat_parameters().flat, b.G.flat_parameters().flat) and torch.equal(a.D.flat_parameters().flat, b.D.flat_parameters().flat)
    assert all(x == 0 for x in a.check_status().values())


@pytest.mark.parametrize('metric', ['siib&estoi', 'siib&haspi&estoi'])
def test_early_prefetch_on_the_second_stream_set_is_bit_identical_to_plain_steps(metric):
    """Small batches: canonical_step(next_batch=...) enqueues the next batch's input-only work at the START of the step, on the second
    set of side streams and metric workspaces (the sets alternate from step to step).  Five steps over three different batches - a
    batch's clean-signal halves are computed while the previous batch's degraded halves, D-step and targets are still in flight -
    must reproduce the plain sequence bit for bit, and both sets must have been used."""
    from nele_gan_amd import synth
    batches = []
    for st in (40, 140, 260):
        c, v = synth.batch(3, 24000, start=st)
        batches.append((torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()))
    a = _trainer(metric, 'bf16')
    b = _trainer(metric, 'bf16')
    assert batches[0][0].shape[0] <= a.early_prefetch_max_batch
    pre, sets = None, set()
    for k in range(5):
        cur, nxt = batches[k % 3], batches[(k + 1) % 3]
        ra = a.canonical_step(cur[0], cur[1], pre=pre, next_batch=nxt)
        pre = a.prefetched
        assert pre is not None and pre['set'] != a._cur_set
        sets.add(a._cur_set)
        rb = b.canonical_step(cur[0], cur[1])
        torch.cuda.synchronize()
        assert torch.equal(ra[2], rb[2]) and float(ra[0]) == float(rb[0]) and float(ra[1]) == float(rb[1]), k
        assert torch.equal(a._last_enh, b._last_enh)
        assert torch.equal(a.G.flat_parameters().flat, b.G.flat_parameters().flat) and torch.equal(a.D.flat_parameters().flat, b.D.flat_parameters().flat)
    assert sets == {0, 1}
    assert all(x == 0 for x in a.check_status().values())


def test_pipelined_streams_are_measured_onto_three_hardware_queues():
    """The pipelined step keeps the next batch's eigen-decomposition on the metric stream for a whole step, so nothing else may share that
    stream's hardware queue (the runtime multiplexes streams onto four).  GanTrainer._shares_queue measures it (an idle wave parked on one
    stream by nele_stream_spin, a trivial kernel timed on the other): a stream shares a queue with itself, and after _pipeline_queues
    neither the feature stream nor the second metric stream shares the metric stream's - even when twelve other streams were
    touched first."""
    tr = _trainer('siib&estoi', 'bf16')
    noise = [torch.cuda.Stream() for _ in range(12)]
    for st in noise:                                        # make the runtime hand out its queues before the trainer asks
        with torch.cuda.stream(st):
            torch.zeros(8, device='cuda').add_(1.0)
    torch.cuda.synchronize()
    tr._pipeline_queues(True)
    assert tr._shares_queue(tr._side, tr._side)
    assert not tr._shares_queue(tr._side, tr._fside)
    assert not tr._shares_queue(tr._side, tr._side2)
    assert tr.D._wstream == (tr._fside, tr._side2) and tr.G._wstream == tr._fside
    tr._pipeline_queues(False)
    assert tr.D._wstream is None or tr._fside not in tr.D._wstream


# tolerances of the bf16 operand mode against the float32 ORACLE (8-bit mantissa operands, float32 accumulation)
BF16 = dict(score_abs=4e-3, loss_rel=3e-2, mask_rel=8e-2, grad_l2=0.12, grad_cos=0.99, enh_rel_l2=5e-2)


@pytest.mark.parametrize('precision', ['f32', 'bf16'])
def test_baseline_step_b32_against_the_oracle(precision):
    """BASELINE configs[1] shape: B = 32, L = 64 000 (T = 251), SIIB + ESTOI, the multi-stream step that bench.py times."""
    from nele_gan_amd import synth
    from oracle.step import CpuStep
    B, L = 32, 64000
    c, v = synth.batch(B, L, start=0)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    tr = _trainer('siib&estoi', precision)
    cpu = CpuStep(_state(tr.G), _state(tr.D), metrics=('siib', 'estoi'))
    g0 = _state(tr.G)
    lg, ld, tgt = tr.canonical_step(cw, nw)
    torch.cuda.synchronize()
    enh = tr._last_enh
    assert all(x == 0 for x in tr.check_status().values())
    assert torch.isfinite(tgt).all() and np.isfinite(float(lg)) and np.isfinite(float(ld))
    enh_h = enh.cpu().numpy()
    # --- oracle, same batch (losses are batch means, so the nets run at B = 32 on the CPU too)
    cb, cm, cp, nb = cpu.features(c, v)
    lg_ref = cpu.g_step(cb, nb)
    enh_ref = cpu.generate(cb, nb, cm, cp)
    sub = [0, 13, 31]                                                     # metric oracle: seconds per utterance
    tgt_ref = cpu.targets(c[sub], [enh_h[i] for i in sub], v[sub])        # on the SAME waveform the GPU scored
    np.testing.assert_allclose(tgt.cpu().numpy()[sub], tgt_ref, rtol=1e-4)
    # batch invariance: the other rows equal small-batch launches on the same waveforms, bit for bit
    for lo, hi in ((0, 4), (17, 18), (28, 32)):
        small = tr.true_metrics(cw[lo:hi], enh[lo:hi], nw[lo:hi])
        assert torch.equal(small, tgt[lo:hi]), (lo, hi)
    ld_ref = cpu.d_step([e for e in enh_h], nb, cb, tgt.cpu().numpy())   # D-step on the GPU's enhanced batch and targets
    if precision == 'f32':
        assert float(lg) == pytest.approx(lg_ref, rel=1e-4)
        assert float(ld) == pytest.approx(ld_ref, rel=5e-4)
        for b in range(B):
            d = np.abs(enh_h[b] - enh_ref[b])
            assert d.max() <= 1.5 / 32768 and np.mean(d > 1e-7) < 0.02    # PCM_16: rare one-LSB rounding flips
        for k, t in tr.G.state_dict().items():
            np.testing.assert_allclose(t.cpu().numpy(), cpu.g[k].detach().numpy(), rtol=1e-3, atol=2e-5, err_msg=k)
    else:
        assert float(lg) == pytest.approx(lg_ref, rel=BF16['loss_rel'])
        assert float(ld) == pytest.approx(ld_ref, rel=BF16['loss_rel'])
        for b in range(B):
            assert np.linalg.norm(enh_h[b] - enh_ref[b]) <= BF16['enh_rel_l2'] * np.linalg.norm(enh_ref[b]), b
        # first Adam step = -lr * g / (|g| + eps) ~ -lr * sign(g): the update directions agree except where bf16 operand noise
        # flips the sign of a near-zero gradient element
        upd = np.concatenate([(t.cpu().numpy() - g0[k].numpy()).ravel() for k, t in tr.G.state_dict().items()])
        ref = np.concatenate([(cpu.g[k].detach().numpy() - g0[k].numpy()).ravel() for k in tr.G.state_dict().keys()])
        assert np.isfinite(upd).all() and np.abs(upd).max() <= 5.1e-4      # |update| <= lr on the first step
        cos = float((upd * ref).sum() / (np.linalg.norm(upd) * np.linalg.norm(ref)))
        assert cos > 0.8, cos


@pytest.mark.parametrize('B,T,precision', [(32, 251, 'f32'), (32, 251, 'bf16'), (4, 501, 'f32'), (4, 501, 'bf16')])
def test_generator_discriminator_at_baseline_frame_counts(B, T, precision):
    """G -> energy normalisation -> D -> MSE, forward and backward, at the BASELINE shapes (T = 251 fills conv_tile16_kernel<4,8>'s
    four tile columns and the 7-step weight-gradient tiles; T = 501 is the inference length) against oracle/nets.py."""
    from nele_gan_amd import model as mods
    from oracle import nets

    def load(module, seed):
        sd = module.state_dict()
        arrs = seeded_state_arrays([(k, tuple(x.shape)) for k, x in sd.items()], seed)
        module.load_state_dict({k: torch.from_numpy(x) for k, x in arrs.items()})
        return module.cuda()

    rs = np.random.RandomState(B * 1000 + T)
    G = load(mods.Generator_Conv1D_cLN(), 7)
    D = load(mods.Discriminator(nout=2), 8)
    G.precision = D.precision = precision
    x = (0.1 + 0.4 * rs.rand(B, T, 64)).astype(np.float32)
    y = (0.1 + 0.4 * rs.rand(B, T, 64)).astype(np.float32)
    tgt = rs.rand(B, 2).astype(np.float32)
    sdg = {k: t.detach().cpu().clone().requires_grad_(True) for k, t in G.state_dict().items()}
    sdd = {k: t.detach().cpu().clone().requires_grad_(not (k.endswith('_u') or k.endswith('_v'))) for k, t in D.state_dict().items()}
    xo, yo = torch.from_numpy(x), torch.from_numpy(y)
    mo = nets.generator_forward(sdg, xo, yo)
    eo, b2o = nets.energy_norm(mo, xo)
    so, _ = nets.discriminator_forward(sdd, nets.d_inputs(eo, yo, xo), train=True)
    lo = torch.nn.functional.mse_loss(so, torch.from_numpy(tgt))
    lo.backward()
    D.train()
    G.flat_parameters().grad.zero_()
    D.flat_parameters().grad.zero_()
    xc, yc = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    m = G(xc, yc)
    din, b2 = mods.energy_norm_pack(m, xc, yc)
    s = D.forward_packed(din)
    loss = torch.nn.functional.mse_loss(s, torch.from_numpy(tgt).cuda())
    loss.backward()
    torch.cuda.synchronize()
    mh, sh = m.detach().cpu().numpy(), s.detach().cpu().numpy()
    assert np.isfinite(mh).all() and np.isfinite(sh).all()
    if precision == 'f32':
        np.testing.assert_allclose(mh, mo.detach().numpy(), rtol=3e-4)
        np.testing.assert_allclose(b2.cpu().numpy(), b2o.detach().numpy(), rtol=1e-4)
        np.testing.assert_allclose(sh, so.detach().numpy(), rtol=1e-4)
        assert float(loss) == pytest.approx(float(lo), rel=1e-4)
        for name, mod, sd in (('G', G, sdg), ('D', D, sdd)):
            for k, p in mod.named_parameters():
                ref = sd[k].grad.numpy()
                np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=5e-3, atol=2e-4 * np.abs(ref).max(), err_msg=name + ' ' + k)
    else:
        np.testing.assert_allclose(mh, mo.detach().numpy(), rtol=BF16['mask_rel'])
        assert np.abs(sh - so.detach().numpy()).max() < BF16['score_abs']
        assert float(loss) == pytest.approx(float(lo), rel=BF16['loss_rel'])
        for name, mod, sd in (('G', G, sdg), ('D', D, sdd)):
            ga = np.concatenate([sd[k].grad.numpy().ravel() for k, _ in mod.named_parameters()])
            gb = np.concatenate([p.grad.cpu().numpy().ravel() for _, p in mod.named_parameters()])
            assert np.isfinite(gb).all()
            rel = np.linalg.norm(ga - gb) / np.linalg.norm(ga)
            cos = float((ga * gb).sum() / (np.linalg.norm(ga) * np.linalg.norm(gb)))
            assert rel < BF16['grad_l2'] and cos > BF16['grad_cos'], (name, rel, cos)


def test_haspi_16k_at_baseline_length_vs_oracle():
    """L = 64 000: 96 000 samples at 24 kHz = 12 chunks of the chunk-parallel filter banks (warm-up 8192 samples), 47 of the
    gain pass - the lengths at which the warm-up approximation and the chunk seams actually occur."""
    from nele_gan_amd import metrics as mt
    from nele_gan_amd import synth
    from oracle import haspi as H
    c, v = synth.batch(2, 64000, start=60)
    y = c + v
    raw, mapped, info = mt.batch_haspi(c, y, fs=16000, dither=None, return_info=True)
    raw = raw.cpu().numpy()
    assert int(info[:, 1].sum()) == 0
    for b in range(2):
        ref, _ = H.haspi_v2(c[b], 16000, y[b], 16000)
        assert raw[b] == pytest.approx(ref, rel=1e-4), b
    np.testing.assert_allclose(mapped.cpu().numpy(), 1 / (1 + np.exp(-0.95 * (raw.astype(np.float64) - 2.8))), rtol=1e-5)


def test_config3_step_b256_three_metrics_finite_and_batch_invariant():
    """BASELINE configs[2]: B = 256, SIIB + ESTOI + HASPI, bf16 operands, the multi-stream step.  Everything finite, no masked
    optimiser step, and the target rows equal a B = 8 launch of the three metric kernels on the same waveforms bit for bit."""
    from nele_gan_amd import synth
    B, L = 256, 64000
    c, v = synth.batch(B, L, start=0)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    tr = _trainer('siib&haspi&estoi', 'bf16')
    lg, ld, tgt = tr.canonical_step(cw, nw)
    lg2, ld2, tgt2 = tr.canonical_step(cw, nw)
    torch.cuda.synchronize()
    enh = tr._last_enh
    st = tr.check_status()
    assert all(x == 0 for x in st.values()), st
    for t in (lg, ld, lg2, ld2, tgt, tgt2, enh, tr.G.flat_parameters().flat, tr.D.flat_parameters().flat):
        assert torch.isfinite(t).all()
    assert (tgt2 > 0).all() and (tgt2 < 1).all()                          # logistic-mapped scores
    for lo in (0, 120, 248):
        small = tr.true_metrics(cw[lo:lo + 8], enh[lo:lo + 8], nw[lo:lo + 8])
        assert torch.equal(small, tgt2[lo:lo + 8]), lo


def test_nonfinite_target_masks_the_optimiser_step():
    """SIIB of an utterance with too few active frames is undefined (pysiib raises; status 8, NaN score).  A D-step that consumes a
    NaN target must leave D's parameters and Adam moments untouched, and check_status() must report both."""
    from nele_gan_amd import synth
    c, v = synth.batch(3, 24000, start=70)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    tr = _trainer('siib&estoi', 'f32')
    lg, ld, tgt = tr.canonical_step(cw, nw)
    assert all(x == 0 for x in tr.check_status().values())
    f = tr.features(cw, nw)
    din = tr.d_inputs(tr._last_enh, f['noise_band'], f['clean_band'])
    d0 = tr.D.flat_parameters().flat.detach().clone()
    m0, v0 = tr.optimizer_d.m.clone(), tr.optimizer_d.v.clone()
    bad = tgt.clone()
    bad[1, 0] = float('nan')
    tr.d_step(din, bad)
    torch.cuda.synchronize()
    assert torch.equal(tr.D.flat_parameters().flat, d0) and torch.equal(tr.optimizer_d.m, m0) and torch.equal(tr.optimizer_d.v, v0)
    tr.d_step(din, tgt)                                                   # the next good step goes through
    assert not torch.equal(tr.D.flat_parameters().flat, d0) and torch.isfinite(tr.D.flat_parameters().flat).all()
    tr._note_status('main', siib_info=torch.tensor([[14, 100, 3, 8], [14, 100, 90, 0], [40, 100, 90, 1]], dtype=torch.int32, device='cuda'))
    st = tr.check_status(raise_on_error=False)
    assert st['skipped_d_steps'] == 1 and st['siib_undefined'] == 1 and st['siib_clamped'] == 1 and st['skipped_g_steps'] == 0, st
    with pytest.raises(RuntimeError):
        tr.check_status()


def test_canonical_step_can_be_captured_in_a_hip_graph_and_replays_bit_identically():
    """Round 4 (verdict item 3): the seven-stream canonical step as ONE HIP graph (torch.cuda.graph stream capture).  Capture needs every
    forked stream joined back into the capturing stream directly - a side stream joining ANOTHER side stream makes ROCm 7's
    hipStreamEndCapture segfault (tools/graph_probe2.py), which is why both metric streams now join the main stream.  One replay on
    the same state equals one eager step bit for bit (same kernels, same order per stream).  Replay is measured 10-20 % SLOWER than
    the eager multi-stream launch at B = 32 / 128 / 256, so the trainer does not use it; this test keeps the step capturable."""
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    B, L = 4, 24000
    c, v = synth.batch(B, L, start=420)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()

    def warm(tr):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                tr.canonical_step(cw, nw)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        return s

    tr_a, tr_b = GanTrainer('siib&haspi&estoi'), GanTrainer('siib&haspi&estoi')
    s = warm(tr_a)
    with torch.cuda.stream(s):
        lg_a, ld_a, tgt_a = tr_a.canonical_step(cw, nw)
    torch.cuda.synchronize()
    warm(tr_b)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        lg_b, ld_b, tgt_b = tr_b.canonical_step(cw, nw)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(tgt_a, tgt_b) and float(lg_a) == float(lg_b) and float(ld_a) == float(ld_b)
    assert torch.equal(tr_a.G.flat_parameters().flat, tr_b.G.flat_parameters().flat)
    assert torch.equal(tr_a.D.flat_parameters().flat, tr_b.D.flat_parameters().flat)
