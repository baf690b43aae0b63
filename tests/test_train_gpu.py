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
    rows = []
    tr.d_step = lambda d, t, *a, **k: (seen.append(k.get('items', d.shape[0])), rows.append(d.shape[0]), orig(d, t, *a, **k))[2]
    tr.history = [item(i) for i in range(60)]                       # 60 // 30 = 2 replayed items
    cur = [item(100 + i) for i in range(5)]
    w0 = tr.D.layers[4].weight_orig.detach().clone()
    tr.d_epoch(cur, batch=4)
    assert sum(seen) == 5 + (2 + 5) + 5                             # items seen by D in passes A, B, C
    assert seen == [4, 1, 4, 3, 4, 1]                               # batches of at most 4 items ...
    assert rows == [4, 4, 4, 4, 4, 4]                               # ... the short ones filled up with all-zero rows outside the loss (one buffer shape)
    assert len(tr.history) == 65
    assert not torch.equal(w0, tr.D.layers[4].weight_orig.detach())


def test_configs0_toy_train_triple_batch2_estoi_only_vs_cpu_oracle():
    """BASELINE configs[0] exactly: the toy_dataset/Train triple duplicated to batch 2 (16 kHz, L = 33 536, T = 132), ESTOI-only loss
    (D out-dim 1): ONE canonical step on the device against the CPU oracle's loop (oracle/step.py = the reference's train_nele.py
    stages) on the same files and the same initial weights."""
    import os
    from nele_gan_amd import dataio
    from nele_gan_amd.train_nele import GanTrainer
    from oracle.step import CpuStep
    toy = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'toy')
    names = sorted(n for n in os.listdir(toy) if n.endswith('.wav'))
    clean = [n for n in names if 'clean' in n.lower() and 'train' in n.lower()]
    noise = [n for n in names if 'noise' in n.lower() and 'train' in n.lower()]
    assert clean and noise, names
    c1, sr = dataio.read_wav(os.path.join(toy, clean[0]))
    v1, _ = dataio.read_wav(os.path.join(toy, noise[0]))
    assert sr == 16000
    L = min(len(c1), len(v1))
    c = np.stack([c1[:L], c1[:L]]).astype(np.float32)
    v = np.stack([v1[:L], v1[:L]]).astype(np.float32)
    assert L == 33536
    tr = GanTrainer('estoi')
    assert tr.D._nout == 1
    g0 = {k: t.detach().cpu().clone() for k, t in tr.G.state_dict().items()}
    d0 = {k: t.detach().cpu().clone() for k, t in tr.D.state_dict().items()}
    cpu = CpuStep(g0, d0, metrics=('estoi',))
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    lg, ld, tgt = tr.canonical_step(cw, nw)
    enh = tr._last_enh
    cb, cm, cp, nb = cpu.features(c, v)
    lg_ref = cpu.g_step(cb, nb)
    assert float(lg) == pytest.approx(lg_ref, rel=1e-4)
    enh_ref = cpu.generate(cb, nb, cm, cp)
    assert enh.shape == (2, 256 * 131)
    for b in range(2):
        d = np.abs(enh[b].cpu().numpy() - enh_ref[b])
        assert d.max() <= 1.5 / 32768 and np.mean(d > 1e-7) < 0.02
    tgt_ref = cpu.targets(c, [e for e in enh.cpu().numpy()], v)
    np.testing.assert_allclose(tgt.cpu().numpy(), tgt_ref, rtol=1e-4)
    assert tgt.shape == (2, 1) and float((tgt[0] - tgt[1]).abs()) == 0.0        # the two copies score alike
    ld_ref = cpu.d_step([e for e in enh.cpu().numpy()], nb, cb, tgt_ref)
    assert float(ld) == pytest.approx(ld_ref, rel=2e-4)
    for k, t in tr.D.state_dict().items():
        np.testing.assert_allclose(t.cpu().numpy(), cpu.d[k].detach().numpy(), rtol=2e-3, atol=2e-5, err_msg=k)
    for k, t in tr.G.state_dict().items():
        np.testing.assert_allclose(t.cpu().numpy(), cpu.g[k].detach().numpy(), rtol=1e-3, atol=2e-5, err_msg=k)


def test_true_metrics_pair_equals_two_calls():
    """The generated and the pre-enhanced example of one clean batch scored with the clean-signal work done once
    (train_nele.py:318-340): bit-identical to two full true_metrics() calls, with and without per-utterance lengths, and the
    fallback when the two comparisons see different lengths."""
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    B, L = 3, 25600
    c, v = synth.batch(B, L, start=40)
    tr = GanTrainer('siib&haspi&estoi')
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    enh = (cw * 1.2).contiguous()
    drc = (cw * 0.7 + 0.1 * nw).contiguous()
    a1 = tr.true_metrics(cw, enh, nw).clone()
    a2 = tr.true_metrics(cw, drc, nw, resynth=False).clone()
    p1, p2 = tr.true_metrics_pair(cw, enh, drc, nw)
    assert torch.equal(a1, p1) and torch.equal(a2, p2)
    assert float((p1 - p2).abs().max()) > 1e-3
    lens = torch.tensor([25600, 20480, 23040], dtype=torch.int32)
    b1 = tr.true_metrics(cw, enh, nw, lengths=lens, norm=False).clone()
    b2 = tr.true_metrics(cw, drc, nw, lengths=lens, resynth=False, norm=False).clone()
    q1, q2 = tr.true_metrics_pair(cw, enh, drc, nw, lengths=lens, drc_lengths=lens, norm=False)
    assert torch.equal(b1, q1) and torch.equal(b2, q2)
    # different truncation (the DRC file is shorter): two full calls, same values as by hand
    dl = torch.tensor([25600, 19000, 23040], dtype=torch.int32)
    ml = torch.minimum(dl, lens)
    e2 = tr.true_metrics(cw, drc, nw, lengths=ml, resynth=False).clone()
    r1, r2 = tr.true_metrics_pair(cw, enh, drc, nw, lengths=lens, drc_lengths=dl)
    assert torch.equal(r2, e2)
    tr.check_status()
    # the three metrics run on their own streams (SIIB / HASPI / ESTOI: nothing shared but the inputs); one stream gives the same bits
    ts = GanTrainer('siib&haspi&estoi')
    ts.metric_streams = False
    s1, s2 = ts.true_metrics_pair(cw, enh, drc, nw, lengths=lens, drc_lengths=lens, norm=False)
    assert torch.equal(s1, q1) and torch.equal(s2, q2)
    assert torch.equal(ts.true_metrics(cw, enh, nw).clone(), a1)
    assert ts._side is None and tr._side is not None and tr._side2 is not None and tr._fside is not None
    ts.check_status()


def test_haspi_dither_per_utterance_id_vs_oracle_and_batch_independent():
    """pyhaspi2.py:362-365 dithers every call.  haspi_dither='utterance': the rows are a function of (seed, utterance id) - the kernel's
    draws equal oracle/haspi.py:dither_rows, the HASPI target with them equals the oracle's haspi_v2 given the same draws (1e-4), and
    an utterance scores the same in another batch position / batch composition (what a sharded run needs)."""
    from nele_gan_amd import metrics as mt
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    from oracle import haspi as H
    B, L = 3, 32000
    c, v = synth.batch(B, L, start=60)
    y = (0.8 * c + v).astype(np.float32)
    ids = torch.tensor([1007, 5, 123456789012], dtype=torch.int64)
    rows = mt.haspi_dither_rows(ids, 11, L)
    nsub = rows.shape[2]
    for k in range(B):
        np.testing.assert_allclose(rows[k].cpu().numpy(), H.dither_rows(int(ids[k]), 11, nsub), rtol=0, atol=1e-12)
    tr = GanTrainer('haspi', haspi_dither='utterance', dither_seed=11)
    cw, yw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(v).cuda()
    zero = torch.zeros_like(nw)
    t = tr.true_metrics(cw, yw, zero, norm=False, resynth=False, utt_ids=ids)[:, 0].cpu().numpy()
    t_plain = GanTrainer('haspi').true_metrics(cw, yw, zero, norm=False, resynth=False)[:, 0].cpu().numpy()
    assert np.abs(t - t_plain).max() > 1e-6                                  # the dither is on
    raw, _, info = mt.batch_haspi(cw, yw, dither=rows, return_info=True)
    for k in range(B):
        na = int(info[k, 0])
        d = rows[k].cpu().numpy()
        ref, _ = H.haspi_v2(c[k], 16000, y[k], 16000, dither_x=d[0, :na], dither_y=d[1, :na])
        assert t[k] == pytest.approx(ref, rel=1e-4)
    # the same utterances in another order / another batch: same scores, bit for bit
    perm = [2, 0]
    t2 = tr.true_metrics(cw[perm].contiguous(), yw[perm].contiguous(), zero[perm].contiguous(), norm=False, resynth=False,
                         utt_ids=ids[perm])[:, 0].cpu().numpy()
    assert t2[0] == t[2] and t2[1] == t[0]
    # and through the canonical step's split path (clean part beside the G-step): finite, and different from the undithered targets
    tr3 = GanTrainer('haspi', haspi_dither='utterance', dither_seed=11)
    tr4 = GanTrainer('haspi')
    _, _, tg3 = tr3.canonical_step(cw, nw, utt_ids=ids)
    _, _, tg4 = tr4.canonical_step(cw, nw)
    assert bool(torch.isfinite(tg3).all()) and float((tg3 - tg4).abs().max()) > 0


def test_adam_after_a_masked_step_matches_torch_adam_that_never_saw_it():
    """nele_adam_step_guarded skips an update whose gradient is not finite; the bias correction of the following updates counts the updates
    that happened (torch.optim.Adam's `step` state), not the calls (round-4 verdict item 8: the masked call used to be counted)."""
    from nele_gan_amd import model as M
    from nele_gan_amd.optim import Adam
    torch.manual_seed(0)
    D = M.Discriminator().cuda()
    opt = Adam(D, lr=2.5e-4)
    fp = D.flat_parameters()
    ref_p = torch.nn.Parameter(fp.flat.detach().clone())
    ref = torch.optim.Adam([ref_p], lr=2.5e-4)
    g = torch.Generator(device='cuda').manual_seed(1)
    grads = [torch.randn(fp.flat.numel(), device='cuda', generator=g) * 0.1 for _ in range(5)]
    for k, gr in enumerate(grads):
        bad = k in (1, 3)
        fp.grad.copy_(gr)
        if bad:
            fp.grad[123] = float('inf')
        opt.step()
        if not bad:
            ref_p.grad = gr.clone()
            ref.step()
    assert opt.skipped_steps() == 2 and opt.step_count == 5
    torch.testing.assert_close(fp.flat, ref_p.detach(), rtol=2e-6, atol=1e-9)
    torch.testing.assert_close(opt.m, ref.state[ref_p]['exp_avg'], rtol=1e-5, atol=1e-7)        # (fused multiply-adds against torch's separate ops)


def test_fill_rows_of_a_padded_d_batch_do_not_touch_the_update():
    """d_step(items=n): rows behind the first n are fill rows (all-zero items) - the loss, hence every gradient, is that of the n items:
    the same D update (to summation order) as the step on the n items alone."""
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    c, v = synth.batch(3, 16000, start=21)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    res = []
    for fill in (0, 5):
        tr = GanTrainer('estoi', seed=11)
        f = tr.features(cw, nw)
        din = tr.d_inputs(cw, f['noise_band'], f['clean_band'])
        tgt = torch.tensor([[0.2], [0.5], [0.7]], device='cuda')
        if fill:
            din = torch.cat([din, din.new_zeros((fill,) + tuple(din.shape[1:]))])
            tgt = torch.cat([tgt, tgt.new_zeros((fill, 1))])
        loss = tr.d_step(din, tgt, items=3)
        res.append((float(loss), tr.D.flat_parameters().flat.clone()))
    assert res[0][0] == pytest.approx(res[1][0], rel=1e-6)
    d = (res[0][1] - res[1][1]).abs()
    assert float(d.max()) <= 2.5e-4 * 1.01 and float(d.mean()) < 1e-6      # Adam's first step moves every weight by lr: equal up to gradient-sign noise at zero


def test_hardware_queue_probe_and_streams_on_distinct_queues():
    """ops.shares_queue: a stream shares a queue with itself, three probed streams do not share one pairwise, and among seven unprobed
    side streams (three hardware queues serve them) some pair does - which the probe reports the same way in both argument orders."""
    from nele_gan_amd import ops
    dev = torch.device('cuda:0')
    s = torch.cuda.Stream(device=dev)
    assert ops.shares_queue(s, s, dev)
    three = ops.streams_on_distinct_queues(dev, 3)
    for i in range(3):
        for j in range(i + 1, 3):
            assert not ops.shares_queue(three[i], three[j], dev)
    many = [torch.cuda.Stream(device=dev) for _ in range(7)]
    pairs = [(i, j) for i in range(7) for j in range(i + 1, 7) if ops.shares_queue(many[i], many[j], dev)]
    assert pairs, "seven side streams on distinct hardware queues?"
    i, j = pairs[0]
    assert ops.shares_queue(many[j], many[i], dev)


def test_replay_history_beyond_its_device_budget_moves_to_host_memory_and_is_still_replayed():
    """GanTrainer.history_hbm_bytes: the D inputs of past epochs beyond the budget live in host memory; pass B uploads the replayed ones."""
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    c, v = synth.batch(2, 16000, start=9)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    tr = GanTrainer('estoi')
    f = tr.features(cw, nw)
    din = tr.d_inputs(cw, f['noise_band'], f['clean_band'])
    item = lambda i: (din[i % 2].clone(), torch.tensor([0.25 + 0.01 * i], device='cuda'))
    per_item = din[0].numel() * 4
    tr.history_hbm_bytes = 10 * per_item                          # room for ten items
    tr.history = [item(i) for i in range(60)]
    seen = []
    orig = tr.d_step
    tr.d_step = lambda d, t, *a, **k: (seen.append(k.get('items', d.shape[0])), orig(d, t, *a, **k))[1]
    tr.d_epoch([item(100 + i) for i in range(5)], batch=4)
    on_dev = sum(1 for it in tr.history if it[0].is_cuda)
    assert len(tr.history) == 65 and on_dev == 10 and all(it[1].device == it[0].device for it in tr.history)
    seen.clear()
    tr.d_epoch([item(200 + i) for i in range(5)], batch=4)        # 65 // 30 = 2 replayed items, most likely from host memory
    assert sum(seen) == 5 + (2 + 5) + 5 and len(tr.history) == 70
    assert tr.check_status()['skipped_d_steps'] == 0


def test_gathered_d_batches_equal_the_per_item_copies():
    """ops.d_gather (nele_d_gather_items): a shuffled list of per-utterance D items - rows of larger padded batches, cut to their own frame
    counts - as one zero-padded batch with device-side frame counts, against the per-item copies it replaces (dataloader.py:54-84)."""
    import torch
    from nele_gan_amd import ops
    g = torch.Generator(device='cuda').manual_seed(5)
    src = [torch.rand((7, 64, T, 4), device='cuda', generator=g) for T in (40, 97, 251)]
    items = []
    for k in range(150):                                       # more than two launches' worth, views with three different row strides
        b = src[k % 3]
        T = b.shape[2]
        fr = T if k % 5 == 0 else 21 + (k * 7) % (T - 20)
        items.append(b[k % 7] if fr == T else b[k % 7, :, :fr])
    rows, Tm = 160, 256
    din, frames = ops.d_gather(items, rows, Tm)
    ref = torch.zeros((rows, 64, Tm, 4), device='cuda')
    for r, it in enumerate(items):
        ref[r, :, :it.shape[1]] = it
    assert torch.equal(din, ref)
    assert frames.tolist() == [it.shape[1] for it in items] + [Tm] * (rows - len(items))
    with pytest.raises(Exception):
        ops.d_gather([src[2][0]], 1, 100)                     # an item longer than the batch
