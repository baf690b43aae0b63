"""GPU: the parts of the loop surface that round 2 had only exercised with stubs or not at all.

* the quality-discriminator training path (train_nele.py:150-152, 362-365): G-step with the 0.5 * MSE(D_Qua) term and a D-step
  that also trains D_Qua, against the oracle (oracle/step.py with a D_Qua state);
* GanTrainer.run_epoch on the real kernels, epochs 1 and 2, over utterances of DIFFERENT lengths (toy Train / Test files + two
  synthetic ones) carried as padded batches with per-utterance lengths: identical, bit for bit, to the same stages called by hand;
  samples, D steps, checkpoint keys, the learning-curve line and the name@epoch.wav files as train_nele.py:110-122, 224-225,
  272-340 prescribe;
* BASELINE configs[4]'s shape: Enhancer.enhance on 8 s utterances (L = 128 000, 501 frames) against the oracle's inference path.
"""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
HERE = os.path.dirname(__file__)
TOY = os.path.join(HERE, 'golden', 'toy')


def _sd(m):
    return {k: t.detach().cpu().clone() for k, t in m.state_dict().items()}


def test_quality_discriminator_training_path_matches_the_oracle():
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    from oracle.step import CpuStep
    B, L = 2, 24000
    c, v = synth.batch(B, L, start=410)
    tr = GanTrainer('siib&estoi', use_quality=True)
    cpu = CpuStep(_sd(tr.G), _sd(tr.D), metrics=('siib', 'estoi'), dq_state=_sd(tr.D_Qua))
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    f = tr.features(cw, nw)
    cb, cm, cp, nb = cpu.features(c, v)
    # G-step: MSE(D, 1) + 0.5 * MSE(D_Qua, 1) (train_nele.py:150-152)
    lg = float(tr.g_step(f['clean_band'], f['noise_band']))
    lg_ref = cpu.g_step(cb, nb)
    assert lg == pytest.approx(lg_ref, rel=1e-4)
    plain = CpuStep(_sd(GanTrainer('siib&estoi').G), _sd(tr.D), metrics=('siib', 'estoi'))   # same seed -> same initial G
    assert abs(lg_ref - plain.g_step(cb, nb)) > 1e-3                  # the quality term is really in the loss
    for k, t in tr.G.state_dict().items():
        np.testing.assert_allclose(t.cpu().numpy(), cpu.g[k].detach().numpy(), rtol=1e-3, atol=2e-5, err_msg=k)
    # neither discriminator moved in the G-step (train_nele.py:153-155) - except their spectral-norm u, v (never in eval mode)
    for name, mod, ref in (('D', tr.D, cpu.d), ('D_Qua', tr.D_Qua, cpu.dq)):
        for k, t in mod.state_dict().items():
            np.testing.assert_allclose(t.cpu().numpy(), ref[k].detach().numpy(), rtol=1e-4, atol=1e-6, err_msg=name + '.' + k)
    # D-step with quality targets (train_nele.py:349-367)
    enh = tr.generate(f['clean_band'], f['noise_band'], f['clean_spec'])
    tgt = tr.true_metrics(cw, enh, nw)
    tq = np.asarray([[0.31, 0.62], [0.74, 0.18]], dtype=np.float32)    # stand-ins for mapped PESQ / ViSQOL scores (absent metrics)
    din = tr.d_inputs(enh, f['noise_band'], f['clean_band'])
    ld = float(tr.d_step(din, tgt, torch.from_numpy(tq).cuda()))
    lq = float(tr.last_loss_qua)
    ld_ref, lq_ref = cpu.d_step([e for e in enh.cpu().numpy()], nb, cb, tgt.cpu().numpy(), tq)
    assert ld == pytest.approx(ld_ref, rel=2e-4) and lq == pytest.approx(lq_ref, rel=2e-4)
    for name, mod, ref in (('D', tr.D, cpu.d), ('D_Qua', tr.D_Qua, cpu.dq)):
        for k, t in mod.state_dict().items():
            np.testing.assert_allclose(t.cpu().numpy(), ref[k].detach().numpy(), rtol=2e-3, atol=2e-5, err_msg=name + '.' + k)
    st = tr.check_status()
    assert st['skipped_dqua_steps'] == 0 and tr.optimizer_dqua.step_count == 1
    # the decision whether D_Qua is stepped is explicit: a batch without quality targets cannot be pushed through a D_Qua step
    with pytest.raises(ValueError):
        tr.d_step(din, tgt, None, has_qua=True)
    tr.d_step(din, tgt, None)                                           # no quality targets -> D only
    assert tr.optimizer_dqua.step_count == 1 and tr.optimizer_d.step_count == 2


def _corpus():
    """Two training batches and one validation batch of utterances of four different lengths, zero-padded, with lengths and names."""
    from nele_gan_amd import dataio, synth
    ld = lambda n: dataio.load(os.path.join(TOY, n))[0]
    c0, n0, e0 = ld('Train_Clean.wav'), ld('Train_Noise.wav'), ld('Train_MultiEnh.wav')        # 33 536 samples
    c1, n1 = ld('Test_Clean.wav'), ld('Test_Noise.wav')                                          # 34 048
    (c2,), (n2,) = synth.batch(1, 40000, start=77)
    (c3,), (n3,) = synth.batch(1, 29000, start=78)
    (c4,), (n4,) = synth.batch(1, 36000, start=79)

    def batch(cl, ns, names, drc=None, qua=None, drc_qua=None):
        cp, lens = dataio.pad_batch(cl)
        npad, _ = dataio.pad_batch(ns)
        b = {'clean': torch.from_numpy(cp).cuda(), 'noise': torch.from_numpy(npad).cuda(), 'lengths': torch.from_numpy(lens).cuda(),
             'names': names}
        if drc is not None:
            dp, dl = dataio.pad_batch(drc)
            b['drc'], b['drc_lengths'] = torch.from_numpy(dp).cuda(), torch.from_numpy(dl).cuda()
        if qua is not None:
            b['qua'], b['drc_qua'] = torch.tensor(qua, device='cuda'), torch.tensor(drc_qua, device='cuda')
        return b
    # pre-enhanced ('DRC') examples: the toy MultiEnh file and, for the synthetic utterances, a fixed spectral tilt of the clean signal
    drc2 = np.convolve(c2, [1.0, -0.6], mode='same').astype(np.float32)
    drc3 = np.convolve(c3, [1.0, -0.6], mode='same').astype(np.float32)[:28950]                   # a little shorter than its clean file (same frame count)
    train = [batch([c0, c2], [n0, n2], ['Train_Clean.wav', 'syn_a.wav'], drc=[e0, drc2]),
             batch([c3, c4], [n3, n4], ['syn_b.wav', 'syn_c.wav'], drc=[drc3, c4 * 0.9])]
    valid = [batch([c1, c3], [n1, n3], ['Test_Clean.wav', 'syn_b.wav'])]
    return train, valid


def _by_hand_epoch(tr, gan_epoch, train, valid):
    """The stages of run_epoch issued one by one through the trainer's public stage methods (same order as train_nele.py:110-429)."""
    from nele_gan_amd import audio_util as au
    feats = []
    if gan_epoch >= 2:
        for b in train:
            f = tr.features(b['clean'], b['noise'], b['lengths'])
            feats.append(f)
            tr.g_step(f['clean_band'], f['noise_band'], f['frames'])
    raw = []
    for b in valid:
        f = tr.features(b['clean'], b['noise'], b['lengths'])
        enh = tr.generate(f['clean_band'], f['noise_band'], f['clean_spec'], frames=f['frames'])
        raw.append(tr.true_metrics(b['clean'], enh, b['noise'], norm=False, lengths=b['lengths']))
    samples, enhs = [], []
    for i, b in enumerate(train):
        f = feats[i] if feats else tr.features(b['clean'], b['noise'], b['lengths'])
        enh = tr.generate(f['clean_band'], f['noise_band'], f['clean_spec'], frames=f['frames'])
        enhs.append(enh)
        tgt = tr.true_metrics(b['clean'], enh, b['noise'], lengths=b['lengths'])
        din = tr.d_inputs(enh, f['noise_band'], f['clean_band'], b['lengths'])
        fr = f['frames'].tolist()
        samples += [(din[k, :, :fr[k]].contiguous(), tgt[k], None) for k in range(din.shape[0])]
        ml = torch.minimum(b['drc_lengths'], b['lengths'])
        tgt_d = tr.true_metrics(b['clean'], b['drc'], b['noise'], lengths=ml, resynth=False)
        din_d = tr.d_inputs(b['drc'], f['noise_band'], f['clean_band'], b['drc_lengths'], resynth=False)
        samples += [(din_d[k, :, :fr[k]].contiguous(), tgt_d[k], None) for k in range(din_d.shape[0])]
    tr.d_epoch(samples, batch=3)
    return torch.cat(raw), samples, enhs


def test_run_epoch_on_the_device_with_utterances_of_different_lengths(tmp_path):
    from nele_gan_amd import dataio
    from nele_gan_amd.train_nele import GanTrainer
    from oracle import step as ostep
    train, valid = _corpus()
    a = GanTrainer('siib&haspi&estoi')
    b = GanTrainer('siib&haspi&estoi')                                # same seed: same weights, same shuffles
    log = str(tmp_path / 'log.txt')
    for ep in (1, 2):
        ck = str(tmp_path / ('chkpt_%d.pt' % ep))
        random.seed(1000 + ep)                                        # the D passes shuffle with the global generator (train_nele.py:347, 377)
        out = a.run_epoch(ep, train, valid, chkpt_path=ck, sample_dir=str(tmp_path / 'out'), log_path=log, d_batch=3)
        random.seed(1000 + ep)
        raw, samples, enhs = _by_hand_epoch(b, ep, train, valid)
        # bookkeeping (train_nele.py:122, 342-426): no G-step in epoch 1; 8 samples (4 generated + 4 pre-enhanced);
        # passes A, B, C over ceil(8/3) = 3 batches each, pass B with len(history) // 30 = 0 replayed items
        assert out['g_steps'] == (0 if ep == 1 else 2) and out['samples'] == 8 and out['d_steps'] == 9
        assert len(a.history) == 8 * ep and a.step_d == b.step_d == 9 * ep and a.step_g == b.step_g
        assert out['status']['skipped_d_steps'] == 0 and out['status']['skipped_g_steps'] == 0
        # same results as the stages called by hand, bit for bit: weights of G and D after the epoch, validation means
        for mod_a, mod_b in ((a.G, b.G), (a.D, b.D)):
            for (k, ta), (_, tb) in zip(mod_a.state_dict().items(), mod_b.state_dict().items()):
                assert torch.equal(ta, tb), k
        means = raw.double().mean(dim=0).cpu().numpy()
        assert [out['valid'][m] for m in ('siib', 'haspi', 'estoi')] == pytest.approx(list(means), rel=1e-12)
        # the learning-curve line (train_nele.py:224-225)
        line = open(log).read().splitlines()[-1]
        assert line == ('SIIB is %.3f, HASPI is %.3f, ESTOI is %.3f, PESQ is %.3f, VISQOL is %.3f, EPOCH:%d ' % (means[0], means[1], means[2], 0, 0, ep))
        # checkpoint keys (train_nele.py:272-277) load into the reference-shaped modules
        sd = torch.load(ck, map_location='cpu')
        assert set(sd.keys()) == {'enhance-model', 'intel-model'}
        assert set(sd['enhance-model'].keys()) == set(a.G.state_dict().keys())
        # name@epoch.wav files (train_nele.py:309-313): PCM_16, 256 * (L // 256) samples, equal to the generated batch rows
        files = out['sample_files']
        assert [os.path.basename(p) for p in files] == ['Train_Clean@%d.wav' % ep, 'syn_a@%d.wav' % ep, 'syn_b@%d.wav' % ep, 'syn_c@%d.wav' % ep]
        k = 0
        for bi, bt in enumerate(train):
            for r, L in enumerate(bt['lengths'].tolist()):
                w, sr = dataio.load(files[k])
                assert sr == 16000 and len(w) == 256 * (L // 256)
                np.testing.assert_array_equal(w, enhs[bi][r, :len(w)].cpu().numpy())
                assert float(enhs[bi][r, len(w):].abs().sum()) == 0.0
                k += 1
        assert os.path.exists(str(tmp_path / 'out' / ('Test_epoch%d' % ep) / ('Test_Clean@%d.wav' % ep)))
    # and against the oracle: the targets of the first generated sample and of the toy pre-enhanced example (the only reference-supplied
    # enhanced file) in the last epoch's list, computed per file at the file's own length
    c0 = train[0]['clean'][0, :33536].cpu().numpy()
    n0 = train[0]['noise'][0, :33536].cpu().numpy()
    e0 = enhs[0][0, :256 * (33536 // 256)].cpu().numpy()
    ref = ostep.metric_targets(c0, e0, n0, ('siib', 'haspi', 'estoi'))
    np.testing.assert_allclose(samples[0][1].cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    drc0 = train[0]['drc'][0, :33536].cpu().numpy()
    ref_d = ostep.metric_targets(c0, drc0, n0, ('siib', 'haspi', 'estoi'))
    np.testing.assert_allclose(samples[2][1].cpu().numpy(), ref_d, rtol=1e-4, atol=1e-5)


def test_run_epoch_with_the_quality_discriminator(tmp_path):
    """use_quality=True through the epoch driver: quality targets travel with the batches ('qua' / 'drc_qua'), D_Qua takes one
    optimiser step per D step (train_nele.py:362-365), and the checkpoint carries 'quality-model'."""
    from nele_gan_amd.train_nele import GanTrainer
    train, _ = _corpus()
    for bt in train:
        bt['qua'] = torch.tensor([[0.4, 0.5], [0.6, 0.3]], device='cuda')
        bt['drc_qua'] = torch.tensor([[0.7, 0.8], [0.2, 0.9]], device='cuda')
    tr = GanTrainer('estoi', use_quality=True)
    w0 = tr.D_Qua.layers[4].weight_orig.detach().clone()
    ck = str(tmp_path / 'c.pt')
    out = tr.run_epoch(2, train, (), chkpt_path=ck, d_batch=4)
    assert out['g_steps'] == 2 and out['d_steps'] == 6 and tr.optimizer_dqua.step_count == 6
    assert not torch.equal(w0, tr.D_Qua.layers[4].weight_orig.detach())
    assert 'quality-model' in torch.load(ck, map_location='cpu')
    # mixed presence of quality targets inside one pass is an error, not a silent skip
    train[1].pop('qua')
    with pytest.raises(ValueError):
        tr.run_epoch(3, train, ())


def test_run_epoch_scores_the_quality_targets_with_the_registered_programs(tmp_path):
    """train_nele.py:216-222, 323-324, 336-337: with a quality.Scorer the epoch driver itself obtains D_Qua's targets - PESQ / ViSQOL of the
    GENERATED file's samples (PCM_16, 256 * (L // 256)) against the clean file cut to them, of the pre-enhanced example over
    min(clean, pre-enhanced), mapped - and the raw validation means.  PESQ and ViSQOL are external programs that are not in the image:
    STAND-INS are registered here (functions of both signals), which shows the plumbing, not a score."""
    from nele_gan_amd import quality
    from nele_gan_amd.train_nele import GanTrainer
    seen = []

    def pesq(ref, deg, fs):
        assert fs == 16000 and len(ref) == len(deg) and ref.dtype == np.float32
        seen.append((ref.copy(), deg.copy()))
        r, d = ref.astype(np.float64), deg.astype(np.float64)
        return 1.0 + 3.5 * abs(float(r @ d)) / (np.sqrt(float(r @ r) * float(d @ d)) + 1e-30)

    def visqol(pairs):
        from nele_gan_amd import dataio
        out = []
        for rp, dp in pairs:
            x, y = dataio.load(rp)[0].astype(np.float64), dataio.load(dp)[0].astype(np.float64)
            n = min(len(x), len(y))
            out.append(1.0 + 4.0 / (1.0 + 50.0 * float(np.abs(x[:n] - y[:n]).mean())))
        return out
    quality.clear_backends()
    quality.set_backends(pesq=pesq, visqol=visqol)
    try:
        train, valid = _corpus()
        tr = GanTrainer('estoi', use_quality=True, quality_scorer=quality.Scorer(tmp_root=str(tmp_path)))
        # epoch 1 has no G-step: the generated examples are those of the untrained generator, computable beforehand
        want = []
        for b in train:
            f = tr.features(b['clean'], b['noise'], b['lengths'])
            enh = tr.generate(f['clean_band'], f['noise_band'], f['clean_spec'], frames=f['frames']).cpu().numpy()
            cl, dr = b['clean'].cpu().numpy(), b['drc'].cpu().numpy()
            for k, (L, Ld) in enumerate(zip(b['lengths'].tolist(), b['drc_lengths'].tolist())):
                n = 256 * (L // 256)
                want.append(('gen', cl[k, :n], enh[k, :n]))
                m = min(L, Ld)
                want.append(('drc', cl[k, :m], dr[k, :m]))
        out = tr.run_epoch(1, train, valid, d_batch=4)
        assert out['samples'] == 8 and out['d_steps'] == 6 and tr.optimizer_dqua.step_count == 6
        assert np.isfinite(float(tr.last_loss_qua))
        # 2 validation + 4 generated + 4 pre-enhanced utterances went to the programs, with exactly the samples the reference compares
        assert len(seen) == 10
        for kind, ref, deg in want:
            assert any(len(r) == len(ref) and np.array_equal(r, ref) and np.array_equal(d, deg) for r, d in seen), kind
        # the D_Qua items carry the mapped scores
        items = {}
        for it in tr.history:
            items[tuple(np.round(it[2].cpu().numpy().astype(np.float64), 6))] = True
        for kind, ref, deg in want:
            p = pesq(ref, deg, 16000)
            v = 1.0 + 4.0 / (1.0 + 50.0 * float(np.abs(ref.astype(np.float64) - deg.astype(np.float64)).mean()))
            key = (round(float(quality.mapping_PESQ_harvard(p)), 6), round(float(quality.mapping_VISQOL(v)), 6))
            assert any(abs(a - key[0]) < 2e-6 and abs(b_ - key[1]) < 2e-6 for a, b_ in items), (kind, key)
        # raw validation means for the learning curve (Test_PESQ / Test_VISQOL)
        assert 1.0 <= out['valid']['pesq'] <= 4.5 and 1.0 <= out['valid']['visqol'] <= 5.0
        # targets that come with the batch win: nothing is scored for that batch's generated examples
        n0 = len(seen)
        train[0]['qua'] = torch.tensor([[0.4, 0.5], [0.6, 0.3]], device='cuda')
        train[0]['drc_qua'] = torch.tensor([[0.7, 0.8], [0.2, 0.9]], device='cuda')
        tr.run_epoch(2, train, (), d_batch=4)
        assert len(seen) - n0 == 4
        # and without the programs the call fails loudly instead of training D_Qua on nothing
        quality.clear_backends()
        with pytest.raises(quality.QualityBackendMissing):
            tr.run_epoch(3, train, (), d_batch=4)
    finally:
        quality.clear_backends()


def test_inference_on_eight_second_utterances_matches_the_oracle():
    """BASELINE configs[4]: 8 s utterances (L = 128 000, T = 501: the IMCRA scan over 501 frames, G, iSTFT, RMS 0.03, PCM_16)."""
    from nele_gan_amd import synth
    from nele_gan_amd.inference import Enhancer
    from oracle.step import CpuStep
    B, L = 4, 128000
    c, v = synth.batch(B, L, start=900)
    torch.manual_seed(3)
    e = Enhancer()
    cpu = CpuStep(_sd(e.G), None, metrics=('estoi',))
    out = e.enhance(torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda(), pcm16=True).cpu().numpy()
    ref = cpu.enhance(c, v)
    assert out.shape == (B, 256 * (L // 256))
    for b in range(B):
        d = np.abs(out[b] - ref[b])
        assert d.max() <= 1.5 / 32768 and np.mean(d > 1e-7) < 0.02     # PCM_16: rare one-LSB rounding flips (as test_train_gpu.py:34-36)
        assert np.sqrt(np.mean(out[b].astype(np.float64) ** 2)) == pytest.approx(0.03, rel=2e-4)


@pytest.mark.parametrize('inflight', [1, 3, 4])
def test_enhance_stream_is_bit_identical_to_enhance(inflight):
    """inference.py:79-117 is a loop over independent files: Enhancer.enhance_stream keeps several batches in flight on their own streams
    (own generator activation buffers, weight layouts written once) and must return, in order, exactly what enhance() returns."""
    from nele_gan_amd import synth
    from nele_gan_amd.inference import Enhancer
    torch.manual_seed(5)
    e = Enhancer()
    e.G.precision = 'bf16'
    shapes = [(3, 40000, None), (2, 64000, [64000, 50000]), (3, 40000, None), (4, 128000, None), (2, 64000, [33536, 64000]), (3, 40000, None),
              (4, 128000, None)]
    batches = []
    for k, (B, L, lens) in enumerate(shapes):
        c, v = synth.batch(B, L, start=100 * k)
        item = (torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda())
        if lens is not None:
            item = item + (torch.tensor(lens, dtype=torch.int32),)
        batches.append(item)
    ref = [e.enhance(b[0], b[1], lengths=b[2] if len(b) > 2 else None).clone() for b in batches]
    # a delayed first stream must not change anything either (the spin parks 20 ms of idle work in front of slot 0's kernels)
    outs = []
    for k, o in enumerate(e.enhance_stream(iter(batches), inflight=inflight)):
        outs.append(o)
    assert len(outs) == len(ref)
    for o, r in zip(outs, ref):
        assert o.shape == r.shape and torch.equal(o, r)
    assert e.G.buffer_slot == 0 and e.G._weights_frozen         # an Enhancer that owns its generator keeps the layouts it wrote
    # the generator trains again afterwards: a training pass unfreezes (weight layouts are rewritten per forward pass), the next enhance()
    # freezes the new weights
    e.G.train()
    m = e.G(torch.rand(1, 30, 64, device='cuda'), torch.rand(1, 30, 64, device='cuda'))
    assert not e.G._weights_frozen
    m.sum().backward()
    e.G.eval()
    b = batches[0]
    again = e.enhance(b[0], b[1])
    assert e.G._weights_frozen and torch.equal(again, ref[0])   # (no optimiser step happened: same weights, same result)


def test_enhance_stream_survives_a_consumer_that_stops_early():
    from nele_gan_amd import synth
    from nele_gan_amd.inference import Enhancer
    torch.manual_seed(5)
    e = Enhancer()
    c, v = synth.batch(2, 40000, start=7)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    ref = e.enhance(cw, nw)
    gen = e.enhance_stream([(cw, nw)] * 6, inflight=3)
    first = next(gen)
    gen.close()                                                          # GeneratorExit inside the loop: pending batches are waited for
    assert torch.equal(first, ref) and e.G._weights_frozen and e.G.buffer_slot == 0      # (its own generator: the layouts stay)
    assert torch.equal(e.enhance(cw, nw), ref)
    # a generator handed in (a trainer's) is given back unfrozen - it is the trainer's to change
    e2 = Enhancer(G=e.G)
    e.G.unfreeze_weights()
    gen = e2.enhance_stream([(cw, nw)] * 6, inflight=3)
    first = next(gen)
    gen.close()
    assert torch.equal(first, ref) and not e.G._weights_frozen and e.G.buffer_slot == 0


def test_fit_runs_the_scripts_outer_loop_over_folders(tmp_path):
    """GanTrainer.fit = `for gan_epoch in np.arange(1, GAN_epoch+1)` of train_nele.py:110-428 over folders of wav files: per epoch a shuffled
    draw of the training list, validation with the learning-curve line, chkpt_<epoch>.pt, name@epoch.wav samples, targets of the generated and
    the pre-enhanced examples, three D passes - here with the clean-signal cache on, against a second trainer that recomputes everything."""
    from nele_gan_amd import dataio, synth
    from nele_gan_amd.train_nele import GanTrainer
    root = str(tmp_path)
    rs = np.random.RandomState(3)
    c, v = synth.batch(10, 40000, start=300)
    for sub in ('Train/Clean', 'Train/Noise', 'Train/MultiEnh', 'Test/Clean', 'Test/Noise'):
        os.makedirs(os.path.join(root, sub))
    for i in range(10):
        L = int(rs.randint(30000, 40001))
        part = 'Train' if i < 7 else 'Test'
        dataio.write_wav_pcm16('%s/%s/Clean/u%02d.wav' % (root, part, i), c[i, :L])
        dataio.write_wav_pcm16('%s/%s/Noise/u%02d.wav' % (root, part, i), v[i, :L])
        if i < 7:
            dataio.write_wav_pcm16('%s/Train/MultiEnh/u%02d.wav' % (root, i), (1.4 * c[i, :L]).astype(np.float32))
    train = sorted(dataio.get_filepaths(root + '/Train/Clean/'))
    test = sorted(dataio.get_filepaths(root + '/Test/Clean/'))

    def run(cache, tag):
        tr = GanTrainer('siib&haspi&estoi', seed=666)
        res = tr.fit(train, root + '/Train/Noise/', test, root + '/Test/Noise/', train_enh_path=root + '/Train/MultiEnh/', epochs=3, sampling=5, valid_samples=2, batch=4,
                     output_path=root + '/out_' + tag, pt_dir=root + '/chkpt_' + tag, log_path=root + '/log_%s.txt' % tag, clean_cache=cache)
        return tr, res
    a, ra = run(True, 'a')
    b, rb = run(False, 'b')
    assert [r['g_steps'] for r in ra] == [0, 2, 2] and all(r['samples'] == 10 for r in ra)          # 5 drawn files: generated + pre-enhanced
    assert a.clean_cache.hits > 0
    for (k, ta), (_, tb) in zip(a.G.state_dict().items(), b.G.state_dict().items()):                 # the cache changes nothing
        assert torch.equal(ta, tb), k
    for (k, ta), (_, tb) in zip(a.D.state_dict().items(), b.D.state_dict().items()):
        assert torch.equal(ta, tb), k
    assert [r['valid'] for r in ra] == [r['valid'] for r in rb]
    lines = open(root + '/log_a.txt').read().splitlines()
    assert len(lines) == 3 and lines[2].startswith('SIIB is ') and lines[2].rstrip().endswith('EPOCH:3')
    for ep in (1, 2, 3):
        assert set(torch.load(root + '/chkpt_a/chkpt_%d.pt' % ep, map_location='cpu').keys()) == {'enhance-model', 'intel-model'}
        assert len(os.listdir(root + '/out_a/Test_epoch%d' % ep)) == 2
    files = os.listdir(root + '/out_a/For_discriminator_training')
    assert len(files) == 15 and all('@' in f for f in files)


def test_the_two_scripts_run_as_modules_on_a_toy_corpus(tmp_path):
    """The reference's usage steps 3 and 4 (README: `python train_nele.py`, `python inference.py`) as `python -m nele_gan_amd.train_nele`
    and `python -m nele_gan_amd.inference` over a corpus laid out like ./toy_dataset - the toy files themselves plus synthetic utterances."""
    import shutil
    import subprocess
    import sys
    from nele_gan_amd import dataio, synth
    root = str(tmp_path / 'toy')
    for sub in ('Train/Clean', 'Train/Noise', 'Train/MultiEnh', 'Test/Clean', 'Test/Noise'):
        os.makedirs(os.path.join(root, sub))
    name = 'f_hvd_100#Babble#-11.wav'
    shutil.copy(os.path.join(TOY, 'Train_Clean.wav'), root + '/Train/Clean/' + name)
    shutil.copy(os.path.join(TOY, 'Train_Noise.wav'), root + '/Train/Noise/' + name)
    shutil.copy(os.path.join(TOY, 'Train_MultiEnh.wav'), root + '/Train/MultiEnh/' + name)
    shutil.copy(os.path.join(TOY, 'Test_Clean.wav'), root + '/Test/Clean/f_hvd_669#AirportAnnouncement#-9.wav')
    shutil.copy(os.path.join(TOY, 'Test_Noise.wav'), root + '/Test/Noise/f_hvd_669#AirportAnnouncement#-9.wav')
    c, v = synth.batch(4, 36000, start=40)
    for i in range(3):
        nm = 'syn_%d#Cafeteria#-5.wav' % i
        dataio.write_wav_pcm16(root + '/Train/Clean/' + nm, c[i])
        dataio.write_wav_pcm16(root + '/Train/Noise/' + nm, v[i])
        dataio.write_wav_pcm16(root + '/Train/MultiEnh/' + nm, (1.3 * c[i]).astype(np.float32))
    dataio.write_wav_pcm16(root + '/Test/Clean/syn_3#Cafeteria#-5.wav', c[3])
    dataio.write_wav_pcm16(root + '/Test/Noise/syn_3#Cafeteria#-5.wav', v[3])
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=repo + os.pathsep + os.environ.get('PYTHONPATH', ''))
    out, ck, log = str(tmp_path / 'output'), str(tmp_path / 'chkpt'), str(tmp_path / 'log.txt')
    r = subprocess.run([sys.executable, '-m', 'nele_gan_amd.train_nele', '--data', root, '--epochs', '2', '--sampling', '3', '--valid-samples', '2',
                        '--batch', '2', '--output', out, '--chkpt', ck, '--log', log], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert '4 training files, 2 validation files' in r.stdout and 'epoch 2:' in r.stdout
    assert len(open(log).read().splitlines()) == 2
    sd = torch.load(ck + '/chkpt_2.pt', map_location='cpu')
    assert set(sd.keys()) == {'enhance-model', 'intel-model'}
    assert len(os.listdir(out + '/For_discriminator_training')) == 6                                # 3 drawn files x 2 epochs: name@epoch.wav
    enh = str(tmp_path / 'enh')
    r = subprocess.run([sys.executable, '-m', 'nele_gan_amd.inference', '--chkpt', ck + '/chkpt_2.pt', '--clean', root + '/Test/Clean', '--noise', root + '/Test/Noise',
                        '--output', enh, '--score'], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sorted(os.listdir(enh)) == ['f_hvd_669#AirportAnnouncement#-9@1.wav', 'syn_3#Cafeteria#-5@1.wav']
    lines = r.stdout.splitlines()
    assert 'Cafeteria:' in lines and 'AirportAnnouncement:' in lines and sum(l.startswith('SIIB is ') for l in lines) == 2
    w, sr = dataio.load(enh + '/syn_3#Cafeteria#-5@1.wav')
    assert sr == 16000 and len(w) == 256 * (36000 // 256) and np.sqrt(np.mean(w.astype(np.float64) ** 2)) == pytest.approx(0.03, rel=2e-3)


def test_an_enhancer_that_owns_its_generator_keeps_the_weight_layouts_and_notices_new_weights(tmp_path):
    """inference.py:71-72: the generator is loaded from a checkpoint once; plain enhance() then writes the weight layouts once instead of per
    batch - and a later load_state_dict() must not be served from stale layouts."""
    from nele_gan_amd import model, synth
    from nele_gan_amd.inference import Enhancer
    c, v = synth.batch(3, 40000, start=4000)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    torch.manual_seed(1)
    Ga = model.Generator_Conv1D_cLN()
    torch.manual_seed(2)
    Gb = model.Generator_Conv1D_cLN()
    pa, pb = str(tmp_path / 'a.pt'), str(tmp_path / 'b.pt')
    torch.save({'enhance-model': Ga.state_dict()}, pa)
    torch.save({'enhance-model': Gb.state_dict()}, pb)
    for prec in ('f32', 'bf16'):
        own = Enhancer(pa)
        own.G.precision = prec
        shared = Enhancer(G=Ga.cuda())
        shared.G.precision = prec
        a1 = own.enhance(cw, nw)
        assert own.G._weights_frozen
        a2 = own.enhance(cw, nw)                                  # second call: no weight-layout launches, same result
        assert torch.equal(a1, a2) and torch.equal(a1, shared.enhance(cw, nw))
        assert not shared.G._weights_frozen                       # a generator handed in may be trained between calls
        own.G.load_state_dict(torch.load(pb, map_location='cpu')['enhance-model'])
        assert not own.G._weights_frozen
        want = Enhancer(pb)
        want.G.precision = prec
        assert torch.equal(own.enhance(cw, nw), want.enhance(cw, nw))
        outs = list(own.enhance_stream([(cw, nw)] * 3, inflight=2))
        assert all(torch.equal(o, outs[0]) for o in outs) and torch.equal(outs[0], want.enhance(cw, nw)) and own.G._weights_frozen
