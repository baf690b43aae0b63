"""GPU: metric fan-out and D training items read from reference-format folders (wav PCM_16, name@epoch, score lists)."""
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
HERE = os.path.dirname(__file__)
TOY = os.path.join(HERE, 'golden', 'toy')


@pytest.fixture(scope='module')
def tree(tmp_path_factory):
    """Clean/ Noise/ Enhanced/ folders as train_nele.py lays them out: two source files of different lengths, the
    enhanced versions written as PCM_16 under name@epoch.wav."""
    from nele_gan_amd import dataio
    assert torch.cuda.is_available()
    root = tmp_path_factory.mktemp('set')
    for d in ('Clean', 'Noise', 'Enh'):
        (root / d).mkdir()
    names = []
    for split in ('Train', 'Test'):
        shutil.copy(os.path.join(TOY, split + '_Clean.wav'), root / 'Clean' / (split + '.wav'))
        shutil.copy(os.path.join(TOY, split + '_Noise.wav'), root / 'Noise' / (split + '.wav'))
        c, _ = dataio.load(str(root / 'Clean' / (split + '.wav')))
        enh = (c * np.float32(1.7))[:len(c) // 256 * 256]                 # stands for a resynthesised signal (shorter than the clean file)
        for ep in (1, 2):
            p = dataio.enhanced_name(str(root / 'Enh'), split + '.wav', ep)
            dataio.write_wav_pcm16(p, enh * np.float32(ep))
            names.append(p)
    return str(root / 'Clean') + '/', str(root / 'Noise') + '/', names


def test_read_batch_metrics_from_files_vs_oracle(tree):
    from nele_gan_amd import dataio
    from oracle import step
    clean_root, noise_root, names = tree
    got = {
        ('estoi', True): dataio.read_batch_STOI(clean_root, noise_root, names, norm=True),
        ('estoi', False): dataio.read_batch_STOI(clean_root, noise_root, names, norm=False),
        ('siib', True): dataio.read_batch_SIIB(clean_root, noise_root, names, norm=True),
    }
    for (m, norm), vals in got.items():
        assert len(vals) == len(names) and all(isinstance(v, float) for v in vals)
        for en, v in zip(names, vals):
            name = dataio.wave_name_of(en) + '.wav'
            c, _ = dataio.load(clean_root + name)
            n, _ = dataio.load(noise_root + name)
            e, _ = dataio.load(en)
            ref = step.metric_targets(c, e, n, [m], norm=norm)[0]
            assert v == pytest.approx(ref, rel=2e-4, abs=2e-4)


def test_read_batch_from_device_side_triples_equals_the_per_file_readers(tree, monkeypatch):
    """read_batch_*: PCM_16 files are read a group per library call and paired / mixed on the GPU; the same list read file by file on the
    host (the path other wav flavours take) gives the same pairs bit for bit and the same scores."""
    from nele_gan_amd import dataio
    clean_root, noise_root, names = tree
    x, y, lens = dataio._triples_on_device(clean_root, noise_root, names, False)
    for k, en in enumerate(names):
        c, e = dataio._triple(clean_root, noise_root, en, False)
        assert lens[k] == len(c)
        np.testing.assert_array_equal(x[k, :lens[k]].cpu().numpy(), c)
        np.testing.assert_array_equal(y[k, :lens[k]].cpu().numpy(), e)
        assert not x[k, lens[k]:].any() and not y[k, lens[k]:].any()
    fast = dataio.read_batch_SIIB(clean_root, noise_root, names, norm=False)
    monkeypatch.setattr(dataio, '_triples_on_device', lambda *a, **k: None)
    slow = dataio.read_batch_SIIB(clean_root, noise_root, names, norm=False)
    assert fast == pytest.approx(slow, rel=1e-9)


def test_drc_variant_uses_the_enhanced_files_own_name(tree, tmp_path):
    from nele_gan_amd import dataio
    clean_root, noise_root, names = tree
    p = str(tmp_path / 'Train.wav')                                      # audio_util.py:267-284: same file name as the clean file
    shutil.copy(names[0], p)
    a = dataio.read_batch_STOI_DRC(clean_root, noise_root, [p])
    b = dataio.read_batch_STOI(clean_root, noise_root, [names[0]], norm=True)
    assert a == b


def test_discriminator_items_from_a_score_list_vs_oracle(tree):
    from nele_gan_amd import dataio
    from oracle import features as F
    clean_root, noise_root, names = tree
    scores = [[0.1 * (i + 1) for i in range(len(names))] for _ in range(5)]
    lines = dataio.List_concat(dataio.List_concat_5scores(*scores), names)
    ds = dataio.Discriminator_train_dataset(lines, noise_root, clean_root)
    assert len(ds) == len(names)
    x3, x2, s, q = ds[2]
    e, _ = dataio.load(names[2])
    name = dataio.wave_name_of(names[2]) + '.wav'
    c, _ = dataio.load(clean_root + name)
    n, _ = dataio.load(noise_root + name)
    Te, Tc = 1 + len(e) // 256, 1 + len(c) // 256
    assert x3[0].shape == (64, Te) and x3.shape[0] == 3 if Te == Tc else True
    np.testing.assert_allclose(x2[0].cpu().numpy(), F.sp_and_phase_speech(e, 1 / 6)[0].T, rtol=2e-5)
    np.testing.assert_allclose(x2[1].cpu().numpy(), F.sp_and_phase_speech(c, 1 / 6)[0].T, rtol=2e-5)
    np.testing.assert_allclose(x3[1].cpu().numpy(), F.sp_and_phase_noise(n, 1 / 6)[0].T, rtol=1e-4)
    assert list(s) == pytest.approx([0.3] * 3) and list(q) == pytest.approx([0.3] * 2)
    batches = list(dataio.create_dataloader(lines, noise_root, clean_root, loader='D', seed=0))
    assert len(batches) == len(names) and batches[0][0].shape[:3] == (1, 3, 64) and batches[0][2].shape == (1, 3)
    g = dataio.create_dataloader([clean_root + 'Train.wav'], noise_root, loader='G')
    item = next(iter(g))
    assert item[0].shape[0] == 1 and item[0].shape[2] == 64 and item[8] == ['Train.wav']


def test_generated_samples_through_files_give_the_in_memory_targets(tmp_path):
    """generate -> PCM_16 files -> read_batch_* (the reference's hand-off, train_nele.py:303-340) must give the targets the
    in-HBM path computes from the same batch, and the D items parsed back from the score list must be the D inputs."""
    from nele_gan_amd import dataio, synth
    from nele_gan_amd.train_nele import GanTrainer
    B, L = 3, 24000
    c, v = synth.batch(B, L, start=40)
    names = ['utt%d.wav' % i for i in range(B)]
    clean_root, noise_root = str(tmp_path / 'Clean') + '/', str(tmp_path / 'Noise') + '/'
    for d in (clean_root, noise_root):
        dataio.creatdir(d)
    # the sources themselves must be PCM_16-exact for the two paths to see identical samples
    q = lambda a: (np.rint(a * 32767).clip(-32768, 32767) / 32768).astype(np.float32)
    c, v = q(c), q(v)
    for i, n in enumerate(names):
        dataio.write_wav_pcm16(clean_root + n, c[i], quantised=True)
        dataio.write_wav_pcm16(noise_root + n, v[i], quantised=True)
    tr = GanTrainer('siib&estoi')
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    f = tr.features(cw, nw)
    enh = tr.generate(f['clean_band'], f['noise_band'], f['clean_spec'])
    tgt = tr.true_metrics(cw, enh, nw).cpu().numpy()
    files = tr.write_samples(enh, names, str(tmp_path / 'out' / 'temp'), 7)
    assert files[0].endswith('/temp/utt0@7.wav')
    back, _ = dataio.load(files[1])
    assert np.array_equal(back, enh[1].cpu().numpy())                        # bit-exact through the file
    # the background batch writer (run_epoch's path) and the per-file Python writer store the same bytes, lengths cut to whole hops
    lens = [L, L - 300, L - 1000]
    bg = tr.write_samples(enh, names, str(tmp_path / 'out' / 'bg'), 7, lengths=lens, wait=False)
    tr.flush_writes()
    for k, pth in enumerate(bg):
        ref_p = str(tmp_path / 'out' / ('ref%d.wav' % k))
        dataio.write_wav_pcm16(ref_p, enh[k, :256 * (lens[k] // 256)].cpu().numpy(), quantised=True)
        assert open(pth, 'rb').read() == open(ref_p, 'rb').read()
    siib = dataio.read_batch_SIIB(clean_root, noise_root, files, norm=True)
    estoi = dataio.read_batch_STOI(clean_root, noise_root, files, norm=True)
    np.testing.assert_allclose(np.stack([siib, estoi], 1), tgt, rtol=1e-6)
    lines = tr.score_lines(torch.from_numpy(tgt), files)
    intel, qua, path = dataio.parse_score_line(lines[2])
    assert path == files[2] and intel[0] == pytest.approx(tgt[2, 0]) and intel[2] == pytest.approx(tgt[2, 1]) and intel[1] == 0
    ds = dataio.Discriminator_train_dataset(lines, noise_root, clean_root)
    x3 = ds[2][0]                                                            # [3, 64, T] = enh, noise, clean
    din = tr.d_inputs(enh, f['noise_band'], f['clean_band'])                 # packed [B, 64, T, 4]
    np.testing.assert_allclose(x3.permute(1, 2, 0).cpu().numpy(), din[2, :, :, :3].float().cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert 'EPOCH:7' in tr.validation_log_line(siib, [0.0], estoi, 7)


@pytest.mark.parametrize('int16', [True, False])
def test_file_batches_prefetching_loader(tree, int16):
    """dataio.FileBatches: the corpus on disk as run_epoch's batch dicts - threaded decode, pinned staging, asynchronous upload, batches
    decoded ahead, re-decoded after eviction (the reference re-reads its files in every stage, dataloader.py:30-42).  Contents equal
    plain loads of the same files, in list order, whatever the access pattern."""
    from nele_gan_amd import dataio
    clean_root, noise_root, _ = tree
    files = [clean_root + 'Train.wav', clean_root + 'Test.wav', clean_root + 'Train.wav', clean_root + 'Test.wav', clean_root + 'Train.wav']
    fb = dataio.FileBatches(files, noise_root, batch=2, workers=3, ahead=1, keep=1, pad_to=4096, drc_path=clean_root, int16=int16)
    assert len(fb) == 3

    def check(b, idx):
        assert b['names'] == [files[i].split('/')[-1] for i in idx]
        assert b['clean'].shape == b['noise'].shape and b['clean'].shape[1] % 4096 == 0 and b['clean'].is_cuda
        for r, i in enumerate(idx):
            c, _ = dataio.load(files[i])
            n, _ = dataio.load(noise_root + files[i].split('/')[-1])
            m = min(len(c), len(n))
            assert int(b['lengths'][r]) == m and int(b['drc_lengths'][r]) == len(c)
            np.testing.assert_array_equal(b['clean'][r, :m].cpu().numpy(), c[:m])
            np.testing.assert_array_equal(b['noise'][r, :m].cpu().numpy(), n[:m])
            np.testing.assert_array_equal(b['drc'][r, :len(c)].cpu().numpy(), c)
            assert not b['clean'][r, m:].any() and not b['noise'][r, m:].any()

    for g, idx in ((0, [0, 1]), (1, [2, 3]), (2, [4]), (0, [0, 1]), (2, [4]), (-1, [4])):     # second pass and random access: re-decoded after eviction
        check(fb[g], idx)
    assert [b['names'] for b in fb] == [['Train.wav', 'Test.wav'], ['Train.wav', 'Test.wav'], ['Train.wav']]
    assert fb.decoded_files >= 9
    with pytest.raises(IndexError):
        fb[3]
    fb.close()


def test_file_batches_with_another_wav_flavour_in_a_batch(tree, tmp_path):
    """A batch that holds a float32 wav cannot go through the int16 staging rows: that batch (only) takes the per-file float32 path;
    the contents are what plain loads give either way."""
    import shutil
    import struct
    from nele_gan_amd import dataio
    clean_root, noise_root, _ = tree
    croot, nroot = str(tmp_path / 'c') + '/', str(tmp_path / 'n') + '/'
    os.makedirs(croot), os.makedirs(nroot)
    for nm in ('Train.wav', 'Test.wav'):
        shutil.copy(clean_root + nm, croot + nm)
        shutil.copy(noise_root + nm, nroot + nm)
    x, _ = dataio.load(clean_root + 'Train.wav')
    body = x[:5000].astype('<f4').tobytes()
    for root in (croot, nroot):
        with open(root + 'F32.wav', 'wb') as f:
            f.write(b'RIFF' + struct.pack('<I', 36 + len(body)) + b'WAVE' + b'fmt ' + struct.pack('<IHHIIHH', 16, 3, 1, 16000, 64000, 4, 32) + b'data'
                    + struct.pack('<I', len(body)) + body)
    files = [croot + 'Train.wav', croot + 'F32.wav', croot + 'Test.wav']
    fb = dataio.FileBatches(files, nroot, batch=2, workers=2, ahead=1, keep=2)
    b0, b1 = fb[0], fb[1]
    assert int(b0['lengths'][1]) == 5000
    np.testing.assert_array_equal(b0['clean'][1, :5000].cpu().numpy(), x[:5000])
    c0, _ = dataio.load(croot + 'Train.wav')
    n0, _ = dataio.load(nroot + 'Train.wav')
    m = min(len(c0), len(n0))
    np.testing.assert_array_equal(b0['clean'][0, :m].cpu().numpy(), c0[:m])
    np.testing.assert_array_equal(b0['noise'][0, :m].cpu().numpy(), n0[:m])
    c2, _ = dataio.load(croot + 'Test.wav')
    np.testing.assert_array_equal(b1['clean'][0, :int(b1['lengths'][0])].cpu().numpy(), c2[:int(b1['lengths'][0])])
    fb.close()


def test_pcm16_conversion_kernels():
    """nele_pcm16_to_float: s / 32768 up to lengths[b], zeros behind (any stride / alignment); nele_float_to_pcm16: the integer of a
    device-quantised sample exactly, libsndfile's lrintf(x * 32767) rule (ties to even, saturation) otherwise."""
    from nele_gan_amd import _lib
    rng = np.random.default_rng(11)
    for B, L, stride in ((3, 4096, 4096), (2, 1001, 1003), (1, 7, 8)):
        q = rng.integers(-32768, 32768, size=(B, stride)).astype(np.int16)
        lens = rng.integers(0, L + 1, size=B).astype(np.int32)
        lens[0] = L
        dq, dl = torch.from_numpy(q).cuda(), torch.from_numpy(lens).cuda()
        out = torch.full((B, stride), 9.0, device='cuda')
        _lib.check(_lib.lib.nele_pcm16_to_float(dq.data_ptr(), stride, dl.data_ptr(), B, L, out.data_ptr(), stride, None), 'nele_pcm16_to_float')
        o = out.cpu().numpy()
        for b in range(B):
            assert np.array_equal(o[b, :lens[b]], q[b, :lens[b]].astype(np.float32) / 32768.0)
            assert not o[b, lens[b]:L].any() and (o[b, L:] == 9.0).all()
        back = torch.zeros((B, stride), dtype=torch.int16, device='cuda')
        full = torch.from_numpy(q.astype(np.float32) / 32768.0).cuda()
        _lib.check(_lib.lib.nele_float_to_pcm16(full.data_ptr(), stride, B, L, back.data_ptr(), stride, 1, None), 'nele_float_to_pcm16')
        assert np.array_equal(back.cpu().numpy()[:, :L], q[:, :L]) and not back.cpu().numpy()[:, L:].any()
    x = np.array([[0.5 / 32767, 1.5 / 32767, 2.5 / 32767, -0.5 / 32767, 1.2, -1.2, 0.3, -0.7]], dtype=np.float32)
    dx, di = torch.from_numpy(x).cuda(), torch.zeros((1, 8), dtype=torch.int16, device='cuda')
    _lib.check(_lib.lib.nele_float_to_pcm16(dx.data_ptr(), 8, 1, 8, di.data_ptr(), 8, 0, None), 'nele_float_to_pcm16')
    from oracle.step import pcm16_roundtrip
    assert np.array_equal(di.cpu().numpy()[0].astype(np.float32) / 32768.0, pcm16_roundtrip(x[0]))
