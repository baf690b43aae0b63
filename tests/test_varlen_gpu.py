"""GPU: utterances of DIFFERENT lengths side by side in one padded batch (per-utterance lengths at the C ABI) against the same
utterances launched one at a time at their own length, and against the oracle.  The reference is batch 1 over files of any length
(dataloader.py:30-42, audio_util.py:134-141, intel.py:58-60); a real corpus has no two files of the same length."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
HERE = os.path.dirname(__file__)
TOY = os.path.join(HERE, 'golden', 'toy')


def _mixed():
    """toy Train (33 536 samples), toy Test (34 048) and a synthetic utterance (40 000): clean / noise lists."""
    from nele_gan_amd import dataio, synth
    c0, _ = dataio.load(os.path.join(TOY, 'Train_Clean.wav'))
    n0, _ = dataio.load(os.path.join(TOY, 'Train_Noise.wav'))
    c1, _ = dataio.load(os.path.join(TOY, 'Test_Clean.wav'))
    n1, _ = dataio.load(os.path.join(TOY, 'Test_Noise.wav'))
    c2, n2 = synth.batch(1, 40000, start=77)
    cl = [c0, c1, c2[0]]
    ns = [n0[:len(c0)], n1[:len(c1)], n2[0]]
    return cl, ns


def test_features_of_a_mixed_length_batch_equal_the_per_file_features():
    from nele_gan_amd import audio_util as au
    from nele_gan_amd import dataio
    cl, ns = _mixed()
    cp, lens = dataio.pad_batch(cl)
    npad, _ = dataio.pad_batch(ns)
    npad[0, lens[0]:] = 3.0                                               # garbage behind a row's end must not matter
    lengths = torch.from_numpy(lens).cuda()
    frames = au.frames_of(lengths)
    spec, band = au.stft_band(torch.from_numpy(cp).cuda(), 1 / 6, lengths=lengths)
    nspec, _ = au.stft_band(torch.from_numpy(npad).cuda(), 1 / 6, want_band=False, lengths=lengths)
    psd, nband = au.imcra_band(nspec, 1 / 6, frames=frames)
    for b, (c, n) in enumerate(zip(cl, ns)):
        T = 1 + len(c) // 256
        s1, b1 = au.stft_band(torch.from_numpy(c).cuda().unsqueeze(0), 1 / 6)
        ns1, _ = au.stft_band(torch.from_numpy(n).cuda().unsqueeze(0), 1 / 6, want_band=False)
        p1, nb1 = au.imcra_band(ns1, 1 / 6)
        assert int(frames[b]) == T
        assert torch.equal(spec[b, :T], s1[0]) and torch.equal(band[b, :T], b1[0])
        assert torch.equal(psd[b, :T], p1[0]) and torch.equal(nband[b, :T], nb1[0])
        assert float(spec[b, T:].abs().sum()) == 0.0 and float(band[b, T:].abs().sum()) == 0.0
        assert float(nband[b, T:].abs().sum()) == 0.0 and float(psd[b, T:].abs().sum()) == 0.0


def test_metrics_of_a_mixed_length_batch_match_per_file_values_and_the_oracle():
    """One launch per metric for toy Train + toy Test + a synthetic utterance."""
    from nele_gan_amd import dataio
    from nele_gan_amd import metrics as mt
    from oracle import step
    cl, ns = _mixed()
    ys = [c + n for c, n in zip(cl, ns)]
    xp, lens = dataio.pad_batch(cl)
    yp, _ = dataio.pad_batch(ys)
    yp[1, lens[1]:] = -2.0
    lengths = torch.from_numpy(lens)
    got = {'siib': mt.batch_siib(xp, yp, lengths=lengths, return_info=True), 'estoi': mt.batch_estoi(xp, yp, lengths=lengths),
           'haspi': mt.batch_haspi(xp, yp, lengths=lengths)}
    for b, (c, y) in enumerate(zip(cl, ys)):
        r1, m1, i1 = mt.batch_siib(c, y, return_info=True)
        assert float(got['siib'][0][b]) == float(r1[0]) and torch.equal(got['siib'][2][b].cpu(), i1[0].cpu())   # incl. the replication factor M
        assert float(got['estoi'][0][b]) == float(mt.batch_estoi(c, y)[0][0])
        assert float(got['haspi'][0][b]) == pytest.approx(float(mt.batch_haspi(c, y)[0][0]), rel=1e-6)
        for m in ('siib', 'estoi', 'haspi'):
            ref = step.metric_targets(c, c, ns[b], [m], norm=False)[0]                  # metric_targets(clean, enhanced, noise): y = enhanced + noise
            assert float(got[m][0][b]) == pytest.approx(ref, rel=1e-4, abs=1e-4), (m, b)


def test_generator_discriminator_on_a_padded_batch_equal_per_utterance_passes():
    """G -> energy normalisation -> D forward, and D's parameter gradient, on a padded batch of two lengths: every row equals the
    single-utterance pass at its own length; the batch gradient is the mean of the single-utterance gradients."""
    from nele_gan_amd import audio_util as au
    from nele_gan_amd import model as mods
    from nele_gan_amd.train_nele import GanTrainer
    cl, ns = _mixed()
    cl, ns = cl[:2] + [cl[2][:36000]], ns[:2] + [ns[2][:36000]]
    tr = GanTrainer('estoi')
    from nele_gan_amd import dataio
    cp, lens = dataio.pad_batch(cl)
    npad, _ = dataio.pad_batch(ns)
    f = tr.features(torch.from_numpy(cp).cuda(), torch.from_numpy(npad).cuda(), lengths=torch.from_numpy(lens))
    tr.D.eval(); tr.G.eval()                                             # no power iteration: identical weights in every pass
    tgt = torch.tensor([[0.3], [0.6], [0.8]], device='cuda')
    tr.D.flat_parameters().grad.zero_()
    with torch.no_grad():
        mask = tr.G(f['clean_band'], f['noise_band'])
    din, beta2 = mods.energy_norm_pack(mask, f['clean_band'], f['noise_band'])
    score = tr.D.forward_packed(din, f['frames'])
    torch.nn.functional.mse_loss(score, tgt).backward()
    g_batch = tr.D.flat_parameters().grad.clone()
    g_sum = torch.zeros_like(g_batch)
    for b, (c, n) in enumerate(zip(cl, ns)):
        T = 1 + len(c) // 256
        f1 = tr.features(torch.from_numpy(c).cuda().unsqueeze(0), torch.from_numpy(n).cuda().unsqueeze(0))
        with torch.no_grad():
            m1 = tr.G(f1['clean_band'], f1['noise_band'])
        assert torch.equal(mask[b, :T], m1[0])                             # causal generator: padding behind the end changes nothing
        d1, b1 = mods.energy_norm_pack(m1, f1['clean_band'], f1['noise_band'])
        assert float(beta2[b]) == pytest.approx(float(b1[0]), rel=1e-6)
        np.testing.assert_allclose(din[b, :, :T].cpu().numpy(), d1[0].cpu().numpy(), rtol=1e-6, atol=0)
        assert float(din[b, :, T:].abs().sum()) == 0.0
        tr.D.flat_parameters().grad.zero_()
        s1 = tr.D.forward_packed(d1)
        assert float(score[b, 0]) == pytest.approx(float(s1[0, 0]), rel=1e-5)
        torch.nn.functional.mse_loss(s1, tgt[b:b + 1]).backward()
        g_sum += tr.D.flat_parameters().grad
    ref = (g_sum / 3).cpu().numpy()
    np.testing.assert_allclose(g_batch.cpu().numpy(), ref, rtol=2e-3, atol=2e-5 * np.abs(ref).max())


def test_canonical_step_on_a_mixed_length_batch():
    """The multi-stream step with per-utterance lengths: targets equal the per-file metric values of the same enhanced signals,
    losses finite, nothing masked."""
    from nele_gan_amd import dataio
    from nele_gan_amd import metrics as mt
    from nele_gan_amd.train_nele import GanTrainer
    cl, ns = _mixed()
    cp, lens = dataio.pad_batch(cl)
    npad, _ = dataio.pad_batch(ns)
    cw, nw, lengths = torch.from_numpy(cp).cuda(), torch.from_numpy(npad).cuda(), torch.from_numpy(lens)
    tr = GanTrainer('siib&haspi&estoi')
    lg, ld, tgt = tr.canonical_step(cw, nw, lengths=lengths)
    torch.cuda.synchronize()
    assert all(v == 0 for v in tr.check_status().values())
    assert torch.isfinite(lg) and torch.isfinite(ld) and torch.isfinite(tgt).all()
    enh = tr._last_enh
    for b, L in enumerate(lens):
        n = 256 * (int(L) // 256)
        assert float(enh[b, n:].abs().sum()) == 0.0
        x = cw[b:b + 1, :n].contiguous()
        y = (enh[b:b + 1, :n] + nw[b:b + 1, :n]).contiguous()
        assert float(tgt[b, 0]) == float(mt.batch_siib(x, y)[1][0])
        assert float(tgt[b, 1]) == pytest.approx(float(mt.batch_haspi(x, y)[1][0]), rel=1e-6)
        assert float(tgt[b, 2]) == float(mt.batch_estoi(x, y)[1][0])


def test_enhance_files_writes_the_reference_layout(tmp_path):
    """inference.py:79-117 over a file list of two lengths: '<name>@1.wav' PCM_16 files whose samples equal the single-file path."""
    import shutil
    from nele_gan_amd import dataio
    from nele_gan_amd.inference import Enhancer, enhance_files
    for d in ('Clean', 'Noise'):
        (tmp_path / d).mkdir()
    for split in ('Train', 'Test'):
        shutil.copy(os.path.join(TOY, split + '_Clean.wav'), tmp_path / 'Clean' / (split + '.wav'))
        shutil.copy(os.path.join(TOY, split + '_Noise.wav'), tmp_path / 'Noise' / (split + '.wav'))
    files = [str(tmp_path / 'Clean' / 'Train.wav'), str(tmp_path / 'Clean' / 'Test.wav')]
    torch.manual_seed(3)
    e = Enhancer()
    out = enhance_files(e, files, str(tmp_path / 'Noise') + '/', str(tmp_path / 'Enh'), batch=8)
    assert [os.path.basename(p) for p in out] == ['Train@1.wav', 'Test@1.wav']
    for path, src in zip(out, files):
        got, sr = dataio.load(path)
        c, _ = dataio.load(src)
        n, _ = dataio.load(str(tmp_path / 'Noise' / os.path.basename(src)))
        m = min(len(c), len(n))
        assert sr == 16000 and len(got) == 256 * (m // 256)
        ref = e.enhance(torch.from_numpy(c[:m]).cuda().unsqueeze(0), torch.from_numpy(n[:m]).cuda().unsqueeze(0))[0].cpu().numpy()
        np.testing.assert_array_equal(got, ref)                           # PCM_16 grid on both sides: exact
        assert np.sqrt(np.mean(got.astype(np.float64) ** 2)) == pytest.approx(0.03, rel=2e-3)
    # the script's report behind the loop (inference.py:119-146): unmapped means per noise type (a substring of the file name)
    import io
    from nele_gan_amd.inference import score_enhanced
    from nele_gan_amd import quality
    quality.clear_backends()
    buf = io.StringIO()
    rep = score_enhanced(str(tmp_path / 'Clean') + '/', str(tmp_path / 'Noise') + '/', out, groups=('Train', 'Test', 'Cafeteria'), out=buf)
    assert set(rep) == {'Train', 'Test'} and rep['Train']['files'] == 1
    for g, path in (('Train', out[0]), ('Test', out[1])):
        one = [path]
        assert rep[g]['siib'] == pytest.approx(dataio.read_batch_SIIB(str(tmp_path / 'Clean') + '/', str(tmp_path / 'Noise') + '/', one, norm=False)[0], rel=1e-12)
        assert rep[g]['estoi'] == pytest.approx(dataio.read_batch_STOI(str(tmp_path / 'Clean') + '/', str(tmp_path / 'Noise') + '/', one, norm=False)[0], rel=1e-12)
        assert rep[g]['haspi'] == pytest.approx(dataio.read_batch_HASPI(str(tmp_path / 'Clean') + '/', str(tmp_path / 'Noise') + '/', one, norm=False)[0], rel=1e-12)
        assert np.isnan(rep[g]['pesq']) and np.isnan(rep[g]['visqol'])           # the external programs are not registered: not scored, not invented
    lines = buf.getvalue().splitlines()
    assert lines[0] == 'Train:' and lines[1] == ('SIIB is %.3f, HASPI is %.3f, ESTOI is %.3f, PESQ is nan, VISQOL is nan'
                                                 % (rep['Train']['siib'], rep['Train']['haspi'], rep['Train']['estoi']))
    with pytest.raises(quality.QualityBackendMissing):
        score_enhanced(str(tmp_path / 'Clean') + '/', str(tmp_path / 'Noise') + '/', out, groups=None, quality=True, out=buf)
    # one file per batch, three batches in flight, no length bucketing, in list order: the same files byte for byte
    out1 = enhance_files(e, files + files[:1], str(tmp_path / 'Noise') + '/', str(tmp_path / 'Enh1'), batch=1, sort_by_length=False, pad_to=0)
    assert [os.path.basename(p) for p in out1] == ['Train@1.wav', 'Test@1.wav', 'Train@1.wav']
    for a, b in zip(out, out1):
        assert open(a, 'rb').read() == open(b, 'rb').read()
