"""GanTrainer.enable_clean_cache(): the clean-signal halves of SIIB (VAD .. KLT eigen-decomposition) and HASPI (reference-signal chain) of an
utterance are computed once and copied back in later calls (train_nele.py:35-38,119,318-340: the reference scores the same clean files
every epoch).  Copies only - the targets must be BIT-identical to recomputation, at any row of any batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(n, L, start):
    from nele_gan_amd import synth
    c, v = synth.batch(n, L, start=start)
    return torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()


@pytest.fixture(scope='module')
def setup():
    from nele_gan_amd.train_nele import GanTrainer
    L = 40000 + 77
    c, v = _batch(6, L, 900)
    Lr = 256 * (L // 256)
    g = torch.Generator(device='cuda').manual_seed(3)
    enh = (c[:, :Lr] * (1.0 + 0.3 * torch.rand((6, 1), device='cuda', generator=g))).contiguous()
    plain = GanTrainer('siib&haspi&estoi')
    ref = plain.true_metrics(c, enh, v, norm=False)
    return c, v, enh, ref


def test_cached_targets_equal_fresh_ones_bit_for_bit(setup):
    from nele_gan_amd.train_nele import GanTrainer
    c, v, enh, ref = setup
    tr = GanTrainer('siib&haspi&estoi')
    cache = tr.enable_clean_cache(4 << 30)
    keys = ['u%d' % k for k in range(6)]
    first = tr.true_metrics(c, enh, v, norm=False, keys=keys)                 # misses: computed + stored
    assert cache.stats()['stored'] == 12 and cache.hits == 0
    assert torch.equal(first, ref)
    cache.poison = True                                                       # 0xFF over the whole workspace before the state is copied in
    again = tr.true_metrics(c, enh, v, norm=False, keys=keys)
    assert cache.hits == 2
    assert torch.equal(again, ref)
    # another degraded signal against the cached clean halves
    enh2 = (enh * 0.8).contiguous()
    want = GanTrainer('siib&haspi&estoi').true_metrics(c, enh2, v, norm=False)
    got = tr.true_metrics(c, enh2, v, norm=False, keys=keys)
    assert torch.equal(got, want)


def test_cached_rows_in_another_order_and_batch_size(setup):
    from nele_gan_amd.train_nele import GanTrainer
    c, v, enh, ref = setup
    tr = GanTrainer('siib&haspi&estoi')
    cache = tr.enable_clean_cache(4 << 30)
    keys = ['u%d' % k for k in range(6)]
    tr.true_metrics(c, enh, v, norm=False, keys=keys)
    cache.poison = True
    rows = [4, 1, 5]
    got = tr.true_metrics(c[rows].contiguous(), enh[rows].contiguous(), v[rows].contiguous(), norm=False, keys=[keys[r] for r in rows])
    assert cache.hits == 2
    assert torch.equal(got, ref[rows])
    # a batch with one unknown utterance: phase 3 runs for that utterance alone (a batch of one on a second workspace), its state is stored
    # and the whole batch comes from the cache - the known rows unchanged, the new row equal to a fresh computation
    c7, v7 = _batch(1, c.shape[1], 950)
    cc, vv = torch.cat([c[:2], c7]), torch.cat([v[:2], v7])
    ee = torch.cat([enh[:2], c7[:, :enh.shape[1]]])
    mixed = tr.true_metrics(cc, ee, vv, norm=False, keys=['u0', 'u1', 'new'])
    assert torch.equal(mixed[:2], ref[:2])
    assert cache.stats()['stored'] == 14 and cache.partial == 2
    fresh = GanTrainer('siib&haspi&estoi').true_metrics(cc, ee, vv, norm=False)
    assert torch.equal(mixed, fresh)


def test_pair_targets_with_cache_equal_those_without(setup):
    from nele_gan_amd.train_nele import GanTrainer
    c, v, enh, ref = setup
    drc = (c[:, :enh.shape[1]] * 1.5).contiguous()
    want = GanTrainer('siib&haspi&estoi').true_metrics_pair(c, enh, drc, v)
    tr = GanTrainer('siib&haspi&estoi')
    cache = tr.enable_clean_cache(4 << 30)
    keys = list(range(6))
    a = tr.true_metrics_pair(c, enh, drc, v, keys=keys)
    cache.poison = True
    b = tr.true_metrics_pair(c, enh, drc, v, keys=keys)
    for k in range(2):
        assert torch.equal(a[k], want[k]) and torch.equal(b[k], want[k])
    assert cache.hits == 2


def test_budget_exhausted_means_recomputation_not_failure(setup):
    from nele_gan_amd.train_nele import GanTrainer
    c, v, enh, ref = setup
    tr = GanTrainer('siib&haspi&estoi')
    cache = tr.enable_clean_cache(1 << 20)                                    # less than one block of utterances
    keys = ['u%d' % k for k in range(6)]
    for _ in range(2):
        assert torch.equal(tr.true_metrics(c, enh, v, norm=False, keys=keys), ref)
    assert cache.hits == 0 and cache.stats()['declined'] > 0


def test_utterance_dither_is_part_of_the_haspi_key(setup):
    from nele_gan_amd.train_nele import GanTrainer
    c, v, enh, ref = setup
    ids = torch.arange(100, 106, dtype=torch.int64, device='cuda')
    want = GanTrainer('haspi', haspi_dither='utterance', dither_seed=5).true_metrics(c, enh, v, norm=False, utt_ids=ids)
    tr = GanTrainer('haspi', haspi_dither='utterance', dither_seed=5)
    cache = tr.enable_clean_cache(4 << 30)
    keys = [int(i) for i in range(100, 106)]
    a = tr.true_metrics(c, enh, v, norm=False, utt_ids=ids, keys=keys)
    cache.poison = True
    b = tr.true_metrics(c, enh, v, norm=False, utt_ids=ids, keys=keys)
    assert torch.equal(a, want) and torch.equal(b, want) and cache.hits == 1
    tr.dither_seed = 6                                                        # another dither: another key, no stale hit
    tr.true_metrics(c, enh, v, norm=False, utt_ids=ids, keys=keys)
    assert cache.hits == 1


def test_file_batches_key_utterances_by_path_not_by_base_name(tmp_path):
    """Train/Clean/x.wav and Test/Clean/x.wav are different utterances: dataio.FileBatches hands the trainer's caches the clean file's PATH
    (with the noise / pre-enhanced folders it is paired with), so an epoch over one folder never finds the other folder's clean-signal state."""
    import os
    from nele_gan_amd import dataio, synth
    from nele_gan_amd.train_nele import GanTrainer
    c, v = synth.batch(4, 36000, start=500)
    for part, rows in (('A', (0, 1)), ('B', (2, 3))):
        for sub in ('Clean', 'Noise'):
            os.makedirs(str(tmp_path / part / sub))
        for j, r in enumerate(rows):                                          # the SAME two base names in both folders, different signals
            dataio.write_wav_pcm16(str(tmp_path / part / 'Clean' / ('u%d.wav' % j)), c[r])
            dataio.write_wav_pcm16(str(tmp_path / part / 'Noise' / ('u%d.wav' % j)), v[r])
    fa = dataio.FileBatches(sorted(dataio.get_filepaths(str(tmp_path / 'A' / 'Clean'))), str(tmp_path / 'A' / 'Noise') + '/', batch=2)
    fb = dataio.FileBatches(sorted(dataio.get_filepaths(str(tmp_path / 'B' / 'Clean'))), str(tmp_path / 'B' / 'Noise') + '/', batch=2)
    try:
        ba, bb = fa[0], fb[0]
        assert ba['names'] == bb['names'] and ba['keys'] != bb['keys'] and len(set(ba['keys'] + bb['keys'])) == 4
        plain = GanTrainer('siib&haspi&estoi')
        tr = GanTrainer('siib&haspi&estoi')
        tr.enable_clean_cache(2 << 30)
        for b in (ba, bb, ba, bb):
            Lr = 256 * (b['clean'].shape[1] // 256)
            enh = (1.2 * b['clean'][:, :Lr]).contiguous()
            want = plain.true_metrics(b['clean'], enh, b['noise'], norm=False, lengths=b['lengths'])
            got = tr.true_metrics(b['clean'], enh, b['noise'], norm=False, lengths=b['lengths'], keys=b['keys'])
            assert torch.equal(got, want)
        assert tr.clean_cache.hits > 0
    finally:
        fa.close()
        fb.close()
