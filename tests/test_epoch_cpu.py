"""CPU: the epoch driver's ORDER (train_nele.py:110-429) with stubbed stages, the length bucketing of the D passes, and the
data-parallel D epoch on ragged shards (gloo, world size 2): every rank joins the same number of all-reduces and the replicas
stay identical.  No GPU and no HIP kernels are involved: the stages are replaced by recorders / a tiny torch-CPU discriminator."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _bare_trainer():
    from nele_gan_amd.train_nele import GanTrainer
    tr = GanTrainer.__new__(GanTrainer)
    tr.metrics = ['siib', 'estoi']
    tr.device = torch.device('cpu')
    tr.D_Qua = None
    tr.optimizer_dqua = None
    tr.history = []
    tr.step_d = 0
    tr.step_g = 0
    tr.world = 1
    tr.pcm16 = True
    tr._status = {}
    tr._pending_d = None
    tr.overlap_allreduce = True
    return tr


def _recording_trainer(log):
    tr = _bare_trainer()
    T = 30

    def features(c, n, lengths=None):
        log.append('features')
        return {'clean_band': torch.zeros(c.shape[0], T, 64), 'noise_band': torch.zeros(c.shape[0], T, 64), 'clean_spec': None}

    def g_step(cb, nb, frames=None, weight=None):
        log.append('g_step')
        return torch.tensor(0.5)

    def generate(cb, nb, spec, rms_target=0.0, frames=None):
        log.append('generate')
        return torch.zeros(cb.shape[0], 256 * (T - 1))

    class _Done:                                           # what true_metrics(defer=True) returns: targets resolved by .result()
        def __init__(self, v):
            self.v = v

        def result(self):
            return self.v

    def true_metrics(c, e, n, norm=True, lengths=None, resynth=True, utt_ids=None, defer=False):
        log.append('metrics' if norm else 'metrics_raw')
        v = torch.full((c.shape[0], 2), 0.25)
        return _Done(v) if defer else v

    def true_metrics_pair(c, e, d, n, norm=True, lengths=None, drc_lengths=None, utt_ids=None, defer=False):
        log.append('metrics_pair')
        v = torch.full((c.shape[0], 2), 0.25), torch.full((c.shape[0], 2), 0.25)
        return _Done(v) if defer else v

    def d_inputs(e, nb, cb, lengths=None, resynth=True):
        log.append('d_inputs')
        return torch.zeros(e.shape[0], 64, T, 4)

    def d_epoch(samples, batch=32):
        log.append('d_epoch:%d' % len(samples))

    def save_checkpoint(path):
        log.append('checkpoint')

    tr.features, tr.g_step, tr.generate, tr.true_metrics = features, g_step, generate, true_metrics
    tr.d_inputs, tr.d_epoch, tr.save_checkpoint = d_inputs, d_epoch, save_checkpoint
    tr.true_metrics_pair = true_metrics_pair
    tr.check_status = lambda raise_on_error=True: {}
    return tr


def _batches(n, B=2, L=256 * 29, drc=False):
    out = []
    for _ in range(n):
        b = {'clean': torch.zeros(B, L), 'noise': torch.zeros(B, L)}
        if drc:
            b['drc'] = torch.zeros(B, L)
        out.append(b)
    return out


def test_epoch_one_has_no_generator_step_and_the_reference_order(tmp_path):
    """train_nele.py:122: `if gan_epoch >= 2` guards the G-steps; then validation (:159), checkpoint (:272), sample generation
    (:279), true targets of the generated and of the pre-enhanced examples (:318-340), the three D passes (:342)."""
    log = []
    tr = _recording_trainer(log)
    out = tr.run_epoch(1, _batches(2, drc=True), valid_batches=_batches(1), chkpt_path=str(tmp_path / 'c.pt'), log_path=str(tmp_path / 'log.txt'))
    assert 'g_step' not in log and out['g_steps'] == 0 and out['g_loss'] is None
    assert log == ['features', 'generate', 'metrics_raw',                       # validation
                   'checkpoint',
                   'features', 'generate', 'metrics_pair', 'd_inputs', 'd_inputs',   # batch 0: generated + DRC example, clean-signal work once
                   'features', 'generate', 'metrics_pair', 'd_inputs', 'd_inputs',   # batch 1
                   'd_epoch:8']
    assert out['samples'] == 8
    line = open(tmp_path / 'log.txt').read()
    assert line == 'SIIB is 0.250, HASPI is 0.000, ESTOI is 0.250, PESQ is 0.000, VISQOL is 0.000, EPOCH:1 \n'   # train_nele.py:224


def test_epoch_two_runs_generator_steps_first_and_reuses_their_features():
    log = []
    tr = _recording_trainer(log)
    out = tr.run_epoch(2, _batches(3))
    assert log[:6] == ['features', 'g_step'] * 3 and out['g_steps'] == 3 and float(out['g_loss']) == 0.5
    rest = log[6:]
    assert rest == ['generate', 'metrics', 'd_inputs'] * 3 + ['d_epoch:6']       # no second feature pass for the same utterances
    assert 'checkpoint' not in log                                                # no path given


def test_padded_chunks_keep_list_order_and_carry_frame_counts():
    from nele_gan_amd.train_nele import GanTrainer
    items = [(torch.full((64, T, 4), float(i)), torch.tensor([float(i)])) for i, T in enumerate([30, 40, 30, 30, 41, 50, 30])]
    chunks = GanTrainer._padded_chunks(items, 2)
    assert [[int(v) for v in ch[1][:, 0]] for ch in chunks] == [[0, 1], [2, 3], [4, 5], [6]]
    din, tgt, tq, frames, nreal = chunks[0]
    assert nreal == 2 and chunks[3][4] == 1
    assert din.shape == (2, 64, 48, 4) and frames.tolist() == [30, 40] and tq is None
    assert float(din[0, :, 30:].abs().sum()) == 0.0 and float(din[0, :, :30].min()) == 0.0 and float(din[1, :, :40].min()) == 1.0 and not din[1, :, 40:].any()
    assert chunks[1][3] is None and chunks[1][0].shape == (2, 64, 30, 4)          # equal lengths: plain stack, no frame counts
    assert chunks[2][0].shape == (2, 64, 64, 4) and chunks[2][3].tolist() == [41, 50]   # padded to a multiple of 16 (few distinct buffer shapes)
    # fill: the short last batch of a pass is filled up with all-zero items outside the loss (a batch size that has not occurred before costs buffers)
    filled = GanTrainer._padded_chunks(items, 2, fill=True)
    assert [c[0].shape[0] for c in filled] == [2, 2, 2, 2] and [c[4] for c in filled] == [2, 2, 2, 1]
    big = GanTrainer._padded_chunks(items * 5, 32, fill=True)                      # 35 items in batches of 32: the last one (3 items) takes 8 rows, not 32
    assert [c[0].shape[0] for c in big] == [32, 8] and big[1][4] == 3
    assert not filled[3][0][1].any() and filled[3][1].shape == (2, 1) and float(filled[3][1][1, 0]) == 0.0 and filled[3][3] is None
    short = GanTrainer._padded_chunks(items[:1], 2, fill=True)                     # a pass without a full batch keeps its size
    assert short[0][0].shape[0] == 1 and short[0][4] == 1


# ------------------------------------------------------------------------------------------ data-parallel D epoch on ragged shards
class _Flat:
    def __init__(self, n):
        self.flat = torch.zeros(n)
        self.grad = torch.zeros(n)


class _TinyD(torch.nn.Module):
    """Stands for the discriminator: score = sigmoid(mean over (64, T, 4) * w + b); same flat-buffer surface as the real module."""

    def __init__(self):
        super().__init__()
        self._f = _Flat(2)
        self._f.flat[:] = torch.tensor([0.3, -0.1])

    def flat_parameters(self):
        return self._f

    def advance_power_iteration(self, dev=None):
        self.sn_steps = getattr(self, 'sn_steps', 0) + 1       # stands for weight_u / weight_v: one power iteration per training forward

    def forward_packed(self, din, frames=None):
        self.advance_power_iteration()
        self._w = self._f.flat.detach().clone().requires_grad_(True)
        return torch.sigmoid(din.mean(dim=(1, 2, 3)).unsqueeze(1) * self._w[0] + self._w[1])

    def collect(self):
        if getattr(self, '_w', None) is not None and self._w.grad is not None:
            self._f.grad += self._w.grad
            self._w = None


class _Sgd:
    def __init__(self, mod, lr=0.5):
        self.mod, self.lr, self.steps = mod, lr, 0

    def zero_grad(self):
        self.mod.flat_parameters().grad.zero_()

    def step(self):
        f = self.mod.flat_parameters()
        f.flat -= self.lr * f.grad
        self.steps += 1

    def skipped_steps(self):
        return 0


def _dp_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import random
    from nele_gan_amd import dist as nd
    tr = _bare_trainer()
    tr.world = world
    tr.D = _TinyD()
    tr.optimizer_d = _Sgd(tr.D)
    tr.MSELoss = torch.nn.MSELoss()
    orig_allreduce = tr._allreduce_grads

    def allreduce(module, weight=None):                 # the tiny module accumulates its autograd result into the flat bucket first
        module.collect()
        orig_allreduce(module, weight)
    tr._allreduce_grads = allreduce
    random.seed(100 + rank)                             # different shuffles per rank, as with different shards
    rs = np.random.RandomState(7)
    # 11 items of 3 different lengths, sharded 6 / 5: ragged chunk counts per rank
    Ts = [30, 30, 40, 50, 30, 40, 30, 30, 50, 50, 40]
    items = [(torch.from_numpy(rs.rand(64, T, 4).astype(np.float32)), torch.tensor([float(rs.rand())])) for T in Ts]
    lo, hi = nd.shard_range(len(items))
    tr.history = [items[i] for i in range(lo, hi)] * 8   # 48 / 40 items of history: replay positions < 40 // 30 ... drawn on rank 0
    tr.d_epoch(items[lo:hi], batch=2)
    res = {'w': tr.D.flat_parameters().flat.numpy().copy(), 'steps': tr.optimizer_d.steps, 'step_d': tr.step_d,
                 'hist': len(tr.history), 'sn': tr.D.sn_steps}
    # one explicit weighted step: rank 0 contributes 3 items, rank 1 one item -> the global mean over 4 items
    g = torch.tensor([float(rank + 1), 2.0 * (rank + 1)])
    nd.allreduce_weighted_mean_(g, 3 if rank == 0 else 1)
    res['wmean'] = g.numpy().copy()
    e = torch.zeros(2)
    nd.allreduce_weighted_mean_(e, 0)                     # nobody contributes: stays zero, no division by zero
    res['empty'] = e.numpy().copy()
    res['max'] = nd.allreduce_max_int(3 + rank)
    out[rank] = res
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_data_parallel_d_epoch_on_ragged_shards_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0['steps'] == r1['steps'] and r0['step_d'] == r1['step_d']            # same number of optimiser steps / all-reduces
    assert r0['sn'] == r1['sn'] == r0['steps']                                    # empty steps advance the spectral-norm iteration too
    np.testing.assert_array_equal(r0['w'], r1['w'])                               # replicas identical after the epoch
    assert not np.array_equal(r0['w'], np.array([0.3, -0.1], dtype=np.float32))   # and they did move
    np.testing.assert_allclose(r0['wmean'], [(3 * 1 + 1 * 2) / 4.0, (3 * 2 + 1 * 4) / 4.0])
    np.testing.assert_array_equal(r0['wmean'], r1['wmean'])
    np.testing.assert_array_equal(r0['empty'], [0.0, 0.0])
    assert r0['max'] == 4 and r1['max'] == 4


# ------------------------------------------------------------------------------------------ data-parallel epoch: D_Qua + ragged G-step loop
def _dp_epoch_worker(rank, world, port, out):
    """run_epoch on two ranks that hold DIFFERENT numbers of training batches (3 / 1), with the quality discriminator enabled: the
    G-step loop and every D pass must issue the same collectives on both ranks (empty steps on the rank that ran out), D_Qua is
    stepped on both or on neither, and the validation means run over both shards."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import random
    from nele_gan_amd.train_nele import GanTrainer
    tr = _bare_trainer()
    tr.world = world
    tr.G, tr.D, tr.D_Qua = _TinyD(), _TinyD(), _TinyD()
    tr.optimizer_g, tr.optimizer_d, tr.optimizer_dqua = _Sgd(tr.G), _Sgd(tr.D), _Sgd(tr.D_Qua)
    tr.MSELoss = torch.nn.MSELoss()
    orig_allreduce = tr._allreduce_grads

    def allreduce(module, weight=None):
        module.collect()
        orig_allreduce(module, weight)
    tr._allreduce_grads = allreduce
    T = 30
    tr.features = lambda c, n, lengths=None: {'clean_band': c[:, :T * 64].reshape(-1, T, 64), 'noise_band': n[:, :T * 64].reshape(-1, T, 64),
                                              'clean_spec': None, 'frames': None}
    real_g_step = GanTrainer.g_step

    def g_step(cb, nb, frames=None, weight=None):
        if cb is None:
            return real_g_step(tr, None, None, weight=weight)          # the product's empty-step path
        tr.optimizer_g.zero_grad()
        loss = tr.MSELoss(tr.G.forward_packed(cb.reshape(cb.shape[0], 1, T, 64)), torch.ones(cb.shape[0], 1))
        loss.backward()
        tr._allreduce_grads(tr.G, weight)
        tr.optimizer_g.step()
        tr.step_g += 1
        return loss.detach()
    tr.g_step = g_step
    tr.generate = lambda cb, nb, spec, rms_target=0.0, frames=None: cb.reshape(cb.shape[0], -1)[:, :256 * (T - 1)].clone()
    class _Done:
        def __init__(self, v):
            self.v = v

        def result(self):
            return self.v
    tr.true_metrics = lambda c, e, n, norm=True, lengths=None, resynth=True, utt_ids=None, defer=False: (
        _Done(torch.full((c.shape[0], 2), 0.2 + 0.1 * rank)) if defer else torch.full((c.shape[0], 2), 0.2 + 0.1 * rank))
    tr.true_metrics_pair = lambda c, e, d, n, norm=True, lengths=None, drc_lengths=None, utt_ids=None, defer=False: (
        _Done((torch.full((c.shape[0], 2), 0.2 + 0.1 * rank), torch.full((c.shape[0], 2), 0.2 + 0.1 * rank))) if defer else
        (torch.full((c.shape[0], 2), 0.2 + 0.1 * rank), torch.full((c.shape[0], 2), 0.2 + 0.1 * rank)))
    tr.d_inputs = lambda e, nb, cb, lengths=None, resynth=True: cb.reshape(cb.shape[0], T, 64, 1).transpose(1, 2).repeat(1, 1, 1, 4).contiguous()
    tr.check_status = lambda raise_on_error=True: {}
    random.seed(5 + rank)
    rs = np.random.RandomState(11 + rank)
    nb_ = 3 if rank == 0 else 1
    B = 2 if rank == 0 else 3                                           # different batch sizes too: the mean must be item-weighted
    mk = lambda: torch.from_numpy(rs.rand(B, 256 * T).astype(np.float32))
    train = [{'clean': mk(), 'noise': mk(), 'qua': torch.full((B, 2), 0.5), 'drc': mk(), 'drc_qua': torch.full((B, 2), 0.25)} for _ in range(nb_)]
    valid = [{'clean': mk(), 'noise': mk()}] if rank == 0 else []      # only rank 0 holds validation utterances
    res = tr.run_epoch(2, train, valid, d_batch=4)
    out[rank] = {'g': tr.G.flat_parameters().flat.numpy().copy(), 'd': tr.D.flat_parameters().flat.numpy().copy(),
                 'q': tr.D_Qua.flat_parameters().flat.numpy().copy(), 'gs': tr.optimizer_g.steps, 'ds': tr.optimizer_d.steps,
                 'qs': tr.optimizer_dqua.steps, 'valid': res['valid'], 'g_steps': res['g_steps'],
                 'sn_d': tr.D.sn_steps, 'sn_q': tr.D_Qua.sn_steps}
    dist.destroy_process_group()


def test_data_parallel_epoch_with_ragged_batch_counts_and_quality_discriminator_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_epoch_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0['gs'] == r1['gs'] == 3 and r0['g_steps'] == r1['g_steps'] == 3     # rank 1 joined two empty G-steps
    assert r0['ds'] == r1['ds'] and r0['qs'] == r1['qs'] == r0['ds']              # D_Qua stepped with every D step, on both ranks
    for k in ('g', 'd', 'q'):
        np.testing.assert_array_equal(r0[k], r1[k])                               # replicas identical
        assert not np.array_equal(r0[k], np.array([0.3, -0.1], dtype=np.float32))
    assert r0['valid'] == r1['valid'] and r0['valid']['siib'] == pytest.approx(0.2)   # rank 1 logs the global mean (rank 0's shard only)
    # spectral-norm buffers in lock-step: the empty G-steps / D-steps of the rank that ran out advanced u, v like the real ones
    # (the real G-step's D / D_Qua forward passes are stubbed out here, so only the empty steps of rank 1 count on the G side)
    assert r0['sn_d'] - r0['ds'] == 0 and r1['sn_d'] - r1['ds'] == 2 and r1['sn_q'] - r1['qs'] == 2


def _dp_mixed_worker(rank, world, port, out):
    """Rank 0's samples carry quality targets, rank 1's do not: BOTH ranks must raise (a rank that raises alone leaves the other in the
    next all-reduce until the collective times out)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    tr = _bare_trainer()
    tr.world = world
    tr.D, tr.D_Qua = _TinyD(), _TinyD()
    tr.optimizer_d, tr.optimizer_dqua = _Sgd(tr.D), _Sgd(tr.D_Qua)
    tr.MSELoss = torch.nn.MSELoss()
    items = [(torch.zeros(64, 30, 4), torch.tensor([0.5]), torch.tensor([0.5, 0.5]) if rank == 0 else None) for _ in range(3)]
    try:
        tr._d_pass(items, 2)
        out[rank] = 'no error'
    except ValueError as e:
        out[rank] = str(e)
    dist.destroy_process_group()


def test_mixed_quality_targets_across_ranks_raise_on_every_rank_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_mixed_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert 'some ranks carry quality targets' in out[0] and 'some ranks carry quality targets' in out[1]


# ------------------------------------------------------------------------------------------ world 4 and 8: collective ORDER per rank
def _dp_worker_n(rank, world, port, out):
    """d_epoch + two G-steps on `world` ranks with ragged shards (the last ranks hold fewer items, one rank holds a different batch size),
    every torch.distributed collective logged: a rank whose sequence differs from the others' is the realistic multi-GPU failure (a hang in
    the next collective), and it cannot be seen on one GPU."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import random
    from nele_gan_amd import dist as nd
    from nele_gan_amd.train_nele import GanTrainer
    log = []
    real_ar, real_bc = dist.all_reduce, dist.broadcast

    def all_reduce(t, op=dist.ReduceOp.SUM, **kw):
        log.append(('all_reduce', int(t.numel()), str(op)))
        return real_ar(t, op=op, **kw)

    def broadcast(t, src, **kw):
        log.append(('broadcast', int(t.numel()), int(src)))
        return real_bc(t, src, **kw)
    dist.all_reduce, dist.broadcast = all_reduce, broadcast
    tr = _bare_trainer()
    tr.world = world
    tr.G, tr.D = _TinyD(), _TinyD()
    tr.optimizer_g, tr.optimizer_d = _Sgd(tr.G), _Sgd(tr.D)
    tr.MSELoss = torch.nn.MSELoss()
    orig_allreduce = tr._allreduce_grads

    def allreduce(module, weight=None):
        module.collect()
        orig_allreduce(module, weight)
    tr._allreduce_grads = allreduce
    random.seed(100 + rank)
    rs = np.random.RandomState(7)
    n_items = 4 * world + 5                                # world 8: shards of 5, 5, 5, 5, 5, 4, 4, 4 -> 3 / 2 chunks of two
    Ts = [30 + 10 * (i % 3) for i in range(n_items)]
    items = [(torch.from_numpy(rs.rand(64, T, 4).astype(np.float32)), torch.tensor([float(rs.rand())])) for T in Ts]
    lo, hi = nd.shard_range(len(items))
    tr.history = [items[i] for i in range(lo, hi)] * 8
    tr.d_epoch(items[lo:hi], batch=2)
    # the G-step loop of run_epoch on ragged batch counts: ranks below world / 2 hold two batches, the others one
    nb = 2 if rank < world // 2 else 1
    n_g = nd.allreduce_max_int(nb)
    for i in range(n_g):
        if i < nb:
            tr.optimizer_g.zero_grad()
            x = torch.from_numpy(rs.rand(2 + rank % 2, 1, 30, 64).astype(np.float32))
            loss = tr.MSELoss(tr.G.forward_packed(x), torch.ones(x.shape[0], 1))
            tr.D.advance_power_iteration()             # stands for D's training-mode forward pass inside a real G-step
            loss.backward()
            tr._allreduce_grads(tr.G, x.shape[0])
            tr.optimizer_g.step()
        else:
            GanTrainer.g_step(tr, None, None, weight=0)   # the product's empty step: joins the collective, advances D's u / v
    out[rank] = {'log': log, 'd': tr.D.flat_parameters().flat.numpy().copy(), 'g': tr.G.flat_parameters().flat.numpy().copy(),
                 'ds': tr.optimizer_d.steps, 'gs': tr.optimizer_g.steps, 'sn': tr.D.sn_steps, 'shard': (lo, hi)}
    dist.all_reduce, dist.broadcast = real_ar, real_bc
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [4, 8])
def test_every_rank_issues_the_same_collectives_in_the_same_order(world):
    """BASELINE configs[3] runs on 8 ranks; the builder's and the driver's boxes have one GPU.  What can be checked without a second GPU:
    on ragged shards (different item counts, batch sizes and chunk counts per rank) every rank of a 4- and an 8-rank gloo job issues the
    same sequence of collectives (kind, payload size, reduce op), takes the same number of optimiser steps, advances the spectral-norm
    iteration as often, and ends with bit-identical replicas."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker_n, args=(world, _free_port(), out), nprocs=world, join=True)
    r0 = out[0]
    assert len(r0['log']) > 10
    shards = sorted(out[r]['shard'] for r in range(world))
    assert shards[0][0] == 0 and all(shards[i][1] == shards[i + 1][0] for i in range(world - 1)) and shards[-1][1] == 4 * world + 5
    assert len({s[1] - s[0] for s in shards}) == 2                                # ragged on purpose
    for r in range(1, world):
        assert out[r]['log'] == r0['log'], 'rank %d issued other collectives than rank 0' % r
        assert out[r]['ds'] == r0['ds'] and out[r]['gs'] == r0['gs'] == 2 and out[r]['sn'] == r0['sn']
        np.testing.assert_array_equal(out[r]['d'], r0['d'])
        np.testing.assert_array_equal(out[r]['g'], r0['g'])


# ------------------------------------------------------------------------------------------ deferred D update (overlap_allreduce)
def _dp_defer_worker(rank, world, port, out):
    """The canonical step ends by STARTING the all-reduce of D's gradients (async) and leaves wait + Adam-D to whatever touches D next
    (GanTrainer._flush_d).  Same weights as the synchronous form, the step counter advances at the flush, and every entry point that reads
    or writes D flushes first."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nele_gan_amd.train_nele import GanTrainer
    res = {}
    for mode in ('sync', 'defer'):
        tr = _bare_trainer()
        tr.world = world
        tr.D = _TinyD()
        tr.optimizer_d = _Sgd(tr.D)
        tr.MSELoss = torch.nn.MSELoss()
        orig = tr._allreduce_grads

        def allreduce(module, weight=None, _o=orig):
            module.collect()
            _o(module, weight)
        tr._allreduce_grads = allreduce
        rs = np.random.RandomState(3 + rank)
        trace = []
        for k in range(3):
            x = torch.from_numpy(rs.rand(2, 64, 30, 4).astype(np.float32))
            tgt = torch.from_numpy(rs.rand(2, 1).astype(np.float32))
            tr.optimizer_d.zero_grad()
            score = tr.D.forward_packed(x)
            if mode == 'defer':
                loss = tr.MSELoss(score, tgt)
                loss.backward()
                tr.D.collect()                                          # (the stub keeps its gradient outside the flat bucket until asked)
                tr._pending_d = __import__('nele_gan_amd.dist', fromlist=['x']).PendingMean(tr.D.flat_parameters().grad)
                trace.append((tr.step_d, tr.optimizer_d.steps))         # nothing applied yet
                if k == 1:
                    tr.check_status = GanTrainer.check_status.__get__(tr)   # any reader of D flushes: here through the next loop turn
                tr._flush_d()
            else:
                tr._d_finish(score, tgt)
            trace.append((tr.step_d, tr.optimizer_d.steps))
        res[mode] = (tr.D.flat_parameters().flat.numpy().copy(), trace)
    out[rank] = res
    dist.destroy_process_group()


def test_deferred_d_update_equals_the_synchronous_one_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_defer_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in (0, 1):
        np.testing.assert_array_equal(out[r]['sync'][0], out[r]['defer'][0])
        assert out[r]['sync'][1] == [(1, 1), (2, 2), (3, 3)]
        assert out[r]['defer'][1] == [(0, 0), (1, 1), (1, 1), (2, 2), (2, 2), (3, 3)]
    np.testing.assert_array_equal(out[0]['defer'][0], out[1]['defer'][0])
