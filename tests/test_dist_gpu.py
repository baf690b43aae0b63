"""GPU, two processes: replicas fed different utterance shards stay bit-identical over three canonical steps - the flat gradient
all-reduce, the rank-0 broadcast and the deterministic kernels.  Over RCCL on two devices (skipped when fewer are visible), and over gloo
with BOTH ranks on device 0 (runs on a 1-GPU box: the same trainer code with the real kernels; gloo stages the buckets through the host),
where the sharded run is also compared with one process that takes the whole batch."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, one_device=False, occupy=0):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = 0 if one_device else rank
    torch.cuda.set_device(dev)
    if one_device:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    else:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    from nele_gan_amd import dist as nd
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    lo, hi = nd.shard_range(8)
    c, v = synth.batch(hi - lo, 24000, start=lo)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    tr = GanTrainer('siib&estoi', device='cuda:%d' % dev, seed=666 + rank)       # different seeds: rank 0's weights must win
    side = torch.cuda.Stream() if occupy else None
    if occupy:                                     # a first step without company: buffers, plans and probes are set up on the host for a while
        tr.canonical_step(cw, nw)
        torch.cuda.synchronize()
    for _ in range(3):
        if occupy and rank == 1:
            # `occupy` CUs of the shared device held by resident 160 KB workgroups for the whole step: what a collective library's
            # persistent channel kernels do beside a step on a real multi-GPU node (the eigensolver's cluster launches need co-residency)
            import ctypes
            from nele_gan_amd._lib import call
            call('nele_stream_occupy', occupy, 160 * 1024, 60000.0, ctypes.c_void_p(side.cuda_stream))
        tr.canonical_step(cw, nw)
        torch.cuda.synchronize()
    st = tr.check_status()
    out[rank] = (tr.G.flat_parameters().flat.cpu(), tr.D.flat_parameters().flat.cpu(), [t.cpu() for t in tr.D.buffers()], st)
    dist.destroy_process_group()


def test_two_replicas_stay_bit_identical_over_three_steps():
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    g0, d0, b0, s0 = out[0]
    g1, d1, b1, s1 = out[1]
    assert torch.equal(g0, g1) and torch.equal(d0, d1)
    assert all(torch.equal(a, b) for a, b in zip(b0, b1))                        # spectral-norm u, v
    assert all(x == 0 for x in s0.values()) and all(x == 0 for x in s1.values())


def test_a_world_of_one_over_rccl_is_the_single_process_step():
    """The RCCL code path on a 1-GPU box: init_process_group('nccl', device_id=...), the bucketed all-reduce of the flat gradient buffers on
    the collective stream beside the metric streams, the rank-0 broadcast - with one rank the collectives are identities, so three steps must
    leave exactly the weights of a process that never initialised torch.distributed."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    g0, d0, b0, s0 = out[0]
    mp.spawn(_run_whole, args=(out,), nprocs=1, join=True)
    gw, dw = out['whole']
    assert torch.equal(g0, gw) and torch.equal(d0, dw)
    assert all(x == 0 for x in s0.values())


def _whole_batch(out):
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    c, v = synth.batch(8, 24000, start=0)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    tr = GanTrainer('siib&estoi', device='cuda:0', seed=666)
    for _ in range(3):
        tr.canonical_step(cw, nw)
    torch.cuda.synchronize()
    out['whole'] = (tr.G.flat_parameters().flat.cpu(), tr.D.flat_parameters().flat.cpu())


def test_two_ranks_on_one_device_over_gloo_match_each_other_and_the_whole_batch():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out, True), nprocs=2, join=True)
    g0, d0, b0, s0 = out[0]
    g1, d1, b1, s1 = out[1]
    assert torch.equal(g0, g1) and torch.equal(d0, d1)
    assert all(torch.equal(a, b) for a, b in zip(b0, b1))
    assert all(x == 0 for x in s0.values()) and all(x == 0 for x in s1.values())
    # two shards of 4 against one process with all 8 utterances: the mean of two shard means is the batch mean up to float32 rounding,
    # the metric targets are per-utterance (batch-invariant kernels); three Adam steps amplify rounding differences, hence the tolerance
    mp.spawn(_run_whole, args=(out,), nprocs=1, join=True)
    gw, dw = out['whole']
    # (an Adam step moves a weight by at most ~lr whatever the gradient's size: elements whose gradient is rounding noise may differ by that)
    eg, ed = (g0 - gw).abs(), (d0 - dw).abs()
    print('sharded vs whole: G max %.3e mean %.3e, D max %.3e mean %.3e' % (float(eg.max()), float(eg.mean()), float(ed.max()), float(ed.mean())))
    assert float(eg.max()) <= 3 * 5e-4 * 1.5 and float(eg.mean()) <= 1e-5
    assert float(ed.max()) <= 3 * 2.5e-4 * 1.5 and float(ed.mean()) <= 1e-4


def _run_whole(_rank, out):
    _whole_batch(out)


def test_two_ranks_on_one_device_with_64_compute_units_held_for_the_whole_step():
    """Verdict r05 item 7: rank 1 keeps 64 CUs busy with resident workgroups (nele_stream_occupy: a stand-in for RCCL's resident channel
    kernels) while both ranks step.  The replicas stay bit-identical and no covariance takes the eigensolver's repair path."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out, True, 64), nprocs=2, join=True)
    g0, d0, b0, s0 = out[0]
    g1, d1, b1, s1 = out[1]
    assert torch.equal(g0, g1) and torch.equal(d0, d1)
    assert all(torch.equal(a, b) for a, b in zip(b0, b1))
    assert s0['eigh_repaired'] == 0 and s1['eigh_repaired'] == 0, (s0, s1)
    assert all(x == 0 for x in s0.values()) and all(x == 0 for x in s1.values())


def test_process_can_be_bound_to_the_gpus_numa_node():
    """dist.bind_to_gpu_numa_node: the CPUs of the NUMA node the GPU hangs on (never more than the launcher allowed), or None where the
    topology cannot be read; the file-fed paths are host-bound and an unbound process copies across the socket link (tools/files_sweep.py)."""
    from nele_gan_amd import dist as nd
    before = os.sched_getaffinity(0)
    try:
        cpus = nd.bind_to_gpu_numa_node(0)
        if cpus is not None:
            assert cpus and cpus <= before and os.sched_getaffinity(0) == cpus
    finally:
        os.sched_setaffinity(0, before)
