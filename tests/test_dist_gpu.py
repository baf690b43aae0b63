"""GPU, two processes over RCCL (skipped when fewer than two devices are visible): replicas fed different utterance shards stay
bit-identical over three canonical steps - the flat gradient all-reduce, the rank-0 broadcast and the deterministic kernels."""
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


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    from nele_gan_amd import dist as nd
    from nele_gan_amd import synth
    from nele_gan_amd.train_nele import GanTrainer
    lo, hi = nd.shard_range(8)
    c, v = synth.batch(hi - lo, 24000, start=lo)
    cw, nw = torch.from_numpy(c).cuda(), torch.from_numpy(v).cuda()
    tr = GanTrainer('siib&estoi', device='cuda:%d' % rank, seed=666 + rank)      # different seeds: rank 0's weights must win
    for _ in range(3):
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
