"""CPU, gloo, world_size 2: the data-parallel glue (utterance sharding, flat gradient all-reduce,
replica broadcast, score gather, shared replay indices)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nele_gan_amd import dist as nd
    res = {}
    # sharding: contiguous, disjoint, complete
    res['shard'] = nd.shard_range(11)
    # flat gradient bucket: mean over ranks
    g = torch.arange(8, dtype=torch.float32) * (rank + 1)
    nd.allreduce_mean_(g)
    res['grad'] = g.numpy().copy()
    # replicas: parameters + buffers follow rank 0
    lin = torch.nn.utils.spectral_norm(torch.nn.Linear(4, 3))
    with torch.no_grad():
        for p in lin.parameters():
            p.add_(rank)
    nd.broadcast_module_(lin, 0)
    res['w'] = lin.weight_orig.detach().numpy().copy()
    res['u'] = lin.weight_u.detach().numpy().copy()
    # metric scores gathered in rank order, ragged shards
    sc = torch.full((2 + rank, 2), float(rank))
    res['scores'] = nd.gather_rows(sc).numpy().copy()
    res['replay'] = nd.replay_indices(90, 30, seed=3)
    out[rank] = res
    dist.destroy_process_group()


def test_data_parallel_glue_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0['shard'] == (0, 6) and r1['shard'] == (6, 11)
    np.testing.assert_allclose(r0['grad'], np.arange(8) * 1.5)
    np.testing.assert_allclose(r1['grad'], r0['grad'])
    np.testing.assert_array_equal(r0['w'], r1['w'])
    np.testing.assert_array_equal(r0['u'], r1['u'])
    assert r0['scores'].shape == (5, 2) and np.array_equal(r0['scores'], r1['scores'])
    assert list(r0['scores'][:, 0]) == [0, 0, 1, 1, 1]
    assert r0['replay'] == r1['replay'] and len(r0['replay']) == 3


def test_shard_range_covers_everything():
    from nele_gan_amd import dist as nd
    for n in (1, 7, 32, 1024, 10000):
        for w in (1, 2, 4, 8):
            spans = [nd.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
