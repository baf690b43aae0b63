"""Data parallelism over utterances (the only parallelism the path has): one process per GPU,
contiguous utterance shards, ONE flat all-reduce (RCCL over xGMI; backend 'nccl' on ROCm) of a model's
gradient bucket per optimiser step, replicas kept identical by broadcasting rank 0's parameters and
spectral-norm buffers at start and drawing the replay indices on rank 0.  Payloads are small
(G 8.37 MB, D 1.37 MB fp32) so the collective is latency-bound: a single bucket per model, no bucketing
heuristics.  Works with any initialised torch.distributed backend (tests use gloo on CPU tensors)."""
import random

import torch
import torch.distributed as dist


def bind_to_gpu_numa_node(device_index=0):
    """Restrict this process (and the threads / child processes it starts from now on) to the CPUs of the NUMA node its GPU hangs on, so
    that page-locked staging buffers, the page cache of the files it writes and the threads that copy between them are local to the GPU's
    PCIe root.  One process per GPU is the launch model; on the two-socket MI355X boxes of this pool an unbound process floats
    over both sockets and the wav-file paths (dataio.FileBatches, inference.enhance_files) run at 36 k instead of 45 k utterances/s of 8 s
    files (tools/files_sweep.py; the PCIe link itself gives 56 GB/s per direction, tools/pcie_time.py - the host side is the bound).
    Call it before anything is allocated.  -> the CPU set, or None when the topology cannot be read (nothing is changed then)."""
    import os
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
            a, _, b = part.partition('-')
            cpus |= set(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)                 # never widen what the launcher allowed
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return cpus
    except Exception:
        return None


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


def shard_range(n_total, rk=None, world=None):
    """Contiguous shard [start, stop) of n_total utterances for rank rk (remainder to the first ranks)."""
    rk = rank() if rk is None else rk
    world = world_size() if world is None else world
    q, r = divmod(n_total, world)
    start = rk * q + min(rk, r)
    return start, start + q + (1 if rk < r else 0)


def allreduce_mean_(flat):
    """In-place mean of a flat gradient bucket over all ranks (sum all-reduce, then scale)."""
    w = world_size()
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / w)
    return flat


class PendingMean:
    """A gradient bucket's mean over the ranks, started with async_op=True (RCCL runs it on its own stream: it proceeds beside whatever the
    caller enqueues next) and completed by wait(): the caller's stream then waits for the collective and applies the 1 / world scale."""

    def __init__(self, flat):
        self.flat = flat
        self.w = world_size()
        self.work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True) if self.w > 1 else None

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            self.flat.mul_(1.0 / self.w)
        return self.flat


def allreduce_weighted_mean_(flat, weight):
    """In-place item-weighted mean over ranks: flat holds this rank's mean gradient over ``weight`` items (0 = none; the buffer is then
    all zeros).  One all-reduce for the bucket and one for the scalar count; ranks with nothing to add still take part."""
    if world_size() > 1:
        cnt = torch.tensor([float(weight)], dtype=torch.float32, device=flat.device)
        flat.mul_(float(weight))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        flat.div_(torch.clamp(cnt, min=1.0))
    return flat


def allreduce_max_int(value, device='cpu'):
    """max over ranks of a host integer (e.g. the number of optimiser steps each rank's shard needs)."""
    if world_size() == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def allreduce_max_ints(values, device='cpu'):
    """Element-wise max over ranks of a short list of host integers in ONE collective (e.g. step count, flags and an error flag that every
    rank must see before any of them raises: a rank that raises alone leaves the others waiting in the next all-reduce)."""
    if world_size() == 1:
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(v) for v in t.tolist()]


def backend_is_nccl():
    return is_dist() and dist.get_backend() == 'nccl'


def broadcast_module_(module, src=0):
    """Make every replica identical to rank src: parameters and buffers (spectral-norm u, v)."""
    if world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


def gather_rows(t):
    """All-gather per-utterance rows (e.g. metric scores) in rank order -> [sum_B, ...]."""
    w = world_size()
    if w == 1:
        return t
    sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(w)]
    dist.all_gather(sizes, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device))
    mx = int(max(int(s) for s in sizes))
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    outs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:int(s)] for o, s in zip(outs, sizes)], dim=0)


def replay_indices(n_history, divisor=30, seed=None, device='cpu'):
    """train_nele.py:373-376: a random 1/30 of the history, drawn on rank 0 and broadcast so that
    every rank replays the same items."""
    k = n_history // divisor
    idx = torch.zeros(k, dtype=torch.int64, device=device)
    if rank() == 0 and k > 0:
        rng = random.Random(seed)
        idx.copy_(torch.tensor(rng.sample(range(n_history), k), dtype=torch.int64))
    if world_size() > 1 and k > 0:
        dist.broadcast(idx, 0)
    return idx.tolist()
