#!/usr/bin/env python
"""Host <-> device copy rates of the box, to tell a PCIe bound from a host-side one in the file paths (bench.py: inference_from_files,
epoch_from_files).  Pinned buffers of 64 MB (one batch of 128 x 8 s int16 pairs is 65 MB in, 33 MB out):
  H2D alone, D2H alone, two H2D copies on two streams, H2D + D2H at the same time, and a host memcpy (page cache -> pinned stand-in)."""
import time

import numpy as np
import torch


def rate(fn, nbytes, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e9


def main():
    n = 64 << 20
    h1, h2 = torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
    d1, d2 = torch.empty(n, dtype=torch.uint8, device='cuda'), torch.empty(n, dtype=torch.uint8, device='cuda')
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def h2d():
        d1.copy_(h1, non_blocking=True)

    def d2h():
        h2.copy_(d2, non_blocking=True)

    def h2d_two():
        with torch.cuda.stream(s1):
            d1.copy_(h1, non_blocking=True)
        with torch.cuda.stream(s2):
            d2.copy_(h2, non_blocking=True)

    def both():
        with torch.cuda.stream(s1):
            d1.copy_(h1, non_blocking=True)
        with torch.cuda.stream(s2):
            h2.copy_(d2, non_blocking=True)

    print('H2D 64 MB pinned:            %.1f GB/s' % rate(h2d, n))
    print('D2H 64 MB pinned:            %.1f GB/s' % rate(d2h, n))
    print('2 x H2D on two streams:      %.1f GB/s in total' % rate(h2d_two, 2 * n))
    print('H2D + D2H at the same time:  %.1f GB/s in total' % rate(both, 2 * n))
    a, b = np.empty(n, dtype=np.uint8), h1.numpy()
    a[:] = 1
    t0 = time.perf_counter()
    for _ in range(10):
        np.copyto(b, a)
    print('host memcpy pageable -> pinned, one thread: %.1f GB/s' % (n * 10 / (time.perf_counter() - t0) / 1e9))
    p = torch.empty(n, dtype=torch.uint8)
    print('H2D 64 MB pageable:          %.1f GB/s' % rate(lambda: d1.copy_(p), n, reps=5))


if __name__ == '__main__':
    main()
