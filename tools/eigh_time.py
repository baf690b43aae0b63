#!/usr/bin/env python
"""Batched symmetric eigensolver alone: ms per call of nele_eigh_sym_batched on B covariance-like n x n float64 matrices, accuracy
against numpy on a sample.  usage: python tools/eigh_time.py [B=256] [n=420] [reps=5]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nele_gan_amd import _lib, metrics as mt          # noqa: E402
from nele_gan_amd._lib import call, ptr, stream       # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 420
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    g = torch.Generator(device='cuda').manual_seed(1)
    X = torch.randn((B, n, 2 * n), dtype=torch.float64, device='cuda', generator=g)
    X = X * torch.logspace(0, -3, n, dtype=torch.float64, device='cuda')[None, :, None]     # decaying spectrum like a stacked-frame covariance
    A0 = X @ X.transpose(1, 2) / (2 * n - 1)
    A0 = 0.5 * (A0 + A0.transpose(1, 2))
    lam = torch.empty((B, n), dtype=torch.float64, device='cuda')
    U = torch.empty_like(A0)
    ws = torch.empty(int(_lib.lib.nele_eigh_workspace_bytes(B, n)), dtype=torch.uint8, device='cuda')
    A = A0.clone()
    call('nele_eigh_sym_batched', ptr(A), n, B, ptr(lam), ptr(U), ptr(ws), ws.numel(), stream())
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        A.copy_(A0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call('nele_eigh_sym_batched', ptr(A), n, B, ptr(lam), ptr(U), ptr(ws), ws.numel(), stream())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    k = min(B, 4)
    ref = np.linalg.eigvalsh(A0[:k].cpu().numpy())
    lam_h, U_h, A_h = lam[:k].cpu().numpy(), U[:k].cpu().numpy(), A0[:k].cpu().numpy()
    err_l = max(np.abs(lam_h[i] - ref[i]).max() / np.abs(ref[i]).max() for i in range(k))
    res = max(np.abs(A_h[i] @ U_h[i].T - U_h[i].T * lam_h[i][None, :]).max() / np.abs(ref[i]).max() for i in range(k))
    orth = max(np.abs(U_h[i] @ U_h[i].T - np.eye(n)).max() for i in range(k))
    if hasattr(_lib.lib, 'nele_ecs_prof_read'):        # -DECS_PROF build (tools/variants.sh eigh prof:"-DECS_PROF"): phase clocks of the symmetric first stage
        import ctypes
        buf = (ctypes.c_ulonglong * 8)()
        _lib.lib.nele_ecs_prof_read(buf)
        names = ['vector work + 3 barriers', 'sweep 1 (update, column sums)', 'pivot column publish', 'sweep 2 (row sums, reduce-scatter)', 'barrier, column reduce, publish', 'poll']
        tot = float(sum(buf[:6])) or 1.0
        print('symmetric first stage, shader clocks per phase over its steps: ' + '; '.join('%s %.0f k (%.0f %%)' % (nm, buf[k] / 1e3, 100.0 * buf[k] / tot) for k, nm in enumerate(names)))
    print('eigh B=%d n=%d: ms per call %s (min %.3f); eigenvalue err %.2e residual %.2e orthogonality %.2e' % (
        B, n, ' '.join('%.3f' % t for t in ts), min(ts), err_l, res, orth))


if __name__ == '__main__':
    main()
