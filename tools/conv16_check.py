"""D's conv layers on the bf16-activation tile kernel (csrc/conv16.hip), forward and data-gradient geometries: numerics against torch
(bf16-rounded operands, float32 accumulate) and isolated launch time.  usage: python tools/conv16_check.py [B] [layers e.g. 5f,5b]"""
import ctypes
import sys

import torch
import torch.nn.functional as F

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nele_gan_amd import ops
from nele_gan_amd._lib import c_void_p, call, stream
from nele_gan_amd.ops import EPI_BIAS_LRELU, EPI_MASK_LRELU_GRAD, Geom

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
want = sys.argv[2].split(',') if len(sys.argv) > 2 else None
T = 251
CONVS = [(8, 1), (16, 3), (32, 5), (48, 7), (64, 9)]
torch.manual_seed(0)
dev = 'cuda'
H, W, C = 64, T, 4
dims = [(H, W, C)]
for cout, k in CONVS:
    H, W, C = H - k + 1, W - k + 1, cout
    dims.append((H, W, C))


def prep(Wg, N, Ktot, seglen, KH):
    wf = torch.zeros(ops.conv16_wfrag_elems(N, seglen, KH), dtype=torch.bfloat16, device=dev)
    pj = (c_void_p * 2)(Wg.data_ptr(), wf.data_ptr())
    dj = (ctypes.c_int * 4)(N, Ktot, seglen, KH)
    call('nele_conv16_weight_prep_batch', pj, dj, 1, stream())
    return wf


def timeit(fn, flops, name, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)[n // 2]
    print('%-12s median %.3f ms  %.1f TFLOP/s (%.3f of 2500)' % (name, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500))
    from nele_gan_amd._lib import lib as _l
    if hasattr(_l, 'nele_conv16_prof_read'):                 # a -DC16_PROF build: phase clocks of wave 0, averaged per workgroup
        buf = (ctypes.c_ulonglong * 8)()
        _l.nele_conv16_prof_read(buf, 1)
        fn(); 
        _l.nele_conv16_prof_read(buf, 1)
        wg = max(buf[5], 1)
        print('             per workgroup (shader clocks): total %.0f  prologue %.0f  DMA wait %.0f  barrier %.0f  epilogue %.0f  (%d workgroups)' % (
            buf[0] / wg, buf[1] / wg, buf[2] / wg, buf[3] / wg, buf[4] / wg, buf[5]))


for l in range(1, 5):
    cout, k = CONVS[l]
    Hi, Wi, Ci = dims[l]
    Ho, Wo, _ = dims[l + 1]
    p = k - 1
    w = (torch.randn(cout, Ci, k, k, device=dev) * (1.0 / (Ci * k * k) ** 0.5))
    bias = torch.randn(cout, device=dev) * 0.1
    w16 = w.bfloat16().float()
    # ---------------- forward
    name = '%df' % (l + 1)
    if want is None or name in want:
        x = torch.randn(B, Hi, Wi, Ci, device=dev).bfloat16()
        g = Geom(Hi, Wi, Ci, Ho, Wo, k, k, Ho, Wo, cout)
        assert ops.conv16_supported(B, cout, g), name
        Wg = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()          # [n][kh][kw][c]
        wf = prep(Wg, cout, k * k * Ci, k * Ci, k)
        for odt in (torch.bfloat16, torch.float32):
            out = torch.empty(B, Ho, Wo, cout, dtype=odt, device=dev)
            ops.conv16(x, wf, bias, None, out, B, cout, EPI_BIAS_LRELU, g)
            ref = F.leaky_relu(F.conv2d(x.float().permute(0, 3, 1, 2), w16, bias), 0.3).permute(0, 2, 3, 1)
            if odt == torch.bfloat16:
                ref = ref.bfloat16()
            err = (out.float() - ref.float()).abs().max().item()
            print(name, odt, 'max abs err %.3e (ref max %.3f)' % (err, ref.float().abs().max().item()))
        timeit(lambda: ops.conv16(x, wf, bias, None, out, B, cout, EPI_BIAS_LRELU, g), 2.0 * B * Ho * Wo * cout * k * k * Ci, name + ' fwd')
    # ---------------- data gradient: input = zero-bordered dOut [B][Ho+2p][Wo+2p][cout], output = dIn interior of a bordered buffer
    name = '%db' % (l + 1)
    if want is None or name in want:
        pp = CONVS[l - 1][1] - 1
        gb = torch.zeros(B, Ho + 2 * p, Wo + 2 * p, cout, device=dev).bfloat16()
        dy = torch.randn(B, Ho, Wo, cout, device=dev).bfloat16()
        gb[:, p:p + Ho, p:p + Wo] = dy
        act = torch.randn(B, Hi, Wi, Ci, device=dev).bfloat16()
        OH, OW = Hi + 2 * pp, Wi + 2 * pp
        gdg = Geom(Ho + 2 * p, Wo + 2 * p, cout, Hi, Wi, k, k, OH, OW, Ci, 0, 0, pp, pp)
        assert ops.conv16_supported(B, Ci, gdg), name
        Wb = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Ci, -1).contiguous()   # [c][kh'][kw'][n]
        wfb = prep(Wb, Ci, k * k * cout, k * cout, k)
        for odt in (torch.bfloat16, torch.float32):
            out = torch.zeros(B, OH, OW, Ci, dtype=odt, device=dev)
            ops.conv16(gb, wfb, None, act, out, B, Ci, EPI_MASK_LRELU_GRAD, gdg)
            ref = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w16).permute(0, 2, 3, 1)
            ref = torch.where(act.float() > 0, ref, 0.3 * ref)
            if odt == torch.bfloat16:
                ref = ref.bfloat16()
            got = out[:, pp:pp + Hi, pp:pp + Wi]
            err = (got.float() - ref.float()).abs().max().item()
            border = out.float().abs().sum().item() - got.float().abs().sum().item()
            print(name, odt, 'max abs err %.3e (ref max %.3f) border sum %.1e' % (err, ref.float().abs().max().item(), border))
        timeit(lambda: ops.conv16(gb, wfb, None, act, out, B, Ci, EPI_MASK_LRELU_GRAD, gdg), 2.0 * B * Hi * Wi * Ci * k * k * cout, name + ' dgrad')
