"""The generator and the discriminators driven through the C ABI alone: ctypes on libnele_hip.so + torch tensors as device memory.  No
nele_gan_amd import - this is the binding a reference-side maintainer (or any other host language) would write against include/nele_hip.h
(INTEGRATION.md section 2b): flat parameter / gradient buffers in nn.Module.parameters() order, one workspace per shape, the plans of
nele_gen_plan_build / nele_disc_plan_build, and the composite calls nele_gen_fwd / nele_gen_bwd / nele_disc_fwd / nele_disc_bwd."""
import ctypes
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = ctypes.CDLL(os.path.join(ROOT, 'nele_gan_amd', 'libnele_hip.so'))
P, I, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
LIB.nele_last_error_string.restype = ctypes.c_char_p
LIB.nele_gen_param_count.restype = LL
LIB.nele_gen_workspace_bytes.restype = LL
LIB.nele_gen_workspace_bytes.argtypes = [I, I, I, I]
LIB.nele_gen_param_layout.argtypes = [ctypes.POINTER(LL), I]
LIB.nele_gen_plan_build.argtypes = [I, I, I, I, I, P, P, P, LL, P, ctypes.POINTER(P), ctypes.POINTER(P)]
LIB.nele_disc_param_count.restype = LL
LIB.nele_disc_param_count.argtypes = [I, I]
LIB.nele_disc_param_layout.argtypes = [I, I, ctypes.POINTER(LL), I]
LIB.nele_disc_workspace_bytes.restype = LL
LIB.nele_disc_workspace_bytes.argtypes = [I, I, I, I]
LIB.nele_disc_workspace_ddin.restype = P
LIB.nele_disc_workspace_ddin.argtypes = [P, I, I, I, I]
LIB.nele_disc_plan_build.argtypes = [I, I, I, I, I, I, I, I, I, P, P, ctypes.POINTER(P), P, LL, P, ctypes.POINTER(P), ctypes.POINTER(P)]
LIB.nele_gen_fwd.argtypes = [P, P, P, P, ctypes.c_uint, ctypes.POINTER(P), I]
LIB.nele_gen_bwd.argtypes = [P, P, P, ctypes.POINTER(P), I]
LIB.nele_disc_fwd.argtypes = [P, P, P, P, ctypes.POINTER(P), I]
LIB.nele_disc_bwd.argtypes = [P, P, P, P, P, ctypes.POINTER(P), I]
LIB.nele_plan_destroy.argtypes = [P]
LIB.nele_plan_run_sized.argtypes = [P, ctypes.POINTER(P), I, ctypes.POINTER(LL), ctypes.POINTER(LL), I]
LIB.nele_plan_slot_bytes.restype = LL
LIB.nele_plan_slot_bytes.argtypes = [P, I]

G_LAYERS = [(128, 256, 5), (256, 256, 7), (256, 256, 7), (256, 256, 7), (256, 256, 7), (256, 64, 5)]
D_CONVS = [(8, 1), (16, 3), (32, 5), (48, 7), (64, 9)]


def ck(st, what):
    if st != 0:
        raise RuntimeError('%s: status %d: %s' % (what, st, LIB.nele_last_error_string().decode()))


def g_keys_shapes():
    """state_dict keys / shapes of model.py:43-82 in parameters() order"""
    out = []
    for l, (cin, cout, k) in enumerate(G_LAYERS):
        out += [('convolutions.%d.0.conv.weight' % l, (cout, cin, k)), ('convolutions.%d.0.conv.bias' % l, (cout,)),
                ('convolutions.%d.2.gain0' % l, (1, cout, 1)), ('convolutions.%d.2.bias0' % l, (1, cout, 1))]
    out += [('fc1.weight', (64, 64)), ('fc1.bias', (64,)), ('fc2.weight', (64, 64)), ('fc2.bias', (64,))]
    return out


def d_keys_shapes(cin=3, nout=3):
    """state_dict keys / shapes of model.py:101-116 (torch.nn.utils.spectral_norm: bias, weight_orig, weight_u, weight_v per layer)"""
    out, ci = [], cin
    for l, (co, k) in enumerate(D_CONVS):
        out += [('layers.%d.bias' % l, (co,)), ('layers.%d.weight_orig' % l, (co, ci, k, k)), ('layers.%d.weight_u' % l, (co,)),
                ('layers.%d.weight_v' % l, (ci * k * k,))]
        ci = co
    for name, (n, kk) in (('fc1', (64, 64)), ('fc2', (16, 64)), ('fc3', (nout, 16))):
        out += [('%s.bias' % name, (n,)), ('%s.weight_orig' % name, (n, kk)), ('%s.weight_u' % name, (n,)), ('%s.weight_v' % name, (kk,))]
    return out


def _stream():
    return P(torch.cuda.current_stream().cuda_stream)


class CapiGenerator:
    def __init__(self, B, T, state, bf16=False, need_bwd=True, overlap=True):
        self.B, self.T = B, T
        n = int(LIB.nele_gen_param_count())
        offs = (LL * 28)()
        assert LIB.nele_gen_param_layout(offs, 28) == 28
        self.names = [k for k, _ in g_keys_shapes()]
        self.offsets = {k: int(offs[i]) for i, k in enumerate(self.names)}
        self.shapes = dict(g_keys_shapes())
        flat = np.zeros(n, dtype=np.float32)
        for k in self.names:
            a = np.asarray(state[k], dtype=np.float32).ravel()
            flat[self.offsets[k]:self.offsets[k] + a.size] = a
        self.flat = torch.from_numpy(flat).cuda()
        self.grad = torch.zeros_like(self.flat)
        nb = int(LIB.nele_gen_workspace_bytes(B, T, int(bf16), int(need_bwd)))
        self.ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
        self.fwd, self.bwd = P(), P()
        ck(LIB.nele_gen_plan_build(B, T, int(bf16), int(need_bwd), int(overlap), P(self.flat.data_ptr()), P(self.grad.data_ptr()), P(self.ws.data_ptr()), nb, _stream(),
                                   ctypes.byref(self.fwd), ctypes.byref(self.bwd)), 'nele_gen_plan_build')
        self.side = torch.cuda.Stream()
        self.token = 0

    def _streams(self):
        return (P * 2)(torch.cuda.current_stream().cuda_stream, self.side.cuda_stream)

    def forward(self, x, y):
        mask = torch.empty((self.B, self.T, 64), device='cuda')
        self.token += 6
        ck(LIB.nele_gen_fwd(self.fwd, P(x.data_ptr()), P(y.data_ptr()), P(mask.data_ptr()), self.token - 5, self._streams(), 2), 'nele_gen_fwd')
        return mask

    def backward(self, dmask, mask):
        ck(LIB.nele_gen_bwd(self.bwd, P(dmask.data_ptr()), P(mask.data_ptr()), self._streams(), 2), 'nele_gen_bwd')

    def grad_of(self, k):
        o = self.offsets[k]
        return self.grad[o:o + int(np.prod(self.shapes[k]))].view(self.shapes[k])

    def close(self):
        for h in (self.fwd, self.bwd):
            if h:
                LIB.nele_plan_destroy(h)
        self.fwd = self.bwd = P()


class CapiDiscriminator:
    def __init__(self, B, T, state, cin=3, nout=3, bf16=False, train=True, need_din=True, weight_grads=True, overlap=True, backward=True):
        self.B, self.T, self.cin, self.nout, self.bf16 = B, T, cin, nout, bf16
        n = int(LIB.nele_disc_param_count(cin, nout))
        offs = (LL * 16)()
        assert LIB.nele_disc_param_layout(cin, nout, offs, 16) == 16
        ks = d_keys_shapes(cin, nout)
        self.pnames = [k for k, _ in ks if k.endswith('bias') or k.endswith('weight_orig')]
        self.shapes = dict(ks)
        self.offsets = {k: int(offs[i]) for i, k in enumerate(self.pnames)}
        flat = np.zeros(n, dtype=np.float32)
        for k in self.pnames:
            a = np.asarray(state[k], dtype=np.float32).ravel()
            flat[self.offsets[k]:self.offsets[k] + a.size] = a
        self.flat = torch.from_numpy(flat).cuda()
        self.grad = torch.zeros_like(self.flat)
        self.uv = [torch.from_numpy(np.asarray(state[k], dtype=np.float32)).cuda() for k, _ in ks if k.endswith('weight_u') or k.endswith('weight_v')]
        self.uv_names = [k for k, _ in ks if k.endswith('weight_u') or k.endswith('weight_v')]
        uvp = (P * 16)(*[t.data_ptr() for t in self.uv])
        nb = int(LIB.nele_disc_workspace_bytes(B, T, cin, int(bf16)))
        self.ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
        self.fwd, self.bwd = P(), P()
        ck(LIB.nele_disc_plan_build(B, T, cin, nout, int(bf16), int(train), int(need_din), int(weight_grads), int(overlap), P(self.flat.data_ptr()), P(self.grad.data_ptr()), uvp,
                                    P(self.ws.data_ptr()), nb, _stream(), ctypes.byref(self.fwd), ctypes.byref(self.bwd) if backward else None), 'nele_disc_plan_build')
        self.side = (torch.cuda.Stream(), torch.cuda.Stream())

    def _streams(self):
        return (P * 3)(torch.cuda.current_stream().cuda_stream, self.side[0].cuda_stream, self.side[1].cuda_stream)

    def forward(self, din, wvalid=None):
        score = torch.empty((self.B, self.nout), device='cuda')
        ck(LIB.nele_disc_fwd(self.fwd, P(din.data_ptr()), P(wvalid.data_ptr()) if wvalid is not None else None, P(score.data_ptr()), self._streams(), 3), 'nele_disc_fwd')
        return score

    def backward(self, dscore, score, din, wvalid=None):
        ck(LIB.nele_disc_bwd(self.bwd, P(dscore.data_ptr()), P(score.data_ptr()), P(wvalid.data_ptr()) if wvalid is not None else None, P(din.data_ptr()), self._streams(), 3),
           'nele_disc_bwd')

    def ddin(self):
        """view of the input gradient [B][64][T][4] inside the workspace"""
        p = LIB.nele_disc_workspace_ddin(P(self.ws.data_ptr()), self.B, self.T, self.cin, int(self.bf16))
        off = int(p) - self.ws.data_ptr()
        n = self.B * 64 * self.T * 4
        return self.ws[off:off + 4 * n].view(torch.float32).view(self.B, 64, self.T, 4)

    def grad_of(self, k):
        o = self.offsets[k]
        return self.grad[o:o + int(np.prod(self.shapes[k]))].view(self.shapes[k])

    def close(self):
        for h in (self.fwd, self.bwd):
            if h:
                LIB.nele_plan_destroy(h)
        self.fwd = self.bwd = P()
