"""Generator / discriminators of NELE-GAN on the MI355X: the host-side mirror of the reference ``model.py``.

Same class names, ``forward`` signatures, shapes and ``state_dict`` keys as the reference
(``Generator_Conv1D_cLN.forward(x[B,T,64], y[B,T,64]) -> [B,T,64]`` model.py:83;
``Discriminator.forward(x[B,3,64,T]) -> [B,3]`` model.py:118; ``Discriminator_Quality`` model.py:152),
so reference checkpoints ('enhance-model' / 'intel-model', train_nele.py:274-277) load unchanged.

The torch ``nn.Module`` objects only OWN the parameters (torch's Conv1d / Conv2d / Linear /
spectral_norm containers are used so that keys, shapes and initialisation match by construction; they
are never called).  Every forward and backward pass runs in libnele_hip.so: implicit-GEMM convolutions
on the f32 matrix cores (csrc/dense.hip), cLN scan, spectral-norm power iteration, pooling + MLP head
(csrc/gen.hip, csrc/disc.hip).  ``torch.autograd.Function`` is the glue that lets the reference's loop
(``loss.backward()``) drive those kernels; parameter gradients are accumulated straight into a flat
gradient buffer (one bucket per model: what Adam and the RCCL all-reduce operate on).
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from . import ops
from . import _lib
from ._lib import c_void_p, call, ptr, stream
from .ops import EPI_BIAS, EPI_BIAS_EXPTANH, EPI_BIAS_LRELU, EPI_MASK_LRELU_GRAD, EPI_NONE, SLOPE, Geom


# ------------------------------------------------------------------ parameter containers (reference names)
class ConvNorm(nn.Module):
    """model.py:10-28 (parameter container; the convolution itself runs in nele_conv_gemm)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1, bias=True,
                 w_init_gain='linear'):
        super().__init__()
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(w_init_gain))


class Chomp1d(nn.Module):
    """model.py:31-40: dropping the right overhang of a conv padded by k-1 == causal left padding."""

    def __init__(self, chomp_size):
        super().__init__()
        self.chomp_size = chomp_size


class cLN(nn.Module):
    """model.py:168-205 (parameter container; the scan runs in nele_cln_fwd / nele_cln_bwd)."""

    def __init__(self, dimension, eps=1e-8, trainable=True):
        super().__init__()
        self.eps = eps
        self.gain0 = nn.Parameter(torch.ones(1, dimension, 1), requires_grad=trainable)
        self.bias0 = nn.Parameter(torch.zeros(1, dimension, 1), requires_grad=trainable)


def _norm_dev(dev):
    """torch.device('cuda') != torch.device('cuda:0'): an index-less device would make every cache below look stale (and re-home the
    flat parameter buffer on every call).  Normalise to the indexed device."""
    dev = torch.device(dev)
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    return dev


class FlatParams:
    """All parameters of a module as views into ONE flat buffer (+ one flat gradient buffer)."""

    def __init__(self, module):
        self.module = module
        self.flat = None
        self.grad = None
        self._plist = None       # [(parameter, offset)] of the flattened module (ensure() fills it)

    def params(self):
        return [p for p in self.module.parameters()]

    def valid(self, device):
        """True while every parameter (and its gradient) is still the view into the flat buffers that ensure() made.  Checked on every
        forward / prepare (ten times per training step), so it walks the list ensure() kept instead of module.parameters() - the
        recursive enumeration was 75 of this call's 100 us.  Replacing a parameter's storage (.to(), .data = ...) is noticed here;
        ADDING a parameter to a module after its first forward pass is not supported."""
        if self.flat is None or self._plist is None:
            return False
        device = _norm_dev(device)
        if self.flat.device != device:
            return False
        f0, g0 = self.flat.data_ptr(), self.grad.data_ptr()
        for p, off in self._plist:
            g = p.grad
            if g is None or p.data_ptr() != f0 + off or g.data_ptr() != g0 + off:
                return False
        return True

    def ensure(self, device):
        device = _norm_dev(device)
        if self.valid(device):
            return
        ps = self.params()
        n = sum(p.numel() for p in ps)
        flat = torch.empty(n, dtype=torch.float32, device=device)
        grad = torch.zeros(n, dtype=torch.float32, device=device)
        off = 0
        cur = torch.cuda.current_stream(device) if device.type == 'cuda' else None
        for p in ps:
            k = p.numel()
            flat[off:off + k].copy_(p.data.reshape(-1).to(device))
            if p.grad is not None:
                grad[off:off + k].copy_(p.grad.reshape(-1).to(device))
            # The old storages lose their last reference below and go back to the allocator at once; it only knows the stream they were
            # ALLOCATED on.  If this runs on another stream (D.prepare on a side stream, first step of a fresh trainer) the copies above
            # may still be pending there when the main stream reuses the blocks: round 2 found D's initial weights silently corrupted
            # whenever the side stream lagged.  record_stream makes the allocator wait for this stream too.
            if cur is not None and p.data.is_cuda:
                p.data.record_stream(cur)
                if p.grad is not None and p.grad.is_cuda:
                    p.grad.record_stream(cur)
            p.data = flat[off:off + k].view(p.shape)
            p.grad = grad[off:off + k].view(p.shape)
            off += k
        self.flat, self.grad = flat, grad
        self._plist, off = [], 0
        for p in ps:
            self._plist.append((p, 4 * off))
            off += p.numel()


class _Anchor:
    """A 1-element leaf that requires grad, so autograd calls our backward even when the data
    inputs do not require grad (G-step: clean/noise features are constants)."""

    def __init__(self):
        self.t = None

    def get(self, device):
        device = _norm_dev(device)
        if self.t is None or self.t.device != device:
            self.t = torch.zeros(1, device=device, requires_grad=True)
        return self.t


def _zeros(shape, dev):
    return torch.zeros(shape, dtype=torch.float32, device=dev)


def _empty(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


# ================================================================== Generator
_G_LAYERS = [(128, 256, 5), (256, 256, 7), (256, 256, 7), (256, 256, 7), (256, 256, 7), (256, 64, 5)]


MAX_BUFFER_SHAPES = int(os.environ.get('NELE_MAX_BUFFER_SHAPES', '16'))   # (a corpus of files of many lengths: D batches of ~18 padded shapes; 288 GB of HBM)


def _lru_get(cache, key, make):
    """Per-(B, T) activation buffer sets, least-recently-used bounded: a long run over padded batches of many different shapes (D epochs
    on a real corpus) would otherwise pin one multi-hundred-MB buffer set per shape until the process ends."""
    bf = cache.pop(key, None)
    if bf is None:
        while len(cache) >= MAX_BUFFER_SHAPES:
            cache.pop(next(iter(cache)))             # dicts keep insertion order: the first key is the least recently used
        bf = make()
    cache[key] = bf                                  # (re-)insert as most recent
    return bf


def _check_generation(ctx, name):
    """The activation buffers are cached per (B, T): a second forward of the same shape overwrites what the first graph's backward
    pass needs.  Each forward bumps the buffer set's generation; a backward whose generation is stale raises instead of silently
    differentiating the wrong activations."""
    if ctx.key not in ctx.module._bufs:
        raise RuntimeError("%s: backward() after %d other (batch, frames) shapes went through the module - the cached activations of "
                           "this graph have been evicted (raise NELE_MAX_BUFFER_SHAPES)" % (name, MAX_BUFFER_SHAPES))
    if getattr(ctx, 'precision', None) is not None and ctx.module.precision != ctx.precision:
        raise RuntimeError("%s: `precision` changed from %r to %r between forward() and backward() - the cached activations were written "
                           "for the forward pass's operand type" % (name, ctx.precision, ctx.module.precision))
    if ctx.module._bufs[ctx.key].gen != ctx.gen:
        raise RuntimeError("%s: backward() after another forward pass of the same (batch, frames) shape - the cached activations of "
                           "this graph have been overwritten (run backward before the next forward of that shape)" % name)


class _GBuffers:
    """Activation buffers of one (B, T) shape.  Two tiers: the float32 set of the per-layer kernels (float32 mode; every backward pass)
    and the bfloat16 inputs of the fused layer kernel (csrc/glayer.hip, bf16 mode).  An evaluation-only user (inference.Enhancer: several
    buffer sets, one per batch in flight) never touches the float32 tier, so it is allocated on first use."""

    def __init__(self, B, T, dev):
        self.B, self.T, self.dev = B, T, dev
        self.gen = 0
        self.a5 = _empty((B, T, 64), dev)
        self.h1 = _empty((B, T, 64), dev)
        self.gfc = Geom(1, T, 64, 1, T, 1, 1, 1, T, 64)
        self.nchunks = int(ops._lib.lib.nele_cln_chunks(T))
        self._f32 = False
        self._stats = False
        self._bwd = False
        self._b16 = False
        self.dY = None
        self.dY16 = None
        self.token = 0
        self.plans = {}                                    # recorded passes of this shape (_lib.Plan), dropped with the buffer set
        self.events = ops.Events()

    def reserve_tokens(self, n):
        """n consecutive carry tokens for the fused layer launches of one forward pass; never 0 (zero-initialised slots must not match)"""
        if self.token + n >= 0xFFFFFFF0:
            self.token = 0
        self.token += n
        return self.token - n + 1

    def need_b16(self):
        """bf16 conv inputs [B][T+K-1][Cin] (time left-padded with K-1 zero rows) + the strip-carry slots of the fused layer kernel"""
        if self._b16:
            return
        B, T, dev = self.B, self.T, self.dev
        self.inp16 = [torch.zeros((B, T + k - 1, cin), dtype=torch.bfloat16, device=dev) for (cin, cout, k) in _G_LAYERS]
        self.carry = torch.zeros(max(32, int(ops._lib.lib.nele_glayer16_carry_bytes(B, T))), dtype=torch.uint8, device=dev)
        self._b16 = True

    def need_stats(self):
        """raw conv outputs [B][T][Cout] + per-frame statistics (what a backward pass reads)"""
        if self._stats:
            return
        B, T, dev = self.B, self.T, self.dev
        self.Y = [_empty((B, T, cout), dev) for (cin, cout, k) in _G_LAYERS]
        self.mean = [_empty((B, T), dev) for _ in _G_LAYERS]
        self.rstd = [_empty((B, T), dev) for _ in _G_LAYERS]
        self.cln_scratch = torch.empty((B, T, 2), dtype=torch.float64, device=dev)
        self._stats = True

    def need_f32(self):
        """float32 tier of the per-layer kernels: time-padded float32 inputs of each conv [B][T+K-1][Cin] (+ need_stats)"""
        if self._f32:
            return
        self.need_stats()
        B, T, dev = self.B, self.T, self.dev
        self.inp = [_zeros((B, T + k - 1, cin), dev) for (cin, cout, k) in _G_LAYERS]
        # forward geometries: H=1 conv over the padded time axis
        self.gf = [Geom(1, T + k - 1, cin, 1, T, 1, k, 1, T, cout) for (cin, cout, k) in _G_LAYERS]
        self._f32 = True

    def need_bwd(self, fused=False):
        """fused (bf16 mode on the fused layer kernels): the convolution operands of the backward pass are the bf16 buffers - no float32
        inputs, no float32 output gradients"""
        if not fused and not self._f32:
            self.need_f32()
        if not fused and self.dY is None:
            self.dY = [_zeros((self.B, self.T + k - 1, cout), self.dev) for (cin, cout, k) in _G_LAYERS]   # END-padded
        if fused:
            self.need_stats()
            self.need_dgrad16()
        if self._bwd:
            return
        B, T, dev = self.B, self.T, self.dev
        self.dA = [_empty((B, T, cin), dev) for (cin, cout, k) in _G_LAYERS]              # grad wrt conv input (index l: input of layer l)
        self.da5 = _empty((B, T, 64), dev)
        self.do2 = _empty((B, T, 64), dev)
        self.dpre1 = _empty((B, T, 64), dev)
        self.gpart = _empty((B * self.nchunks, 256), dev)
        self.bpart = _empty((B * self.nchunks, 256), dev)
        # data-gradient geometries: input = END-padded dY_l [B][T+k-1][cout], output = dA_l [B][T][cin]
        self.gb = [Geom(1, T + k - 1, cout, 1, T, 1, k, 1, T, cin) for (cin, cout, k) in _G_LAYERS]
        # weight-gradient geometries: A = padded input of layer l, dOut = dY_l (buffer width T+k-1)
        self.gw = [Geom(1, T + k - 1, cin, 1, T, 1, k, 1, T + k - 1, cout) for (cin, cout, k) in _G_LAYERS]
        self.gwfc = Geom(1, T, 64, 1, T, 1, 1, 1, T, 64)
        nws = max([ops.wgrad_workspace_floats(B, cout, g) for (cin, cout, k), g in zip(_G_LAYERS, self.gw)] +
                  [ops.wgrad_workspace_floats(B, 64, self.gwfc)])
        self.ws = _empty((nws,), dev)
        self._bwd = True

    def need_dgrad16(self):
        """END-padded bf16 output gradients [B][T+K-1][Cout]: what nele_glayer16_conv multiplies with the flipped weights"""
        if self.dY16 is None:
            self.dY16 = [torch.zeros((self.B, self.T + k - 1, cout), dtype=torch.bfloat16, device=self.dev) for (cin, cout, k) in _G_LAYERS]


class _GFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, anchor, module):
        ctx.module = module
        ctx.precision = module.precision
        ctx.key = module._forward_impl(x, y, need_bwd=True)
        ctx.gen = module._bufs[ctx.key].gen
        mask = module._last_mask
        module._last_mask = None             # the output must not stay reachable from ctx except through save_for_backward:
        ctx.save_for_backward(mask)          # output -> grad_fn -> ctx -> output is a cycle the collector cannot free (2 MB per G-step)
        return mask

    @staticmethod
    def backward(ctx, dmask):
        (mask,) = ctx.saved_tensors
        _check_generation(ctx, 'Generator_Conv1D_cLN')
        ctx.module._backward_impl(dmask.contiguous(), ctx.key, mask)
        return None, None, None, None


class Generator_Conv1D_cLN(nn.Module):
    """model.py:43-98."""

    def __init__(self):
        super().__init__()
        self.convolutions = nn.ModuleList()
        self.convolutions.append(nn.Sequential(
            ConvNorm(64 * 2, 256, kernel_size=5, stride=1, padding=int(5 - 1), dilation=1, w_init_gain='tanh'),
            Chomp1d(5 - 1), cLN(256)))
        for _ in range(1, 6 - 1):
            self.convolutions.append(nn.Sequential(
                ConvNorm(256, 256, kernel_size=7, stride=1, padding=int(7 - 1), dilation=1, w_init_gain='tanh'),
                Chomp1d(7 - 1), cLN(256)))
        self.convolutions.append(nn.Sequential(
            ConvNorm(256, 64, kernel_size=5, stride=1, padding=int(5 - 1), dilation=1, w_init_gain='linear'),
            Chomp1d(5 - 1), cLN(64)))
        self.LReLU = nn.LeakyReLU(0.3)
        self.fc1 = nn.Linear(64, 64)
        self.fc2 = nn.Linear(64, 64)
        self._flat = FlatParams(self)
        self._anchor = _Anchor()
        self._bufs = {}
        self._wf = None
        self._span_ok = {}
        self._wstream = None
        self.overlap_wgrad = True          # weight gradients on a second stream beside the data-gradient chain
        self._last_mask = None
        self.precision = 'f32'            # 'bf16': bf16 MFMA operands (f32 accumulate) in the Conv1d / Linear GEMMs (fwd, dgrad, wgrad)
        # Several forward passes in flight on different streams (inference.Enhancer.enhance_stream): each stream names its own activation
        # buffer set through `buffer_slot`, and the weight layouts - shared by all of them - are written ONCE by freeze_weights() instead
        # of at the head of every forward pass (a rewrite, even of identical values, beside another stream's reads is a race).
        self.buffer_slot = 0
        self._weights_frozen = False
        self.fused = True                 # bf16 mode: one fused launch per layer (conv + bias + cLN + LeakyReLU, csrc/glayer.hip) where the shapes allow
        self.fused_ok = False
        # new weights invalidate frozen layouts (inference.Enhancer keeps the layouts of a generator it owns across calls)
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.unfreeze_weights())

    # ---- plumbing
    def train(self, mode=True):
        """The sub-modules are parameter containers that are never called: only this module's own flag means anything, and
        nn.Module.train's walk over the 35 of them cost 0.12 ms per G.eval() / G.train() pair of every generate() call."""
        self.training = bool(mode)
        return self

    def flat_parameters(self, device=None):
        device = device or next(self.parameters()).device
        self._flat.ensure(device)
        return self._flat

    def _weights(self, dev):
        dev = _norm_dev(dev)
        if self._wf is None or self._wf[0][0].device != dev:
            self._wgen = getattr(self, '_wgen', 0) + 1       # generation of the weight-layout tensors: part of every recorded plan's key
            wf, wb = [], []
            for (cin, cout, k) in _G_LAYERS:
                wf.append(_zeros((cout, k * cin), dev))
                wb.append(_zeros((cin, k * cout), dev))
            wf.append(_zeros((64, 64), dev)); wb.append(_zeros((64, 64), dev))   # fc1
            wf.append(_zeros((64, 64), dev)); wb.append(_zeros((64, 64), dev))   # fc2
            self._wf = (wf, wb)
            # bf16 fragment streams for the Conv1d / Linear tile kernel: (N, seglen) per layer, forward and data-gradient
            dims = [(cout, k * cin, cin, k * cout) for (cin, cout, k) in _G_LAYERS] + [(64, 64, 64, 64), (64, 64, 64, 64)]
            self._wf16 = ([torch.zeros(ops.frag16_elems(nf, sf, 1), dtype=torch.bfloat16, device=dev) for (nf, sf, nb_, sb_) in dims],
                          [torch.zeros(ops.frag16_elems(nb_, sb_, 1), dtype=torch.bfloat16, device=dev) for (nf, sf, nb_, sb_) in dims])
            self._wf16dims = dims
            # fragment streams of the fused layer kernel (csrc/glayer.hip): forward [cout][k*cin] and flipped data-gradient [cin][k*cout]
            self.fused_ok = all(ops.glayer16_supported(cin, cout, k) for (cin, cout, k) in _G_LAYERS)
            self._wgl = None
            if self.fused_ok:
                self._wgl = ([torch.zeros(ops.glayer16_wfrag_elems(cin, cout, k), dtype=torch.bfloat16, device=dev) for (cin, cout, k) in _G_LAYERS],
                             [torch.zeros(ops.glayer16_wfrag_elems(cout, cin, k), dtype=torch.bfloat16, device=dev) if l > 0 else None
                              for l, (cin, cout, k) in enumerate(_G_LAYERS)])
        return self._wf

    def _prep_weights(self, dev):
        wf, wb = self._weights(dev)
        ws = [self.convolutions[l][0].conv.weight for l in range(len(_G_LAYERS))] + [self.fc1.weight, self.fc2.weight]
        ck = (ws[0].data_ptr(), ws[-1].data_ptr(), wf[0].data_ptr())
        if getattr(self, '_prepjobs', None) is None or self._prepjobs[0] != ck:
            pj, dj = [], []
            for l, (cin, cout, k) in enumerate(_G_LAYERS):
                pj += [ws[l].data_ptr(), None, wf[l].data_ptr(), wb[l].data_ptr()]
                dj += [cout, cin, cin, 1, k]
            for q in (6, 7):
                pj += [ws[q].data_ptr(), None, wf[q].data_ptr(), wb[q].data_ptr()]
                dj += [64, 64, 64, 1, 1]
            self._prepjobs = (ck, (c_void_p * len(pj))(*pj), (ctypes.c_int * len(dj))(*dj), len(dj) // 5)
        _, pja, dja, npj = self._prepjobs
        call('nele_weight_prep_batch', pja, dja, npj, stream())       # all 8 layers in one launch
        if self.precision == 'bf16':
            if getattr(self, '_fragjobs', None) is None or self._fragjobs[0] != ck:
                fj, ej = [], []
                for q, (nf, sf, nb_, sb_) in enumerate(self._wf16dims):
                    fj += [wf[q].data_ptr(), self._wf16[0][q].data_ptr()]
                    ej += [nf, sf, sf, 1]
                for q, (nf, sf, nb_, sb_) in enumerate(self._wf16dims):
                    fj += [wb[q].data_ptr(), self._wf16[1][q].data_ptr()]
                    ej += [nb_, sb_, sb_, 1]
                self._fragjobs = (ck, (c_void_p * len(fj))(*fj), (ctypes.c_int * len(ej))(*ej), len(ej) // 4)
            _, fja, eja, nfj = self._fragjobs
            call('nele_weight_prep_frag16_batch', fja, eja, nfj, stream())   # 16 fragment streams in one launch
            if self.fused and self.fused_ok:
                if getattr(self, '_gljobs', None) is None or self._gljobs[0] != ck:
                    gj, hj = [], []
                    for l, (cin, cout, k) in enumerate(_G_LAYERS):
                        gj += [wf[l].data_ptr(), self._wgl[0][l].data_ptr()]
                        hj += [cout, cin, k]
                    for l, (cin, cout, k) in enumerate(_G_LAYERS):       # flipped layouts [cin][k * cout]: the data gradients of layers 1 .. 5
                        if l > 0:
                            gj += [wb[l].data_ptr(), self._wgl[1][l].data_ptr()]
                            hj += [cin, cout, k]
                    self._gljobs = (ck, (c_void_p * len(gj))(*gj), (ctypes.c_int * len(hj))(*hj), len(hj) // 3)
                _, gja, hja, ngj = self._gljobs
                call('nele_glayer16_weight_prep_batch', gja, hja, ngj, stream())
        return wf, wb

    def _get_bufs(self, B, T, dev):
        key = (B, T, str(_norm_dev(dev)), self.buffer_slot)
        return key, _lru_get(self._bufs, key, lambda: _GBuffers(B, T, dev))

    def freeze_weights(self, dev=None):
        """Write the GEMM / fragment layouts of the current parameters now (current stream) and skip that step in the forward passes
        that follow, until unfreeze_weights().  For evaluation loops whose parameters do not change (inference.py:79-117): forward
        passes on several streams then only READ the layouts.  Returns an event recorded behind the layout kernels."""
        dev = _norm_dev(dev or next(self.parameters()).device)
        self._flat.ensure(dev)
        self._weights_frozen = False
        self._prep_weights(dev)
        self._weights_frozen = True
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        return ev

    def unfreeze_weights(self):
        self._weights_frozen = False

    def _gemm(self, A, q, back, bias, aux, out, B, N, epi, g):
        """Conv1d / Linear GEMM of layer q (forward or data-gradient weights): the strip tile kernel in bf16 mode where the
        geometry fits it, the generic implicit GEMM otherwise."""
        b16 = self.precision == 'bf16'
        if b16:
            sk = (q, back, B, g.Wout)
            if sk not in self._span_ok:
                self._span_ok[sk] = ops.span16_supported(B, N, g)
            if self._span_ok[sk]:
                ops.conv_span_bf16(A, self._wf16[1 if back else 0][q], bias, aux, out, B, N, epi, g)
                return
        ops.conv_gemm(A, (self._wf[1] if back else self._wf[0])[q], bias, aux, out, B, N, epi, g, bf16=b16)

    def _fused_for(self, T, need_bwd):
        """bf16 mode: one fused launch per layer (csrc/glayer.hip) - for evaluation at any length; with a backward pass to follow only from
        32 frames on (the weight gradient's tile kernel, which takes the bf16 operands the fused forward pass leaves, needs rows of >= 32
        output positions; shorter utterances keep the float32 buffers of the per-layer kernels)."""
        return self.precision == 'bf16' and self.fused and self.fused_ok and (not need_bwd or T >= 32)

    # ---- forward (model.py:83-98)
    def _forward_impl(self, x, y, need_bwd=False):
        if x.dim() != 3 or x.shape[2] != 64 or y.shape != x.shape:
            raise ValueError("Generator_Conv1D_cLN.forward: x and y must both be [B, T, 64]")
        if not x.is_cuda:
            raise RuntimeError("nele_gan_amd: the generator runs on the GPU only (no CPU fallback)")
        dev = x.device
        self._flat.ensure(dev)
        B, T, _ = x.shape
        key, bf = self._get_bufs(B, T, dev)
        bf.gen += 1
        if self._weights_frozen and need_bwd:
            self.unfreeze_weights()           # a pass that will be differentiated: the layouts follow the weights again (an evaluation loop
                                              # freezes them anew)
        self._weights(dev)
        fused = self._fused_for(T, need_bwd)
        xs, ys = x.contiguous().float(), y.contiguous().float()
        mask = _empty((B, T, 64), dev)
        tok0 = bf.reserve_tokens(len(_G_LAYERS))
        # the pass as ONE call (nele_gen_fwd on the job table recorded the first time this shape / mode came by), or call by call
        pkey = ('fwd', bool(need_bwd), self.precision, fused, self._weights_frozen, self._flat.flat.data_ptr(), self._wf[0][0].data_ptr(), self._wgen)
        plan = bf.plans.get(pkey) if ops.plans_enabled() else None
        if plan is not None:
            plan.streams[0] = stream()
            call('nele_gen_fwd', plan.handle, ptr(xs), ptr(ys), ptr(mask), tok0, plan.streams, len(plan.streams))
        elif ops.plans_enabled():
            with _lib.recording([ops.rng(xs), ops.rng(ys), ops.rng(mask), None]) as rec:
                self._forward_live(xs, ys, mask, bf, need_bwd, fused, tok0)
            bf.plans[pkey] = rec.finish()
        else:
            self._forward_live(xs, ys, mask, bf, need_bwd, fused, tok0)
        self._last_mask = mask
        return key

    def _forward_live(self, xs, ys, mask, bf, need_bwd, fused, tok0):
        B, T, dev = bf.B, bf.T, xs.device
        if not self._weights_frozen:
            self._prep_weights(dev)
        if fused:
            # one launch per layer; the float32 copies of the activations, the raw convolutions and the per-frame statistics are written
            # only for a backward pass
            bf.need_b16()
            if need_bwd:
                bf.need_bwd(fused=True)
            call('nele_g_pack16', ptr(xs), ptr(ys), ptr(bf.inp16[0]), B, T, _G_LAYERS[0][2] - 1, stream())
            nl = len(_G_LAYERS)
            for l, (cin, cout, k) in enumerate(_G_LAYERS):
                seq = self.convolutions[l]
                last = l + 1 == nl
                padn = 0 if last else _G_LAYERS[l + 1][2] - 1
                out16 = None if last else bf.inp16[l + 1]
                out32 = bf.a5 if last else None              # (the Linear tail reads float32; every Conv1d operand is the bf16 buffer)
                call('nele_glayer16_fwd', ptr(bf.inp16[l]), ptr(self._wgl[0][l]), ptr(seq[0].conv.bias), ptr(seq[2].gain0), ptr(seq[2].bias0),
                     ptr(bf.Y[l]) if need_bwd else None, ptr(bf.mean[l]) if need_bwd else None, ptr(bf.rstd[l]) if need_bwd else None,
                     ptr(out16), ptr(out32), ptr(bf.carry), _lib.DynInt(3, l, tok0), B, T, cin, cout, k, padn, SLOPE, stream())
        else:
            bf.need_f32()
            if need_bwd:
                bf.need_bwd()
            call('nele_g_pack', ptr(xs), ptr(ys), ptr(bf.inp[0]), B, T, _G_LAYERS[0][2] - 1, stream())
            for l, (cin, cout, k) in enumerate(_G_LAYERS):
                seq = self.convolutions[l]
                self._gemm(bf.inp[l], l, False, seq[0].conv.bias, None, bf.Y[l], B, cout, EPI_BIAS, bf.gf[l])
                if l + 1 < len(_G_LAYERS):
                    nxt, pad = bf.inp[l + 1], _G_LAYERS[l + 1][2] - 1
                else:
                    nxt, pad = bf.a5, 0
                call('nele_cln_fwd', ptr(bf.Y[l]), ptr(seq[2].gain0), ptr(seq[2].bias0), ptr(nxt), ptr(bf.mean[l]), ptr(bf.rstd[l]),
                     ptr(bf.cln_scratch), B, T, cout, pad, SLOPE, stream())
        self._gemm(bf.a5, 6, False, self.fc1.bias, None, bf.h1, B, 64, EPI_BIAS_LRELU, bf.gfc)
        self._gemm(bf.h1, 7, False, self.fc2.bias, None, mask, B, 64, EPI_BIAS_EXPTANH, bf.gfc)

    def forward(self, x, y):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _GFn.apply(x, y, self._anchor.get(x.device), self)
        self._forward_impl(x, y)
        mask, self._last_mask = self._last_mask, None
        return mask

    # ---- backward: accumulates into the flat gradient buffer
    def _backward_impl(self, dmask, key, mask):
        bf = self._bufs[key]
        fused = self._fused_for(bf.T, True)
        bf.need_bwd(fused=fused)
        wst = None
        if self.overlap_wgrad:
            if self._wstream is None:
                self._wstream = ops.side_stream(dmask.device)
            wst = self._wstream
        pkey = ('bwd', self.precision, fused, None if wst is None else wst.cuda_stream, self._flat.flat.data_ptr(), self._flat.grad.data_ptr(), self._wf[0][0].data_ptr(),
                self._wgen)
        plan = bf.plans.get(pkey) if ops.plans_enabled() else None
        if plan is not None:
            plan.streams[0] = stream()
            call('nele_gen_bwd', plan.handle, ptr(dmask), ptr(mask), plan.streams, len(plan.streams))
        elif ops.plans_enabled():
            with _lib.recording([ops.rng(dmask), ops.rng(mask)]) as rec:
                self._backward_live(dmask, mask, bf, wst, fused)
            bf.plans[pkey] = rec.finish()
        else:
            self._backward_live(dmask, mask, bf, wst, fused)

    def _backward_live(self, dmask, mask, bf, wst, fused=False):
        B, T = bf.B, bf.T
        # data-gradient chain on the current stream, weight gradients on a second stream beside it (see _DiscriminatorBase)
        main = torch.cuda.current_stream()
        if wst is not None and wst == main:
            wst = None
        b16w = self.precision == 'bf16'
        ev = bf.events
        ev.start()

        def wgrad(A, dOut, N, g, Cvalid, dW, db, bf16=False):
            if wst is None:
                ops.conv_wgrad(A, dOut, bf.ws, B, N, g, Cvalid, dW, db, bf16=bf16)
                return
            ops.hand_over(ev, main, wst)
            with torch.cuda.stream(wst):
                ops.conv_wgrad(A, dOut, bf.ws, B, N, g, Cvalid, dW, db, bf16=bf16)

        call('nele_exptanh_bwd', ptr(dmask), ptr(mask), ptr(bf.do2), dmask.numel(), stream())
        # fc2
        wgrad(bf.h1, bf.do2, 64, bf.gwfc, 64, self.fc2.weight.grad, self.fc2.bias.grad)
        self._gemm(bf.do2, 7, True, None, bf.h1, bf.dpre1, B, 64, EPI_MASK_LRELU_GRAD, bf.gfc)
        # fc1
        wgrad(bf.a5, bf.dpre1, 64, bf.gwfc, 64, self.fc1.weight.grad, self.fc1.bias.grad)
        self._gemm(bf.dpre1, 6, True, None, None, bf.da5, B, 64, EPI_NONE, bf.gfc)
        dact = bf.da5
        for l in range(len(_G_LAYERS) - 1, -1, -1):
            cin, cout, k = _G_LAYERS[l]
            seq = self.convolutions[l]
            # fused path: the output gradient exists as bf16 only - the operand of the weight gradient (beside the bf16 input) and of the
            # data gradient; per-layer path: float32, converted while staged
            d16 = bf.dY16[l] if fused else None
            call('nele_cln_bwd', ptr(dact), ptr(bf.Y[l]), ptr(seq[2].gain0), ptr(seq[2].bias0), ptr(bf.mean[l]), ptr(bf.rstd[l]),
                 None if fused else ptr(bf.dY[l]), ptr(d16), ptr(bf.gpart), ptr(bf.bpart), ptr(bf.cln_scratch), B, T, cout, k - 1, SLOPE, stream())
            call('nele_colsum2', ptr(bf.gpart), ptr(seq[2].gain0.grad), ptr(bf.bpart), ptr(seq[2].bias0.grad), B * bf.nchunks, cout, 1, stream())
            if fused:
                wgrad(bf.inp16[l], d16, cout, bf.gw[l], cin, seq[0].conv.weight.grad, seq[0].conv.bias.grad, bf16=True)
            else:
                wgrad(bf.inp[l], bf.dY[l], cout, bf.gw[l], cin, seq[0].conv.weight.grad, seq[0].conv.bias.grad, bf16=b16w)
            if l > 0:
                if fused:
                    # data gradient = the layer kernel's plain-convolution form over the bf16 output gradient with the flipped weights
                    call('nele_glayer16_conv', ptr(d16), ptr(self._wgl[1][l]), ptr(bf.dA[l]), B, T, cout, cin, k, stream())
                else:
                    self._gemm(bf.dY[l], l, True, None, None, bf.dA[l], B, cin, EPI_NONE, bf.gb[l])
                dact = bf.dA[l]
        if wst is not None:
            ops.hand_over(ev, wst, main)


# ================================================================== Discriminators
_D_CONVS = [(8, 1), (16, 3), (32, 5), (48, 7), (64, 9)]   # (Cout, k) ; Cin of layer 0 is 3 (D) or 2 (D_Qua), padded to 4


class _DBuffers:
    def __init__(self, B, T, dev, cin0, mode16=False):
        """mode16 (bf16 operand mode): where every layer's kernels support it (csrc/conv16.hip + the weight-gradient tile kernel) the
        activations of conv1..conv4 and the output gradients of conv2..conv5 live in memory as bfloat16 - their consumers round them to
        bf16 anyway.  conv5's activation (pooled in float32) and conv1's output gradient (float32 pointwise kernels) stay float32."""
        self.B, self.T = B, T
        self.gen = 0
        self.wvalid = None
        self.prepjobs = None               # (key, ctypes job tables) of the batched weight-layout launches for this shape
        self.plans = {}                    # recorded passes of this shape (_lib.Plan), dropped with the buffer set
        self.events = ops.Events()
        H, W, C = 64, T, 4
        self.dims = [(H, W, C)]
        self.act, self.gf, self.gbuf, self.gb, self.gw, self.pad = [], [], [], [], [], []
        shapes = []
        for (cout, k) in _D_CONVS:
            Ho, Wo = H - k + 1, W - k + 1
            if Ho < 1 or Wo < 1:
                raise ValueError("Discriminator: T=%d is too short (needs T >= 21, model.py:105-109)" % T)
            self.gf.append(Geom(H, W, C, Ho, Wo, k, k, Ho, Wo, cout))
            p = k - 1
            self.pad.append(p)
            shapes.append(((B, Ho, Wo, cout), (B, Ho + 2 * p, Wo + 2 * p, cout)))
            self.dims.append((Ho, Wo, cout))
            H, W, C = Ho, Wo, cout
        self._shapes = shapes
        self.ddin = _empty((B, 64, T, 4), dev)
        for l, (cout, k) in enumerate(_D_CONVS):
            Hi, Wi, Ci = self.dims[l]
            Ho, Wo, _ = self.dims[l + 1]
            p = k - 1
            if l == 0:
                OH, OW, OC, o0 = Hi, Wi, 4, 0
            else:
                pp = self.pad[l - 1]
                OH, OW, OC, o0 = Hi + 2 * pp, Wi + 2 * pp, Ci, pp
            # data gradient: input = gbuf[l] (padded), output positions = layer input positions
            self.gb.append(Geom(Ho + 2 * p, Wo + 2 * p, cout, Hi, Wi, k, k, OH, OW, OC, 0, 0, o0, o0))
            # weight gradient: A = layer input, dOut = gbuf[l] interior
            self.gw.append(Geom(Hi, Wi, Ci, Ho, Wo, k, k, Ho + 2 * p, Wo + 2 * p, cout, 0, 0, p, p))
        cins = [4] + [c for (c, k) in _D_CONVS[:-1]]
        nl = len(_D_CONVS)
        self.c16 = bool(mode16) and all(ops.conv16_supported(B, _D_CONVS[l][0], self.gf[l]) and ops.conv16_supported(B, cins[l], self.gb[l]) and
                                        ops.wgrad_tile_supported(B, _D_CONVS[l][0], self.gw[l]) for l in range(1, nl))
        b16 = torch.bfloat16
        for l, (ashape, gshape) in enumerate(shapes):
            # (the last layer's activation too: its pooling happens in the conv kernel's epilogue on the float32 results - nele_conv16_gap -
            #  and the backward pass only needs the sign of the stored value)
            a16, g16 = self.c16, self.c16 and l >= 1
            self.act.append(torch.empty(ashape, dtype=b16 if a16 else torch.float32, device=dev))
            # zero-bordered gradient buffer of this layer's OUTPUT
            self.gbuf.append(torch.zeros(gshape, dtype=b16 if g16 else torch.float32, device=dev))
        self.span_f = [ops.span_supported(B, cout, g) for (cout, k), g in zip(_D_CONVS, self.gf)]
        self.span_b = [ops.span_supported(B, cin, g) for cin, g in zip(cins, self.gb)]
        self.span16_f = [ops.span16_supported(B, cout, g) for (cout, k), g in zip(_D_CONVS, self.gf)]
        self.span16_b = [ops.span16_supported(B, cin, g) for cin, g in zip(cins, self.gb)]
        # bf16 mode: the last layer's output gradient (the pooling gradient) as bfloat16 - its data gradient re-stages it once per kernel
        # row; allocated on first use (same shape and zero border as gbuf[-1])
        self.grad16_ok = (not self.c16) and ops.grad16_supported(B, cins[-1], self.gb[-1], _D_CONVS[-1][0], self.gw[-1])
        self.gbuf16 = None
        self.P = self.dims[-1][0] * self.dims[-1][1]
        self.gap_parts = ops.conv16_gap_parts(_D_CONVS[-1][0], self.gf[-1]) if self.c16 else 0
        self.gap_part = torch.empty((B, self.gap_parts, 64), dtype=torch.float64, device=dev) if self.c16 else None
        self.pooled, self.h1, self.h2 = _empty((B, 64), dev), _empty((B, 64), dev), _empty((B, 16), dev)
        self.dz1, self.dz2, self.dz3, self.dpooled = _empty((B, 64), dev), _empty((B, 16), dev), _empty((B, 4), dev), _empty((B, 64), dev)
        nws = max(ops.wgrad_workspace_floats(B, cout, g) for (cout, k), g in zip(_D_CONVS, self.gw))
        self.ws = _empty((nws,), dev)
        self.tmpw = _empty((64 * 48 * 81 + 64,), dev)
        self.scratch64 = torch.empty((max(B * 32 * 64, 128),), dtype=torch.float64, device=dev)   # largest weight tensor (sigma-normalised gradient staging)
        # second set of weight-gradient temporaries: the backward pass runs the layers' weight gradients on two streams
        self.ws2 = _empty((nws,), dev)
        self.tmpw2 = _empty((64 * 48 * 81 + 64,), dev)
        self.scratch64b = torch.empty((max(B * 32 * 64, 128),), dtype=torch.float64, device=dev)


class _DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, din, anchor, module, frames=None):
        ctx.module = module
        ctx.precision = module.precision
        ctx.key = module._forward_impl(din, frames)
        ctx.wvalid = module._bufs[ctx.key].wvalid
        ctx.gen = module._bufs[ctx.key].gen
        ctx.need_din = din.requires_grad
        score = module._last_score
        module._last_score = None
        ctx.save_for_backward(score)
        return score

    @staticmethod
    def backward(ctx, dscore):
        (score,) = ctx.saved_tensors
        _check_generation(ctx, 'Discriminator')
        ddin = ctx.module._backward_impl(dscore.contiguous(), ctx.key, ctx.need_din, score, ctx.wvalid)
        return ddin, None, None, None


class _NchwToPacked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.cin = x.shape[1]
        return ops.nchw_to_nhwc4(x.float())

    @staticmethod
    def backward(ctx, ddin):
        return ops.nhwc4_to_nchw(ddin.contiguous(), ctx.cin)


class _DiscriminatorBase(nn.Module):
    def __init__(self, cin, nout):
        super().__init__()
        self._cin, self._nout = cin, nout
        layers = [spectral_norm(nn.Conv2d(cin, 8, (1, 1))),
                  spectral_norm(nn.Conv2d(8, 16, (3, 3))),
                  spectral_norm(nn.Conv2d(16, 32, (5, 5))),
                  spectral_norm(nn.Conv2d(32, 48, (7, 7))),
                  spectral_norm(nn.Conv2d(48, 64, (9, 9)))]
        self.layers = nn.ModuleList(layers)
        self.GAPool = nn.AdaptiveAvgPool2d((1, 1))
        self.LReLU = nn.LeakyReLU(0.3)
        self.fc1 = spectral_norm(nn.Linear(64, 64))
        self.fc2 = spectral_norm(nn.Linear(64, 16))
        self.fc3 = spectral_norm(nn.Linear(16, nout))
        self._flat = FlatParams(self)
        self._anchor = _Anchor()
        self._bufs = {}
        self._w = None
        self._last_score = None
        self._prepared = None
        self._wstream = None
        self.overlap_wgrad = True          # weight gradients on a second stream beside the data-gradient chain
        self.profile_prefix = ''           # prepended to the ops.PROFILE tags of this module's launches (bench.py)
        self.weight_grad_enabled = True   # reference computes (unused) D weight grads in the G-step too
        self.precision = 'f32'            # 'bf16': bf16 MFMA operands (f32 accumulate) in the conv forward / data-gradient passes

    def flat_parameters(self, device=None):
        device = device or next(self.parameters()).device
        self._flat.ensure(device)
        return self._flat

    def _sn_modules(self):
        return list(self.layers) + [self.fc1, self.fc2, self.fc3]

    def train(self, mode=True):
        """see Generator_Conv1D_cLN.train (the spectral-norm hooks of the sub-modules never run: the power iteration is nele_spectral_norm)"""
        self.training = bool(mode)
        return self

    def _weights(self, dev):
        dev = _norm_dev(dev)
        if self._w is None or self._w['sigma'].device != dev:
            self._wgen = getattr(self, '_wgen', 0) + 1       # generation of the weight-layout tensors: part of every recorded plan's key
            w = {'sigma': _zeros((8,), dev), 'wf': [], 'wb': [], 'wff': [], 'wbf': [], 'wff16': [], 'wbf16': [], 'wf16c': [], 'wb16c': []}
            cin = 4
            for (cout, k) in _D_CONVS:
                # fragment streams of csrc/conv16.hip (bf16 activations in memory): forward [cout][k*k*cin], data gradient [cin][k*k*cout]
                w['wf16c'].append(torch.zeros(ops.conv16_wfrag_elems(cout, k * cin, k), dtype=torch.bfloat16, device=dev) if k > 1 else None)
                w['wb16c'].append(torch.zeros(ops.conv16_wfrag_elems(cin, k * cout, k), dtype=torch.bfloat16, device=dev) if k > 1 else None)
                w['wf'].append(_zeros((cout, k * k * cin), dev))
                w['wb'].append(_zeros((cin, k * k * cout), dev))
                w['wff'].append(_zeros((ops.frag_floats(cout, k * k * cin),), dev) if (k * k * cin) % 8 == 0 else None)
                w['wbf'].append(_zeros((ops.frag_floats(cin, k * k * cout),), dev) if (k * k * cout) % 8 == 0 else None)
                w['wff16'].append(torch.zeros(ops.frag16_elems(cout, k * cin, k), dtype=torch.bfloat16, device=dev))
                w['wbf16'].append(torch.zeros(ops.frag16_elems(cin, k * cout, k), dtype=torch.bfloat16, device=dev))
                cin = cout
            self._w = w
        return self._w

    def _get_bufs(self, B, T, dev):
        key = (B, T, str(_norm_dev(dev)), self.precision)
        return key, _lru_get(self._bufs, key, lambda: _DBuffers(B, T, dev, self._cin, mode16=self.precision == 'bf16'))

    def _mlp_ptrs(self, w):
        arr = (c_void_p * 9)()
        for i, m in enumerate((self.fc1, self.fc2, self.fc3)):
            arr[3 * i + 0] = m.weight_orig.data_ptr()
            arr[3 * i + 1] = m.bias.data_ptr()
            arr[3 * i + 2] = w['sigma'].data_ptr() + 4 * (5 + i)
        return arr

    def prepare(self, B, T, dev):
        """Spectral-norm power iteration + sigma (model.py:105-116) and every weight layout of a forward / backward pass at batch B,
        frames T, enqueued on the CURRENT stream.  The next forward of that shape waits for it instead of doing it inline:
        this work only depends on the parameters, so a training loop can run it on another stream ahead of the forward pass
        (GanTrainer does, beside the generator's forward pass and beside generate)."""
        self._flat.ensure(dev)
        key, bf = self._get_bufs(B, T, dev)
        w = self._weights(dev)
        self._prepare_inline(key, bf, w)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._prepared = (key, self.training, ev)

    def advance_power_iteration(self, dev=None):
        """One spectral-norm power iteration of every layer (what a training-mode forward pass does first, model.py:105-116), no data.
        An empty data-parallel step (this rank has run out of batches) calls it so that weight_u / weight_v advance in lock-step with
        the ranks that ran a real forward pass: all replicas then normalise by the same sigma and rank 0's checkpoint is what a
        single-GPU run of the same steps would hold."""
        if not self.training:
            return
        dev = _norm_dev(dev or next(self.parameters()).device)
        self._flat.ensure(dev)
        w = self._weights(dev)
        mods = self._sn_modules()
        pp = (c_void_p * (3 * len(mods)))()
        dd = (ctypes.c_int * (2 * len(mods)))()
        for i, m in enumerate(mods):
            pp[3 * i], pp[3 * i + 1], pp[3 * i + 2] = m.weight_orig.data_ptr(), m.weight_u.data_ptr(), m.weight_v.data_ptr()
            dd[2 * i] = m.weight_orig.shape[0]
            dd[2 * i + 1] = m.weight_orig.numel() // m.weight_orig.shape[0]
        call('nele_spectral_norm', pp, dd, len(mods), ptr(w['sigma']), 1, stream())

    def _prepare_inline(self, key, bf, w, power_iter=True):
        n_iter = 1 if (self.training and power_iter) else 0
        # spectral norm: power iteration (train mode) + sigma for all 8 layers in one launch (model.py:105-116)
        mods = self._sn_modules()
        pp = (c_void_p * (3 * len(mods)))()
        dd = (ctypes.c_int * (2 * len(mods)))()
        for i, m in enumerate(mods):
            pp[3 * i], pp[3 * i + 1], pp[3 * i + 2] = m.weight_orig.data_ptr(), m.weight_u.data_ptr(), m.weight_v.data_ptr()
            dd[2 * i] = m.weight_orig.shape[0]
            dd[2 * i + 1] = m.weight_orig.numel() // m.weight_orig.shape[0]
        call('nele_spectral_norm', pp, dd, len(mods), ptr(w['sigma']), n_iter, stream())
        use16 = self.precision == 'bf16'
        if use16:
            # every layer's weight layouts in two launches (float32 GEMM / data-gradient layouts, then the bf16 fragment streams)
            # the ctypes job tables of this shape live on its buffer set, so the LRU eviction of the set drops them too
            jk = (id(w), self.layers[0].weight_orig.data_ptr(), self.layers[-1].weight_orig.data_ptr())
            if bf.prepjobs is None or bf.prepjobs[0] != jk:
                cin, cpad = self._cin, 4
                pj, dj, fj, ej = [], [], [], []
                cj, gj = [], []
                for l, (cout, k) in enumerate(_D_CONVS):
                    m = self.layers[l]
                    pj += [m.weight_orig.data_ptr(), w['sigma'][l:l + 1].data_ptr(), w['wf'][l].data_ptr(), w['wb'][l].data_ptr()]
                    dj += [cout, cin, cpad, k, k]
                    if bf.c16:
                        if l >= 1:
                            cj += [w['wf'][l].data_ptr(), w['wf16c'][l].data_ptr(), w['wb'][l].data_ptr(), w['wb16c'][l].data_ptr()]
                            gj += [cout, k * k * cpad, k * cpad, k, cpad, k * k * cout, k * cout, k]
                        cin = cpad = cout
                        continue
                    if bf.span16_f[l]:
                        fj += [w['wf'][l].data_ptr(), w['wff16'][l].data_ptr()]
                        ej += [cout, k * k * cpad, k * cpad, k]
                    if bf.span16_b[l]:
                        fj += [w['wb'][l].data_ptr(), w['wbf16'][l].data_ptr()]
                        ej += [cpad, k * k * cout, k * cout, k]
                    cin = cpad = cout
                bf.prepjobs = (jk, ((c_void_p * len(pj))(*pj), (ctypes.c_int * len(dj))(*dj), len(dj) // 5,
                                    (c_void_p * max(1, len(fj)))(*fj), (ctypes.c_int * max(1, len(ej)))(*ej), len(ej) // 4,
                                    (c_void_p * max(1, len(cj)))(*cj), (ctypes.c_int * max(1, len(gj)))(*gj), len(gj) // 4))
            pja, dja, npj, fja, eja, nfj, cja, gja, ncj = bf.prepjobs[1]
            call('nele_weight_prep_batch', pja, dja, npj, stream())
            if nfj:
                call('nele_weight_prep_frag16_batch', fja, eja, nfj, stream())
            if ncj:
                call('nele_conv16_weight_prep_batch', cja, gja, ncj, stream())
                return
            # layers the bf16 kernels decline still need the float32 fragment layout
            cin, cpad = self._cin, 4
            for l, (cout, k) in enumerate(_D_CONVS):
                if not bf.span16_f[l] and bf.span_f[l]:
                    ops.weight_prep_frag(w['wf'][l], cout, k * k * cpad, w['wff'][l])
                if not bf.span16_b[l] and bf.span_b[l]:
                    ops.weight_prep_frag(w['wb'][l], cpad, k * k * cout, w['wbf'][l])
                cin = cpad = cout
        else:
            cin, cpad = self._cin, 4
            for l, (cout, k) in enumerate(_D_CONVS):
                m = self.layers[l]
                ops.weight_prep(m.weight_orig, w['sigma'][l:l + 1], cout, cin, cpad, k, k, w['wf'][l], w['wb'][l])
                if bf.span_f[l]:
                    ops.weight_prep_frag(w['wf'][l], cout, k * k * cpad, w['wff'][l])
                if bf.span_b[l]:
                    ops.weight_prep_frag(w['wb'][l], cpad, k * k * cout, w['wbf'][l])
                cin = cpad = cout

    def _forward_impl(self, din, frames=None):
        if not din.is_cuda:
            raise RuntimeError("nele_gan_amd: the discriminator runs on the GPU only (no CPU fallback)")
        dev = din.device
        self._flat.ensure(dev)
        B, H, T, C4 = din.shape
        assert H == 64 and C4 == 4
        key, bf = self._get_bufs(B, T, dev)
        w = self._weights(dev)
        prep, self._prepared = self._prepared, None
        if prep is not None and prep[1] == self.training:
            torch.cuda.current_stream().wait_event(prep[2])
            if prep[0] != key:
                # prepared for another shape: u, v have already advanced once for this forward pass (the reference advances them once per
                # training forward, model.py:105-116) - only the shape-dependent weight layouts are redone
                self._prepare_inline(key, bf, w, power_iter=False)
        else:
            self._prepare_inline(key, bf, w)
        bf.gen += 1
        a = din.contiguous()
        bf.din = a
        score = _empty((B, self._nout), dev)
        # padded batch of utterances of different lengths: the pooling runs over each utterance's own valid output columns (frames - 20)
        bf.wvalid = None if frames is None else (frames.to(device=dev, dtype=torch.int32) - 20).contiguous()
        # the conv stack + head as ONE call (nele_disc_fwd on the job table recorded the first time this shape / mode came by), or call by call
        pkey = ('fwd', self.precision, self.profile_prefix, bf.wvalid is None, self._flat.flat.data_ptr(), w['sigma'].data_ptr(), self._wgen)
        plan = bf.plans.get(pkey) if ops.plans_enabled() else None
        if plan is not None:
            plan.streams[0] = stream()
            call('nele_disc_fwd', plan.handle, ptr(a), ptr(bf.wvalid), ptr(score), plan.streams, len(plan.streams))
        elif ops.plans_enabled():
            with _lib.recording([ops.rng(a), None if bf.wvalid is None else ops.rng(bf.wvalid), ops.rng(score)]) as rec:
                self._forward_live(a, bf, w, score)
            bf.plans[pkey] = rec.finish()
        else:
            self._forward_live(a, bf, w, score)
        self._last_score = score
        return key

    def _forward_live(self, a, bf, w, score):
        B = bf.B
        for l, (cout, k) in enumerate(_D_CONVS):
            if bf.c16:
                # bf16 activations in memory: conv1 as a float32 pointwise stream, conv2..conv5 on the ring / DMA tile kernel
                if l == 0:
                    ops.conv16_pointwise_fwd(a, w['wf'][0], self.layers[0].bias, bf.act[0])
                elif l == len(_D_CONVS) - 1:
                    ops.conv16_gap(a, w['wf16c'][l], self.layers[l].bias, bf.act[l], B, cout, bf.gf[l], bf.wvalid, bf.gap_part,
                                   tag=self.profile_prefix + 'D.conv%d.fwd' % (l + 1))
                else:
                    ops.conv16(a, w['wf16c'][l], self.layers[l].bias, None, bf.act[l], B, cout, EPI_BIAS_LRELU, bf.gf[l],
                               tag=self.profile_prefix + 'D.conv%d.fwd' % (l + 1))
            elif self.precision == 'bf16' and bf.span16_f[l]:
                ops.conv_span_bf16(a, w['wff16'][l], self.layers[l].bias, None, bf.act[l], B, cout, EPI_BIAS_LRELU, bf.gf[l], tag=self.profile_prefix + 'D.conv%d.fwd' % (l + 1))
            elif bf.span_f[l]:
                ops.conv_span(a, w['wff'][l], self.layers[l].bias, None, bf.act[l], B, cout, EPI_BIAS_LRELU, bf.gf[l], tag=self.profile_prefix + 'D.conv%d.fwd' % (l + 1))
            else:
                ops.conv_gemm(a, w['wf'][l], self.layers[l].bias, None, bf.act[l], B, cout, EPI_BIAS_LRELU, bf.gf[l], tag=self.profile_prefix + 'D.conv%d.fwd' % (l + 1))
            a = bf.act[l]
        if bf.c16:
            call('nele_gap_mlp_fwd_parts', ptr(bf.gap_part), bf.gap_parts, B, bf.P, bf.dims[-1][1], ptr(bf.wvalid), self._mlp_ptrs(w), self._nout, SLOPE,
                 ptr(bf.pooled), ptr(bf.h1), ptr(bf.h2), ptr(score), stream())
        else:
            call('nele_gap_mlp_fwd_var', ptr(a), B, bf.P, bf.dims[-1][1], ptr(bf.wvalid), self._mlp_ptrs(w), self._nout, SLOPE, ptr(bf.pooled), ptr(bf.h1),
                 ptr(bf.h2), ptr(score), ptr(bf.scratch64), stream())

    def forward_packed(self, din, frames=None):
        """din: channels-last [B,64,T,4] (ops.d_pack / energy-norm output).  frames [B] (optional): STFT frames of each utterance
        inside a padded batch (every utterance needs >= 21, like T)."""
        if torch.is_grad_enabled() and (din.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _DFn.apply(din, self._anchor.get(din.device), self, frames)
        self._forward_impl(din, frames)
        score, self._last_score = self._last_score, None
        return score

    def forward(self, x):
        """x: [B, Cin, 64, T] as in the reference (model.py:118)."""
        if x.dim() != 4 or x.shape[1] != self._cin or x.shape[2] != 64:
            raise ValueError("%s.forward: x must be [B, %d, 64, T]" % (type(self).__name__, self._cin))
        return self.forward_packed(_NchwToPacked.apply(x))

    def _backward_impl(self, dscore, key, need_din, score, wvalid=None):
        bf = self._bufs[key]
        w = self._w
        wgrad = self.weight_grad_enabled
        wsts = None
        if wgrad and self.overlap_wgrad:
            if self._wstream is None:
                self._wstream = (ops.side_stream(dscore.device), ops.side_stream(dscore.device))
            wsts = self._wstream
        g16 = self.precision == 'bf16' and (bf.grad16_ok or bf.c16)
        if g16 and not bf.c16 and bf.gbuf16 is None:
            bf.gbuf16 = torch.zeros(bf.gbuf[-1].shape, dtype=torch.bfloat16, device=bf.gbuf[-1].device)
        pkey = ('bwd', self.precision, bool(need_din), bool(wgrad), None if wsts is None else tuple(q.cuda_stream for q in wsts), wvalid is None,
                self._flat.flat.data_ptr(), self._flat.grad.data_ptr(), w['sigma'].data_ptr(), self._wgen)
        plan = bf.plans.get(pkey) if ops.plans_enabled() else None
        if plan is not None:
            plan.streams[0] = stream()
            call('nele_disc_bwd', plan.handle, ptr(dscore), ptr(score), ptr(wvalid), ptr(bf.din), plan.streams, len(plan.streams))
        elif ops.plans_enabled():
            with _lib.recording([ops.rng(dscore), ops.rng(score), None if wvalid is None else ops.rng(wvalid), ops.rng(bf.din)]) as rec:
                self._backward_live(dscore, bf, w, need_din, score, wvalid, wgrad, wsts, g16)
            bf.plans[pkey] = rec.finish()
        else:
            self._backward_live(dscore, bf, w, need_din, score, wvalid, wgrad, wsts, g16)
        return bf.ddin if need_din else None

    def _backward_live(self, dscore, bf, w, need_din, score, wvalid, wgrad, wsts, g16):
        B = bf.B
        nout = self._nout
        Ho, Wo, _ = bf.dims[-1]
        p5 = bf.pad[-1]
        glast = bf.gbuf16 if (g16 and not bf.c16) else bf.gbuf[-1]      # the last layer's output gradient, as its two consumers read it
        call(('nele_gap_mlp_bwd_var16a' if bf.c16 else 'nele_gap_mlp_bwd_var16') if g16 else 'nele_gap_mlp_bwd_var', ptr(dscore), ptr(score), ptr(bf.h1), ptr(bf.h2), ptr(bf.act[-1]),
             self._mlp_ptrs(w), nout, SLOPE, B, Ho, Wo, ptr(wvalid), Ho + 2 * p5, Wo + 2 * p5, p5, p5, ptr(bf.dz3), ptr(bf.dz2), ptr(bf.dz1),
             ptr(bf.dpooled), ptr(glast), stream())
        # The data-gradient chain (layer l's needs layer l+1's) stays on the current stream; a layer's weight gradient (+ its
        # spectral-norm chain rule and bias gradient) only needs that layer's output gradient, so those run on two more streams
        # beside the chain and join at the end.  In the D-step this tail is fully exposed (nothing else is left to overlap).
        main = torch.cuda.current_stream()
        if wsts is not None and (wsts[0] == main or wsts[1] == main):
            wsts = None
        ev = bf.events
        ev.start()
        if wsts is not None:
            for q in wsts:
                ops.hand_over(ev, main, q)                # dz1..dz3 and gbuf[-1] exist
        if wgrad:
            # the three FC layers' weight gradients: a dozen small launches, on the second weight-gradient stream (its temporaries)
            ctx = None
            if wsts is not None:
                ctx = torch.cuda.stream(wsts[1])
                ctx.__enter__()
            tw, sc = (bf.tmpw2, bf.scratch64b) if wsts is not None else (bf.tmpw, bf.scratch64)
            for i, (m, dz, xin, N, K) in enumerate(((self.fc3, bf.dz3, bf.h2, nout, 16), (self.fc2, bf.dz2, bf.h1, 16, 64),
                                                    (self.fc1, bf.dz1, bf.pooled, 64, 64))):
                li = 7 - i
                tmpb = tw[N * K:N * K + N]
                call('nele_mlp_wgrad', ptr(dz), ptr(xin), B, N, K, ptr(tw), c_void_p(tmpb.data_ptr()), stream())
                call('nele_sn_grad', ptr(tw), ptr(m.weight_orig), ptr(m.weight_u), ptr(m.weight_v),
                     c_void_p(w['sigma'].data_ptr() + 4 * li), N, K, ptr(m.weight_orig.grad), 1, ptr(sc), stream())
                ops.vec_add_(m.bias.grad, tmpb)
            if ctx is not None:
                ctx.__exit__(None, None, None)
        for l in range(len(_D_CONVS) - 1, -1, -1):
            cout, k = _D_CONVS[l]
            m = self.layers[l]
            Hi, Wi, Ci = bf.dims[l]
            cin_valid = self._cin if l == 0 else Ci
            a_in = bf.din if l == 0 else bf.act[l - 1]
            if wgrad:
                N, K = cout, cin_valid * k * k
                # two streams alternate over the layers, each with its own temporaries
                # (conv5, conv3) on one, (conv4, conv2, conv1) on the other: about equal isolated time, both shorter than the data-gradient chain
                q = 1 if l == 0 else (l & 1)
                tw, wsb, sc = (bf.tmpw, bf.ws, bf.scratch64) if q == 0 else (bf.tmpw2, bf.ws2, bf.scratch64b)
                tmpb = tw[N * K:N * K + N]
                wst = wsts[q] if wsts is not None else None
                ctx = None
                if wst is not None:
                    ops.hand_over(ev, main, wst)          # gbuf[l] is complete at this point of the current stream
                    ctx = torch.cuda.stream(wst)
                    ctx.__enter__()
                ops.conv_wgrad(a_in, glast if l == len(_D_CONVS) - 1 else bf.gbuf[l], wsb, B, cout, bf.gw[l], cin_valid, tw, tmpb, accumulate=False,
                               bf16=(self.precision == 'bf16' and l > 0),
                               tag='D.conv%d.wgrad' % (l + 1))
                call('nele_sn_grad', ptr(tw), ptr(m.weight_orig), ptr(m.weight_u), ptr(m.weight_v),
                     c_void_p(w['sigma'].data_ptr() + 4 * l), N, K, ptr(m.weight_orig.grad), 1, ptr(sc), stream())
                ops.vec_add_(m.bias.grad, tmpb)
                if ctx is not None:
                    ctx.__exit__(None, None, None)
            if l > 0 and bf.c16:
                ops.conv16(bf.gbuf[l], w['wb16c'][l], None, bf.act[l - 1], bf.gbuf[l - 1], B, Ci, EPI_MASK_LRELU_GRAD, bf.gb[l], tag='D.conv%d.dgrad' % (l + 1))
            elif l > 0:
                if self.precision == 'bf16' and bf.span16_b[l]:
                    ops.conv_span_bf16(glast if l == len(_D_CONVS) - 1 else bf.gbuf[l], w['wbf16'][l], None, bf.act[l - 1], bf.gbuf[l - 1], B, Ci,
                                       EPI_MASK_LRELU_GRAD, bf.gb[l], tag='D.conv%d.dgrad' % (l + 1))
                elif bf.span_b[l]:
                    ops.conv_span(bf.gbuf[l], w['wbf'][l], None, bf.act[l - 1], bf.gbuf[l - 1], B, Ci, EPI_MASK_LRELU_GRAD, bf.gb[l], tag='D.conv%d.dgrad' % (l + 1))
                else:
                    ops.conv_gemm(bf.gbuf[l], w['wb'][l], None, bf.act[l - 1], bf.gbuf[l - 1], B, Ci, EPI_MASK_LRELU_GRAD, bf.gb[l], tag='D.conv%d.dgrad' % (l + 1))
            elif need_din:
                ops.conv_gemm(bf.gbuf[0], w['wb'][0], None, None, bf.ddin, B, 4, EPI_NONE, bf.gb[0])
        if wsts is not None:
            for q in wsts:
                ops.hand_over(ev, q, main)


class Discriminator(_DiscriminatorBase):
    """model.py:101-132: input channels (enhanced, noise, clean); outputs (SIIB, HASPI, ESTOI) scores.
    ``nout`` is configurable (build-side feature) for metric subsets, e.g. 2 for SIIB+ESTOI."""

    def __init__(self, nout=3):
        super().__init__(3, nout)


class Discriminator_Quality(_DiscriminatorBase):
    """model.py:135-166: input channels (enhanced, clean); outputs (PESQ, ViSQOL) scores."""

    def __init__(self, nout=2):
        super().__init__(2, nout)


# ================================================================== energy normalisation glue (train_nele.py:133-146)
class _EnergyNormPack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, clean, noise, p, inv_p):
        beta2, s2, din, _ = ops.energy_norm_fwd(clean, mask, noise, p, inv_p, want_din=True)
        ctx.save_for_backward(mask, clean, beta2, s2, din)
        ctx.p, ctx.inv_p = p, inv_p
        ctx.mark_non_differentiable(beta2)
        return din, beta2

    @staticmethod
    def backward(ctx, ddin, dbeta2):
        mask, clean, beta2, s2, din = ctx.saved_tensors
        dmask = ops.energy_norm_bwd(clean, mask, beta2, s2, ddin.contiguous(), ctx.p, ctx.inv_p, din=din)
        return dmask, None, None, None, None


def energy_norm_pack(mask, clean_band, noise_band, p_power=1.0 / 6, inv_p=6.0):
    """mask, clean_band, noise_band [B,T,64] -> (D input [B,64,T,4] = (enh, noise, clean, 0), beta2 [B]).
    Utterance-level (per sample) normalisation, as the batch-1 reference (train_nele.py:133-146)."""
    return _EnergyNormPack.apply(mask.contiguous(), clean_band.contiguous(), noise_band.contiguous(), float(p_power), float(inv_p))


def normed_alpha2(mask, clean_band, inv_p=6.0, frames=None):
    """mask * beta_2 (train_nele.py:303-307; inference.py:99-104), no autograd.  frames [B]: STFT frames of each utterance inside a
    padded batch (the energy sums of beta_2 run over those frames only)."""
    _, _, _, alpha2 = ops.energy_norm_fwd(clean_band.contiguous(), mask.contiguous(), None, 1.0 / inv_p, float(inv_p), want_din=False,
                                          want_alpha2=True, frames=frames)
    return alpha2
