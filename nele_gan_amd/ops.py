"""Thin Python wrappers over the dense / glue entry points of libnele_hip.so (no arithmetic here)."""
import ctypes
import os

import torch

from . import _lib
from ._lib import c_float, c_int, c_longlong, c_void_p, call, declare, ptr, stream

EPI_NONE, EPI_BIAS, EPI_BIAS_LRELU, EPI_MASK_LRELU_GRAD, EPI_BIAS_EXPTANH = 0, 1, 2, 3, 4
SLOPE = 0.3  # nn.LeakyReLU(0.3), model.py:79,112

_P = c_void_p
_SIGS = {
    'nele_conv_gemm': [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, ctypes.POINTER(c_int), _P],
    'nele_conv_gemm_bf16': [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, ctypes.POINTER(c_int), _P],
    'nele_conv_wgrad': [_P, _P, _P, c_longlong, c_int, c_int, ctypes.POINTER(c_int), c_int, c_int, c_int, _P, _P, c_int, _P],
    'nele_conv_wgrad_bf16': [_P, _P, _P, c_longlong, c_int, c_int, ctypes.POINTER(c_int), c_int, c_int, c_int, _P, _P, c_int, _P],
    'nele_weight_prep': [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P],
    'nele_weight_prep_frag': [_P, c_int, c_int, _P, _P],
    'nele_conv_span_supported': [c_int, c_int, ctypes.POINTER(c_int), c_int, c_int],
    'nele_conv_span': [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, ctypes.POINTER(c_int), c_int, c_int, c_longlong, _P],
    'nele_conv_span_bf16': [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, ctypes.POINTER(c_int), c_int, c_int, c_longlong, _P],
    'nele_conv_span_bf16_supported': [c_int, c_int, ctypes.POINTER(c_int), c_int, c_int],
    'nele_conv_span_bf16_a16_supported': [c_int, c_int, ctypes.POINTER(c_int), c_int, c_int],
    'nele_conv_span_bf16_a16': [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, ctypes.POINTER(c_int), c_int, c_int, c_longlong, _P],
    'nele_conv_wgrad_bf16_d16_supported': [c_int, c_int, ctypes.POINTER(c_int), c_int, c_int],
    'nele_conv_wgrad_bf16_d16': [_P, _P, _P, c_longlong, c_int, c_int, ctypes.POINTER(c_int), c_int, c_int, c_int, _P, _P, c_int, _P],
    'nele_weight_prep_frag16': [_P, c_int, c_int, c_int, c_int, _P, _P],
    'nele_conv16_supported': [c_int, c_int, ctypes.POINTER(c_int), c_int, c_int],
    'nele_conv16_weight_prep_batch': [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int, _P],
    'nele_conv16': [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, ctypes.POINTER(c_int), c_int, c_int, _P],
    'nele_conv16_pointwise_fwd': [_P, _P, _P, _P, c_longlong, c_int, c_float, _P],
    'nele_conv16_gap': [_P, _P, _P, _P, c_int, c_int, c_float, ctypes.POINTER(c_int), c_int, c_int, _P, _P, _P],
    'nele_conv16_gap_parts': [c_int, ctypes.POINTER(c_int), c_int, c_int],
    'nele_conv_wgrad_bf16_a16d16': [_P, _P, _P, c_longlong, c_int, c_int, ctypes.POINTER(c_int), c_int, c_int, c_int, _P, _P, c_int, _P],
    'nele_g_pack': [_P, _P, _P, c_int, c_int, c_int, _P],
    'nele_g_pack16': [_P, _P, _P, c_int, c_int, c_int, _P],
    'nele_glayer16_supported': [c_int, c_int, c_int],
    'nele_glayer16_weight_prep_batch': [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int, _P],
    'nele_glayer16_fwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_uint, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P],
    'nele_glayer16_conv': [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    'nele_cln_fwd': [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P],
    'nele_cln_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P],
    'nele_cln_chunks': [c_int],
    'nele_colsum': [_P, c_int, c_int, _P, c_int, _P],
    'nele_weight_prep_batch': [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int, _P],
    'nele_weight_prep_frag16_batch': [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int, _P],
    'nele_colsum2': [_P, _P, _P, _P, c_int, c_int, c_int, _P],
    'nele_exptanh_bwd': [_P, _P, _P, c_longlong, _P],
    'nele_energy_norm_fwd': [_P, _P, _P, c_float, c_float, _P, _P, _P, _P, c_int, c_int, _P],
    'nele_energy_norm_bwd': [_P, _P, _P, _P, _P, _P, c_float, c_float, _P, c_int, c_int, _P],
    'nele_d_pack': [_P, _P, _P, _P, c_int, c_int, _P],
    'nele_d_layout': [_P, _P, c_int, c_int, c_int, c_int, _P],
    'nele_d_gather_items': [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), ctypes.POINTER(c_longlong), c_int, c_int, c_int, _P, _P, _P],
    'nele_spectral_norm': [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int, _P, c_int, _P],
    'nele_sn_grad': [_P, _P, _P, _P, _P, c_int, c_int, _P, c_int, _P, _P],
    'nele_sn_grad_scratch_doubles': [c_int],
    'nele_gap_mlp_fwd': [_P, c_int, c_int, ctypes.POINTER(c_void_p), c_int, c_float, _P, _P, _P, _P, _P, _P],
    'nele_gap_mlp_bwd': [_P, _P, _P, _P, _P, ctypes.POINTER(c_void_p), c_int, c_float, c_int, c_int, c_int, c_int, c_int, c_int,
                         c_int, _P, _P, _P, _P, _P, _P],
    'nele_gap_mlp_fwd_var': [_P, c_int, c_int, c_int, _P, ctypes.POINTER(c_void_p), c_int, c_float, _P, _P, _P, _P, _P, _P],
    'nele_gap_mlp_bwd_var': [_P, _P, _P, _P, _P, ctypes.POINTER(c_void_p), c_int, c_float, c_int, c_int, c_int, _P, c_int, c_int, c_int,
                             c_int, _P, _P, _P, _P, _P, _P],
    'nele_gap_mlp_bwd_var16': [_P, _P, _P, _P, _P, ctypes.POINTER(c_void_p), c_int, c_float, c_int, c_int, c_int, _P, c_int, c_int, c_int,
                               c_int, _P, _P, _P, _P, _P, _P],
    'nele_gap_mlp_bwd_var16a': [_P, _P, _P, _P, _P, ctypes.POINTER(c_void_p), c_int, c_float, c_int, c_int, c_int, _P, c_int, c_int, c_int,
                                c_int, _P, _P, _P, _P, _P, _P],
    'nele_gap_mlp_fwd_parts': [_P, c_int, c_int, c_int, c_int, _P, ctypes.POINTER(c_void_p), c_int, c_float, _P, _P, _P, _P, _P],
    'nele_mlp_wgrad': [_P, _P, c_int, c_int, c_int, _P, _P, _P],
    'nele_adam_step': [_P, _P, _P, _P, c_longlong, c_float, c_float, c_float, c_float, c_int, _P],
    'nele_adam_step_guarded': [_P, _P, _P, _P, c_longlong, c_float, c_float, c_float, c_float, c_int, _P, _P],
}
for _n, _a in _SIGS.items():
    declare(_n, _a)
    _lib._SIGS[_n] = _a
_lib.lib.nele_weight_frag16_elems.argtypes = [c_int, c_int, c_int]
_lib.lib.nele_weight_frag16_elems.restype = c_longlong
_lib._SIGS['nele_weight_frag16_elems'] = _lib.lib.nele_weight_frag16_elems.argtypes
_lib.lib.nele_conv16_wfrag_elems.argtypes = [c_int, c_int, c_int]
_lib.lib.nele_conv16_wfrag_elems.restype = c_longlong
_lib._SIGS['nele_conv16_wfrag_elems'] = _lib.lib.nele_conv16_wfrag_elems.argtypes
for _n in ('nele_glayer16_wfrag_elems', 'nele_glayer16_carry_bytes'):
    getattr(_lib.lib, _n).argtypes = [c_int, c_int, c_int] if _n.endswith('elems') else [c_int, c_int]
    getattr(_lib.lib, _n).restype = c_longlong
    _lib._SIGS[_n] = getattr(_lib.lib, _n).argtypes
_lib.lib.nele_conv_wgrad_workspace_floats.argtypes = [c_int, c_int, c_int, ctypes.POINTER(c_int)]
_lib.lib.nele_conv_wgrad_workspace_floats.restype = c_longlong
_lib._SIGS['nele_conv_wgrad_workspace_floats'] = _lib.lib.nele_conv_wgrad_workspace_floats.argtypes


class Geom:
    """ConvGeom of csrc/dense.hip: how output position (b,ho,wo) maps into the channels-last input
    buffer [B][H][W][C] and the output buffer [B][OH][OW][OC]."""

    def __init__(self, H, W, C, Hout, Wout, KH, KW, OH, OW, OC, ih0=0, iw0=0, oh0=0, ow0=0):
        self.KH, self.KW = KH, KW
        self.Hout, self.Wout = Hout, Wout
        self.Ktot = KH * KW * C
        vals = [H, W, C, ih0, iw0, Hout, Wout, KW * C, W * C, self.Ktot, OH, OW, OC, oh0, ow0]
        self.arr = (c_int * 15)(*vals)


# Module-level test knobs (tests monkeypatch them; nothing reads the environment): False = the discriminator keeps float32 activations /
# a float32 pooling gradient in memory - the round-2 kernels, which remain the path for geometries the bf16-in-memory kernels decline.
CONV16 = True
GRAD16 = True


def side_stream(device):
    """A new side stream - or, with NELE_SERIAL=1 (diagnostic: every kernel then runs alone and a rocprofv3 kernel trace shows
    isolated durations), the current stream itself, which serialises the whole step.
    (CU-masked streams - hipExtStreamCreateWithCUMask, to keep the metric streams off part of the GPU - were tried in rounds 1 and 2: a
    masked queue's kernels stop overlapping with the other queues altogether: 96-128 ms per B = 256 step against 78.)"""
    if os.environ.get('NELE_SERIAL', '0') == '1':
        return torch.cuda.current_stream(device)
    return torch.cuda.Stream(device=device)


_probe_warm = False


def shares_queue(a, b, device, spin_us=300.0):
    """Does a kernel on stream ``b`` wait behind a kernel running on stream ``a`` - do the two sit on the same hardware queue?  (The HIP
    runtime multiplexes all streams onto four hardware queues - the default stream's and three for the side streams, assigned at
    creation; streams on one queue run in submission order, whatever their events say.)  Parks one idle wave on ``a`` for ``spin_us`` and
    times a trivial kernel on ``b``: on another queue it finishes long before the spin does.  Asked in BOTH directions (the first probe
    of a process misreads one of them - first kernel launches - and streams that do share a queue wait for each other either way).
    Synchronises the device a few times: for set-up code, once per object."""
    global _probe_warm
    t = torch.zeros(64, device=device)

    def probe(p, q):
        torch.cuda.synchronize(device)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(p):
            _lib.lib.nele_stream_spin(float(spin_us), c_void_p(p.cuda_stream))
            ea.record(p)
        with torch.cuda.stream(q):
            t.add_(1.0)
            eb.record(q)
        torch.cuda.synchronize(device)
        return eb.elapsed_time(ea) < 0.5 * spin_us * 1e-3       # q's kernel ended less than half a spin before the spin did (or after it)
    if not _probe_warm:
        _probe_warm = True
        probe(a, b)                                             # first launches of both kernels: result discarded
    return probe(a, b) and probe(b, a)


def streams_on_distinct_queues(device, n, have=()):
    """``n`` (at most three) side streams on pairwise different hardware queues, none of them the queue of a stream in ``have``."""
    if os.environ.get('NELE_SERIAL', '0') == '1' or torch.cuda.is_current_stream_capturing():
        return [side_stream(device) for _ in range(n)]
    out = []
    for _ in range(n):
        st, tries = torch.cuda.Stream(device=device), 0
        while tries < 8 and any(shares_queue(o, st, device) for o in list(have) + out):
            st, tries = torch.cuda.Stream(device=device), tries + 1
        out.append(st)
    return out


class Events:
    """Hand-over events between the streams of one pass, created once per buffer set and addressed by position: a recorded plan
    (_lib.PlanRecorder) holds their handles, so the n-th fork of a pass must always use the n-th event."""

    def __init__(self):
        self.ev = []
        self.k = 0

    def start(self):
        self.k = 0

    def next(self):
        if self.k == len(self.ev):
            h = c_void_p()
            _lib.check(_lib.lib.nele_event_create(ctypes.byref(h)), 'nele_event_create')
            self.ev.append(h)
        self.k += 1
        return self.ev[self.k - 1]

    def __del__(self):
        try:
            for h in self.ev:
                _lib.lib.nele_event_destroy(h)
        except Exception:
            pass


def hand_over(events, src, dst):
    """stream ``dst`` waits for everything enqueued on stream ``src`` so far (two plan-recordable library calls)."""
    ev = events.next()
    call('nele_event_record', ev, c_void_p(src.cuda_stream))
    call('nele_stream_wait_event', ev, c_void_p(dst.cuda_stream))


def vec_add_(dst, src):
    """dst += src (float32, same length) as a plan-recordable library call."""
    call('nele_vec_add', c_void_p(dst.data_ptr()), c_void_p(src.data_ptr()), dst.numel(), stream())


def plans_enabled():
    """Passes are replayed from recorded job tables unless switched off (_lib.PLANS) or a torch-event profile of single launches is armed."""
    return _lib.PLANS and PROFILE is None


def rng(t):
    """(address, bytes) of a tensor for PlanRecorder's dynamic slots"""
    return (t.data_ptr(), t.numel() * t.element_size())


# bench.py sets PROFILE = {tag: [(start_event, end_event), ...]} to time one tagged kernel with HIP events
# recorded on the stream the kernel is launched on (torch's current stream).
PROFILE = None


def conv_gemm(A, Wg, bias, aux, out, B, N, epi, g, tag=None, bf16=False):
    M = B * g.Hout * g.Wout
    prof = PROFILE is not None and tag in PROFILE
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call('nele_conv_gemm_bf16' if (bf16 and N > 48) else 'nele_conv_gemm', ptr(A), ptr(Wg), ptr(bias), ptr(aux), ptr(out), M, N, epi, SLOPE, g.arr, stream())
    if prof:
        e1.record()
        PROFILE[tag].append((e0, e1, 2.0 * M * N * g.Ktot))


def span_supported(B, N, g):
    return bool(_lib.lib.nele_conv_span_supported(B * g.Hout * g.Wout, N, g.arr, g.KH, g.KW))


def frag_floats(N, Ktot):
    return (Ktot // 8) * ((N + 15) // 16) * 128


def weight_prep_frag(Wg, N, Ktot, Wfrag):
    call('nele_weight_prep_frag', ptr(Wg), N, Ktot, ptr(Wfrag), stream())


def conv_span(A, Wfrag, bias, aux, out, B, N, epi, g, tag=None):
    M = B * g.Hout * g.Wout
    prof = PROFILE is not None and tag in PROFILE
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call('nele_conv_span', ptr(A), ptr(Wfrag), ptr(bias), ptr(aux), ptr(out), M, N, epi, SLOPE, g.arr, g.KH, g.KW, A.numel(), stream())
    if prof:
        e1.record()
        PROFILE[tag].append((e0, e1, 2.0 * M * N * g.Ktot))


def span16_supported(B, N, g):
    return bool(_lib.lib.nele_conv_span_bf16_supported(B * g.Hout * g.Wout, N, g.arr, g.KH, g.KW))


def grad16_supported(B, N_dgrad, g_dgrad, N_wgrad, g_wgrad):
    """Can this layer's output gradient live in memory as bfloat16?  (data gradient on the span kernel, weight gradient on the tile kernel)"""
    if not GRAD16:
        return False
    return bool(_lib.lib.nele_conv_span_bf16_a16_supported(B * g_dgrad.Hout * g_dgrad.Wout, N_dgrad, g_dgrad.arr, g_dgrad.KH, g_dgrad.KW)) and \
        bool(_lib.lib.nele_conv_wgrad_bf16_d16_supported(B * g_wgrad.Hout * g_wgrad.Wout, N_wgrad, g_wgrad.arr, g_wgrad.KH, g_wgrad.KW))


def frag16_elems(N, seglen, KH):
    return int(_lib.lib.nele_weight_frag16_elems(N, seglen, KH))


def weight_prep_frag16(Wg, N, Ktot, seglen, KH, Wfrag):
    call('nele_weight_prep_frag16', ptr(Wg), N, Ktot, seglen, KH, ptr(Wfrag), stream())


def conv_span_bf16(A, Wfrag, bias, aux, out, B, N, epi, g, tag=None):
    """A float32, or bfloat16 (nele_conv_span_bf16_a16: the span kernel only)."""
    M = B * g.Hout * g.Wout
    prof = PROFILE is not None and tag in PROFILE
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call('nele_conv_span_bf16_a16' if A.dtype == torch.bfloat16 else 'nele_conv_span_bf16', ptr(A), ptr(Wfrag), ptr(bias), ptr(aux), ptr(out), M, N, epi, SLOPE,
         g.arr, g.KH, g.KW, A.numel(), stream())
    if prof:
        e1.record()
        PROFILE[tag].append((e0, e1, 2.0 * M * N * g.Ktot))


def conv16_supported(B, N, g):
    """Can this layer run on the bf16-activation tile kernel (csrc/conv16.hip)?"""
    if not CONV16:
        return False
    return bool(_lib.lib.nele_conv16_supported(B * g.Hout * g.Wout, N, g.arr, g.KH, g.KW))


def wgrad_tile_supported(B, N, g):
    """Does this layer's weight gradient run on the 2-D tile kernel (the one that takes bf16 operands from memory)?"""
    return bool(_lib.lib.nele_conv_wgrad_bf16_d16_supported(B * g.Hout * g.Wout, N, g.arr, g.KH, g.KW))


def conv16_pointwise_fwd(din, Wf, bias, out16):
    """D's first layer (1 x 1, 4 -> 8 channels) on the packed float32 input, bf16 activation out."""
    M = din.numel() // 4
    call('nele_conv16_pointwise_fwd', ptr(din), ptr(Wf), ptr(bias), ptr(out16), M, 8, SLOPE, stream())


def conv16_wfrag_elems(N, seglen, KH):
    return int(_lib.lib.nele_conv16_wfrag_elems(N, seglen, KH))


def conv16(A16, Wfrag, bias, aux16, out, B, N, epi, g, tag=None):
    """Conv2d forward / data gradient on bfloat16 activations: A16 [B][H][W][C] bf16, out bf16 or float32 (by its dtype),
    aux16 = forward activation (bf16) for the LeakyReLU mask of a data gradient."""
    M = B * g.Hout * g.Wout
    if A16.dtype != torch.bfloat16 or (aux16 is not None and aux16.dtype != torch.bfloat16):
        raise ValueError('conv16: activations must be bfloat16')
    prof = PROFILE is not None and tag in PROFILE
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call('nele_conv16', ptr(A16), ptr(Wfrag), ptr(bias), ptr(aux16), ptr(out), int(out.dtype == torch.bfloat16), M, N, epi, SLOPE, g.arr, g.KH, g.KW,
         stream())
    if prof:
        e1.record()
        PROFILE[tag].append((e0, e1, 2.0 * M * N * g.Ktot))


def conv16_gap_parts(N, g):
    """Pooled partial sums per image that conv16_gap writes for this geometry (0: unsupported)"""
    return int(_lib.lib.nele_conv16_gap_parts(N, g.arr, g.KH, g.KW))


def conv16_gap(A16, Wfrag, bias, out16, B, N, g, wvalid, gap_part, tag=None):
    """The last conv layer with the global average pooling fused (model.py:109,121-123): out16 = bf16(LeakyReLU(conv + bias)),
    gap_part [B][parts][N] float64 pooled partial sums over each image's valid columns."""
    M = B * g.Hout * g.Wout
    if A16.dtype != torch.bfloat16 or out16.dtype != torch.bfloat16 or gap_part.dtype != torch.float64:
        raise ValueError('conv16_gap: bfloat16 activations and float64 partial sums')
    prof = PROFILE is not None and tag in PROFILE
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call('nele_conv16_gap', ptr(A16), ptr(Wfrag), ptr(bias), ptr(out16), M, N, SLOPE, g.arr, g.KH, g.KW, ptr(wvalid), ptr(gap_part), stream())
    if prof:
        e1.record()
        PROFILE[tag].append((e0, e1, 2.0 * M * N * g.Ktot))


def glayer16_supported(cin, cout, k):
    """Does the fused layer kernel (conv + bias + cLN + LeakyReLU on bf16 activations, csrc/glayer.hip) take this Conv1d?"""
    return bool(_lib.lib.nele_glayer16_supported(cin, cout, k))


def glayer16_wfrag_elems(cin, cout, k):
    return int(_lib.lib.nele_glayer16_wfrag_elems(cin, cout, k))


def wgrad_workspace_floats(B, N, g):
    M = B * g.Hout * g.Wout
    return int(_lib.lib.nele_conv_wgrad_workspace_floats(M, N, g.Ktot, None))


def conv_wgrad(A, dOut, ws, B, N, g, Cvalid, dW, db, accumulate=True, bf16=False, tag=None):
    M = B * g.Hout * g.Wout
    prof = PROFILE is not None and tag in PROFILE
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    fn = 'nele_conv_wgrad_bf16' if bf16 else 'nele_conv_wgrad'
    if dOut.dtype == torch.bfloat16:
        if not bf16:
            raise ValueError('conv_wgrad: a bfloat16 output gradient needs bf16=True')
        fn = 'nele_conv_wgrad_bf16_a16d16' if A.dtype == torch.bfloat16 else 'nele_conv_wgrad_bf16_d16'
    elif A.dtype == torch.bfloat16:
        raise ValueError('conv_wgrad: a bfloat16 activation needs a bfloat16 output gradient')
    call(fn, ptr(A), ptr(dOut), ptr(ws), ws.numel(), M, N, g.arr, g.KH, g.KW, Cvalid, ptr(dW), ptr(db), int(accumulate), stream())
    if prof:
        e1.record()
        # algorithmic bytes: the input and the output gradient read once (float32), the weight gradient written once
        PROFILE[tag].append((e0, e1, 4.0 * (B * g.arr[0] * g.arr[1] * g.arr[2] + M * N + N * g.Ktot)))


def weight_prep(Wt, sigma, N, Cvalid, C, KH, KW, Wf, Wb):
    call('nele_weight_prep', ptr(Wt), ptr(sigma), N, Cvalid, C, KH, KW, ptr(Wf), ptr(Wb), stream())


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, guard=None):
    if guard is not None:
        call('nele_adam_step_guarded', ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step, ptr(guard), stream())
    else:
        call('nele_adam_step', ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step, stream())


def energy_norm_fwd(clean, mask, noise, p, inv_p, want_din=True, want_alpha2=False, frames=None):
    """frames [B] is accepted for symmetry with the other padded-batch entry points and needs no kernel support: the band features of
    frames behind an utterance's end are exactly zero (nele_stft_band_var / nele_imcra_band_var), so they add nothing to beta_2's sums,
    their D input is zero and their mask gradient is zero."""
    B, T, _ = clean.shape
    dev = clean.device
    beta2 = torch.empty(B, device=dev)
    s2 = torch.empty(B, device=dev)
    din = torch.empty((B, 64, T, 4), device=dev) if want_din else None
    alpha2 = torch.empty((B, T, 64), device=dev) if want_alpha2 else None
    call('nele_energy_norm_fwd', ptr(clean), ptr(mask), ptr(noise), p, inv_p, ptr(beta2), ptr(s2), ptr(din), ptr(alpha2), B, T, stream())
    return beta2, s2, din, alpha2


def energy_norm_bwd(clean, mask, beta2, s2, ddin, p, inv_p, din=None):
    B, T, _ = clean.shape
    dmask = torch.empty_like(mask)
    call('nele_energy_norm_bwd', ptr(clean), ptr(mask), ptr(beta2), ptr(s2), ptr(ddin), ptr(din), p, inv_p, ptr(dmask), B, T, stream())
    return dmask


def d_pack(c0, c1, c2=None):
    """three (two) band-feature tensors [B,T,64] -> channels-last D input [B,64,T,4] (dataloader.py:76-84)."""
    B, T, _ = c0.shape
    din = torch.empty((B, 64, T, 4), device=c0.device)
    call('nele_d_pack', ptr(c0.contiguous()), ptr(c1.contiguous()), ptr(c2.contiguous() if c2 is not None else None), ptr(din), B, T,
         stream())
    return din


def d_gather(items, rows, Tm, want_frames=True):
    """A list of per-utterance D items [64, T_k, 4] (float32, device; contiguous or rows of a larger padded batch: band rows
    item.stride(0) floats apart) -> (din [rows, 64, Tm, 4] zero-padded, frames [rows] int32 or None): one launch per 64 items
    (nele_d_gather_items) instead of one copy per item."""
    n = len(items)
    dev = items[0].device
    Ts = [int(t.shape[1]) for t in items]
    for t in items:
        if t.dtype != torch.float32 or t.dim() != 3 or t.shape[0] != 64 or t.shape[2] != 4 or t.stride(2) != 1 or t.stride(1) != 4:
            raise ValueError('d_gather: items must be float32 [64, T, 4] with contiguous (T, 4) rows')
    din = torch.empty((rows, 64, Tm, 4), device=dev)
    frames = torch.empty((rows,), dtype=torch.int32, device=dev) if want_frames else None
    pa = (c_void_p * n)(*[t.data_ptr() for t in items])
    fa = (c_int * n)(*Ts)
    sa = (c_longlong * n)(*[int(t.stride(0)) for t in items])
    call('nele_d_gather_items', pa, fa, sa, n, rows, Tm, ptr(din), ptr(frames), stream())
    return din, frames


def nchw_to_nhwc4(x):
    B, Cin, H, T = x.shape
    assert H == 64
    din = torch.empty((B, 64, T, 4), device=x.device)
    call('nele_d_layout', ptr(x.contiguous()), ptr(din), B, Cin, T, 1, stream())
    return din


def nhwc4_to_nchw(ddin, Cin):
    B, H, T, _ = ddin.shape
    dx = torch.empty((B, Cin, 64, T), device=ddin.device)
    call('nele_d_layout', ptr(ddin), ptr(dx), B, Cin, T, 0, stream())
    return dx
