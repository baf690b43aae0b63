"""ctypes binding of libnele_hip.so (the C ABI of include/nele_hip.h).

The HIP library IS the product path: if it is missing or fails to load this module raises; there
is no CPU or eager-PyTorch fallback anywhere in the package.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NELE_LIB') or os.path.join(_HERE, 'libnele_hip.so')     # NELE_LIB: another build of the same ABI (A/B measurements)

c_int = ctypes.c_int
c_float = ctypes.c_float
c_double = ctypes.c_double
c_void_p = ctypes.c_void_p
c_size_t = ctypes.c_size_t
c_longlong = ctypes.c_longlong


class NeleError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "nele_gan_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C nele_gan_amd/csrc` (hipcc, gfx950). There is no fallback path." % LIB_PATH)
    return ctypes.CDLL(LIB_PATH)


lib = _load()
lib.nele_version.restype = c_int
lib.nele_last_error_string.restype = ctypes.c_char_p

# name -> argtypes (all return int status); mirrors include/nele_hip.h
_SIGS = {
    'nele_device_info': [ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.c_char_p, c_int],
    'nele_build_has_ab_switches': [],
    'nele_stream_spin': [ctypes.c_double, c_void_p],
    'nele_stream_occupy': [c_int, c_int, ctypes.c_double, c_void_p],
    'nele_stft_band': [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    'nele_imcra_band': [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    'nele_gain_istft': [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
    'nele_wav_post': [c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p],
    'nele_profile_begin': [ctypes.c_char_p],
    'nele_profile_collect': [c_void_p, c_int],
    'nele_profile_collect_tag': [ctypes.c_char_p, c_void_p, c_int],
    'nele_stft_band_var': [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    'nele_imcra_band_var': [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    'nele_imcra_band_ws': [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p],
    'nele_imcra_workspace_bytes': [c_int, c_int],
    'nele_stft_pow_var': [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    'nele_imcra_band_pw': [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p],
    'nele_gain_istft_var': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
    'nele_wav_post_var': [c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p],
    'nele_compute_band_E': [c_void_p, c_int, c_void_p, c_void_p],
    'nele_interp_band_gain': [c_void_p, c_int, c_void_p, c_void_p],
}


def declare(name, argtypes):
    fn = getattr(lib, name)
    fn.argtypes = argtypes
    fn.restype = c_int
    return fn


for _n, _a in _SIGS.items():
    declare(_n, _a)
_SIGS.update({
    'nele_wav_decode_pcm16': [ctypes.c_char_p, c_void_p, c_longlong, ctypes.POINTER(c_longlong), ctypes.POINTER(c_int)],
    'nele_wav_write_pcm16': [ctypes.c_char_p, c_void_p, c_longlong, c_int, c_int],
    'nele_wav_read_pcm16_batch': [c_void_p, c_int, c_void_p, c_longlong, c_longlong, c_void_p, c_void_p, c_int],
    'nele_wav_write_pcm16_batch': [c_void_p, c_int, c_void_p, c_longlong, c_void_p, c_int, c_int],
    'nele_wav_probe_pcm16_batch': [c_void_p, c_int, c_void_p, c_int],
    'nele_pcm16_to_float': [c_void_p, c_longlong, c_void_p, c_int, c_longlong, c_void_p, c_longlong, c_void_p],
    'nele_float_to_pcm16': [c_void_p, c_longlong, c_int, c_longlong, c_void_p, c_longlong, c_int, c_void_p],
    'nele_event_record': [c_void_p, c_void_p],
    'nele_stream_wait_event': [c_void_p, c_void_p],
    'nele_vec_add': [c_void_p, c_void_p, c_longlong, c_void_p],
    'nele_event_create': [ctypes.POINTER(c_void_p)],
    'nele_event_destroy': [c_void_p],
    'nele_plan_op_id': [ctypes.c_char_p],
    'nele_plan_op_nargs': [c_int],
    'nele_plan_create': [ctypes.c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p)],
    'nele_plan_run': [c_void_p, ctypes.POINTER(c_void_p), c_int, ctypes.POINTER(c_longlong), c_int],
    'nele_plan_destroy': [c_void_p],
    'nele_plan_declare_slot': [c_void_p, c_int, c_longlong, c_int],
    'nele_plan_slot_bytes': [c_void_p, c_int],
    'nele_plan_run_sized': [c_void_p, ctypes.POINTER(c_void_p), c_int, ctypes.POINTER(c_longlong), ctypes.POINTER(c_longlong), c_int],
    'nele_gen_param_count': [],
    'nele_gen_param_layout': [ctypes.POINTER(c_longlong), c_int],
    'nele_gen_workspace_bytes': [c_int, c_int, c_int, c_int],
    'nele_gen_plan_build': [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p)],
    'nele_disc_param_count': [c_int, c_int],
    'nele_disc_param_layout': [c_int, c_int, ctypes.POINTER(c_longlong), c_int],
    'nele_disc_workspace_bytes': [c_int, c_int, c_int, c_int],
    'nele_disc_workspace_ddin': [c_void_p, c_int, c_int, c_int, c_int],
    'nele_disc_plan_build': [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_void_p, c_longlong, c_void_p,
                             ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p)],
    'nele_gen_fwd': [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_uint, ctypes.POINTER(c_void_p), c_int],
    'nele_gen_bwd': [c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_int],
    'nele_disc_fwd': [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_int],
    'nele_disc_bwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_int],
})
for _n in ('nele_wav_decode_pcm16', 'nele_wav_write_pcm16', 'nele_wav_read_pcm16_batch', 'nele_wav_write_pcm16_batch', 'nele_wav_probe_pcm16_batch', 'nele_pcm16_to_float', 'nele_float_to_pcm16', 'nele_event_record', 'nele_stream_wait_event', 'nele_vec_add', 'nele_event_create', 'nele_event_destroy', 'nele_plan_op_id', 'nele_plan_op_nargs',
           'nele_plan_create', 'nele_plan_run', 'nele_plan_destroy', 'nele_gen_fwd', 'nele_gen_bwd', 'nele_disc_fwd', 'nele_disc_bwd'):
    declare(_n, _SIGS[_n])
for _n in ('nele_plan_declare_slot', 'nele_plan_run_sized', 'nele_gen_param_layout', 'nele_gen_plan_build', 'nele_disc_param_layout', 'nele_disc_plan_build'):
    declare(_n, _SIGS[_n])
for _n in ('nele_plan_slot_bytes', 'nele_gen_param_count', 'nele_gen_workspace_bytes', 'nele_disc_param_count', 'nele_disc_workspace_bytes'):
    getattr(lib, _n).argtypes = _SIGS[_n]
    getattr(lib, _n).restype = c_longlong
lib.nele_disc_workspace_ddin.argtypes = _SIGS['nele_disc_workspace_ddin']
lib.nele_disc_workspace_ddin.restype = c_void_p
lib.nele_imcra_workspace_bytes.restype = c_longlong
lib.nele_wav_post_workspace_doubles.argtypes = [c_int, c_int]
lib.nele_wav_post_workspace_doubles.restype = c_longlong
_SIGS['nele_wav_post_workspace_doubles'] = lib.nele_wav_post_workspace_doubles.argtypes


def check(status, name=''):
    if status != 0:
        msg = lib.nele_last_error_string().decode('utf-8', 'replace')
        if status == -1:
            raise ValueError("%s: %s" % (name, msg))
        raise NeleError("%s failed (status %d): %s" % (name, status, msg))


# ------------------------------------------------------------------ job tables (csrc/plan.hip): record a pass once, replay it with one call
PLAN_MAXARGS = 24


class nele_plan_job(ctypes.Structure):
    _fields_ = [('op', c_int), ('nargs', c_int), ('stream', c_int), ('slot', c_int * PLAN_MAXARGS), ('ival', c_longlong * PLAN_MAXARGS),
                ('fval', c_double * PLAN_MAXARGS)]


class DynInt:
    """An integer argument that changes from call to call (a counter): value = slot value + offset.  ctypes passes ``_as_parameter_``."""

    def __init__(self, slot, off, base):
        self.slot, self.off = int(slot), int(off)
        self._as_parameter_ = int(base) + int(off)


def _addr(a, keep):
    if a is None:
        return 0
    if isinstance(a, int):
        return a
    if isinstance(a, c_void_p):
        return a.value or 0
    if isinstance(a, ctypes.Array):
        keep.append(a)                                   # host arrays (geometries, pointer tables) must outlive the plan
        return ctypes.addressof(a)
    if isinstance(a, ctypes._Pointer):
        keep.append(a)
        return ctypes.cast(a, c_void_p).value or 0
    raise TypeError("plan recorder: cannot take the address of %r" % (a,))


class PlanRecorder:
    """Collects the library calls of a pass while they execute (call() below appends to the active recorder).  ``dyn``: per slot either
    (address, nbytes) of a tensor whose address changes from call to call - every pointer argument inside that range is recorded as
    slot + offset - or None for a scalar slot (DynInt arguments)."""

    def __init__(self, main_stream, dyn):
        self.main = main_stream or 0
        self.streams = [self.main]
        self.dyn = list(dyn)
        self.jobs, self.keep = [], []

    def add(self, name, args):
        sig = _SIGS.get(name)
        op = lib.nele_plan_op_id(name.encode())
        if sig is None or op < 0:
            raise NeleError("plan recorder: %s is not a plan operation (inside a recorded pass every call must be a stream-enqueuing entry point)" % name)
        if len(args) != len(sig) or len(args) > PLAN_MAXARGS:
            raise NeleError("plan recorder: %s takes %d arguments, got %d" % (name, len(sig), len(args)))
        j = nele_plan_job()
        j.op, j.nargs = op, len(args)
        for i, (a, t) in enumerate(zip(args, sig)):
            j.slot[i] = -1
            if i == len(args) - 1:                        # the stream
                h = _addr(a, self.keep)
                if h not in self.streams:
                    self.streams.append(h)
                j.stream = self.streams.index(h)
            elif isinstance(a, DynInt):
                j.slot[i], j.ival[i] = a.slot, a.off
            elif t in (c_float, c_double):
                j.fval[i] = float(a)
            elif t in (c_int, c_longlong, ctypes.c_uint, ctypes.c_ulonglong):
                j.ival[i] = int(a)
            else:                                          # a pointer
                v = _addr(a, self.keep)
                j.ival[i] = v
                if v:
                    for k, rng in enumerate(self.dyn):
                        if rng is not None and rng[0] <= v < rng[0] + max(1, rng[1]):
                            j.slot[i], j.ival[i] = k, v - rng[0]
                            break
        self.jobs.append(j)

    def finish(self):
        return Plan(self)


class Plan:
    def __init__(self, rec):
        self.n = len(rec.jobs)
        self._jobs = (nele_plan_job * self.n)(*rec.jobs)
        self._keep = rec.keep
        self.nslots = len(rec.dyn)
        self.streams = (c_void_p * len(rec.streams))(*rec.streams)
        self.side = tuple(rec.streams[1:])
        h = c_void_p()
        check(lib.nele_plan_create(self._jobs, self.n, self.nslots, len(rec.streams), ctypes.byref(h)), 'nele_plan_create')
        self.handle = h
        for k, rng in enumerate(rec.dyn):                  # what a call must provide behind each pointer slot (nele_plan_run refuses a NULL there)
            if rng is not None and rng[1] > 0:
                check(lib.nele_plan_declare_slot(h, k, int(rng[1]), 0), 'nele_plan_declare_slot')

    def run(self, *slots):
        """Enqueue the recorded pass: streams[0] = the current stream, the side streams as recorded."""
        self.streams[0] = _raw_stream(_raw_device()) if _raw_stream is not None else torch.cuda.current_stream().cuda_stream
        arr = (c_longlong * self.nslots)(*slots)
        check(lib.nele_plan_run(self.handle, self.streams, len(self.streams), arr, self.nslots), 'nele_plan_run')

    def __del__(self):
        try:
            if self.handle:
                lib.nele_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


_recorder = None
PLANS = True          # False: every pass is driven call by call (tests compare the two; bench.py reports both host times)


class recording:
    """with recording(dyn) as rec: ...library calls...   ->  rec.finish() is the plan of what ran."""

    def __init__(self, dyn):
        self.dyn = dyn

    def __enter__(self):
        global _recorder
        if _recorder is not None:
            raise NeleError("plan recorder: recordings do not nest")
        _recorder = PlanRecorder(stream().value, self.dyn)
        return _recorder

    def __exit__(self, *exc):
        global _recorder
        _recorder = None
        return False


def call(name, *args):
    if _recorder is not None:
        _recorder.add(name, args)
    check(getattr(lib, name)(*args), name)


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("nele_gan_amd: expected a tensor on the GPU (got %s)" % t.device)
    if not t.is_contiguous():
        raise ValueError("nele_gan_amd: expected a contiguous tensor")
    return c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def stream():
    """The current torch HIP stream as a hipStream_t.  (torch.cuda.current_stream() builds a Stream object through three Python layers:
    10 us a call, 0.5 ms of a B = 32 step's 4 ms of host time; the raw accessor is the same lookup without the object.)"""
    if _raw_stream is not None and _raw_device is not None:
        return c_void_p(_raw_stream(_raw_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def profile_begin(tag):
    """Arm the library's HIP-event hook for the launch sites tagged ``tag`` (a name or a comma-separated list; None disarms)."""
    lib.nele_profile_begin(tag.encode() if tag else None)


def profile_collect_tag(tag, max_n=8192):
    """-> list of durations (ms) of the launches tagged ``tag`` since profile_begin (synchronises on them; the hook stays armed)."""
    buf = (c_float * max_n)()
    n = lib.nele_profile_collect_tag(tag.encode(), buf, max_n)
    return [float(buf[i]) for i in range(n)]


def profile_collect(max_n=4096):
    """-> list of durations (ms) of the tagged launches since profile_begin (synchronises on them)."""
    buf = (c_float * max_n)()
    n = lib.nele_profile_collect(buf, max_n)
    return [float(buf[i]) for i in range(n)]


def device_info():
    cu = c_int(0)
    ws = c_int(0)
    buf = ctypes.create_string_buffer(64)
    call('nele_device_info', ctypes.byref(cu), ctypes.byref(ws), buf, 64)
    return {'cu_count': cu.value, 'wave_size': ws.value, 'arch': buf.value.decode()}
