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
lib.nele_wav_post_workspace_doubles.argtypes = [c_int, c_int]
lib.nele_wav_post_workspace_doubles.restype = c_longlong
_SIGS['nele_wav_post_workspace_doubles'] = lib.nele_wav_post_workspace_doubles.argtypes


def check(status, name=''):
    if status != 0:
        msg = lib.nele_last_error_string().decode('utf-8', 'replace')
        if status == -1:
            raise ValueError("%s: %s" % (name, msg))
        raise NeleError("%s failed (status %d): %s" % (name, status, msg))


def call(name, *args):
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
