"""Batched objective metrics on the GPU: host-side mirror of the reference ``intel.py`` wrappers and
the ``read_batch_*`` fan-out of ``audio_util.py:120-203`` (there: 32 joblib processes over wav files;
here: one batched launch over utterances resident in HBM).

    ESTOI_Wrapper[_raw]_harvard(x, y, fs)   intel.py:122-134
    SIIB_Wrapper[_raw]_harvard(x, y, fs)    intel.py:57-100
    HASPI_Wrapper[_raw]_harvard(x, y, fs)   intel.py:108-114
    batch_estoi / batch_siib / batch_haspi (clean[B,L], degraded[B,L]) -> (raw[B], mapped[B])

x is the clean reference, y the degraded signal (enhanced + noise, audio_util.py:139-141).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import c_int, c_longlong, c_void_p, call, declare, ptr, stream

_P = c_void_p
declare('nele_metric_estoi', [_P, _P, c_int, c_int, _P, c_longlong, _P, _P, _P])
_lib._SIGS['nele_metric_estoi'] = _lib.lib.nele_metric_estoi.argtypes
declare('nele_metric_estoi_var', [_P, _P, _P, c_int, c_int, _P, c_longlong, _P, _P, _P])
_lib._SIGS['nele_metric_estoi_var'] = _lib.lib.nele_metric_estoi_var.argtypes
declare('nele_metric_siib_var', [_P, _P, _P, c_int, c_int, _P, c_longlong, _P, _P, _P, c_int, _P])
_lib._SIGS['nele_metric_siib_var'] = _lib.lib.nele_metric_siib_var.argtypes
_lib.lib.nele_metric_estoi_workspace_bytes.argtypes = [c_int, c_int]
_lib.lib.nele_metric_estoi_workspace_bytes.restype = c_longlong
_lib._SIGS['nele_metric_estoi_workspace_bytes'] = _lib.lib.nele_metric_estoi_workspace_bytes.argtypes

declare('nele_metric_siib', [_P, _P, c_int, c_int, _P, c_longlong, _P, _P, _P, _P])
declare('nele_metric_siib_phase', [_P, _P, c_int, c_int, _P, c_longlong, _P, _P, _P, c_int, _P])
_lib._SIGS['nele_metric_siib_phase'] = _lib.lib.nele_metric_siib_phase.argtypes
_lib._SIGS['nele_metric_siib'] = _lib.lib.nele_metric_siib.argtypes
_lib.lib.nele_metric_siib_workspace_bytes.argtypes = [c_int, c_int]
_lib.lib.nele_metric_siib_workspace_bytes.restype = c_longlong
_lib._SIGS['nele_metric_siib_workspace_bytes'] = _lib.lib.nele_metric_siib_workspace_bytes.argtypes

declare('nele_metric_haspi', [_P, _P, c_int, c_int, c_int, _P, _P, c_longlong, _P, _P, _P, _P])
_lib._SIGS['nele_metric_haspi'] = _lib.lib.nele_metric_haspi.argtypes
declare('nele_metric_haspi_var', [_P, _P, _P, c_int, c_int, c_int, _P, _P, c_longlong, _P, _P, _P, c_int, _P])
_lib._SIGS['nele_metric_haspi_var'] = _lib.lib.nele_metric_haspi_var.argtypes
declare('nele_metric_haspi_var_hl', [_P, _P, _P, c_int, c_int, c_int, _P, ctypes.POINTER(ctypes.c_double), c_int, _P, c_longlong, _P, _P, _P, c_int, _P])
_lib._SIGS['nele_metric_haspi_var_hl'] = _lib.lib.nele_metric_haspi_var_hl.argtypes
_lib.lib.nele_metric_haspi_workspace_bytes.argtypes = [c_int, c_int, c_int]
_lib.lib.nele_metric_haspi_workspace_bytes.restype = c_longlong
_lib._SIGS['nele_metric_haspi_workspace_bytes'] = _lib.lib.nele_metric_haspi_workspace_bytes.argtypes
_lib.lib.nele_metric_haspi_quality_workspace_bytes.argtypes = [c_int, c_int, c_int]
_lib.lib.nele_metric_haspi_quality_workspace_bytes.restype = c_longlong
_lib._SIGS['nele_metric_haspi_quality_workspace_bytes'] = _lib.lib.nele_metric_haspi_quality_workspace_bytes.argtypes
declare('nele_metric_haspi_quality', [_P, _P, _P, c_int, c_int, c_int, c_int, ctypes.c_ulonglong, ctypes.c_double, _P, c_longlong, _P, _P, _P])
_lib._SIGS['nele_metric_haspi_quality'] = _lib.lib.nele_metric_haspi_quality.argtypes
declare('nele_metric_haspi_quality_hl', [_P, _P, _P, c_int, c_int, c_int, c_int, ctypes.c_ulonglong, ctypes.c_double, ctypes.POINTER(ctypes.c_double), c_int, _P,
                                         c_longlong, _P, _P, _P])
_lib._SIGS['nele_metric_haspi_quality_hl'] = _lib.lib.nele_metric_haspi_quality_hl.argtypes
_lib.lib.nele_metric_haspi_nsub.argtypes = [c_int, c_int]
_lib.lib.nele_metric_haspi_nsub.restype = c_int
_lib._SIGS['nele_metric_haspi_nsub'] = _lib.lib.nele_metric_haspi_nsub.argtypes

declare('nele_metric_siib_clean_sections', [c_int, c_int, ctypes.POINTER(c_longlong), c_int])
_lib._SIGS['nele_metric_siib_clean_sections'] = _lib.lib.nele_metric_siib_clean_sections.argtypes
declare('nele_metric_haspi_clean_sections', [c_int, c_int, c_int, ctypes.POINTER(c_longlong), c_int])
_lib._SIGS['nele_metric_haspi_clean_sections'] = _lib.lib.nele_metric_haspi_clean_sections.argtypes

declare('nele_haspi_dither_rows', [_P, ctypes.c_ulonglong, c_int, c_int, _P, _P])
_lib._SIGS['nele_haspi_dither_rows'] = _lib.lib.nele_haspi_dither_rows.argtypes

declare('nele_eigh_sym_batched', [_P, c_int, c_int, _P, _P, _P, c_longlong, _P])
_lib._SIGS['nele_eigh_sym_batched'] = _lib.lib.nele_eigh_sym_batched.argtypes
_lib.lib.nele_eigh_repaired.argtypes = [_P, c_int, c_int]
_lib.lib.nele_eigh_repaired.restype = c_int
_lib._SIGS['nele_eigh_repaired'] = _lib.lib.nele_eigh_repaired.argtypes
_lib.lib.nele_eigh_workspace_bytes.argtypes = [c_int, c_int]
_lib.lib.nele_eigh_workspace_bytes.restype = c_longlong
_lib._SIGS['nele_eigh_workspace_bytes'] = _lib.lib.nele_eigh_workspace_bytes.argtypes

_ws_cache = {}


def _workspace(kind, nbytes, dev, cache=None):
    """Grow-only scratch buffer per (kind, device).  ``cache``: the dict that owns it - the module-global one for the one-shot batch_*
    calls, or a dict held by the caller (a GanTrainer) for the split objects' multi-GB workspaces, so that they are released with
    their owner instead of living until the process exits."""
    cache = _ws_cache if cache is None else cache
    key = (kind, str(dev))
    t = cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        cache[key] = t
    return t


def release_workspaces():
    """Drop the module-global scratch buffers (they are re-created on demand)."""
    _ws_cache.clear()


class CleanStateCache:
    """Per-utterance clean-signal state of the split metrics, kept across calls (GanTrainer.enable_clean_cache).

    The reference scores the same <= 720 clean training files in every one of its 500 GAN epochs (train_nele.py:35-38,119,318-340) and
    recomputes the clean file's half of every metric each time.  That half is a pure function of the clean waveform: SIIB's VAD,
    clean spectra, covariance and KLT eigen-decomposition (phase 3 of nele_metric_siib_var: the step's single most expensive
    kernel chain), HASPI's whole reference-signal chain (phase 3 of nele_metric_haspi_var).  The library says which byte ranges of a
    workspace make up that state (nele_metric_{siib,haspi}_clean_sections); this class copies them out after phase 3 - one row per
    utterance and section in pooled device buffers - and copies them back in place of phase 3 when the same utterances come by again,
    at whatever rows of whatever batch.  Copies only: the scores are bit-identical to recomputation (tests/test_clean_cache_gpu.py).
    Keys: (kind, geometry = padded length L (+ fs, dither mode for HASPI), utterance key); the utterance key (file name, corpus id)
    stands for the clean waveform including its own length.  Never invalidated (clean files are immutable); once ``budget_bytes`` of device memory are in use, new utterances are simply not stored (they are recomputed)."""

    BLOCK_ROWS = 64

    def __init__(self, budget_bytes):
        self.budget = int(budget_bytes)
        self.used = 0
        self.pools = {}             # (kind, geometry) -> pool dict
        self.hits = self.misses = self.stored = self.declined = self.partial = 0
        self.poison = False         # tests: fill the workspace with 0xFF bytes before a restore (a forgotten section cannot go unnoticed)

    @staticmethod
    def _sections(kind, B, L, fs):
        buf = (c_longlong * (3 * 24))()
        if kind == 'siib':
            n = _lib.lib.nele_metric_siib_clean_sections(int(B), int(L), buf, 24)
        else:
            n = _lib.lib.nele_metric_haspi_clean_sections(int(B), int(L), int(fs), buf, 24)
        if n < 0:
            return None
        return [(int(buf[3 * k]), int(buf[3 * k + 1]), int(buf[3 * k + 2])) for k in range(n)]

    def _pool(self, kind, geom, L, fs):
        key = (kind, geom)
        p = self.pools.get(key)
        if p is None:
            sec = self._sections(kind, 1, L, fs)
            if sec is None:
                return None
            p = self.pools[key] = {'row_bytes': [b for (_o, st, b) in sec if st > 0], 'shared': None, 'blocks': [], 'index': {}, 'free': 0,
                                   'event': None, 'per_utt': sum(b for (_o, st, b) in sec if st > 0)}
        return p

    def lookup(self, kind, geom, keys, count=True):
        """-> list of (block, row) slots when EVERY utterance of the batch is cached, else None."""
        p = self.pools.get((kind, geom))
        if p is None or p['shared'] is None:
            self.misses += int(count)
            return None
        idx = p['index']
        slots = [idx.get(k) for k in keys]
        if any(s is None for s in slots):
            self.misses += int(count)
            return None
        self.hits += int(count)
        return slots

    def missing(self, kind, geom, keys):
        """Rows of the batch whose utterances are not cached yet, or None when the batch is better computed as a whole (nothing of it is
        cached, or the pool of this geometry does not exist yet)."""
        p = self.pools.get((kind, geom))
        if p is None or p['shared'] is None:
            return None
        rows = [i for i, k in enumerate(keys) if k not in p['index']]
        return rows if 0 < len(rows) < len(keys) else None

    @staticmethod
    def _runs(slots):
        """[(block, row)] in batch order -> runs (first batch row, block, first block row, count) of consecutive rows"""
        runs, k = [], 0
        while k < len(slots):
            b, r = slots[k]
            n = 1
            while k + n < len(slots) and slots[k + n] == (b, r + n):
                n += 1
            runs.append((k, b, r, n))
            k += n
        return runs

    def restore(self, kind, geom, ws, B, L, fs, slots):
        """Copy the cached state of the batch's utterances into workspace ``ws`` (uint8 tensor) on the current stream."""
        p = self.pools[(kind, geom)]
        sec = self._sections(kind, B, L, fs)
        cur = torch.cuda.current_stream(ws.device)
        if p['event'] is not None:
            cur.wait_event(p['event'])           # the stores may have run on another stream
        if self.poison:
            ws.fill_(255)
        runs = self._runs(slots)
        j = sh = 0
        for (off, st, nb) in sec:
            if st == 0:
                ws[off:off + nb].copy_(p['shared'][sh])
                sh += 1
                continue
            view = ws[off:off + B * st].view(B, st)[:, :nb]
            for (k0, blk, r0, n) in runs:
                view[k0:k0 + n].copy_(p['blocks'][blk][j][r0:r0 + n])
            j += 1

    def store(self, kind, geom, ws, B, L, fs, keys):
        """Copy the state phase 3 just left in ``ws`` into the pool for the utterances not cached yet (current stream)."""
        p = self._pool(kind, geom, L, fs)
        if p is None:
            return False
        sec = self._sections(kind, B, L, fs)
        if p['shared'] is None:
            p['shared'] = [ws[off:off + nb].clone() for (off, st, nb) in sec if st == 0]
            self.used += sum(t.numel() for t in p['shared'])
        new = [(k, key) for k, key in enumerate(keys) if key not in p['index']]
        if not new:
            return True
        n_new = len(new)
        slots = []
        for _k, _key in new:
            if p['free'] == 0:
                need = self.BLOCK_ROWS * p['per_utt']
                if self.used + need > self.budget:
                    self.declined += len(new) - len(slots)
                    break
                p['blocks'].append([torch.empty((self.BLOCK_ROWS, nb), dtype=torch.uint8, device=ws.device) for nb in p['row_bytes']])
                p['free'] = self.BLOCK_ROWS
                self.used += need
            slots.append((len(p['blocks']) - 1, self.BLOCK_ROWS - p['free']))
            p['free'] -= 1
        new = new[:len(slots)]
        if not new:
            return False
        # runs of consecutive batch rows that landed in consecutive pool rows
        runs, k = [], 0
        while k < len(new):
            n = 1
            while (k + n < len(new) and new[k + n][0] == new[k][0] + n and slots[k + n] == (slots[k][0], slots[k][1] + n)):
                n += 1
            runs.append((new[k][0], slots[k][0], slots[k][1], n))
            k += n
        j = 0
        for (off, st, nb) in sec:
            if st == 0:
                continue
            view = ws[off:off + B * st].view(B, st)[:, :nb]
            for (k0, blk, r0, n) in runs:
                p['blocks'][blk][j][r0:r0 + n].copy_(view[k0:k0 + n])
            j += 1
        for (_k, key), slot in zip(new, slots):
            p['index'][key] = slot
        self.stored += len(new)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(ws.device))
        p['event'] = ev
        return len(new) == n_new

    def stats(self):
        return {'hits': self.hits, 'misses': self.misses, 'partial': self.partial, 'stored': self.stored, 'declined': self.declined, 'bytes': self.used,
                'utterances': {'%s@%s' % k: len(p['index']) for k, p in self.pools.items()}}


def eigh_batched(A, return_repaired=False):
    """A [B,n,n] float64 symmetric (device) -> (eigenvalues [B,n] ascending, U [B,n,n] with rows = eigenvectors).
    return_repaired: also the number of matrices the cluster tridiagonalisation gave up on and the repair kernel redid (synchronises)."""
    A = A.clone().contiguous()
    B, n, _ = A.shape
    lam = torch.empty((B, n), dtype=torch.float64, device=A.device)
    U = torch.empty_like(A)
    ws = _workspace('eigh', _lib.lib.nele_eigh_workspace_bytes(B, n), A.device)
    call('nele_eigh_sym_batched', ptr(A), n, B, ptr(lam), ptr(U), ptr(ws), ws.numel(), stream())
    if return_repaired:
        return lam, U, int(_lib.lib.nele_eigh_repaired(ptr(ws), B, n))
    return lam, U


def _pair(x, y):
    def dev(a):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a))
        if not a.is_cuda:
            a = a.cuda()
        return a.float()
    x, y = dev(x), dev(y)
    single = x.dim() == 1
    if single:
        x, y = x.unsqueeze(0), y.unsqueeze(0)
    L = min(x.shape[1], y.shape[1])                    # intel.py:58-60 (minL truncation)
    return x[:, :L].contiguous(), y[:, :L].contiguous(), single


def _lens(lengths, device):
    return None if lengths is None else torch.as_tensor(lengths).to(device=device, dtype=torch.int32).contiguous()


def batch_estoi(x, y, lengths=None):
    """clean x [B,L], degraded y [B,L] (16 kHz) -> (raw [B], mapped [B]) float32 device tensors.
    lengths [B]: samples of each utterance inside the padded batch (one launch for files of different lengths)."""
    x, y, _ = _pair(x, y)
    B, L = x.shape
    nb = _lib.lib.nele_metric_estoi_workspace_bytes(B, L)
    ws = _workspace('estoi', nb, x.device)
    raw = torch.empty(B, device=x.device)
    mapped = torch.empty(B, device=x.device)
    call('nele_metric_estoi_var', ptr(x), ptr(y), ptr(_lens(lengths, x.device)), B, L, ptr(ws), ws.numel(), ptr(raw), ptr(mapped), stream())
    return raw, mapped


class SiibSplit:
    """batch_siib in pieces.  front()/back(): wide kernels (VAD .. covariance) / latency-bound eigen-decomposition .. score, so
    that independent work can be enqueued in between on another stream.  clean_part()/degraded_part(y): split by data
    dependence - SIIB's Karhunen-Loeve basis comes from the clean signal alone, so VAD, the clean spectra, the covariance and its
    eigen-decomposition can run before the degraded signal exists."""

    def __init__(self, x, y=None, lengths=None, owner=None, ws_kind='siib_split'):
        self.lengths = None
        self.owner = owner
        if y is None:
            self.x = x.contiguous().float()
            self.y = None
        else:
            self.x, self.y, _ = _pair(x, y)
        B, L = self.x.shape
        # own workspace: the clean-signal state must survive until degraded_part(), whatever else calls batch_siib() meanwhile
        # (``owner``: a dict held by the caller, e.g. one per trainer, that owns the workspace - released with the trainer)
        self.ws = _workspace(ws_kind, _lib.lib.nele_metric_siib_workspace_bytes(B, L), self.x.device, cache=owner)
        self.raw = torch.empty(B, device=self.x.device)
        self.mapped = torch.empty(B, device=self.x.device)
        self.info = torch.zeros((B, 4), dtype=torch.int32, device=self.x.device)
        self.lengths = _lens(lengths, self.x.device)

    def _call(self, phase):
        B, L = self.x.shape
        call('nele_metric_siib_var', ptr(self.x), ptr(self.y), ptr(self.lengths), B, L, ptr(self.ws), self.ws.numel(), ptr(self.raw),
             ptr(self.mapped), ptr(self.info), phase, stream())

    def front(self):
        self._call(1)

    def back(self):
        self._call(2)
        return self.raw, self.mapped

    def clean_part(self, cache=None, keys=None):
        """cache (CleanStateCache) + keys (one hashable per utterance, e.g. the file name): when every utterance's clean-signal state is
        cached it is copied into the workspace instead of being recomputed; when some are, phase 3 runs for the others only (a batch of
        their own on a second workspace) and everything is copied in; otherwise phase 3 runs for the batch and its utterances are stored."""
        if cache is None or keys is None:
            self._call(3)
            return False
        B, L = self.x.shape
        geom = ('L', L)
        full = list(keys)                       # a key stands for the clean waveform, its own length included
        slots = cache.lookup('siib', geom, full)
        if slots is not None:
            cache.restore('siib', geom, self.ws, B, L, 0, slots)
            return True
        rows = cache.missing('siib', geom, full)
        if rows is not None:
            # some utterances of the batch are known (a loop that re-draws its batches every epoch): phase 3 for the others only, as a batch
            # of their own on a second workspace, then everything comes from the cache (the kernels are batch-invariant: an utterance's state
            # does not depend on its row or on its batch-mates)
            sub = SiibSplit(self.x[rows], lengths=None if self.lengths is None else self.lengths[rows], owner=self.owner, ws_kind='siib_split_sub')
            sub._call(3)
            if cache.store('siib', geom, sub.ws, len(rows), L, 0, [full[i] for i in rows]):
                slots = cache.lookup('siib', geom, full, count=False)
                if slots is not None:
                    cache.partial += 1
                    cache.restore('siib', geom, self.ws, B, L, 0, slots)
                    return True
        self._call(3)
        cache.store('siib', geom, self.ws, B, L, 0, full)
        return False

    def degraded_part(self, y):
        assert y.shape == self.x.shape
        self.y = y.contiguous().float()
        self._call(4)
        return self.raw, self.mapped


def batch_siib(x, y, return_info=False, lengths=None):
    """clean x [B,L], degraded y [B,L] (16 kHz) -> (raw [B] bits/s, mapped [B]); the replication rule of
    intel.py:93-97 is applied per utterance on the device.  info [B,4] = (M, tiled frames, active frames, status)."""
    x, y, _ = _pair(x, y)
    B, L = x.shape
    nb = _lib.lib.nele_metric_siib_workspace_bytes(B, L)
    ws = _workspace('siib', nb, x.device)
    raw = torch.empty(B, device=x.device)
    mapped = torch.empty(B, device=x.device)
    info = torch.zeros((B, 4), dtype=torch.int32, device=x.device)
    call('nele_metric_siib_var', ptr(x), ptr(y), ptr(_lens(lengths, x.device)), B, L, ptr(ws), ws.numel(), ptr(raw), ptr(mapped), ptr(info), 0,
         stream())
    if return_info:
        return raw, mapped, info
    return raw, mapped


def _siib_checked(x, y):
    raw, mapped, info = batch_siib(x, y, return_info=True)
    st = int(info[0, 3])
    if st & 8:
        raise ValueError("SIIB: not enough active speech frames")       # pysiib / reference raise here
    if st & 1:
        raise ValueError("SIIB: replication factor above the supported maximum")
    return raw, mapped


def SIIB_Wrapper_raw_harvard(x, y, fs):
    assert fs == 16000
    return float(_siib_checked(x, y)[0][0])


def SIIB_Wrapper_harvard(x, y, fs):
    assert fs == 16000
    return float(_siib_checked(x, y)[1][0])


def _hl6(HL):
    """Audiogram argument of the reference's pyHASPI functions (HL = np.zeros(6) by default) -> ctypes double[6] or None (normal hearing)."""
    if HL is None:
        return None
    v = [float(t) for t in np.asarray(HL, dtype=np.float64).reshape(-1)]
    if len(v) != 6:
        raise ValueError('HL must hold the six audiogram values at 250, 500, 1000, 2000, 4000, 6000 Hz (pyhaspi2.py:780)')
    if not any(v):
        return None
    return (ctypes.c_double * 6)(*v)


def batch_haspi(x, y, fs=16000, dither=None, seed=None, return_info=False, lengths=None, HL=None):
    """clean x [B,L], degraded y [B,L] -> (raw HASPI v2 [B], mapped [B]).
    HL: audiogram of the listener (six values in dB HL, pyhaspi2.py:76, 779-807); None / zeros = normal hearing (what the loop uses).
    dither: None -> no IHC firing jitter (deterministic score); True -> standard normals drawn on the device
    (torch generator, optional ``seed``) as the reference does with np.random.randn (pyhaspi2.py:362-365); or a
    float64 tensor [B,2,nsub,32] whose row k perturbs the k-th active frame (used by the parity tests)."""
    x, y, _ = _pair(x, y)
    B, L = x.shape
    nsub = _lib.lib.nele_metric_haspi_nsub(L, fs)
    if dither is True:
        g = None
        if seed is not None:
            g = torch.Generator(device=x.device)
            g.manual_seed(int(seed))
        dither = torch.randn((B, 2, nsub, 32), dtype=torch.float64, device=x.device, generator=g)
    if dither is not None:
        if tuple(dither.shape) != (B, 2, nsub, 32) or dither.dtype != torch.float64:
            raise ValueError("batch_haspi: dither must be float64 [B, 2, %d, 32]" % nsub)
        dither = dither.to(x.device).contiguous()
    nb = _lib.lib.nele_metric_haspi_workspace_bytes(B, L, fs)
    ws = _workspace('haspi', nb, x.device)
    raw = torch.empty(B, device=x.device)
    mapped = torch.empty(B, device=x.device)
    info = torch.zeros((B, 2), dtype=torch.int32, device=x.device)
    if lengths is not None:
        lengths = torch.as_tensor(lengths).to(device=x.device, dtype=torch.int32).contiguous()
    hl = _hl6(HL)
    if hl is None:
        call('nele_metric_haspi_var', ptr(x), ptr(y), ptr(lengths), B, L, int(fs), ptr(dither), ptr(ws), ws.numel(), ptr(raw), ptr(mapped), ptr(info), 0,
             stream())
    else:
        call('nele_metric_haspi_var_hl', ptr(x), ptr(y), ptr(lengths), B, L, int(fs), ptr(dither), hl, 0, ptr(ws), ws.numel(), ptr(raw), ptr(mapped),
             ptr(info), 0, stream())
    if return_info:
        return raw, mapped, info
    return raw, mapped


def haspi_dither_rows(utt_ids, seed, L, fs=16000, device=None):
    """Dither rows [B, 2, nsub, 32] float64 for batch_haspi / HaspiSplit, drawn per UTTERANCE ID (pyhaspi2.py:362-365 draws
    np.random.randn rows on every call; here a row is a pure function of (seed, id, signal, frame, channel), so an utterance is
    scored alike on any rank and in any batch - SURVEY 8e).  utt_ids: [B] integers."""
    ids = torch.as_tensor(utt_ids)
    device = device or (ids.device if ids.is_cuda else 'cuda')
    ids = ids.to(device=device, dtype=torch.int64).contiguous()
    B = ids.shape[0]
    nsub = _lib.lib.nele_metric_haspi_nsub(int(L), int(fs))
    out = torch.empty((B, 2, nsub, 32), dtype=torch.float64, device=ids.device)
    call('nele_haspi_dither_rows', ptr(ids), int(seed) & 0xFFFFFFFFFFFFFFFF, B, nsub, ptr(out), stream())
    return out


class HaspiSplit:
    """batch_haspi split by data dependence: clean_part() runs the whole reference-signal chain (ear model, envelope filter,
    silence gate, group-delay shifts, cepstra, modulation filters) before the degraded signal exists; degraded_part(y) does the same
    for y and correlates.  Own workspace: the clean-signal state must survive until degraded_part()."""

    def __init__(self, x, fs=16000, lengths=None, owner=None, ws_kind='haspi_split'):
        self.x = x.contiguous().float()
        self.fs = int(fs)
        self.owner = owner
        B, L = self.x.shape
        self.lengths = None if lengths is None else lengths.to(device=self.x.device, dtype=torch.int32).contiguous()
        self.ws = _workspace(ws_kind, _lib.lib.nele_metric_haspi_workspace_bytes(B, L, self.fs), self.x.device, cache=owner)
        self.raw = torch.empty(B, device=self.x.device)
        self.mapped = torch.empty(B, device=self.x.device)
        self.info = torch.zeros((B, 2), dtype=torch.int32, device=self.x.device)

    def _call(self, y, dither, phase):
        B, L = self.x.shape
        call('nele_metric_haspi_var', ptr(self.x), ptr(y), ptr(self.lengths), B, L, self.fs, ptr(dither), ptr(self.ws), self.ws.numel(),
             ptr(self.raw), ptr(self.mapped), ptr(self.info), phase, stream())

    def clean_part(self, dither=None, cache=None, keys=None, dither_tag=None):
        """cache / keys: as SiibSplit.clean_part.  ``dither_tag``: hashable that identifies the reference's dither rows (None = no dither;
        e.g. ('utterance', seed) for rows drawn per utterance id) - a cached state is only valid for the dither it was computed with;
        a dither with no tag is never cached."""
        if cache is None or keys is None or (dither is not None and dither_tag is None):
            self._call(None, dither, 3)
            return False
        B, L = self.x.shape
        geom = ('L', L, 'fs', self.fs, 'dither', dither_tag)
        full = list(keys)
        slots = cache.lookup('haspi', geom, full)
        if slots is not None:
            cache.restore('haspi', geom, self.ws, B, L, self.fs, slots)
            return True
        rows = cache.missing('haspi', geom, full)
        if rows is not None:                                      # (see SiibSplit.clean_part: the unknown utterances as a batch of their own)
            sub = HaspiSplit(self.x[rows], fs=self.fs, lengths=None if self.lengths is None else self.lengths[rows], owner=self.owner, ws_kind='haspi_split_sub')
            sub._call(None, None if dither is None else dither[rows].contiguous(), 3)
            if cache.store('haspi', geom, sub.ws, len(rows), L, self.fs, [full[i] for i in rows]):
                slots = cache.lookup('haspi', geom, full, count=False)
                if slots is not None:
                    cache.partial += 1
                    cache.restore('haspi', geom, self.ws, B, L, self.fs, slots)
                    return True
        self._call(None, dither, 3)
        cache.store('haspi', geom, self.ws, B, L, self.fs, full)
        return False

    def degraded_part(self, y, dither=None):
        assert y.shape == self.x.shape
        self.y = y.contiguous().float()
        self._call(self.y, dither, 4)
        return self.raw, self.mapped


QUALITY_FIELDS = ('haspi', 'CepCorr', 'cov3_low', 'cov3_mid', 'cov3_high', 'hasqi', 'Nonlin', 'Linear', 'BMsync5', 'Dloud', 'Dslope', 'avecov')


def batch_haspi_quality(x, y, fs=16000, lengths=None, noise=True, seed=None, alpha=-1.0, return_info=False, HL=None, itype=0):
    """reference x [B,L], processed y [B,L] -> float64 [B, 12] (columns QUALITY_FIELDS): HASPI version 1 (pyhaspi2.py:109-157) and
    HASQI v2 (pyhaspi2.py:32-74) from one launch chain.  noise: the reference's eb_BMaddnoise (device generator, ``seed`` or a fresh
    one per call); False = none (deterministic, the parity tests).  HL: audiogram (six values in dB HL); with a loss the two models
    differ in how the reference signal is heard (pyhaspi2.py:1162-1166): itype 0 -> the `haspi` columns are valid, itype 2 -> the
    `hasqi_v2` columns."""
    x, y, _ = _pair(x, y)
    B, L = x.shape
    ws = _workspace('haspi_quality', _lib.lib.nele_metric_haspi_quality_workspace_bytes(B, L, int(fs)), x.device)
    out = torch.empty((B, len(QUALITY_FIELDS)), dtype=torch.float64, device=x.device)
    info = torch.zeros((B, 4), dtype=torch.int32, device=x.device)
    if lengths is not None:
        lengths = torch.as_tensor(lengths).to(device=x.device, dtype=torch.int32).contiguous()
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if noise else 0
    hl = _hl6(HL)
    if hl is None:
        call('nele_metric_haspi_quality', ptr(x), ptr(y), ptr(lengths), B, L, int(fs), int(bool(noise)), int(seed), float(alpha), ptr(ws), ws.numel(),
             ptr(out), ptr(info), stream())
    else:
        call('nele_metric_haspi_quality_hl', ptr(x), ptr(y), ptr(lengths), B, L, int(fs), int(bool(noise)), int(seed), float(alpha), hl, int(itype),
             ptr(ws), ws.numel(), ptr(out), ptr(info), stream())
    return (out, info) if return_info else out


def _quality_checked(x, fx, y, fy, alpha=-1.0, HL=None, itype=0):
    if fx != fy:
        raise ValueError('haspi / hasqi_v2: both signals must have the same sampling rate here (the reference resamples each to 24 kHz)')
    L = min(len(x), len(y))
    out, info = batch_haspi_quality(x[:L], y[:L], fx, alpha=alpha, return_info=True, HL=HL, itype=itype)
    st = int(info[0, 1])
    if st & 1:
        raise Exception('Function eb_melcor: Signal below threshold, outputs set to 0.')        # pyhaspi2.py:723-724
    return out[0].cpu().numpy(), st


def haspi(x, fx, y, fy, HL=None, alpha=-1.0):
    """pyhaspi2.py:109-157 (HASPI version 1): -> (Intel, raw = [CepCorr, cov3 low, mid, high]).  HL: audiogram of the listener."""
    o, st = _quality_checked(x, fx, y, fy, alpha, HL=HL, itype=0)
    if st & 2:
        raise Exception('Function eb_3LevelCovary: Signal below threshold, outputs set to 0.')    # pyhaspi2.py:427-428
    return float(o[0]), o[1:5].copy()


def hasqi_v2(x, fx, y, fy, HL=None):
    """pyhaspi2.py:32-74: -> (Combined, Nonlin, Linear, raw = [CepCorr, BMsync5, Dloud, Dslope]).  HL: audiogram (both signals are heard
    with it: eq = 2, pyhaspi2.py:41-43)."""
    o, st = _quality_checked(x, fx, y, fy, HL=HL, itype=2)
    if st & 2:
        raise TypeError("'int' object is not subscriptable")                                       # eb_AveCovary2's (0, 0) return at pyhaspi2.py:54
    return float(o[5]), float(o[6]), float(o[7]), [float(o[1]), float(o[8]), float(o[9]), float(o[10])]


def haspi_v2(x, fx, y, fy, HL=None):
    """pyhaspi2.py:76-107: -> Intel (the modulation-band correlations `raw` stay on the device).  The reference dithers every call
    (pyhaspi2.py:362-365): so does this wrapper; batch_haspi(dither=None) is the deterministic form."""
    if fx != fy:
        raise ValueError('haspi_v2: both signals must have the same sampling rate here (the reference resamples each to 24 kHz)')
    L = min(len(x), len(y))
    raw, mapped, info = batch_haspi(x[:L], y[:L], fx, dither=True, return_info=True, HL=HL)
    if int(info[0, 1]):
        raise Exception('Function ebm_CepCoef: Signal below threshold')     # pyhaspi2.py:357-358
    return float(raw[0])


def _haspi_checked(x, y, fs):
    raw, mapped, info = batch_haspi(x, y, fs, dither=True, return_info=True)
    if int(info[0, 1]):
        raise Exception('Function ebm_CepCoef: Signal below threshold')     # pyhaspi2.py:357-358
    return raw, mapped


def HASPI_Wrapper_raw_harvard(x, y, fs):
    return float(_haspi_checked(x, y, fs)[0][0])


def HASPI_Wrapper_harvard(x, y, fs):
    return float(_haspi_checked(x, y, fs)[1][0])


def ESTOI_Wrapper_raw_harvard(x, y, fs):
    assert fs == 16000
    return float(batch_estoi(x, y)[0][0])


def ESTOI_Wrapper_harvard(x, y, fs):
    assert fs == 16000
    return float(batch_estoi(x, y)[1][0])


def mapping_ESTOI_harvard(x):
    return 1 / (1 + np.exp(-8.0 * (x - 0.25)))


def mapping_SIIB_harvard(x):
    return 1 / (1 + np.exp(-0.06 * (x - 32)))


def mapping_HASPI_harvard(x):
    return 1 / (1 + np.exp(-0.95 * (x - 2.8)))
