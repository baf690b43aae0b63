"""On-disk hand-off formats of the reference loop, host side.

The reference exchanges everything between its stages through files: enhanced speech is written as PCM_16
wav (``train_nele.py:198,313``, ``inference.py:115``) under ``name@epoch.wav``, the metric fan-out reads the
(clean, noise, enhanced) triple back per file (``audio_util.py:120-203``, ``267-321``), and the D-step consumes
``"s0,s1,s2,s3,s4,path"`` strings (``audio_util.py:367-389``, ``dataloader.py:54-84``).  This module reads and
writes those formats so that folders and lists produced by either implementation are interchangeable, and it
feeds the batched GPU kernels from them: files are decoded on the host, padded side by side with their per-file lengths,
and each metric is one batched launch (the reference: one joblib process per file, ``audio_util.py:146``).

No third-party audio library: RIFF/WAVE PCM is parsed here (libsndfile / librosa are not in the image).
PCM_16 semantics follow libsndfile: write ``rint(x * 0x7FFF)`` (round half to even), read ``s / 0x8000``
(PARITY UNPINNED: libsndfile itself is absent; same rule as ``nele_wav_post`` on the device).
"""
import os
import threading
import struct

import numpy as np

fs = 16000
power_law = (1 / 6)
from ._lib import lib as _nele_lib                     # host-side entry points of libnele_hip.so (no GPU needed to call them; no library, no package)
_native_decode = _nele_lib.nele_wav_decode_pcm16


# ------------------------------------------------------------------------------------------------ wav files
def read_wav(path):
    """-> (float32 [L] mono, sample_rate).  PCM 8/16/24/32-bit and IEEE float32/64 RIFF files; multi-channel
    files are averaged to mono (librosa.load(mono=True))."""
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 12 or data[0:4] != b'RIFF' or data[8:12] != b'WAVE':
        raise ValueError('%s: not a RIFF/WAVE file' % path)
    pos, fmt, pcm = 12, None, None
    view = memoryview(data)                                              # chunk bodies are views: the sample data is not copied before numpy reads it
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from('<I', data, pos + 4)[0]
        body = view[pos + 8:pos + 8 + size]
        if cid == b'fmt ':
            fmt = struct.unpack_from('<HHIIHH', body, 0)
            if fmt[0] == 0xFFFE and len(body) >= 26:                     # WAVE_FORMAT_EXTENSIBLE: sub-format tag
                fmt = (struct.unpack_from('<H', body, 24)[0],) + fmt[1:]
        elif cid == b'data':
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError('%s: missing fmt or data chunk' % path)
    tag, nch, sr, _, _, bits = fmt
    if tag == 1:
        if bits == 16:
            x = np.frombuffer(pcm[:len(pcm) // 2 * 2], dtype='<i2').astype(np.float32) / np.float32(32768.0)
        elif bits == 8:
            x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / np.float32(128.0)
        elif bits == 24:
            b = np.frombuffer(pcm[:len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16))
            v = np.where(v & 0x800000, v - 0x1000000, v)
            x = v.astype(np.float32) / np.float32(8388608.0)
        elif bits == 32:
            x = (np.frombuffer(pcm[:len(pcm) // 4 * 4], dtype='<i4').astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise ValueError('%s: unsupported PCM width %d' % (path, bits))
    elif tag == 3:
        x = np.frombuffer(pcm, dtype='<f4' if bits == 32 else '<f8').astype(np.float32)
    else:
        raise ValueError('%s: unsupported wav format tag %d' % (path, tag))
    if nch > 1:
        x = x[:len(x) // nch * nch].reshape(-1, nch).mean(axis=1).astype(np.float32)
    return x, sr


def read_wav_into(path, out_row):
    """Decode a mono PCM_16 file straight into ``out_row`` (float32 numpy row, e.g. a pinned staging buffer); zeros behind the signal.
    -> (samples written, sample_rate), or None when the file is anything else (the caller falls back to read_wav)."""
    if _native_decode is not None and out_row.flags['C_CONTIGUOUS'] and out_row.dtype == np.float32:
        # the library's C reader: the whole call runs outside the interpreter lock (loader threads decode in parallel)
        import ctypes
        n, sr = ctypes.c_longlong(0), ctypes.c_int(0)
        st = _native_decode(path.encode(), out_row.ctypes.data, out_row.shape[0], ctypes.byref(n), ctypes.byref(sr))
        return (int(n.value), int(sr.value)) if st == 0 else None
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 44 or data[0:4] != b'RIFF' or data[8:12] != b'WAVE':
        return None
    pos, fmt, off, size = 12, None, -1, 0
    while pos + 8 <= len(data):
        cid, csz = data[pos:pos + 4], struct.unpack_from('<I', data, pos + 4)[0]
        if cid == b'fmt ':
            fmt = struct.unpack_from('<HHIIHH', data, pos + 8)
        elif cid == b'data':
            off, size = pos + 8, min(csz, len(data) - pos - 8)
        pos += 8 + csz + (csz & 1)
    if fmt is None or off < 0 or fmt[0] != 1 or fmt[1] != 1 or fmt[5] != 16:
        return None
    n = min(size // 2, out_row.shape[0])
    np.multiply(np.frombuffer(data, dtype='<i2', count=n, offset=off), np.float32(1.0 / 32768.0), out=out_row[:n])   # == s / 32768 exactly (power of two)
    out_row[n:] = 0.0
    return n, fmt[2]


def load(path, sr=None):
    """librosa.load(path, sr=...) as the reference uses it: native rate only (every call site asserts 16 kHz)."""
    x, file_sr = read_wav(path)
    if sr is not None and sr != file_sr:
        raise ValueError('%s: sample rate %d, expected %d (resampling on load is not part of the path)' % (path, file_sr, sr))
    return x, file_sr


def pcm16_quantise(wav):
    """float -> int16 as libsndfile's PCM_16 writer without clipping control: rint(x * 32767), saturated."""
    q = np.rint(np.asarray(wav, dtype=np.float32) * np.float32(32767.0))
    return np.clip(q, -32768, 32767).astype('<i2')


def write_wav_pcm16_native(path, row, n, sr=fs, quantised=False):
    """The same file through the library's C writer (nele_wav_write_pcm16): ``row`` a contiguous float32 numpy row (e.g. of a pinned staging
    buffer), its first ``n`` samples.  Runs outside the interpreter lock - what the writer threads of inference.enhance_files call."""
    import ctypes
    st = _nele_lib.nele_wav_write_pcm16(path.encode(), ctypes.c_void_p(row.ctypes.data), int(n), int(sr), int(bool(quantised)))
    if st != 0:
        raise IOError(_nele_lib.nele_last_error_string().decode('utf-8', 'replace'))


def _c_paths(paths):
    import ctypes
    arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
    return arr


def read_wav_batch_pcm16(paths, out_rows, threads=8):
    """A batch of mono PCM_16 files -> the rows of ``out_rows`` ([n][cap] int16 numpy view of e.g. a pinned staging buffer), read by the
    library's own threads in ONE foreign call (no interpreter lock, no per-file Python).  -> (samples int32 [n] (-1: another wav flavour,
    -2: unreadable), sample rates int32 [n])."""
    import ctypes
    n = len(paths)
    assert out_rows.dtype == np.int16 and out_rows.ndim == 2 and out_rows.shape[0] >= n and out_rows.strides[1] == 2
    got, sr = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    st = _nele_lib.nele_wav_read_pcm16_batch(ctypes.cast(_c_paths(paths), ctypes.c_void_p), n, ctypes.c_void_p(out_rows.ctypes.data),
                                             out_rows.strides[0] // 2, out_rows.shape[1], ctypes.c_void_p(got.ctypes.data),
                                             ctypes.c_void_p(sr.ctypes.data), max(1, min(256, int(threads))))
    if st != 0:
        raise IOError(_nele_lib.nele_last_error_string().decode('utf-8', 'replace'))
    return got, sr


def probe_wav_batch_pcm16(paths, threads=8):
    """Sample counts of a batch of wav files from their RIFF headers in ONE foreign call (-1: not mono PCM_16, -2: unreadable) - what a
    loop over os.path.getsize tells a loader, without 2 x B stat() calls per batch in the interpreter."""
    import ctypes
    n = len(paths)
    got = np.zeros(n, dtype=np.int32)
    st = _nele_lib.nele_wav_probe_pcm16_batch(ctypes.cast(_c_paths(paths), ctypes.c_void_p), n, ctypes.c_void_p(got.ctypes.data), max(1, min(256, int(threads))))
    if st != 0:
        raise IOError(_nele_lib.nele_last_error_string().decode('utf-8', 'replace'))
    return got


def write_wav_batch_pcm16(paths, rows, n_samples, sr=fs, threads=8):
    """Rows of ``rows`` ([n][>= max n_samples] int16: the sample values themselves) -> n mono PCM_16 files, written by the library's own
    threads in one foreign call."""
    import ctypes
    n = len(paths)
    assert rows.dtype == np.int16 and rows.ndim == 2 and rows.shape[0] >= n and rows.strides[1] == 2
    ns = np.ascontiguousarray(np.asarray(n_samples, dtype=np.int32))
    st = _nele_lib.nele_wav_write_pcm16_batch(ctypes.cast(_c_paths(paths), ctypes.c_void_p), n, ctypes.c_void_p(rows.ctypes.data), rows.strides[0] // 2,
                                              ctypes.c_void_p(ns.ctypes.data), int(sr), max(1, min(256, int(threads))))
    if st != 0:
        raise IOError(_nele_lib.nele_last_error_string().decode('utf-8', 'replace'))


def write_wav_pcm16(path, wav, sr=fs, quantised=False):
    """sf.write(path, wav, sr, 'PCM_16').  ``quantised=True``: ``wav`` already went through the device-side PCM_16
    emulation (values k / 32768), so the samples are recovered exactly instead of being rounded a second time."""
    if hasattr(wav, 'detach'):
        wav = wav.detach().cpu().numpy()
    wav = np.asarray(wav, dtype=np.float32).reshape(-1)
    s = np.rint(wav * np.float32(32768.0)).clip(-32768, 32767).astype('<i2') if quantised else pcm16_quantise(wav)
    body = s.tobytes()
    hdr = b'RIFF' + struct.pack('<I', 36 + len(body)) + b'WAVE' + b'fmt ' + struct.pack('<IHHIIHH', 16, 1, 1, sr, sr * 2, 2, 16)
    with open(path, 'wb') as f:
        f.write(hdr + b'data' + struct.pack('<I', len(body)) + body)


# ------------------------------------------------------------------------------------------------ names and lists
def wave_name_of(enhanced_file):
    """audio_util.py:121-126: 'dir/name@12.wav' -> 'name', 'dir/name.wav' -> 'name'."""
    f = enhanced_file.split('/')[-1]
    return f.split('@')[0] if '@' in f else f[:-4]


def enhanced_name(directory, wave_name, gan_epoch):
    """train_nele.py:190-193, 309-312: '<directory>/<stem>@<epoch><ext>'."""
    return directory + '/' + wave_name[0:-4] + '@' + str(gan_epoch) + wave_name[-4:]


def List_concat(score, enhanced_list):
    """audio_util.py:367-371."""
    return [str(score[i]) + ',' + enhanced_list[i] for i in range(len(score))]


def List_concat_score(score, score2):
    return [str(score[i]) + ',' + str(score2[i]) for i in range(len(score))]


def List_concat_3scores(score1, score2, score3):
    return [str(score1[i]) + ',' + str(score2[i]) + ',' + str(score3[i]) for i in range(len(score1))]


def List_concat_5scores(score1, score2, score3, score4, score5):
    """audio_util.py:385-389: the 's0,s1,s2,s3,s4' prefix of a D training item."""
    return [','.join(str(s[i]) for s in (score1, score2, score3, score4, score5)) for i in range(len(score1))]


def parse_score_line(line):
    """dataloader.py:56-77: 's0,s1,s2,s3,s4,path' -> (intel targets [3], quality targets [2], path)."""
    p = line.split(',')
    if len(p) < 6:
        raise ValueError('D training item needs five scores and a path: %r' % line)
    return (np.asarray([float(p[0]), float(p[1]), float(p[2])], dtype=np.float32),
            np.asarray([float(p[3]), float(p[4])], dtype=np.float32), p[5])


def creatdir(directory):
    os.makedirs(directory, exist_ok=True)


def ListRead(filelist):
    with open(filelist, 'r') as f:
        return [line[0:-1] for line in f]


def get_filepaths(directory):
    out = []
    for root, _, files in os.walk(directory):
        out += [os.path.join(root, n) for n in files if '.wav' in n]
    return out


# ------------------------------------------------------------------------------------------------ metric fan-out from files
def _triple(clean_root, noise_root, enhanced_file, drc):
    name = enhanced_file.split('/')[-1] if drc else wave_name_of(enhanced_file) + '.wav'
    clean, sr = load(clean_root + name, sr=fs)
    assert sr == 16000
    enh, _ = load(enhanced_file, sr=fs)
    noise, _ = load(noise_root + name, sr=fs)
    n = min(len(clean), len(enh))                                          # audio_util.py:134-137
    return clean[:n], enh[:n] + noise[:n]


def pad_batch(signals):
    """list of float32 [L_i] -> (padded [n, Lmax] float32 with zeros behind every row's end, lengths int32 [n])."""
    lens = np.asarray([len(a) for a in signals], dtype=np.int32)
    out = np.zeros((len(signals), int(lens.max())), dtype=np.float32)
    for i, a in enumerate(signals):
        out[i, :len(a)] = a
    return out, lens


def _triples_on_device(clean_root, noise_root, enhanced_files, drc, threads=8):
    """The (clean, enhanced + noise) pairs of ``_triple`` for a group of files, built on the GPU: the 3 n files are read by ONE library
    call into pinned int16 rows, uploaded as int16, converted there (s / 32768, exact), rows cut to min(len(clean), len(enhanced))
    (audio_util.py:134-137) and the noise added in float32 as numpy does.  -> (x [n, L], y [n, L] device float32, lengths int32 [n]),
    or None when a file is not mono PCM_16 (the caller then reads the group file by file)."""
    import torch
    from . import _lib
    names = [en.split('/')[-1] if drc else wave_name_of(en) + '.wav' for en in enhanced_files]
    paths = [clean_root + nm for nm in names] + list(enhanced_files) + [noise_root + nm for nm in names]
    n = len(names)
    L = max(max(1, (os.path.getsize(p) - 44 + 1) // 2) for p in paths)
    L = (L + 7) // 8 * 8
    host = pinned_get((3 * n, L), torch.int16)
    try:
        got, sr = read_wav_batch_pcm16(paths, host.numpy(), threads)
        if (got == -2).any():
            raise IOError('cannot read ' + paths[int(np.argmax(got == -2))])
        if (got < 0).any():
            return None
        if (sr != fs).any():
            raise ValueError('%s: sample rate %d, expected %d' % (paths[int(np.argmax(sr != fs))], int(sr[np.argmax(sr != fs)]), fs))
        lens = np.minimum(got[:n], got[n:2 * n]).astype(np.int32)
        if (got[2 * n:] < lens).any():
            raise ValueError('%s: noise file shorter than the utterance' % paths[2 * n + int(np.argmax(got[2 * n:] < lens))])
        raw = host.cuda(non_blocking=True)
        dl = torch.from_numpy(np.tile(lens, 3)).cuda()
        f = torch.empty((3 * n, L), dtype=torch.float32, device='cuda')
        _lib.check(_lib.lib.nele_pcm16_to_float(raw.data_ptr(), L, dl.data_ptr(), 3 * n, L, f.data_ptr(), L, _lib.stream()), 'nele_pcm16_to_float')
        x, y = f[:n], f[n:2 * n] + f[2 * n:]
        torch.cuda.current_stream().synchronize()                          # the pinned rows go back to the pool below
        return x, y, lens
    finally:
        pinned_put(host)


def _read_batch(kind, clean_root, noise_root, enhanced_list, norm, drc=False, max_batch=256):
    """Files of ANY lengths go side by side into one padded batch with per-utterance lengths: one kernel launch per metric and
    ``max_batch`` files (the reference: one joblib process per file, audio_util.py:146).  Mono PCM_16 files (what the reference
    writes) never become float32 on the host: see _triples_on_device."""
    import torch
    from . import metrics as mt
    fn = {'estoi': mt.batch_estoi, 'siib': mt.batch_siib, 'haspi': mt.batch_haspi}[kind]
    out = [None] * len(enhanced_list)
    for k in range(0, len(enhanced_list), max_batch):
        sel = list(range(k, min(k + max_batch, len(enhanced_list))))
        dev = _triples_on_device(clean_root, noise_root, [enhanced_list[i] for i in sel], drc) if torch.cuda.is_available() else None
        if dev is not None:
            x, y, lens = dev
        else:
            pairs = [_triple(clean_root, noise_root, enhanced_list[i], drc) for i in sel]
            xp, lens = pad_batch([p[0] for p in pairs])
            yp, _ = pad_batch([p[1] for p in pairs])
            x, y = torch.from_numpy(xp).cuda(), torch.from_numpy(yp).cuda()
        raw, mapped = fn(x, y, lengths=torch.from_numpy(lens))[:2]
        vals = (mapped if norm else raw).double().cpu().numpy()
        for i, v in zip(sel, vals):
            if not np.isfinite(v):
                raise ValueError('%s: metric undefined for %s (the reference raises here)' % (kind, enhanced_list[i]))
            out[i] = float(v)
    return out


def read_batch_STOI(clean_root, noise_root, enhanced_list, norm=True):
    """audio_util.py:120-147.  One batched launch per group of equal-length files; values in list order."""
    return _read_batch('estoi', clean_root, noise_root, enhanced_list, norm)


def read_batch_SIIB(clean_root, noise_root, enhanced_list, norm=True):
    """audio_util.py:149-175."""
    return _read_batch('siib', clean_root, noise_root, enhanced_list, norm)


def read_batch_HASPI(clean_root, noise_root, enhanced_list, norm=True):
    """audio_util.py:177-203."""
    return _read_batch('haspi', clean_root, noise_root, enhanced_list, norm)


def read_batch_STOI_DRC(clean_root, noise_root, enhanced_list):
    """audio_util.py:267-284 (pre-enhanced examples: the enhanced file carries the clean file's name)."""
    return _read_batch('estoi', clean_root, noise_root, enhanced_list, True, drc=True)


def read_batch_SIIB_DRC(clean_root, noise_root, enhanced_list):
    """audio_util.py:286-303."""
    return _read_batch('siib', clean_root, noise_root, enhanced_list, True, drc=True)


def read_batch_HASPI_DRC(clean_root, noise_root, enhanced_list):
    """audio_util.py:305-322."""
    return _read_batch('haspi', clean_root, noise_root, enhanced_list, True, drc=True)


# PESQ / ViSQOL fan-out (audio_util.py:205-265, 323-364): external programs, see quality.py
from .quality import (read_PESQ, read_batch_PESQ, read_batch_VISQOL, read_PESQ_DRC, read_batch_PESQ_DRC,   # noqa: E402,F401
                      read_batch_VISQOL_DRC)


# ------------------------------------------------------------------------------------------------ pinned staging buffers
# Page-locked host buffers cost milliseconds to allocate (tens for a 64 MB batch): they are pooled per shape for the life of the process and
# shared by every loader / writer (FileBatches, inference.enhance_files).
_PINNED = {}
_PINNED_LOCK = threading.Lock()
_PINNED_BYTES = 0
PINNED_POOL_MAX_BYTES = 16 << 30     # page-locked host memory kept for reuse; buffers returned beyond it are freed (a cap of 2 GB evicted the staging buffers of the streamed file path between its passes: 42 k -> 17 k utterances/s)


def pinned_get(shape, dtype=None):
    """A page-locked host buffer of this shape from the pool (hipHostMalloc per batch costs milliseconds), or a new one.  The pool is shared
    by the loader threads, the background writer thread and the main thread: one lock around the check-and-pop."""
    global _PINNED_BYTES
    import torch
    dtype = torch.float32 if dtype is None else dtype
    with _PINNED_LOCK:
        lst = _PINNED.get((tuple(shape), dtype))
        if lst:
            t = lst.pop()
            _PINNED_BYTES -= t.numel() * t.element_size()
            return t
    return torch.empty(tuple(shape), dtype=dtype).pin_memory()


def pinned_put(t):
    """Return a buffer to the pool; beyond PINNED_POOL_MAX_BYTES the oldest pooled buffers are dropped first (shapes that no longer occur -
    padded lengths vary from batch to batch - must not pin host memory for good)."""
    global _PINNED_BYTES
    if t is None:
        return
    nb = t.numel() * t.element_size()
    with _PINNED_LOCK:
        _PINNED.setdefault((tuple(t.shape), t.dtype), []).append(t)
        _PINNED_BYTES += nb
        if _PINNED_BYTES > PINNED_POOL_MAX_BYTES:
            for key in list(_PINNED.keys()):                # insertion order: the shapes seen first
                lst = _PINNED[key]
                while lst and _PINNED_BYTES > PINNED_POOL_MAX_BYTES and not (key == (tuple(t.shape), t.dtype) and len(lst) == 1):
                    d = lst.pop(0)
                    _PINNED_BYTES -= d.numel() * d.element_size()
                if not lst:
                    del _PINNED[key]
                if _PINNED_BYTES <= PINNED_POOL_MAX_BYTES:
                    break


def pinned_release():
    """Drop the pooled buffers (they are re-created on demand)."""
    global _PINNED_BYTES
    with _PINNED_LOCK:
        _PINNED.clear()
        _PINNED_BYTES = 0


# ------------------------------------------------------------------------------------------------ batches from files, prefetched
class FileBatches:
    """A corpus on disk as the sequence of batch dicts GanTrainer.run_epoch takes ({'clean', 'noise', 'lengths', 'names'[, 'drc',
    'drc_lengths']}, device tensors, files of any lengths padded side by side).  The reference feeds its loop from 8 DataLoader worker
    processes (dataloader.py:86-98) and re-reads the wav files in every stage; here wav decoding runs on ``workers`` threads
    (numpy's frombuffer / astype release the GIL), ``ahead`` batches are decoded beyond the one being asked for, and each batch is staged
    through pinned host memory with an asynchronous copy on its own stream (the consumer's stream waits for that copy only).
    ``seq[i]`` may be asked for repeatedly and in any order (run_epoch walks the list once for the G-steps and once for the sample
    generation): a batch is decoded again when it is no longer cached, like the reference."""

    def __init__(self, file_list, noise_path, batch=32, drc_path=None, workers=8, ahead=2, device='cuda', pad_to=4096, keep=4, int16=True):
        """``int16`` (default): a batch is read by ONE library call (nele_wav_read_pcm16_batch, ``workers`` library threads) into pinned int16
        rows, uploaded as int16 and converted on the device (nele_pcm16_to_float: s / 32768, the padding and the samples behind the shorter
        of clean / noise zeroed there).  A batch that holds any other wav flavour - and every batch with ``int16=False`` - takes the
        per-file float32 path (read_wav_into / load on ``workers`` Python threads)."""
        import concurrent.futures as cf
        self.int16 = bool(int16)
        self.files, self.noise_path, self.drc_path = list(file_list), noise_path, drc_path
        self.names = [f.split('/')[-1] for f in self.files]
        self.groups = [list(range(k, min(k + batch, len(self.files)))) for k in range(0, len(self.files), batch)]
        self.workers = max(1, int(workers))
        self.pool = cf.ThreadPoolExecutor(max_workers=self.workers)
        self.ahead, self.pad_to, self.keep, self.device = int(ahead), int(pad_to), int(keep), device
        self._pending, self._ready = {}, {}
        self._bounds = {}
        self._copy = None
        self.decoded_files = 0

    def __len__(self):
        return len(self.groups)

    def _keys(self, g):
        """One hashable per utterance of group g for the trainer's caches (GanTrainer.enable_clean_cache): the clean FILE (not its base name:
        Train/Clean/x.wav and Test/Clean/x.wav are different utterances) with the noise and pre-enhanced folders it is paired with - the
        cached features and D items of an utterance depend on those files too."""
        return [(self.files[i], self.noise_path, self.drc_path) for i in self.groups[g]]

    def _bound(self, path):
        """upper bound of a file's samples from its size (exact for the 44-byte-header PCM_16 files the reference writes)"""
        b = self._bounds.get(path)
        if b is None:
            b = self._bounds[path] = max(1, (os.path.getsize(path) - 44 + 1) // 2)
        return b

    def _decode_into(self, idx, rows):
        """decode file idx into its rows of the group's pinned buffers -> (samples clean/noise, samples drc, name)"""
        name = self.files[idx].split('/')[-1]
        srcs = [self.files[idx], self.noise_path + name] + ([self.drc_path + name] if self.drc_path is not None else [])
        got = []
        for path, row in zip(srcs, rows):
            r = read_wav_into(path, row)
            if r is None:                                                  # not plain mono PCM_16: the general reader
                x, sr = load(path)
                n = min(len(x), row.shape[0])
                row[:n] = x[:n]
                row[n:] = 0.0
                r = (n, sr)
            assert r[1] == 16000                                           # dataloader.py:35
            got.append(r[0])
        m = min(got[0], got[1])                                            # the noise file is cut to the clean file's length and vice versa
        rows[0][m:] = 0.0
        rows[1][m:] = 0.0
        return m, (got[2] if len(got) > 2 else None), name

    def _pinned(self, shape):
        """pinned staging buffer from a small pool (cudaHostAlloc per batch costs milliseconds); a buffer returns to the pool when the
        batch that used it leaves the cache"""
        return pinned_get(shape)

    def _submit(self, g, force_f32=False):
        if not (0 <= g < len(self.groups)) or g in self._pending or g in self._ready:
            return
        idxs = self.groups[g]
        names = [self.names[i] for i in idxs]
        if self.int16 and not force_f32:
            # the files' sample counts in one library call (round 6: 2 x B stat() calls per batch were 0.8 of the ~3.1 ms of interpreter time
            # that bound the streamed file path - tools/files_sweep.py: the same 41 k utterances/s with 4, 8 or 16 reader threads, with or
            # without writing)
            need = [p_ for p_ in [self.files[i] for i in idxs] + [self.noise_path + nm for nm in names] +
                    ([self.drc_path + nm for nm in names] if self.drc_path is not None else []) if p_ not in self._bounds]
            if need:
                for p_, n_ in zip(need, probe_wav_batch_pcm16(need, self.workers)):
                    if n_ > 0:
                        self._bounds[p_] = int(n_)
        Lmax = max(max(self._bound(self.files[i]), self._bound(self.noise_path + nm)) for i, nm in zip(idxs, names))
        if self.pad_to:
            Lmax = (Lmax + self.pad_to - 1) // self.pad_to * self.pad_to
        n = len(idxs)
        Ld = 0
        if self.drc_path is not None:
            Ld = max(self._bound(self.drc_path + nm) for nm in names)
            if self.pad_to:
                Ld = (Ld + self.pad_to - 1) // self.pad_to * self.pad_to
            Ld = max(Ld, Lmax)
        if self.int16 and not force_f32:
            import torch
            hc, hn = pinned_get((n, Lmax), torch.int16), pinned_get((n, Lmax), torch.int16)
            hd = pinned_get((n, Ld), torch.int16) if self.drc_path is not None else None
            sets = [([self.files[i] for i in idxs], hc), ([self.noise_path + nm for nm in names], hn)]
            if hd is not None:
                sets.append(([self.drc_path + nm for nm in names], hd))

            def run_batch():                                               # one task per batch: the library's threads do the rest
                out = []
                for paths, h in sets:
                    got, sr = read_wav_batch_pcm16(paths, h.numpy(), self.workers)
                    if (got == -2).any():
                        raise IOError('cannot read ' + paths[int(np.argmax(got == -2))])
                    if (got < 0).any():
                        return None                                        # another wav flavour in the batch: the float32 path
                    assert (sr == 16000).all()                             # dataloader.py:35
                    out.append(got)
                return out
            self._pending[g] = ('i16', self.pool.submit(run_batch), hc, hn, hd, names)
            return
        hc, hn = self._pinned((n, Lmax)), self._pinned((n, Lmax))
        hd = self._pinned((n, Ld)) if self.drc_path is not None else None
        ac, an, ad = hc.numpy(), hn.numpy(), (hd.numpy() if hd is not None else None)
        # a task = a run of rows (submitting one task per file costs the submitting thread ~20 us each: more than the decode itself)
        nt = max(1, min(self.workers, n))
        per = (n + nt - 1) // nt

        def run(r0):
            return [self._decode_into(idxs[r], [ac[r], an[r]] + ([ad[r]] if ad is not None else [])) for r in range(r0, min(n, r0 + per))]
        futs = [self.pool.submit(run, r0) for r0 in range(0, n, per)]
        self._pending[g] = ('f32', futs, hc, hn, hd, names)

    def _stage_i16(self, g, got, hc, hn, hd, names):
        """int16 staging rows -> device: upload, then s / 32768 with the rows cut to the shorter of clean / noise, all on the copy stream"""
        import torch
        from . import _lib
        self.decoded_files += len(names)
        lens = np.minimum(got[0], got[1]).astype(np.int32)
        n, L = hc.shape
        if self._copy is None:
            self._copy = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._copy):
            sh = self._copy.cuda_stream

            dls, small = {}, []

            def up(h, ln):
                raw = h.to(self.device, non_blocking=True)
                dl = dls.get(id(ln))
                if dl is None:                                             # (clean and noise rows share their lengths: one upload, through a POOLED
                    hl = pinned_get((len(ln),), torch.int32)               #  page-locked buffer: .pin_memory() allocated one per call, and a copy from
                    hl.numpy()[:] = ln                                     #  pageable memory would make this thread wait for the stream's 65 MB uploads)
                    small.append(hl)
                    dl = dls[id(ln)] = hl.to(self.device, non_blocking=True)
                out = torch.empty(tuple(h.shape), dtype=torch.float32, device=self.device)
                _lib.check(_lib.lib.nele_pcm16_to_float(raw.data_ptr(), h.shape[1], dl.data_ptr(), h.shape[0], h.shape[1], out.data_ptr(), h.shape[1], sh),
                           'nele_pcm16_to_float')       # (not _lib.call: never part of a recorded pass)
                return out, dl
            clean, dl = up(hc, lens)
            noise, _ = up(hn, lens)
            b = {'clean': clean, 'noise': noise, 'lengths': dl, 'names': list(names), 'lengths_host': lens, 'keys': self._keys(g)}
            if hd is not None:
                dlens = got[2].astype(np.int32)
                b['drc'], b['drc_lengths'] = up(hd, dlens)
                b['drc_lengths_host'] = dlens
            ev = torch.cuda.Event()
            ev.record(self._copy)
        self._ready[g] = (b, ev, (hc, hn, hd) + tuple(small))

    def _stage(self, g):
        import torch
        kind, futs, hc, hn, hd, names = self._pending.pop(g)
        if kind == 'i16':
            got = futs.result()
            if got is not None:
                self._stage_i16(g, got, hc, hn, hd, names)
                self._trim()
                return
            for t in (hc, hn, hd):                                        # not all plain PCM_16: this batch again through the general readers
                pinned_put(t)
            self._submit(g, force_f32=True)
            kind, futs, hc, hn, hd, names = self._pending.pop(g)
        res = [r for f in futs for r in f.result()]
        self.decoded_files += len(res)
        lens = np.asarray([r[0] for r in res], dtype=np.int32)
        if self._copy is None:
            self._copy = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._copy):
            b = {'clean': hc.to(self.device, non_blocking=True), 'noise': hn.to(self.device, non_blocking=True),
                 'lengths': torch.from_numpy(lens).pin_memory().to(self.device, non_blocking=True), 'names': [r[2] for r in res],
                 'lengths_host': lens, 'keys': self._keys(g)}
            if hd is not None:
                dlens = np.asarray([r[1] for r in res], dtype=np.int32)
                b['drc'] = hd.to(self.device, non_blocking=True)
                b['drc_lengths'] = torch.from_numpy(dlens).pin_memory().to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy)
        self._ready[g] = (b, ev, (hc, hn, hd))
        self._trim()

    def _trim(self):
        while len(self._ready) > self.keep:
            _, ev_old, bufs = self._ready.pop(next(iter(self._ready)))
            ev_old.synchronize()                                           # its upload is long done; the pinned buffers go back to the pool
            for t in bufs:
                pinned_put(t)

    def __getitem__(self, g):
        import torch
        if g < 0:
            g += len(self.groups)
        if not 0 <= g < len(self.groups):
            raise IndexError(g)
        for k in range(g, g + 1 + self.ahead):
            self._submit(k)
        if g not in self._ready:
            self._stage(g)
        b, ev, _ = self._ready[g]
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for t in b.values():                                               # allocated on the copy stream, consumed on this one
            if isinstance(t, torch.Tensor):
                t.record_stream(cur)
        return b

    def __iter__(self):
        for g in range(len(self.groups)):
            yield self[g]

    def close(self):
        """Stop the decode threads and hand the staging buffers back to the process-wide pool."""
        self.pool.shutdown(wait=True)
        import torch
        torch.cuda.synchronize()                                           # uploads still in flight read the pinned buffers
        for _, _, bufs in self._ready.values():
            for t in bufs:
                pinned_put(t)
        for _, _, hc, hn, hd, _ in self._pending.values():
            for t in (hc, hn, hd):
                pinned_put(t)
        self._ready, self._pending = {}, {}


# ------------------------------------------------------------------------------------------------ datasets (dataloader.py)
class Generator_train_dataset:
    """dataloader.py:19-42: item = (clean_band [T,64], clean_mag [257,T], clean_phase [257,T], noise_band, noise_mag,
    noise_phase, target_score [3], target_qua [2], filename); features come from the GPU kernels (device tensors)."""

    def __init__(self, file_list, noise_path, rir_path=None):
        self.file_list, self.noise_path, self.rir_path = file_list, noise_path, rir_path
        self.target_score = np.asarray([1.0, 1.0, 1.0], dtype=np.float32)
        self.target_qua = np.asarray([1.0, 1.0], dtype=np.float32)

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, idx):
        from . import audio_util as au
        filename = self.file_list[idx].split('/')[-1]
        clean_wav, sr = load(self.file_list[idx])
        assert sr == 16000
        noise_wav, sr = load(self.noise_path + filename)
        assert sr == 16000
        cb, cm, cp = au.Sp_and_phase_Speech(clean_wav, power=power_law, Normalization=True)
        nb, nm, np_ = au.Sp_and_phase_Noise(noise_wav, power=power_law, Normalization=True)
        return cb, cm, cp, nb, nm, np_, self.target_score, self.target_qua, filename


class Discriminator_train_dataset:
    """dataloader.py:44-84: item = ([enh, noise, clean] [3,64,T], [enh, clean] [2,64,T], True_score [3], True_score_Qua [2])."""

    def __init__(self, file_list, noise_path, clean_path, rir_path=None):
        self.file_list, self.noise_path, self.clean_path, self.rir_path = file_list, noise_path, clean_path, rir_path

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, idx):
        import torch
        from . import audio_util as au
        score, score_qua, path = parse_score_line(self.file_list[idx])
        enh, sr = load(path)
        assert sr == 16000
        f = self.file_list[idx].split('/')[-1]
        if '@' in f:
            f = f.split('@')[0] + '.wav'
        noise, sr = load(self.noise_path + f)
        assert sr == 16000
        clean, sr = load(self.clean_path + f)
        assert sr == 16000
        eb = au.Sp_and_phase_Speech(enh, power=power_law)[0].transpose(0, 1)
        nb = au.Sp_and_phase_Noise(noise, power=power_law)[0].transpose(0, 1)
        cb = au.Sp_and_phase_Speech(clean, power=power_law)[0].transpose(0, 1)
        return torch.stack((eb, nb, cb)), torch.stack((eb, cb)), score, score_qua


class _Loader:
    """batch_size = 1, shuffle = True, drop_last = True (dataloader.py:86-101).  No worker processes: the features are
    computed by the GPU kernels, the host only decodes wav files."""

    def __init__(self, dataset, seed=None):
        self.dataset, self.rng = dataset, np.random.RandomState(seed)

    def __len__(self):
        return len(self.dataset)

    def __iter__(self):
        import torch
        for i in self.rng.permutation(len(self.dataset)):
            item = self.dataset[int(i)]
            yield tuple(v.unsqueeze(0) if isinstance(v, torch.Tensor) else
                        (torch.from_numpy(v).unsqueeze(0) if isinstance(v, np.ndarray) else [v]) for v in item)


def create_dataloader(filelist, noise_path, clean_path=None, rir_path=None, loader='G', seed=None):
    if loader == 'G':
        return _Loader(Generator_train_dataset(filelist, noise_path, rir_path), seed)
    if loader == 'D':
        return _Loader(Discriminator_train_dataset(filelist, noise_path, clean_path, rir_path), seed)
    raise Exception("No such dataloader type!")
