"""PESQ / ViSQOL targets of Discriminator_Quality: the reference's call sites with the two EXTERNAL programs plugged in.

The reference scores every generated and every pre-enhanced training example with two programs that are not part of it:
``pesq(ref, deg, fs)`` of the ``pypesq`` C extension (intel.py:14, 142-154) and the ViSQOL command-line binary in batch mode
(audio_util.py:228-247: a CSV of reference,degraded paths in, a CSV with a ``moslqo`` column out).  Neither is in this build's image and
neither is restated here - an own PESQ or ViSQOL could not be pinned against the programs the reference calls, and a training target
that only resembles them would be a different model.  What this module holds is everything AROUND them, so that a user who has both
gets the reference's D_Qua training with two lines (INTEGRATION.md section 2b'''):

* the logistic maps of the raw scores (intel.py:156-160, audio_util.py:253-256),
* the file fan-out with the reference's names and conventions (``read_batch_PESQ`` / ``read_batch_VISQOL`` and their ``_DRC`` forms:
  '<name>@<epoch>.wav' -> clean file '<name>.wav', both cut to the shorter one for PESQ, paths handed over untouched for ViSQOL),
* ``Scorer``: the same two scores for a BATCH IN MEMORY (what ``GanTrainer.run_epoch`` has: the clean batch and the PCM_16-rounded
  generated batch) - PESQ on the rows, ViSQOL through wav files written for the call.

Nothing here runs on the GPU: both programs are host programs, as in the reference; the trainer calls ``Scorer`` on a background thread
while the next batch's kernels are enqueued (train_nele.py:323-324, 336-337, 216-222 in ``run_epoch``).
A call without a registered program raises - there is no stand-in score.
"""
import concurrent.futures as cf
import csv
import os
import shutil
import subprocess
import tempfile

import numpy as np

fs = 16000

_BACKENDS = {'pesq': None, 'visqol': None, 'pesq_batch': None}
PESQ_WORKERS = 32                     # audio_util.py:224: Parallel(n_jobs=32)


class QualityBackendMissing(RuntimeError):
    pass


def set_backends(pesq=None, visqol=None, pesq_batch=None):
    """Register the external programs (process-wide, like the reference's module-level imports).

    pesq(ref, deg, fs) -> float: raw PESQ of one pair of equally long float arrays - ``pypesq.pesq`` has this signature (intel.py:14,147).
    pesq_batch(refs, degs, fs) -> sequence of floats (optional): all pairs of a call at once, for a caller that has its own process
    pool; without it the pairs are scored by ``PESQ_WORKERS`` threads (a C extension that releases the interpreter lock scales, one that
    does not runs serially - the reference uses 32 joblib PROCESSES, which a process that has initialised the GPU must not fork).
    visqol(pairs) -> sequence of floats: MOS-LQO of every (reference_path, degraded_path) pair, in order - ``visqol_program`` builds
    one from the binary's path as the reference invokes it.  ``None`` leaves an entry as it is; ``clear_backends()`` removes all."""
    if pesq is not None:
        _BACKENDS['pesq'] = pesq
    if pesq_batch is not None:
        _BACKENDS['pesq_batch'] = pesq_batch
    if visqol is not None:
        _BACKENDS['visqol'] = visqol


def add_cli_arguments(ap):
    """--pesq / --visqol / --visqol-model for the two module entry points (train_nele, inference)."""
    ap.add_argument('--pesq', default=None, metavar='MODULE:FUNCTION', help='pesq(ref, deg, fs) of an installed PESQ package, e.g. pypesq:pesq (intel.py:14)')
    ap.add_argument('--visqol', default=None, metavar='PROGRAM', help='the ViSQOL binary (audio_util.py:230)')
    ap.add_argument('--visqol-model', default=None, metavar='FILE', help='its --similarity_to_quality_model file (audio_util.py:231)')


def backends_from_cli(a):
    """Register what --pesq / --visqol name; -> True when both programs are there afterwards."""
    if getattr(a, 'pesq', None):
        import importlib
        mod, _, fn = a.pesq.partition(':')
        set_backends(pesq=getattr(importlib.import_module(mod), fn or 'pesq'))
    if getattr(a, 'visqol', None):
        if not getattr(a, 'visqol_model', None):
            raise SystemExit('--visqol needs --visqol-model')
        set_backends(visqol=visqol_program(a.visqol, a.visqol_model))
    return _BACKENDS['visqol'] is not None and (_BACKENDS['pesq'] is not None or _BACKENDS['pesq_batch'] is not None)


def clear_backends():
    for k in _BACKENDS:
        _BACKENDS[k] = None


def _need(kind):
    b = _BACKENDS[kind]
    if b is None and not (kind == 'pesq' and _BACKENDS['pesq_batch'] is not None):
        raise QualityBackendMissing(
            "nele_gan_amd.quality: no %s program registered.  PESQ (pypesq) and ViSQOL are external programs the reference calls "
            "(intel.py:147, audio_util.py:228-247); they are not part of this build.  quality.set_backends(pesq=pypesq.pesq, "
            "visqol=quality.visqol_program('/path/to/visqol', '/path/to/libsvm_nu_svr_model.txt'))" % ('PESQ' if kind == 'pesq' else 'ViSQOL'))
    return b


# ------------------------------------------------------------------------------------------------ maps
def mapping_PESQ_harvard(x):
    """intel.py:156-160: 1 / (1 + exp(-1.5 (x - 2.5)))."""
    a, b = -1.5, 2.5
    return 1 / (1 + np.exp(a * (x - b)))


def mapping_VISQOL(x):
    """audio_util.py:253-256 (and :359-362): 1 / (1 + exp(-2.5 (x - 2.2)))."""
    a, b = -2.5, 2.2
    return 1 / (1 + np.exp(a * (x - b)))


# ------------------------------------------------------------------------------------------------ PESQ
def PESQ_Wrapper_raw_harvard(ref, deg, fs_=fs):
    """intel.py:142-147."""
    n = min(len(ref), len(deg))
    return float(_pesq_pairs([ref[:n]], [deg[:n]], fs_)[0])


def PESQ_Wrapper_harvard(ref, deg, fs_=fs):
    """intel.py:149-154."""
    return float(mapping_PESQ_harvard(PESQ_Wrapper_raw_harvard(ref, deg, fs_)))


def _pesq_pairs(refs, degs, fs_):
    _need('pesq')
    if _BACKENDS['pesq_batch'] is not None:
        out = list(_BACKENDS['pesq_batch'](refs, degs, fs_))
    elif len(refs) == 1:
        out = [_BACKENDS['pesq'](refs[0], degs[0], fs_)]
    else:
        fn = _BACKENDS['pesq']
        with cf.ThreadPoolExecutor(max_workers=min(PESQ_WORKERS, len(refs))) as pool:
            out = list(pool.map(lambda p: fn(p[0], p[1], fs_), zip(refs, degs)))
    if len(out) != len(refs):
        raise ValueError('PESQ program returned %d scores for %d pairs' % (len(out), len(refs)))
    return [float(v) for v in out]


def _clean_name(enhanced_file, drc):
    from . import dataio
    return enhanced_file.split('/')[-1] if drc else dataio.wave_name_of(enhanced_file) + '.wav'


def _pesq_files(clean_root, enhanced_list, drc):
    from . import dataio
    refs, degs = [], []
    for en in enhanced_list:
        clean, sr = dataio.load(clean_root + _clean_name(en, drc), sr=fs)
        assert sr == 16000                                                  # audio_util.py:214
        enh, _ = dataio.load(en, sr=fs)
        n = min(len(clean), len(enh))                                       # :216-218
        refs.append(clean[:n])
        degs.append(enh[:n])
    return _pesq_pairs(refs, degs, fs)


def read_PESQ(clean_root, enhanced_file, norm):
    """audio_util.py:205-222."""
    v = _pesq_files(clean_root, [enhanced_file], False)[0]
    return float(mapping_PESQ_harvard(v)) if norm else v


def read_batch_PESQ(clean_root, enhanced_list, norm=True):
    """audio_util.py:224-226: scores in list order."""
    v = _pesq_files(clean_root, list(enhanced_list), False)
    return [float(mapping_PESQ_harvard(x)) for x in v] if norm else v


def read_PESQ_DRC(clean_root, enhanced_file):
    """audio_util.py:323-335 (pre-enhanced examples carry the clean file's name; always mapped)."""
    return float(mapping_PESQ_harvard(_pesq_files(clean_root, [enhanced_file], True)[0]))


def read_batch_PESQ_DRC(clean_root, enhanced_list):
    """audio_util.py:337-339."""
    return [float(mapping_PESQ_harvard(x)) for x in _pesq_files(clean_root, list(enhanced_list), True)]


# ------------------------------------------------------------------------------------------------ ViSQOL
def visqol_program(program, model_path, extra_args=()):
    """The reference's invocation of the ViSQOL binary (audio_util.py:228-247, 341-356) as a ``visqol(pairs)`` backend:
    ``<program> --use_speech_mode --similarity_to_quality_model <model> --batch_input_csv <in> --results_csv <out>`` on a CSV of
    'reference,degraded' rows; the ``moslqo`` column of the result, in row order.  The CSVs live in a directory of their own per call
    (the reference names them by wall-clock second in /tmp: two calls in one second collide)."""
    def run(pairs):
        pairs = list(pairs)
        if not pairs:
            return []
        work = tempfile.mkdtemp(prefix='nele-visqol-')
        try:
            inp, res = os.path.join(work, 'input.csv'), os.path.join(work, 'result.csv')
            with open(inp, 'w') as f:
                f.write('reference,degraded\n')                            # :234
                for r, d in pairs:
                    f.write(r + ',' + d + '\n')
            cmd = [program, '--use_speech_mode', '--similarity_to_quality_model', model_path, '--batch_input_csv', inp, '--results_csv', res]
            ret = subprocess.run(cmd + list(extra_args), stdout=subprocess.DEVNULL)
            if ret.returncode != 0:                                         # :247 `assert ret==0`
                raise RuntimeError('ViSQOL exited with %d: %s' % (ret.returncode, ' '.join(cmd)))
            with open(res, newline='') as f:
                rows = list(csv.DictReader(f))
            out = [float(r['moslqo']) for r in rows]
            if len(out) != len(pairs):                                      # :250
                raise RuntimeError('ViSQOL returned %d rows for %d pairs' % (len(out), len(pairs)))
            return out
        finally:
            shutil.rmtree(work, ignore_errors=True)
    return run


def _visqol_files(clean_root, enhanced_list, drc):
    fn = _need('visqol')
    pairs = [(clean_root + _clean_name(en, drc), en) for en in enhanced_list]
    out = [float(v) for v in fn(pairs)]
    if len(out) != len(pairs):
        raise ValueError('ViSQOL program returned %d scores for %d pairs' % (len(out), len(pairs)))
    return out


def read_batch_VISQOL(clean_root, enhanced_list, norm=True):
    """audio_util.py:228-265: the clean and the enhanced PATHS go to the program as they are (it aligns and trims itself)."""
    v = _visqol_files(clean_root, list(enhanced_list), False)
    return [float(mapping_VISQOL(x)) for x in v] if norm else v


def read_batch_VISQOL_DRC(clean_root, enhanced_list):
    """audio_util.py:341-364."""
    return [float(mapping_VISQOL(x)) for x in _visqol_files(clean_root, list(enhanced_list), True)]


# ------------------------------------------------------------------------------------------------ a batch in memory
class Scorer:
    """Raw (PESQ, ViSQOL MOS-LQO) of every row of a batch that is in memory - what the reference gets by writing the generated batch
    to disk and calling read_batch_PESQ / read_batch_VISQOL on the files (train_nele.py:309-324).

    ``raw(refs, degs, fs)``: lists of float32 arrays, row k of both already cut to the samples the reference compares (the enhanced
    file's 256 * (L // 256) samples - what ``librosa.load`` returns for the PCM_16 file - and the clean file cut to them, audio_util.py:216-218)
    -> float64 [n, 2].  PESQ sees the arrays; ViSQOL sees PCM_16 wav files of the same samples written for the call (the clean rows come
    from PCM_16 files, the generated rows are PCM_16-rounded already: both are written without loss) unless ``files`` - the (reference,
    degraded) paths of files that are on disk already - is given.
    ``mapped(...)``: the D_Qua training targets [n, 2] (mapping_PESQ_harvard, mapping_VISQOL).
    ``use``: ('pesq', 'visqol') by default; a subset leaves the other column at 0 raw / its map of 0 (the reference has no such mode)."""

    def __init__(self, use=('pesq', 'visqol'), tmp_root=None):
        self.use = tuple(use)
        for u in self.use:
            if u not in ('pesq', 'visqol'):
                raise ValueError("Scorer: use must name 'pesq' and / or 'visqol'")
        self.tmp_root = tmp_root
        self.calls = 0
        self.pairs = 0

    def raw(self, refs, degs, fs_=fs, files=None):
        from . import dataio
        n = len(refs)
        if len(degs) != n:
            raise ValueError('Scorer.raw: %d reference rows, %d degraded rows' % (n, len(degs)))
        out = np.zeros((n, 2), dtype=np.float64)
        if n == 0:
            return out
        for u in self.use:                                                  # fail before any work if a program is missing
            _need(u)
        self.calls += 1
        self.pairs += n
        if 'pesq' in self.use:
            cut = [min(len(r), len(d)) for r, d in zip(refs, degs)]
            out[:, 0] = _pesq_pairs([np.asarray(r[:c], dtype=np.float32) for r, c in zip(refs, cut)],
                                    [np.asarray(d[:c], dtype=np.float32) for d, c in zip(degs, cut)], fs_)
        if 'visqol' in self.use:
            fn = _need('visqol')
            if files is not None:
                out[:, 1] = [float(v) for v in fn(list(files))]
            else:
                work = tempfile.mkdtemp(prefix='nele-quality-', dir=self.tmp_root)
                try:
                    pairs = []
                    for k, (r, d) in enumerate(zip(refs, degs)):
                        rp, dp = os.path.join(work, 'ref%05d.wav' % k), os.path.join(work, 'deg%05d.wav' % k)
                        # (rows on the PCM_16 grid k / 32768 - read from such files, or PCM_16-rounded on the device - are written exactly)
                        dataio.write_wav_pcm16(rp, np.asarray(r, dtype=np.float32), fs_, quantised=True)
                        dataio.write_wav_pcm16(dp, np.asarray(d, dtype=np.float32), fs_, quantised=True)
                        pairs.append((rp, dp))
                    v = [float(x) for x in fn(pairs)]
                    if len(v) != n:
                        raise ValueError('ViSQOL program returned %d scores for %d pairs' % (len(v), n))
                    out[:, 1] = v
                finally:
                    shutil.rmtree(work, ignore_errors=True)
        return out

    def mapped(self, refs, degs, fs_=fs, files=None):
        return self.map(self.raw(refs, degs, fs_, files))

    @staticmethod
    def map(raw):
        raw = np.asarray(raw, dtype=np.float64)
        return np.stack([mapping_PESQ_harvard(raw[:, 0]), mapping_VISQOL(raw[:, 1])], axis=1)
