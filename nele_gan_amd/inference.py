"""Enhancement path of the reference ``inference.py`` (inference.py:79-117), batched on the GPU:
features -> G (eval) -> mask * beta2 -> resynthesis -> enh / rms(enh) * 0.03 -> PCM_16.

``Enhancer.enhance``         one padded batch resident in HBM (optional per-utterance lengths), on the current stream
``Enhancer.enhance_stream``  a sequence of batches with several of them in flight on their own streams: the noise branch of a batch
                             (STFT -> IMCRA, a scan that is serial over frames and fills a fraction of the chip) runs under the
                             generator of the batch before it; results come out in order, bit-identical to ``enhance``
``enhance_files``            the reference's loop over a file list (inference.py:79-117): files of any lengths are decoded on a
                             small thread pool, padded side by side in batches, staged through pinned host buffers, enhanced with
                             ``enhance_stream`` and written as '<name>@1.wav' PCM_16; with torch.distributed initialised every
                             rank takes a contiguous shard of the list (BASELINE configs[4]: data-parallel enhancement, pure
                             replicas, no collective).
"""
import collections
import os

import torch

from . import audio_util as au
from . import dist as ndist
from . import model as M
from . import ops
from . import _lib

p_power = (1 / 6)
inv_p = 6
fs = 16000


class Enhancer:
    def __init__(self, chkpt_path=None, device='cuda', G=None):
        self.device = M._norm_dev(device)
        self.G = G if G is not None else M.Generator_Conv1D_cLN()
        if chkpt_path is not None:
            self.G.load_state_dict(torch.load(chkpt_path, map_location='cpu')['enhance-model'])   # inference.py:71-72
        self.G = self.G.to(self.device)
        self.G.eval()
        # An Enhancer that owns its generator (loaded from a checkpoint, never trained through this object) writes the weight layouts once and
        # keeps them: plain enhance() then skips the three weight-layout launches per batch as enhance_stream always did (0.06 of 1.9 ms and
        # 187 MB of traffic per 128 x 8 s batch).  A generator handed in (G=trainer.G) may be stepped between calls: its layouts are redone.
        self._own_G = G is None
        self._frozen_precision = None
        self._slots = []                # streams of enhance_stream, created on first use and kept (the runtime maps streams to hardware queues once)

    def _enhance_impl(self, clean_wav, noise_wav, pcm16, lengths):
        if self._own_G and not torch.cuda.is_current_stream_capturing() and (not self.G._weights_frozen or self._frozen_precision != self.G.precision):
            torch.cuda.current_stream(self.device).wait_event(self.G.freeze_weights(self.device))
            self._frozen_precision = self.G.precision
        lengths = au._i32(lengths, self.device)
        frames = au.frames_of(lengths)
        clean_spec, clean_band = au.stft_band(clean_wav, p_power, lengths=lengths)
        noise_band = au.noise_band(noise_wav, p_power, lengths=lengths, frames=frames)
        mask = self.G(clean_band, noise_band)
        alpha2 = M.normed_alpha2(mask, clean_band, inv_p, frames=frames)
        return au.gain_istft(alpha2, clean_spec, rms_target=0.030, pcm16=pcm16, frames=frames)

    @torch.no_grad()
    def enhance(self, clean_wav, noise_wav, pcm16=True, lengths=None):
        """clean_wav, noise_wav [B,L] -> enhanced wav [B, 256*(T-1)] at RMS 0.03 (inference.py:99-115).
        lengths [B] (optional): samples of each utterance inside the padded batch; row b of the result then holds
        256 * (lengths[b] // 256) samples followed by zeros, and its RMS is taken over those samples."""
        return self._enhance_impl(clean_wav, noise_wav, pcm16, lengths)

    @torch.no_grad()
    def enhance_stream(self, batches, inflight=4, pcm16=True):
        """batches: iterable of (clean_wav [B,L], noise_wav [B,L]) or (clean_wav, noise_wav, lengths) device tensors (shapes may differ
        from batch to batch).  Generator of the enhanced batches, in order; each equals ``enhance`` of the same batch bit for bit.

        Up to ``inflight`` batches are enqueued before the oldest result is handed out, each on a stream of its own with its own set
        of generator activation buffers (``Generator_Conv1D_cLN.buffer_slot``); the generator's weight layouts are written once
        (``freeze_weights``) and only read afterwards.  One batch by itself is a chain of kernels of which the IMCRA scan (serial over
        the frames, 1 / 8 .. 1 / 4 of the chip) and the per-utterance tails leave most of the GPU idle: with several batches in flight those
        phases run under another batch's generator (inference.py:79-117 is a loop over independent files).  Throughput = batches in
        flight / a batch's latency as long as every batch sits on a hardware queue of its own; the runtime has four, so the default is
        four batches - three side streams and the caller's own stream (round 6: 102 k -> 111 k utterances/s at 128 x 8 s; a fifth: 97 k).
        The consumer's current stream waits for a result before it is yielded; the result stays valid until the consumer drops it."""
        if self.device.type != 'cuda':
            raise RuntimeError("nele_gan_amd: the enhancement path runs on the GPU only (no CPU fallback)")
        inflight = max(1, int(inflight))
        caller = torch.cuda.current_stream(self.device)
        n_side = inflight if inflight <= 3 else inflight - 1
        if len(self._slots) < n_side:
            # a hardware queue each for the first three (which of a process's streams share one is decided at their creation: two slots
            # on one queue run their batches one after the other - 65 k instead of 100 k utterances/s)
            if min(n_side, 3) > len(self._slots):
                # (round 6: and none of them on the CALLER's queue - "the default stream has a queue of its own" only holds in a fresh
                #  process; after a trainer has created its side streams a new stream may land on it: 93 k instead of 112 k with four in flight)
                self._slots += ops.streams_on_distinct_queues(self.device, min(n_side, 3) - len(self._slots), have=[caller] + self._slots)
            while len(self._slots) < n_side:
                self._slots.append(ops.side_stream(self.device))
        # the runtime has four hardware queues: the three side streams above + the one of the caller's own stream.  A FOURTH batch in
        # flight therefore runs on the caller's stream (a fourth side stream would share a queue with one of the three: 66 k utterances/s)
        slots = list(self._slots[:min(n_side, 3)]) + ([caller] if inflight >= 4 else []) + list(self._slots[3:n_side])
        G = self.G
        slot0 = G.buffer_slot
        was_frozen = G._weights_frozen and self._own_G and self._frozen_precision == G.precision
        frozen = G.freeze_weights(self.device)
        self._frozen_precision = G.precision
        pending = collections.deque()

        def hand_out():
            out, done, st = pending.popleft()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)
            out.record_stream(cur)
            return out

        try:
            for k, batch in enumerate(batches):
                if len(pending) >= inflight:
                    yield hand_out()
                clean_wav, noise_wav = batch[0], batch[1]
                lengths = batch[2] if len(batch) > 2 else None
                st = slots[k % inflight]
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(self.device))       # the inputs (and, first round, the frozen layouts) exist
                G.buffer_slot = ('stream', k % inflight)
                with torch.cuda.stream(st):
                    st.wait_event(ready)
                    if k < inflight:
                        st.wait_event(frozen)
                    out = self._enhance_impl(clean_wav, noise_wav, pcm16, lengths)
                    done = torch.cuda.Event()
                    done.record(st)
                for t in (clean_wav, noise_wav):
                    if t.is_cuda:
                        t.record_stream(st)
                pending.append((out, done, st))
            while pending:
                yield hand_out()
        finally:
            G.buffer_slot = slot0
            if not (self._own_G or was_frozen):
                G.unfreeze_weights()
            for _, done, _ in pending:                                   # a consumer that stopped early: nothing may still write when we return
                done.synchronize()


def enhance_files(enhancer, file_list, noise_path, output_path, batch=32, epoch_tag=1, sort_by_length=True, inflight=3, workers=8,
                  pad_to=4096, ahead=2, write=True):
    """inference.py:79-117 over ``file_list`` (clean wav paths; the noise file of each has the same name under ``noise_path``).
    Returns the list of written files ('<output_path>/<stem>@<epoch_tag>.wav'), in list order, for this rank's shard.

    Host side (the reference: one file at a time, librosa.load + sf.write in the main process): dataio.FileBatches reads a batch per
    library call (``workers`` library threads) straight into pinned int16 staging rows (batches padded to a multiple of ``pad_to``
    samples: few distinct shapes, the generator's buffers are cached per shape), uploads asynchronously, two batches ahead, and
    converts to float32 on the device;
    ``inflight`` batches are on the GPU at a time (enhance_stream); results are converted to their int16 sample values on the device, come
    back through pinned buffers on a copy stream and are written a batch per call by the library's own threads (nele_wav_write_pcm16_batch)
    while later batches run.  The host moves bytes only: no sample is touched by the interpreter or converted on a CPU core."""
    import concurrent.futures as cf
    import numpy as np
    from . import dataio
    dataio.creatdir(output_path)
    lo, hi = ndist.shard_range(len(file_list))                      # contiguous shard per rank; the whole list on one GPU
    mine = list(range(lo, hi))
    order = mine
    if sort_by_length:
        # batches of similar lengths waste less padding; the output list keeps the caller's order
        sizes = {i: os.path.getsize(file_list[i]) for i in mine}
        order = sorted(mine, key=lambda i: sizes[i])
    dev = enhancer.device
    fb = dataio.FileBatches([file_list[i] for i in order], noise_path, batch=batch, workers=workers, ahead=ahead, device=dev, pad_to=pad_to,
                            keep=inflight + ahead)
    if sort_by_length:
        for i in mine:                                                   # the sizes are known already: no second stat() per clean file
            fb._bounds[file_list[i]] = max(1, (sizes[i] - 44 + 1) // 2)
    written = {}
    names = {i: file_list[i].split('/')[-1] for i in mine}
    copy_out = torch.cuda.Stream(device=dev)
    pool = cf.ThreadPoolExecutor(max_workers=inflight + 2)           # a task = one batch's write call
    meta, outq, writes = [], collections.deque(), []

    def batches():
        for g in range(len(fb)):
            b = fb[g]
            meta.append((order[g * batch:(g + 1) * batch], b['lengths_host']))
            yield b['clean'], b['noise'], b['lengths']

    def flush(block_all):
        while outq and (block_all or outq[0][1].query()):
            host, ev, sel, lens = outq.popleft()
            ev.synchronize()
            paths = []
            for i in sel:
                path = dataio.enhanced_name(output_path, names[i], epoch_tag)
                paths.append(path)
                written[i] = path
            ns = [256 * (int(n) // 256) for n in lens]
            # one task per batch: the library's own threads write the files (nele_wav_write_pcm16_batch), no per-file work in the interpreter
            if write:
                writes.append((pool.submit(dataio.write_wav_batch_pcm16, paths, host.numpy(), ns, fs, workers), host))
            else:                                                        # (diagnostic: everything but the file writes)
                dataio.pinned_put(host)
        while writes and (block_all or writes[0][0].done()):
            fut, host = writes.pop(0)
            fut.result()                                                 # re-raises a writer's error
            dataio.pinned_put(host)                                     # the pinned buffer goes back to the pool

    try:
        for g, enh in enumerate(enhancer.enhance_stream(batches(), inflight=inflight, pcm16=True)):
            sel, lens = meta[g]
            meta[g] = None
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
            with torch.cuda.stream(copy_out):
                copy_out.wait_event(done)
                # the PCM_16 emulation left values k / 32768: k itself goes to the host (half the bytes), the writer threads add headers
                q = torch.empty(tuple(enh.shape), dtype=torch.int16, device=enh.device)
                _lib.check(_lib.lib.nele_float_to_pcm16(enh.data_ptr(), enh.shape[1], enh.shape[0], enh.shape[1], q.data_ptr(), enh.shape[1], 1,
                                                        copy_out.cuda_stream), 'nele_float_to_pcm16')
                enh.record_stream(copy_out)
                host = dataio.pinned_get(q.shape, torch.int16)
                host.copy_(q, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_out)
            outq.append((host, ev, sel, lens))
            flush(False)
        flush(True)
    finally:
        pool.shutdown(wait=True)
        fb.close()
    return [written[i] for i in mine]


def score_enhanced(clean_path, noise_path, enhanced_names, groups=('Cafeteria', 'AirportAnnouncement'), quality=None, out=None):
    """inference.py:119-146: the true (unmapped) scores of the enhanced test files, per noise type - ``groups``: substrings of the file
    names (the reference's test set carries the noise type in the name); ``None`` / an empty tuple scores the whole list as one group ''.
    SIIB, HASPI and ESTOI through dataio.read_batch_* (one launch per metric and 256 files); PESQ and ViSQOL - external host programs,
    quality.py - when ``quality`` is true (None: when both programs are registered), else reported as nan.
    -> {group: {'siib', 'haspi', 'estoi', 'pesq', 'visqol': mean, 'files': n}}; the reference's report lines are printed to ``out``
    (a file object; default sys.stdout)."""
    import sys
    import numpy as np
    from . import dataio
    from . import quality as q
    if quality is None:
        quality = q._BACKENDS['visqol'] is not None and (q._BACKENDS['pesq'] is not None or q._BACKENDS['pesq_batch'] is not None)
    out = sys.stdout if out is None else out
    res = {}
    for g in (tuple(groups) if groups else ('',)):
        names = [x for x in enhanced_names if g in x]                               # :120
        if not names:
            continue
        r = {'haspi': float(np.mean(dataio.read_batch_HASPI(clean_path, noise_path, names, norm=False))),      # :122-131
             'estoi': float(np.mean(dataio.read_batch_STOI(clean_path, noise_path, names, norm=False))),
             'siib': float(np.mean(dataio.read_batch_SIIB(clean_path, noise_path, names, norm=False))),
             'pesq': float('nan'), 'visqol': float('nan'), 'files': len(names)}
        if quality:                                                                  # :134-141
            r['pesq'] = float(np.mean(q.read_batch_PESQ(clean_path, names, norm=False)))
            r['visqol'] = float(np.mean(q.read_batch_VISQOL(clean_path, names, norm=False)))
        res[g] = r
        out.write(g + ':\n')                                                         # :142-145
        out.write('SIIB is %.3f, HASPI is %.3f, ESTOI is %.3f, PESQ is %.3f, VISQOL is %.3f\n\n' % (r['siib'], r['haspi'], r['estoi'], r['pesq'], r['visqol']))
        out.write('======\n')
    return res


def main(argv=None):
    """``python -m nele_gan_amd.inference --chkpt ./trained_model/chkpt_GD.pt --clean <dir> --noise <dir> --output <dir>`` = the
    reference's usage step 4 (`python inference.py`, README; inference.py:28-146): every wav under --clean is enhanced against the noise file
    of the same name, written as '<output>/<name>@1.wav' (PCM_16), and - ``--score`` - the per-noise report of :119-146 is printed."""
    import argparse
    from . import dataio
    ap = argparse.ArgumentParser(prog='python -m nele_gan_amd.inference', description=main.__doc__)
    ap.add_argument('--chkpt', default=None, help="checkpoint with an 'enhance-model' entry (reference checkpoints load unchanged); default: random weights")
    ap.add_argument('--clean', required=True)
    ap.add_argument('--noise', required=True)
    ap.add_argument('--output', default='./output_inference')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--precision', choices=('bf16', 'f32'), default='bf16')
    ap.add_argument('--score', action='store_true', help='SIIB / HASPI / ESTOI (PESQ / ViSQOL when registered) of the written files')
    ap.add_argument('--groups', default='Cafeteria,AirportAnnouncement', help='noise types (substrings of the file names) reported separately; empty = all files together')
    from . import quality as _q
    _q.add_cli_arguments(ap)
    a = ap.parse_args(argv)
    _q.backends_from_cli(a)
    e = Enhancer(a.chkpt)
    e.G.precision = a.precision
    files = sorted(dataio.get_filepaths(a.clean))
    noise = a.noise.rstrip('/') + '/'
    out = enhance_files(e, files, noise, a.output, batch=a.batch)
    print('%d files -> %s' % (len(out), a.output))
    if a.score:
        score_enhanced(a.clean.rstrip('/') + '/', noise, out, groups=tuple(g for g in a.groups.split(',') if g))
    return 0


if __name__ == '__main__':
    import sys
    sys.exit(main())
