"""Enhancement path of the reference ``inference.py`` (inference.py:79-117), batched on the GPU:
features -> G (eval) -> mask * beta2 -> resynthesis -> enh / rms(enh) * 0.03 -> PCM_16.

``Enhancer.enhance``      one padded batch resident in HBM (optional per-utterance lengths)
``enhance_files``         the reference's loop over a file list (inference.py:79-117): files of any lengths are decoded on the
                          host, padded side by side in batches, enhanced, and written as '<name>@1.wav' PCM_16; with
                          torch.distributed initialised every rank takes a contiguous shard of the list (BASELINE configs[4]:
                          data-parallel enhancement, pure replicas, no collective).
"""
import torch

from . import audio_util as au
from . import dist as ndist
from . import model as M
from . import ops

p_power = (1 / 6)
inv_p = 6
fs = 16000


class Enhancer:
    def __init__(self, chkpt_path=None, device='cuda', G=None):
        self.device = torch.device(device)
        self.G = G if G is not None else M.Generator_Conv1D_cLN()
        if chkpt_path is not None:
            self.G.load_state_dict(torch.load(chkpt_path, map_location='cpu')['enhance-model'])   # inference.py:71-72
        self.G = self.G.to(self.device)
        self.G.eval()

    @torch.no_grad()
    def enhance(self, clean_wav, noise_wav, pcm16=True, lengths=None):
        """clean_wav, noise_wav [B,L] -> enhanced wav [B, 256*(T-1)] at RMS 0.03 (inference.py:99-115).
        lengths [B] (optional): samples of each utterance inside the padded batch; row b of the result then holds
        256 * (lengths[b] // 256) samples followed by zeros, and its RMS is taken over those samples."""
        lengths = au._i32(lengths, self.device)
        frames = au.frames_of(lengths)
        clean_spec, clean_band = au.stft_band(clean_wav, p_power, lengths=lengths)
        noise_spec, _ = au.stft_band(noise_wav, p_power, want_band=False, lengths=lengths)
        _, noise_band = au.imcra_band(noise_spec, p_power, frames=frames)
        mask = self.G(clean_band, noise_band)
        alpha2 = M.normed_alpha2(mask, clean_band, inv_p, frames=frames)
        return au.gain_istft(alpha2, clean_spec, rms_target=0.030, pcm16=pcm16, frames=frames)


def enhance_files(enhancer, file_list, noise_path, output_path, batch=32, epoch_tag=1, sort_by_length=True):
    """inference.py:79-117 over ``file_list`` (clean wav paths; the noise file of each has the same name under ``noise_path``).
    Returns the list of written files ('<output_path>/<stem>@<epoch_tag>.wav'), in list order, for this rank's shard."""
    import numpy as np
    from . import dataio
    dataio.creatdir(output_path)
    lo, hi = ndist.shard_range(len(file_list))                      # contiguous shard per rank; the whole list on one GPU
    mine = list(range(lo, hi))
    order = mine
    if sort_by_length:
        # batches of similar lengths waste less padding; the output list keeps the caller's order
        sizes = {i: __import__('os').path.getsize(file_list[i]) for i in mine}
        order = sorted(mine, key=lambda i: sizes[i])
    written = {}
    for k in range(0, len(order), batch):
        sel = order[k:k + batch]
        cl, ns = [], []
        for i in sel:
            c, sr = dataio.load(file_list[i])
            assert sr == 16000                                       # inference.py:85-88
            n, sr = dataio.load(noise_path + file_list[i].split('/')[-1])
            assert sr == 16000
            m = min(len(c), len(n))
            cl.append(c[:m]); ns.append(n[:m])
        cp, lens = dataio.pad_batch(cl)
        npad, _ = dataio.pad_batch(ns)
        enh = enhancer.enhance(torch.from_numpy(cp).to(enhancer.device), torch.from_numpy(npad).to(enhancer.device), pcm16=True,
                               lengths=torch.from_numpy(lens)).cpu().numpy()
        for r, i in enumerate(sel):
            name = file_list[i].split('/')[-1]
            path = dataio.enhanced_name(output_path, name, epoch_tag)
            dataio.write_wav_pcm16(path, enh[r, :256 * (int(lens[r]) // 256)], fs, quantised=True)
            written[i] = path
    return [written[i] for i in mine]
