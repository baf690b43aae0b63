"""Enhancement path of the reference ``inference.py`` (inference.py:79-117), batched on the GPU:
features -> G (eval) -> mask * beta2 -> resynthesis -> enh / rms(enh) * 0.03 -> PCM_16."""
import torch

from . import audio_util as au
from . import model as M

p_power = (1 / 6)
inv_p = 6


class Enhancer:
    def __init__(self, chkpt_path=None, device='cuda', G=None):
        self.device = torch.device(device)
        self.G = G if G is not None else M.Generator_Conv1D_cLN()
        if chkpt_path is not None:
            self.G.load_state_dict(torch.load(chkpt_path, map_location='cpu')['enhance-model'])   # inference.py:71-72
        self.G = self.G.to(self.device)
        self.G.eval()

    @torch.no_grad()
    def enhance(self, clean_wav, noise_wav, pcm16=True):
        """clean_wav, noise_wav [B,L] -> enhanced wav [B, 256*(T-1)] at RMS 0.03 (inference.py:99-115)."""
        clean_spec, clean_band = au.stft_band(clean_wav, p_power)
        noise_spec, _ = au.stft_band(noise_wav, p_power, want_band=False)
        _, noise_band = au.imcra_band(noise_spec, p_power)
        mask = self.G(clean_band, noise_band)
        alpha2 = M.normed_alpha2(mask, clean_band, inv_p)
        return au.gain_istft(alpha2, clean_spec, rms_target=0.030, pcm16=pcm16)
