"""Signal features / resynthesis on the GPU: the host-side mirror of the reference's ``audio_util.py``.

Same function names, argument order and shapes as the reference (``Sp_and_phase_Speech``,
``Sp_and_phase_Noise``, ``SP_to_wav``, ``compute_band_E`` via ``stft_band``, ``NoisePSD``, ``rms``),
plus batched entry points that take/return device tensors with a leading batch dimension.  All
arithmetic happens in libnele_hip.so (csrc/features.hip); torch only owns the memory.

Device layouts (frame-major): spec [B,T,257] complex64, band [B,T,64] f32,
psd [B,T,257] f32.  The single-utterance reference-shaped wrappers return mag/phase as [257,T].
"""
import os

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream

NB_BANDS = 64
N_FFT = 512
HOP = 256
N_BINS = 257
fs = 16000
TWO_KERNEL_IMCRA = os.environ.get('NELE_IMCRA_SPLIT', '1') != '0'      # imcra_band through nele_imcra_band_ws (0: the one-kernel form, A/B)
# audio_util.py:23
gmtband = [0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 28, 30, 32, 34,
           36, 38, 41, 43, 46, 49, 52, 55, 58, 62, 66, 70, 74, 79, 83, 88, 93, 99, 105, 111, 117, 124, 131, 139, 147,
           156, 165, 174, 184, 195, 206, 218, 230, 243, 257]


def _dev(x, device=None):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not x.is_cuda:
        x = x.to(device or 'cuda')
    return x.contiguous()


def n_frames(L):
    """T = 1 + L // 256 (centred STFT, audio_util.py:53-58)."""
    return 1 + L // HOP


def _i32(t, device):
    """lengths / frame counts -> contiguous device int32 tensor (or None)."""
    if t is None:
        return None
    return torch.as_tensor(t).to(device=device, dtype=torch.int32).contiguous()


def frames_of(lengths):
    """STFT frames of utterances with ``lengths`` samples: T = 1 + L // 256 (device int32 in, device int32 out)."""
    return None if lengths is None else (1 + torch.div(lengths, HOP, rounding_mode='floor')).to(torch.int32)


# ------------------------------------------------------------------ batched device API
def stft_band(wav, power=1.0 / 6, want_spec=True, want_band=True, lengths=None):
    """wav [B,L] f32 (device) -> (spec [B,T,257] complex64 | None, band [B,T,64] f32 | None).
    band = compute_band_E(|STFT|) ** power (audio_util.py:426-433).  lengths [B]: samples of each utterance inside the padded
    batch (frames behind a row's own end come out as zeros)."""
    wav = _dev(wav).float()
    if wav.dim() != 2:
        raise ValueError("stft_band: wav must be [B, L]")
    B, L = wav.shape
    T = n_frames(L)
    spec = torch.empty((B, T, N_BINS), dtype=torch.complex64, device=wav.device) if want_spec else None
    band = torch.empty((B, T, NB_BANDS), dtype=torch.float32, device=wav.device) if want_band else None
    call('nele_stft_band_var', ptr(wav), ptr(_i32(lengths, wav.device)), B, L, float(power), ptr(spec), ptr(band), stream())
    return spec, band


def imcra_band(spec, power=1.0 / 6, want_psd=False, frames=None):
    """spec [B,T,257] complex64 -> (psd [B,T,257] f32 | None, band [B,T,64] f32) with
    band = compute_band_E(sqrt(NoisePSD(spec))) ** power (audio_util.py:445-451)."""
    if spec.dtype != torch.complex64 or spec.dim() != 3 or spec.shape[2] != N_BINS:
        raise ValueError("imcra_band: spec must be [B, T, 257] complex64")
    spec = spec.contiguous()
    B, T, _ = spec.shape
    # the PSD buffer is always handed over: with it the kernel computes the band feature from the PSD after the serial scan instead of inside it
    psd = torch.empty((B, T, N_BINS), dtype=torch.float32, device=spec.device)
    band = torch.empty((B, T, NB_BANDS), dtype=torch.float32, device=spec.device)
    if TWO_KERNEL_IMCRA:
        # the prior q (per utterance, bins through LDS) and the tracker (per utterance and bin) as two kernels: bit-identical, shorter
        # (nele_imcra_band_ws); the prior's q / (1 - q) travels through a scratch buffer of the caching allocator
        nb = int(_lib.lib.nele_imcra_workspace_bytes(B, T))
        ws = torch.empty((nb,), dtype=torch.uint8, device=spec.device)
        call('nele_imcra_band_ws', ptr(spec), ptr(_i32(frames, spec.device)), B, T, float(power), ptr(psd), ptr(band), ptr(ws), nb, stream())
    else:
        call('nele_imcra_band_var', ptr(spec), ptr(_i32(frames, spec.device)), B, T, float(power), ptr(psd), ptr(band), stream())
    return psd, band


def noise_band(wav, power=1.0 / 6, lengths=None, frames=None, want_psd=False):
    """Sp_and_phase_Noise's band feature (audio_util.py:439-451: compute_band_E(sqrt(NoisePSD(STFT(noise)))) ** power) of a batch of noise
    signals, wav [B,L] -> band [B,T,64] (and the PSD [B,T,257] with ``want_psd``): IMCRA needs |STFT|^2 only, so the STFT kernel writes
    that (float32, as numpy squares np.abs) and the spectrum never exists in memory (nele_stft_pow_var + nele_imcra_band_pw).  The same
    numbers, bit for bit, as imcra_band(stft_band(wav)) - which stays for callers that want the spectrum."""
    if not TWO_KERNEL_IMCRA:
        spec, _ = stft_band(wav, power, want_band=False, lengths=lengths)
        psd, band = imcra_band(spec, power, want_psd=True, frames=frames if frames is not None else frames_of(_i32(lengths, spec.device)))
        return (psd, band) if want_psd else band
    wav = _dev(wav).float()
    if wav.dim() != 2:
        raise ValueError("noise_band: wav must be [B, L]")
    B, L = wav.shape
    T = n_frames(L)
    lengths = _i32(lengths, wav.device)
    if frames is None:
        frames = frames_of(lengths)
    nb = int(_lib.lib.nele_imcra_workspace_bytes(B, T))
    pw = torch.empty((B, T, N_BINS), dtype=torch.float32, device=wav.device)
    ws = torch.empty((nb,), dtype=torch.uint8, device=wav.device)
    psd = torch.empty((B, T, N_BINS), dtype=torch.float32, device=wav.device)
    band = torch.empty((B, T, NB_BANDS), dtype=torch.float32, device=wav.device)
    call('nele_stft_pow_var', ptr(wav), ptr(lengths), B, L, float(power), None, None, ptr(pw), stream())
    call('nele_imcra_band_pw', ptr(pw), ptr(_i32(frames, wav.device)), B, T, float(power), ptr(psd), ptr(band), ptr(ws), nb, stream())
    return (psd, band) if want_psd else band


def gain_istft(alpha2, spec, rms_target=0.0, pcm16=False, frames=None):
    """alpha2 [B,T,64] f32 (energy gains), spec [B,T,257] complex64 -> wav [B, 256*(T-1)] f32.
    Resyn (audio_util.py:76-90) + optional enh/rms(enh)*target (inference.py:109) + optional
    PCM_16 write/read round trip (train_nele.py:313, dataloader.py:58)."""
    alpha2 = _dev(alpha2).float()
    spec = spec.contiguous()
    B, T, _ = spec.shape
    if alpha2.shape != (B, T, NB_BANDS):
        raise ValueError("gain_istft: alpha2 must be [B, T, 64] matching spec")
    wav = torch.empty((B, HOP * (T - 1)), dtype=torch.float32, device=spec.device)
    frames = _i32(frames, spec.device)
    call('nele_gain_istft_var', ptr(alpha2), ptr(spec), ptr(frames), B, T, ptr(wav), stream())
    if rms_target > 0 or pcm16:
        ws = None
        if rms_target > 0:
            ws = torch.empty(int(_lib.lib.nele_wav_post_workspace_doubles(B, wav.shape[1])), dtype=torch.float64, device=wav.device)
        call('nele_wav_post_var', ptr(wav), ptr(frames), B, wav.shape[1], float(rms_target), int(bool(pcm16)), ptr(ws), stream())
    return wav


def pcm16_roundtrip(wav):
    """In-place emulation of sf.write(..., 'PCM_16') followed by librosa.load."""
    wav = wav.contiguous()
    B = 1 if wav.dim() == 1 else wav.shape[0]
    call('nele_wav_post', ptr(wav), B, wav.shape[-1], 0.0, 1, None, stream())
    return wav


# ------------------------------------------------------------------ reference-shaped wrappers
def STFT(x, nfft=512, nw=512, nm=256):
    """audio_util.py:53-58: [L] -> [257, T] complex64 (device tensor)."""
    spec, _ = stft_band(_dev(x).reshape(1, -1), want_band=False)
    return spec[0].transpose(0, 1)


def ISTFT(X, nw=512, nm=256):
    """audio_util.py:60-65: X [257, T] complex64 -> wav [256*(T-1)] (librosa.istft, hop 256, window 512)."""
    spec = _dev(X).to(torch.complex64).transpose(0, 1).contiguous().unsqueeze(0)
    T = spec.shape[1]
    wav = torch.empty((1, HOP * (T - 1)), dtype=torch.float32, device=spec.device)
    call('nele_gain_istft', None, ptr(spec), 1, T, ptr(wav), stream())
    return wav[0]


def compute_band_E(X):
    """audio_util.py:30-50: X magnitude spectrogram [T, 257] -> [T, 64] float32 band energies (device tensor)."""
    X = _dev(X).float()
    if X.dim() != 2 or X.shape[1] != N_BINS:
        raise ValueError("compute_band_E: X must be [T, 257]")
    out = torch.empty((X.shape[0], NB_BANDS), dtype=torch.float32, device=X.device)
    call('nele_compute_band_E', ptr(X), X.shape[0], ptr(out), stream())
    return out


def interp_band_gain(bandE):
    """audio_util.py:93-110: bandE [64] (or [T, 64]) -> g [257] (or [T, 257]) float64."""
    b = _dev(bandE).float()
    single = b.dim() == 1
    b = b.reshape(-1, NB_BANDS).contiguous()
    g = torch.empty((b.shape[0], N_BINS), dtype=torch.float64, device=b.device)
    call('nele_interp_band_gain', ptr(b), b.shape[0], ptr(g), stream())
    return g[0] if single else g


def Resyn(X, alpha):
    """audio_util.py:76-90: X [257, T] complex spectrogram, alpha [T, 64] energy gains -> wav [256*(T-1)]."""
    spec = _dev(X).to(torch.complex64).transpose(0, 1).contiguous().unsqueeze(0)
    return gain_istft(_dev(alpha).unsqueeze(0), spec)[0]


def clip(x):
    """audio_util.py:67-74: divide by (1.05, 1.10, ...) until max < 1 and min >= -1 (device tensor in, device tensor out)."""
    from . import eval_metrics
    x = _dev(x)
    single = x.dim() == 1
    out = eval_metrics.norm_clip(x.reshape(1, -1) if single else x)
    return out[0] if single else out


def NoisePSD(MIXED, nfft=512):
    """audio_util.py:113-117: MIXED [257, T] complex64 -> estimated noise PSD [257, T] f32."""
    spec = _dev(MIXED).transpose(0, 1).contiguous().unsqueeze(0)
    psd, _ = imcra_band(spec, want_psd=True)
    return psd[0].transpose(0, 1)


def Sp_and_phase_Speech(signal, power, Normalization=True):
    """audio_util.py:422-437 -> (bandE [T,64], mag [257,T], phase [257,T]) device tensors."""
    spec, band = stft_band(_dev(signal).reshape(1, -1), power if Normalization else 1.0)
    if not Normalization:
        print('No normalization for func: Sp_and_phase_Clean')
    F = spec[0].transpose(0, 1)
    return band[0], F.abs(), F.angle()


def Sp_and_phase_Noise(signal, power, Normalization=True):
    """audio_util.py:439-456."""
    spec, _ = stft_band(_dev(signal).reshape(1, -1), want_band=False)
    _, band = imcra_band(spec, power if Normalization else 1.0)
    if not Normalization:
        print('No normalization for func: Sp_and_phase_Noise')
    F = spec[0].transpose(0, 1)
    return band[0], F.abs(), F.angle()


def SP_to_wav(alpha2, mag, phase, signal_length=None):
    """audio_util.py:458-461: alpha2 [T,64], mag/phase [257,T] -> wav [256*(T-1)]."""
    mag = _dev(mag)
    phase = _dev(phase)
    spec = torch.polar(mag.float(), phase.float()).transpose(0, 1).contiguous().unsqueeze(0)
    return gain_istft(_dev(alpha2).unsqueeze(0), spec)[0]


def rms(x):
    """audio_util.py:463-464."""
    x = _dev(x) if not isinstance(x, torch.Tensor) else x
    return torch.sqrt(torch.mean(x ** 2))
