/* libnele_hip.so -- C ABI of the MI355X-native NELE-GAN hot path (gfx950).
 *
 * The reference (nii-yamagishilab/NELE-GAN) has no FFI: its hot path is Python calling numpy /
 * librosa / torch.  Each entry point below replaces the reference function(s) cited next to it
 * (paths relative to the reference checkout) and is what a ctypes binding on the reference side
 * would call.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in _host; the caller owns all
 *     memory, nothing is allocated or freed inside the library;
 *   - `stream` is a hipStream_t (passed as void*); kernels are enqueued on it and the call returns
 *     without synchronising; no internal threads; reductions are fixed-order (bit-reproducible);
 *   - return value: 0 = ok, -1 invalid argument, -2 unsupported shape, -3 signal below threshold /
 *     too short (the reference raises there), -4 HIP error, -5 workspace too small;
 *     nele_last_error_string() describes the last non-zero status of the calling thread;
 *   - row-major layouts; B = utterances, L = samples, T = 1 + L/256 STFT frames.
 */
#ifndef NELE_HIP_H
#define NELE_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

int nele_version(void);
const char* nele_last_error_string(void);
int nele_device_info(int* cu_count, int* wave_size, char* arch, int arch_len);

/* ---- signal features / resynthesis (csrc/features.hip) ---------------------------------------- */

/* audio_util.py:53-58 STFT (librosa 0.7.1: reflect pad 256, periodic Hann 512, hop 256),
 * audio_util.py:30-50 compute_band_E, audio_util.py:422-437 Sp_and_phase_Speech.
 * wav [B][L] f32 -> spec [B][T][257] complex64 (may be NULL), band [B][T][64] f32 = bandE**power (may be NULL). */
int nele_stft_band(const float* wav, int B, int L, float power, void* spec, float* band, void* stream);

/* noise_est/imcra.py:521-577 imcra_est.estimate + :363-484 imcra.update; audio_util.py:113-117 NoisePSD,
 * :439-456 Sp_and_phase_Noise.  spec [B][T][257] complex64 -> psd [B][T][257] f32 (may be NULL),
 * band [B][T][64] f32 = compute_band_E(sqrt(psd))**power (may be NULL). */
int nele_imcra_band(const void* spec, int B, int T, float power, float* psd, float* band, void* stream);

/* audio_util.py:93-110 interp_band_gain, :76-90 Resyn, :458-461 SP_to_wav, :60-65 ISTFT.
 * alpha2 [B][T][64] f32 (energy gains), spec [B][T][257] complex64 -> wav [B][256*(T-1)] f32. */
int nele_gain_istft(const float* alpha2, const void* spec, int B, int T, float* wav, void* stream);

/* inference.py:109 (enh / rms(enh) * target_rms, skipped when target_rms <= 0) and the PCM_16
 * write/read round trip of train_nele.py:313 + dataloader.py:58 (pcm16 != 0).  In place on wav [B][N]. */
int nele_wav_post(float* wav, int B, int N, float target_rms, int pcm16, void* stream);

#ifdef __cplusplus
}
#endif
#endif
