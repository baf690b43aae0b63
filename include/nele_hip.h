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
/* 0 for the product library (libnele_hip.so: one path per operation, no environment switch is read); 1 for the test library built from
 * the same sources with -DNELE_AB (libnele_hip_ab.so), in which the superseded kernel variants exist and NELE_* switches select them. */
int nele_build_has_ab_switches(void);
/* Measurement aid (no reference counterpart): one wave that idles for `microseconds` on `stream`.  The HIP runtime multiplexes streams
 * onto a handful of hardware queues; a caller that keeps a long dependent chain on one stream (train_nele.py's pipelined small-batch
 * step) finds out which of its other streams share that queue by parking this kernel on the first and timing a trivial kernel on the
 * others. */
int nele_stream_spin(double microseconds, void* stream);
/* Test aid: `workgroups` workgroups of 1024 threads, each holding `lds_bytes` of LDS (160 KB = a whole CU for LDS-using kernels), idle for
 * `microseconds` on `stream`: stands for kernels that stay resident on part of the GPU while the path runs (a collective library's channels,
 * another process).  tests/test_metrics_gpu.py runs the batched eigensolver beside it and records time and give-ups. */
int nele_stream_occupy(int workgroups, int lds_bytes, double microseconds, void* stream);

/* Measurement hook (no reference counterpart): HIP-event timing of single kernels that are launched from inside multi-kernel entry
 * points, on the stream they run on.  nele_profile_begin(tags) arms it for the launch sites whose tag is in the comma-separated list
 * (e.g. "eigh_tridiag_cluster,haspi_bank_gain_kernel"; NULL disarms); nele_profile_collect_tag waits for the recorded launches of one
 * tag and writes their durations in milliseconds (returns how many; stays armed); nele_profile_collect = the first tag, then disarm. */
int nele_profile_begin(const char* tags);
int nele_profile_collect_tag(const char* tag, float* ms_out, int max_n);
int nele_profile_collect(float* ms_out, int max_n);

/* ---- composite entry points of the dense part: job tables (csrc/plan.hip) ---------------------------------------------------------
 * model.py:83-98 Generator_Conv1D_cLN.forward, :118-132 Discriminator.forward and their autograd, each as ONE call.
 * A pass is a fixed sequence of the per-layer entry points below over a few streams; a PLAN is that sequence recorded once per shape
 * (nele_plan_create: operation + arguments, each argument a constant or "slot k + offset" of a small per-call array) and
 * nele_plan_run enqueues it: same kernels, order, streams and hand-over events as the per-layer calls, bit-identical results, one
 * foreign-function call instead of 25 .. 60.  The host mirror records its own per-layer loop the first time it sees a shape
 * (nele_gan_amd/_lib.py), so the layer logic exists once; a reference-side binding may build the tables itself the same way.
 * Operations: every entry point of this header whose last parameter is the stream (nele_plan_op_id(name); csrc/plan_ops.inc is generated
 * from this header by tools/gen_plan_ops.py).  streams_host [nstreams]: index 0 = the caller's stream, 1.. = the side streams the plan was
 * recorded with (weight gradients run beside the data-gradient chain). */
#define NELE_PLAN_MAXARGS 24
typedef struct nele_plan_job {
    int op;                               /* nele_plan_op_id("nele_...") */
    int nargs;                            /* parameters of that entry point, the stream included */
    int stream;                           /* index into streams_host */
    int slot[NELE_PLAN_MAXARGS];          /* -1: the argument is ival / fval; k >= 0: it is slots_host[k] + ival (pointers / counters that change per call) */
    long long ival[NELE_PLAN_MAXARGS];    /* integers and pointers (pointers to host arrays must stay valid for the plan's lifetime) */
    double fval[NELE_PLAN_MAXARGS];       /* float / double arguments */
} nele_plan_job;
int nele_plan_op_id(const char* name);    /* -1: not an operation */
int nele_plan_op_nargs(int op);
int nele_plan_create(const nele_plan_job* jobs, int njobs, int nslots, int nstreams, void** plan_out);
int nele_plan_run(void* plan, void* const* streams_host, int nstreams, const long long* slots_host, int nslots);
/* Slot declarations: what a call must provide behind slot k (bytes of device memory; 0 = a scalar slot; nullable != 0: the pointer may be
 * NULL).  nele_plan_run refuses a NULL in a declared non-optional slot; nele_plan_run_sized also takes the sizes of the caller's buffers
 * (sizes_host[k] bytes behind slots_host[k]) and refuses one that is shorter than what the plan touches.  Plans built by
 * nele_gen_plan_build / nele_disc_plan_build come fully declared; a recorded plan is declared by its recorder. */
int nele_plan_declare_slot(void* plan, int slot, long long bytes, int nullable);
long long nele_plan_slot_bytes(void* plan, int slot);
int nele_plan_run_sized(void* plan, void* const* streams_host, int nstreams, const long long* slots_host, const long long* sizes_host, int nslots);
int nele_plan_destroy(void* plan);
/* model.py:83-98.  Slots 0 .. 3 of the plan = x [B][T][64], y [B][T][64], mask [B][T][64] (out), token (nele_glayer16_fwd's counter: the
 * plan uses token, token + 1, ..). */
int nele_gen_fwd(void* plan, const float* x, const float* y, float* mask, unsigned token, void* const* streams_host, int nstreams);
/* autograd of the above: slots 0, 1 = dmask, mask; parameter gradients accumulate into the flat gradient buffer the plan was recorded on */
int nele_gen_bwd(void* plan, const float* dmask, const float* mask, void* const* streams_host, int nstreams);
/* model.py:118-132 (Discriminator / Discriminator_Quality).  Slots 0 .. 2 = din [B][64][T][4], wvalid [B] or NULL, score [B][nout] (out) */
int nele_disc_fwd(void* plan, const float* din, const int* wvalid, float* score, void* const* streams_host, int nstreams);
/* autograd of the above: slots 0 .. 3 = dscore, score, wvalid, din (the forward pass's input: the first layer's weight gradient reads it);
 * the input gradient lands in the buffer the plan was recorded with */
int nele_disc_bwd(void* plan, const float* dscore, const float* score, const int* wvalid, const float* din, void* const* streams_host, int nstreams);
/* ---- plans built inside the library (csrc/netplan.hip, round 6): the composite entry points above need no host-language recorder --------
 * model.py:43-98 (Generator_Conv1D_cLN), :101-166 (Discriminator / Discriminator_Quality) and their autograd as job tables over this
 * header's per-layer entry points: which kernels, layouts, workspaces and stream hand-overs make up a pass is stated by the library.
 * The caller owns (a) the model's parameters and gradients as two flat float32 buffers in nn.Module.parameters() order of the reference
 * modules - nele_{gen,disc}_param_layout give the offsets (G: 6 x {conv.weight, conv.bias, cLN.gain0, cLN.bias0}, fc1.weight, fc1.bias,
 * fc2.weight, fc2.bias; D: 8 x {bias, weight_orig} for conv1..5, fc1..3) - (b) for a discriminator the spectral-norm vectors, sn_uv_host
 * [16] = HOST array of device pointers {weight_u, weight_v} per layer in that order, and (c) ONE workspace of nele_*_workspace_bytes per
 * (B, T, precision) which holds every activation, weight layout and temporary of both passes: the backward plan reads what the forward
 * plan left there.  *_plan_build zero-fills the workspace on `stream` (padding rows, zero borders, carry slots) and returns the plans;
 * run them with nele_gen_fwd / nele_gen_bwd / nele_disc_fwd / nele_disc_bwd (or nele_plan_run[_sized]) and release them with
 * nele_plan_destroy.  bf16 != 0: bf16 MFMA operands / float32 accumulate (BASELINE configs[1]); 0: float32 operands.
 * need_bwd: the forward plan keeps what a backward pass needs (and bwd_out is built).  overlap_wgrad != 0: weight gradients run on
 * side streams beside the data-gradient chain - streams_host then needs 2 (G) / 3 (D) streams - else everything on streams_host[0].
 * Gradients ACCUMULATE into grads_flat (zero it per optimiser step, as optimizer.zero_grad()). */
long long nele_gen_param_count(void);
int nele_gen_param_layout(long long* offsets_host, int n /* >= 28 */);
long long nele_gen_workspace_bytes(int B, int T, int bf16, int need_bwd);
int nele_gen_plan_build(int B, int T, int bf16, int need_bwd, int overlap_wgrad, const float* params_flat, float* grads_flat, void* workspace,
                        long long workspace_bytes, void* stream, void** fwd_out, void** bwd_out);
/* cin = 3 (Discriminator: enhanced, noise, clean) or 2 (Discriminator_Quality); nout = number of scores (<= 4).  train != 0: the forward
 * plan runs one spectral-norm power iteration first (model.py:105-116 in training mode), 0: sigma from the stored u, v.  need_din: the
 * backward plan also produces the input gradient (the G-step), at nele_disc_workspace_ddin(...) [B][64][T][4].  weight_grads == 0: data
 * gradients only.  bwd_out may be NULL (forward only). */
long long nele_disc_param_count(int cin, int nout);
int nele_disc_param_layout(int cin, int nout, long long* offsets_host, int n /* >= 16 */);
long long nele_disc_workspace_bytes(int B, int T, int cin, int bf16);
float* nele_disc_workspace_ddin(void* workspace, int B, int T, int cin, int bf16);
int nele_disc_plan_build(int B, int T, int cin, int nout, int bf16, int train, int need_din, int weight_grads, int overlap_wgrad,
                         const float* params_flat, float* grads_flat, const void* const* sn_uv_host, void* workspace, long long workspace_bytes,
                         void* stream, void** fwd_out, void** bwd_out);
/* Hand-over events between the streams of a pass (hipEvent, timing disabled), recordable as plan operations */
int nele_event_create(void** event_out);
int nele_event_destroy(void* event);
int nele_event_record(void* event, void* stream);
int nele_stream_wait_event(void* event, void* stream);
/* dst[i] += src[i] (bias gradients: the reduced partials of a weight-gradient call added to the flat gradient buffer) */
int nele_vec_add(float* dst, const float* src, long long n, void* stream);

/* Host-side helper of the file hand-off (dataloader.py:34-37 librosa.load -> libsndfile): decodes a mono PCM_16 RIFF file into out_host [cap]
 * float32 HOST memory (samples / 32768, zeros behind them); *n_out = samples, *sample_rate_out = rate.  No GPU work; a foreign-function call
 * runs it outside the host language's interpreter lock, so loader threads decode in parallel.  Other wav flavours: NELE_ERR_UNSUPPORTED. */
int nele_wav_decode_pcm16(const char* path, float* out_host, long long cap, long long* n_out, int* sample_rate_out);
/* ... and the writer (train_nele.py:198,313, inference.py:115 sf.write(..., 'PCM_16')): float32 HOST samples -> mono PCM_16 RIFF file.
 * quantised = 0: lrintf(x * 32767) saturated, as libsndfile; != 0: the samples are values k / 32768 from nele_wav_post's PCM_16 emulation and
 * are recovered exactly.  No GPU work; runs outside the host language's interpreter lock when called through a foreign-function interface. */
int nele_wav_write_pcm16(const char* path, const float* wav_host, long long n, int sample_rate, int quantised);
/* The same hand-off a BATCH at a time, bytes only on the host (round 5; replaces the per-file librosa.load / sf.write of dataloader.py:34-40,
 * inference.py:99-101,115): paths[n] mono PCM_16 files -> rows of out_host [n][row_stride] int16 HOST memory (pinned staging buffer), at most
 * cap samples each, zeros behind them; n_out[i] = samples read, -1 = not mono PCM_16 (the caller's general reader takes that file),
 * -2 = cannot open; sample_rate_out[n] may be NULL.  `threads` (1 .. 256) library threads share the files; no GPU work. */
int nele_wav_read_pcm16_batch(const char* const* paths, int n, short* out_host, long long row_stride, long long cap, int* n_out,
                              int* sample_rate_out, int threads);
/* Sample counts of n wav files from their RIFF headers in one call (what os.path.getsize / sf.info per file would tell the loader, dataloader.py:34):
 * n_out[i] = samples of a mono PCM_16 file, -1 = another flavour, -2 = cannot open.  No GPU work. */
int nele_wav_probe_pcm16_batch(const char* const* paths, int n, int* n_out, int threads);
/* ... rows of in_host [n][row_stride] int16 HOST memory -> n mono PCM_16 RIFF files of n_samples[i] samples (sf.write(path, wav, 16000, 'PCM_16')) */
int nele_wav_write_pcm16_batch(const char* const* paths, int n, const short* in_host, long long row_stride, const int* n_samples, int sample_rate,
                               int threads);
/* Device side of it: int16 rows -> float32 rows, out[b][i] = i < lengths[b] ? in[b][i] / 32768 : 0 for i < L (what sf.read returns for PCM_16;
 * lengths [B] int32 device memory or NULL: the shorter of an utterance and its noise file, dataloader.py:38-40) ... */
int nele_pcm16_to_float(const short* in, long long in_stride, const int* lengths, int B, long long L, float* out, long long out_stride, void* stream);
/* ... and float32 rows -> the int16 sample values sf.write(..., 'PCM_16') stores: quantised != 0: input = values k / 32768 from the device-side
 * PCM_16 emulation (nele_wav_post), recovered exactly; 0: libsndfile's lrintf(x * 32767), saturated. */
int nele_float_to_pcm16(const float* in, long long in_stride, int B, long long L, short* out, long long out_stride, int quantised, void* stream);

/* ---- signal features / resynthesis (csrc/features.hip) ---------------------------------------- */

/* audio_util.py:53-58 STFT (librosa 0.7.1: reflect pad 256, periodic Hann 512, hop 256),
 * audio_util.py:30-50 compute_band_E, audio_util.py:422-437 Sp_and_phase_Speech.
 * wav [B][L] f32 -> spec [B][T][257] complex64 (may be NULL), band [B][T][64] f32 = bandE**power (may be NULL). */
int nele_stft_band(const float* wav, int B, int L, float power, void* spec, float* band, void* stream);
/* Per-utterance lengths.  The reference works on one file of any length at a time (dataloader.py:30-42, audio_util.py:134-141); a
 * batch carries utterances of different lengths side by side in padded [B][L] buffers.  lengths [B] (device int32, may be NULL) =
 * samples of each row; frames [B] = STFT frames of each row, 1 + lengths / 256.  Everything behind a row's own end is written as
 * zeros and never read.  The same convention holds for every *_var entry point below. */
int nele_stft_band_var(const float* wav, const int* lengths, int B, int L, float power, void* spec, float* band, void* stream);

/* noise_est/imcra.py:521-577 imcra_est.estimate + :363-484 imcra.update; audio_util.py:113-117 NoisePSD,
 * :439-456 Sp_and_phase_Noise.  spec [B][T][257] complex64 -> psd [B][T][257] f32 (may be NULL),
 * band [B][T][64] f32 = compute_band_E(sqrt(psd))**power (may be NULL). */
int nele_imcra_band(const void* spec, int B, int T, float power, float* psd, float* band, void* stream);
int nele_imcra_band_var(const void* spec, const int* frames, int B, int T, float power, float* psd, float* band, void* stream);
/* The same recursion without a workgroup-wide serial loop (round 6): the frequency smoothing is the only thing that couples bins, and
 * what it smooths exists ahead of the recursion that consumes it - |Y|^2 for all frames at once, then one THREAD per utterance and bin for
 * S / S_min / the speech indicator (imcra.py:363-412), then one thread per utterance and bin for S~ / S~_min / the speech-absence prior q and
 * the tracker (Gamma, xi, G, p, lambda_D: imcra.py:413-484, 543-557, 22-36).  psd (required) and band are BIT-identical to
 * nele_imcra_band_var's.  workspace: device scratch of nele_imcra_workspace_bytes(B, T) (5 bytes per frame and bin), 16-byte aligned. */
long long nele_imcra_workspace_bytes(int B, int T);
/* The noise file's features without its spectrum in memory: nele_stft_pow_var = nele_stft_band_var + pw [B][T][257] float32 =
 * np.abs(STFT) ** 2 (float32, as audio_util.py:113-117 / imcra.py:523-527 see it; frames behind a short row's end are not written);
 * nele_imcra_band_pw = nele_imcra_band_ws starting from pw.  Bit-identical psd / band. */
int nele_stft_pow_var(const float* wav, const int* lengths, int B, int L, float power, void* spec, float* band, float* pw, void* stream);
int nele_imcra_band_pw(const float* pw, const int* frames, int B, int T, float power, float* psd, float* band, void* workspace,
                       long long workspace_bytes, void* stream);
int nele_imcra_band_ws(const void* spec, const int* frames, int B, int T, float power, float* psd, float* band, void* workspace,
                       long long workspace_bytes, void* stream);

/* audio_util.py:30-50 compute_band_E by itself: magnitude spectrogram mag [N][257] f32 -> band [N][64] f32 (no power law). */
int nele_compute_band_E(const float* mag, int N, float* band, void* stream);

/* audio_util.py:93-110 interp_band_gain by itself: bandE [N][64] f32 -> g [N][257] f64 (bins 0, 1 = 1e-4, bin 256 = 1e-2). */
int nele_interp_band_gain(const float* bandE, int N, double* g, void* stream);

/* audio_util.py:93-110 interp_band_gain, :76-90 Resyn, :458-461 SP_to_wav, :60-65 ISTFT.
 * alpha2 [B][T][64] f32 (energy gains; NULL = plain ISTFT, no gain), spec [B][T][257] complex64 -> wav [B][256*(T-1)] f32. */
int nele_gain_istft(const float* alpha2, const void* spec, int B, int T, float* wav, void* stream);
int nele_gain_istft_var(const float* alpha2, const void* spec, const int* frames, int B, int T, float* wav, void* stream);

/* inference.py:109 (enh / rms(enh) * target_rms, skipped when target_rms <= 0) and the PCM_16
 * write/read round trip of train_nele.py:313 + dataloader.py:58 (pcm16 != 0).  In place on wav [B][N].
 * workspace: nele_wav_post_workspace_doubles(B, N) float64 values (per-chunk sums of squares; only read when target_rms > 0, else NULL). */
long long nele_wav_post_workspace_doubles(int B, int N);
int nele_wav_post(float* wav, int B, int N, float target_rms, int pcm16, double* workspace, void* stream);
int nele_wav_post_var(float* wav, const int* frames, int B, int N, float target_rms, int pcm16, double* workspace,
                      void* stream);   /* row b has 256 (frames[b] - 1) samples */

/* ---- dense layers: convolution as implicit GEMM on the f32 matrix cores (csrc/dense.hip) ------- */

/* ConvGeom, passed as 15 ints {H, W, C, ih0, iw0, Hout, Wout, seglen=KW*C, segstride=W*C, Ktot=KH*KW*C,
 * OH, OW, OC, oh0, ow0}: channels-last input buffer [B][H][W][C]; output position (b,ho,wo) reads the
 * window whose origin is (ho+ih0, wo+iw0) and writes element (ho+oh0, wo+ow0, n) of [B][OH][OW][OC].
 *
 * nele_conv_gemm: out = epi(sum_kk A_view[m][kk] * Wg[n][kk]), m = (b,ho,wo), M = B*Hout*Wout.
 * Replaces torch.nn.Conv1d + Chomp1d (model.py:10-40, 49-77; H = 1 on a left-padded time axis),
 * torch.nn.Conv2d (model.py:105-109), torch.nn.Linear (model.py:81-82, 95-97) and, run over a
 * zero-bordered gradient buffer with flipped weights, their data gradients (autograd in the reference).
 * epi: 0 none, 1 +bias, 2 +bias then LeakyReLU(slope), 3 multiply by LeakyReLU'(aux) (aux = forward
 * activation, unpadded [M][OC]), 4 +bias then exp(3.2*tanh(.)) (model.py:98). */
int nele_conv_gemm(const float* A, const float* Wg, const float* bias, const float* aux, float* out, int M, int N,
                   int epi, float slope, const int* geom_host, void* stream);

/* nele_conv_gemm_bf16: nele_conv_gemm with bf16 MFMA operands (inputs rounded while staged, float32 accumulation), N > 48. */
int nele_conv_gemm_bf16(const float* A, const float* Wg, const float* bias, const float* aux, float* out, int M, int N,
                        int epi, float slope, const int* geom_host, void* stream);

/* Span-staged variant of nele_conv_gemm for 2-D convolutions with long output rows (Wout >= 64, N <= 64): per kernel
 * row the input span of a block's 256 output positions is staged in LDS once and re-read KW times.  Wfrag is Wg
 * [N][Ktot] re-ordered by nele_weight_prep_frag to [Ktot/8][ceil(N/16)][64 lanes][2] (one coalesced load per MFMA
 * operand).  a_elems = elements of the input buffer (32-bit offsets are used inside).  Same epilogues. */
int nele_weight_prep_frag(const float* Wg, int N, int Ktot, float* Wfrag, void* stream);
int nele_conv_span_supported(int M, int N, const int* geom_host, int KH, int KW);
int nele_conv_span(const float* A, const float* Wfrag, const float* bias, const float* aux, float* out, int M, int N, int epi,
                   float slope, const int* geom_host, int KH, int KW, long long a_elems, void* stream);

/* Weight gradient dW[n][ci][kh][kw] (+)= sum_m dOut[m][n] * A_view[m][(kh,kw,ci)], db[n] (+)= sum_m dOut[m][n]
 * (autograd of the layers above).  Split over m across workgroups, partials summed in fixed order.
 * workspace_floats >= nele_conv_wgrad_workspace_floats(M, N, Ktot, &splits). */
long long nele_conv_wgrad_workspace_floats(int M, int N, int Ktot, int* splits_out_host);
/* nele_conv_wgrad_bf16: same contract with bf16 MFMA operands (inputs are rounded to bf16 while staged; float32 accumulation). */
int nele_conv_wgrad_bf16(const float* A, const float* dOut, float* workspace, long long workspace_floats, int M, int N,
                         const int* geom_host, int KH, int KW, int Cvalid, float* dW, float* db, int accumulate, void* stream);
/* bf16-operand variants of the span-staged convolution (float32 in / out, bf16 MFMA operands, float32 accumulation):
 * weights re-ordered by nele_weight_prep_frag16 into nele_weight_frag16_elems(N, KW*C, KH) bf16 elements. */
long long nele_weight_frag16_elems(int N, int seglen, int KH);
int nele_weight_prep_frag16(const float* Wg, int N, int Ktot, int seglen, int KH, void* Wfrag, void* stream);
/* Batched variants: every layer of a model in one launch each.  ptrs / dims are HOST arrays (as for nele_spectral_norm):
 * prep: ptrs [4*jobs] = (Wt, sigma or NULL, Wf, Wb or NULL), dims [5*jobs] = (N, Cvalid, C, KH, KW);
 * frag16: ptrs [2*jobs] = (Wg, Wfrag), dims [4*jobs] = (N, Ktot, seglen, KH).  jobs <= 16. */
int nele_weight_prep_batch(const void* const* ptrs_host, const int* dims_host, int jobs, void* stream);
int nele_weight_prep_frag16_batch(const void* const* ptrs_host, const int* dims_host, int jobs, void* stream);
int nele_conv_span_bf16_supported(int M, int N, const int* geom_host, int KH, int KW);
int nele_conv_span_bf16(const float* A, const void* Wfrag, const float* bias, const float* aux, float* out, int M, int N, int epi,
                        float slope, const int* geom_host, int KH, int KW, long long a_elems, void* stream);
int nele_conv_wgrad(const float* A, const float* dOut, float* workspace, long long workspace_floats, int M, int N,
                    const int* geom_host, int KH, int KW, int Cvalid, float* dW, float* db, int accumulate, void* stream);
/* The pooling gradient of D's last conv layer (autograd of model.py:124-126) stored as bf16: its two consumers - the layer's data and
 * weight gradients - round their operands to bf16 anyway, and the data gradient re-stages it once per kernel row.  Same layouts and zero
 * border as the float32 buffer; only for geometries that run on the span kernel / the 2-D weight-gradient tile kernel
 * (nele_conv_span_bf16_a16_supported; NELE_ERR_UNSUPPORTED otherwise). */
int nele_conv_span_bf16_a16_supported(int M, int N, const int* geom_host, int KH, int KW);
int nele_conv_span_bf16_a16(const void* A16, const void* Wfrag, const float* bias, const float* aux, float* out, int M, int N, int epi,
                            float slope, const int* geom_host, int KH, int KW, long long a_elems, void* stream);
int nele_conv_wgrad_bf16_d16_supported(int M, int N, const int* geom_host, int KH, int KW);
int nele_conv_wgrad_bf16_d16(const float* A, const void* dOut16, float* workspace, long long workspace_floats, int M, int N,
                             const int* geom_host, int KH, int KW, int Cvalid, float* dW, float* db, int accumulate, void* stream);
/* ... and with the input activation stored as bfloat16 too (round 3: every activation of D.conv1-4 and every output gradient of
 * D.conv2-5 lives in memory as bf16 in bf16 mode).  Same kernel, same bf16 operands, same accumulation order as the float32-buffer
 * forms: the weight gradient is bit-identical, the bias gradient sums the bf16-rounded output gradient. */
int nele_conv_wgrad_bf16_a16d16(const void* A16, const void* dOut16, float* workspace, long long workspace_floats, int M, int N,
                                const int* geom_host, int KH, int KW, int Cvalid, float* dW, float* db, int accumulate, void* stream);

/* ---- Conv2d on bfloat16 activations (csrc/conv16.hip): model.py:105-109, 118-122 F.conv2d + LeakyReLU(0.3) and its autograd -------
 * The discriminator's conv2..conv5, forward and data gradient, bf16 MFMA operands / float32 accumulate, with the operands READ AS
 * bf16 FROM MEMORY: A16 [B][H][W][C] bf16 channels-last (C a multiple of 8, <= 64), out [B][OH][OW][OC] bf16 (out_bf16 != 0) or
 * float32, aux16 [B][Hout][Wout][N] bf16 = the forward activation whose sign gates a data gradient (EPI_MASK_LRELU_GRAD).
 * Wfrag: fragment stream of nele_conv16_weight_prep_batch (nele_conv16_wfrag_elems bf16 elements) made from the float32 GEMM layouts
 * nele_weight_prep writes (forward [N][KH][KW][C], or the flipped data-gradient layout).  geom_host as for nele_conv_gemm; the data
 * gradient is the forward kernel over the zero-bordered output-gradient buffer.  One tile = 4 (or 8) output rows x 64 columns x all N
 * channels; the input halo is a ring of tile-rows + 1 rows in LDS filled by global->LDS DMA, weights stream through a two-slot LDS
 * ring, two workgroups share a CU.  nele_conv16_supported: 0 = use nele_conv_span_bf16 / nele_conv_gemm_bf16 (KH or KW == 1, C < 8,
 * N > 64, or NELE_CONV16=0). */
int nele_conv16_supported(int M, int N, const int* geom_host, int KH, int KW);
long long nele_conv16_wfrag_elems(int N, int seglen, int KH);
/* jobs <= 16; ptrs_host[2 j] = Wg float32 [N][Ktot] (device), ptrs_host[2 j + 1] = Wfrag bf16 (device); dims_host[4 j ..] = {N, Ktot, seglen, KH} */
int nele_conv16_weight_prep_batch(const void* const* ptrs_host, const int* dims_host, int jobs, void* stream);
int nele_conv16(const void* A16, const void* Wfrag, const float* bias, const void* aux16, void* out, int out_bf16, int M, int N, int epi,
                float slope, const int* geom_host, int KH, int KW, void* stream);
/* The last conv layer with the pooling fused into its epilogue (model.py:109,121-123: Conv2d -> LeakyReLU -> AdaptiveAvgPool2d(1)): out16
 * [B][Hout][Wout][N] bf16 = LeakyReLU(conv + bias) (kept for the backward pass's LeakyReLU mask); gap_part [B][parts][N] float64 = sums of the
 * float32 LeakyReLU outputs over each wave's positions with column < wvalid[b] (wvalid [B] int32 device memory or NULL: every column);
 * parts = nele_conv16_gap_parts(N, geom, KH, KW) (0: unsupported geometry).  nele_gap_mlp_fwd_parts adds them in a fixed order. */
int nele_conv16_gap_parts(int N, const int* geom_host, int KH, int KW);
int nele_conv16_gap(const void* A16, const void* Wfrag, const float* bias, void* out16, int M, int N, float slope, const int* geom_host, int KH,
                    int KW, const int* wvalid, double* gap_part, void* stream);
/* The discriminator's first layer (1 x 1 Conv2d, 3 (+1 zero) -> 8 channels, model.py:105): in [M][4] float32 (the packed D input),
 * Wf [8][4] float32 (nele_weight_prep's forward layout), out16 [M][8] bf16 = LeakyReLU(W in + bias), exact float32 arithmetic. */
int nele_conv16_pointwise_fwd(const float* in, const float* Wf, const float* bias, void* out16, long long M, int N, float slope, void* stream);

/* PyTorch parameter layout [N][Cvalid][KH][KW] (optionally / sigma[0]) -> GEMM layouts:
 * Wf[n][kh][kw][c] (c zero-padded to C) and, if Wb != NULL, the flipped data-gradient layout
 * Wb[c][KH-1-kh][KW-1-kw][n]. */
int nele_weight_prep(const float* Wt, const float* sigma, int N, int Cvalid, int C, int KH, int KW, float* Wf, float* Wb,
                     void* stream);

/* ---- generator glue (csrc/gen.hip) -------------------------------------------------------------- */

/* model.py:85-86: cat(x,y) [B][T][64]x2 -> left-padded conv input [B][T+pad][128]. */
int nele_g_pack(const float* x, const float* y, float* out, int B, int T, int pad, void* stream);

/* model.py:168-205 cLN followed by LeakyReLU (model.py:88-91): Y [B][T][C] -> out (rows pad.. of
 * [B][T+pad][C]); saves the cumulative mean / 1/std per frame for the backward pass. */
int nele_cln_chunks(int T);  /* frame chunks per utterance: rows of the partial buffers = B * nele_cln_chunks(T) */
int nele_cln_fwd(const float* Y, const float* gain, const float* bias, float* out, float* mean, float* rstd, double* scratch,
                 int B, int T, int C, int pad, float slope, void* stream);   /* scratch: float64 [B][T][2] */
/* Backward of the above: dAct [B][T][C] -> dY (rows 0..T-1 of the END-padded [B][T+pade][C]) and / or dY16 (the same values as bf16 in a
 * buffer of the same layout: the operand of nele_glayer16_conv's data gradient and of nele_conv_wgrad_bf16_a16d16; either may be NULL,
 * not both), gain/bias gradient partials
 * [B * nele_cln_chunks(T)][C] (reduce with nele_colsum). */
int nele_cln_bwd(const float* dAct, const float* Y, const float* gain, const float* bias, const float* mean,
                 const float* rstd, float* dY, void* dY16, float* dgain_part, float* dbias_part, double* scratch, int B, int T, int C,
                 int pade, float slope, void* stream);
int nele_colsum(const float* part, int rows, int cols, float* out, int accumulate, void* stream);
/* two column sums of equal shape in one launch (cLN gain and bias partials) */
int nele_colsum2(const float* part0, float* out0, const float* part1, float* out1, int rows, int cols, int accumulate, void* stream);

/* ---- generator layers on bf16 activations (csrc/glayer.hip), bf16 mode --------------------------------------------------------
 * One launch per layer of model.py:83-91: causal Conv1d (Chomp1d = K - 1 zero rows in front of the time axis) + bias + cumulative layer
 * norm (model.py:168-205) + LeakyReLU, bf16 MFMA operands / float32 accumulate, statistics in float64.  Layers: Cin a multiple of 64,
 * N = 64 or 256 output channels, K <= 9 (nele_glayer16_supported); everything else stays on nele_conv_span_bf16 + nele_cln_fwd. */
int nele_glayer16_supported(int Cin, int N, int K);
long long nele_glayer16_wfrag_elems(int Cin, int N, int K);      /* bf16 elements of a layer's weight fragment stream */
long long nele_glayer16_carry_bytes(int B, int T);               /* strip-carry workspace (T > 256 frames); zero it once */
/* jobs <= 16; ptrs_host[2 j] = Wg float32 [N][K * Cin] in k order (tap, channel) (nele_weight_prep's forward layout, or the flipped one for
 * a data gradient), ptrs_host[2 j + 1] = the fragment stream; dims_host[3 j ..] = {N, Cin, K} */
int nele_glayer16_weight_prep_batch(const void* const* ptrs_host, const int* dims_host, int jobs, void* stream);
/* model.py:85-86 cat(x, y) as the first layer's bf16 input [B][T + pad][128] (rows < pad stay zero) */
int nele_g_pack16(const float* x, const float* y, void* out16, int B, int T, int pad, void* stream);
/* A16 [B][T + K - 1][Cin] bf16 -> out16 [B][T + padn][N] bf16 (rows padn ..: the next layer's input).  Optional outputs (NULL = not
 * written): Y [B][T][N] float32 = convolution + bias, mean / rstd [B][T] (what nele_cln_bwd reads), out32 = the activation in float32,
 * same layout as out16.  carry: nele_glayer16_carry_bytes(B, T) zero-initialised bytes, read and written when T > 256; token: a non-zero
 * value that differs from call to call on the same carry buffer (a counter). */
int nele_glayer16_fwd(const void* A16, const void* Wfrag, const float* bias, const float* gain, const float* beta, float* Y, float* mean,
                      float* rstd, void* out16, float* out32, void* carry, unsigned token, int B, int T, int Cin, int N, int K, int padn,
                      float slope, void* stream);
/* The convolution alone, float32 result [B][T][N] (no bias): a layer's data gradient = this over the END-padded bf16 output gradient
 * [B][T + K - 1][Cin = the layer's output channels] with the flipped weights (N = the layer's input channels). */
int nele_glayer16_conv(const void* A16, const void* Wfrag, float* out32, int B, int T, int Cin, int N, int K, void* stream);

/* Gradient of model.py:98 exp(3.2*tanh(o)) given the mask itself. */
int nele_exptanh_bwd(const float* dmask, const float* mask, float* dout, long long n, void* stream);

/* train_nele.py:133-146 (per utterance): beta2 = sum(clean^inv_p) / sum(mask*clean^inv_p);
 * din [B][64][T][4] = (clean*mask^p*beta2^p, noise, clean, 0) channels-last D input (may be NULL);
 * alpha2 = mask*beta2 (train_nele.py:307, inference.py:104; may be NULL); s2 = sum(mask*clean^inv_p). */
int nele_energy_norm_fwd(const float* clean, const float* mask, const float* noise, float p, float inv_p, float* beta2,
                         float* s2, float* din, float* alpha2, int B, int T, void* stream);
/* Gradient w.r.t. the mask given d(din).  din = the forward pass's output (may be NULL: the enhanced band is then recomputed). */
int nele_energy_norm_bwd(const float* clean, const float* mask, const float* beta2, const float* s2, const float* ddin, const float* din,
                         float p, float inv_p, float* dmask, int B, int T, void* stream);

/* dataloader.py:76-84: band features (enhanced, noise, clean) [B][T][64] -> D input [B][64][T][4]
 * (c2 NULL for Discriminator_Quality's (enhanced, clean)). */
int nele_d_pack(const float* c0, const float* c1, const float* c2, float* din, int B, int T, void* stream);
/* dataloader.py:54-84 + train_nele.py:349-367, batched: a shuffled list of per-utterance D items -> one padded batch.  items_host [n]: HOST
 * array of device pointers to channels-last items [64][T_k][4] float32 whose band rows lie strides_host[i] floats apart (NULL: 4 T_k, a
 * contiguous item; a row of a larger padded batch is a valid item), frames_host [n] = T_k <= Tm.  din_out [rows][64][Tm][4]: item r in row r,
 * columns >= T_k and rows >= n zero; frames_out [rows] (device, may be NULL) = T_k, Tm for the fill rows.  One launch per 64 items. */
int nele_d_gather_items(const void* const* items_host, const int* frames_host, const long long* strides_host, int n, int rows, int Tm,
                        float* din_out, int* frames_out, void* stream);
/* Reference tensor layout [B][Cin][64][T] (model.py:118) <-> channels-last [B][64][T][4]. */
int nele_d_layout(const float* src, float* dst, int B, int Cin, int T, int to_nhwc, void* stream);

/* ---- discriminator glue + optimiser (csrc/disc.hip) --------------------------------------------- */

/* torch.nn.utils.spectral_norm as used at model.py:105-116, all layers of a discriminator in one launch: n_iter (1 in
 * train mode, 0 in eval) power iterations updating u [N], v [K] in place, then sigma[l] = u . (W v); W = weight_orig as
 * [N][K].  ptrs_host: HOST array of 3*layers device pointers {W, u, v}; dims_host: HOST array {N, K} per layer. */
int nele_spectral_norm(const void* const* ptrs_host, const int* dims_host, int layers, float* sigma, int n_iter, void* stream);
/* dst (+)= (dWsn - <dWsn, W/sigma> u v^T) / sigma : gradient through W/sigma with u, v constant. */
int nele_sn_grad_scratch_doubles(int total);
int nele_sn_grad(const float* dW, const float* W, const float* u, const float* v, const float* sigma, int N, int K,
                 float* dst, int accumulate, double* scratch, void* stream);

/* model.py:123-132: AdaptiveAvgPool2d(1) over act [B][P][64] + fc1/fc2/fc3 (spectral-norm Linear) +
 * LeakyReLU + sigmoid.  mlp_host: HOST array of 9 device pointers {w1,b1,sigma1,w2,b2,sigma2,w3,b3,sigma3}. */
int nele_gap_mlp_fwd(const float* act, int B, int P, const float* const* mlp_host, int nout, float slope, float* pooled,
                     float* h1, float* h2, float* score, double* scratch /* float64 [B][32][64] */, void* stream);
/* Backward of the head: dscore [B][nout] -> dz3, dz2, dz1, dpooled and (gbuf != NULL) the gradient of
 * the last conv activation written into the zero-bordered buffer [B][OH][OW][64] at (oh0, ow0). */
int nele_gap_mlp_bwd(const float* dscore, const float* score, const float* h1, const float* h2, const float* act,
                     const float* const* mlp_host, int nout, float slope, int B, int Hout, int Wout, int OH, int OW, int oh0,
                     int ow0, float* dz3, float* dz2, float* dz1, float* dpooled, float* gbuf, void* stream);
/* The same for a padded batch of utterances of different lengths: wvalid [B] (device int32, may be NULL) = valid output columns of
 * each utterance in the last conv layer's output (frames - 20); the pooling mean and its gradient run over those columns only, as
 * the batch-1 reference pools every utterance over its own extent (model.py:123). */
int nele_gap_mlp_fwd_var(const float* act, int B, int P, int Wout, const int* wvalid, const float* const* mlp_host, int nout, float slope,
                         float* pooled, float* h1, float* h2, float* score, double* scratch, void* stream);
int nele_gap_mlp_bwd_var(const float* dscore, const float* score, const float* h1, const float* h2, const float* act,
                         const float* const* mlp_host, int nout, float slope, int B, int Hout, int Wout, const int* wvalid, int OH, int OW,
                         int oh0, int ow0, float* dz3, float* dz2, float* dz1, float* dpooled, float* gbuf, void* stream);
/* ... with the pooling gradient written as bf16 (see nele_conv_span_bf16_a16) */
int nele_gap_mlp_bwd_var16(const float* dscore, const float* score, const float* h1, const float* h2, const float* act,
                           const float* const* mlp_host, int nout, float slope, int B, int Hout, int Wout, const int* wvalid, int OH, int OW,
                           int oh0, int ow0, float* dz3, float* dz2, float* dz1, float* dpooled, void* gbuf16, void* stream);
/* ... with the last conv layer's activation as bf16 as well (nele_conv16_gap's out16) */
int nele_gap_mlp_bwd_var16a(const float* dscore, const float* score, const float* h1, const float* h2, const void* act16,
                            const float* const* mlp_host, int nout, float slope, int B, int Hout, int Wout, const int* wvalid, int OH, int OW,
                            int oh0, int ow0, float* dz3, float* dz2, float* dz1, float* dpooled, void* gbuf16, void* stream);
/* The head of nele_gap_mlp_fwd_var alone, on pooled partial sums [B][nparts][64] float64 the producing conv kernel wrote (nele_conv16_gap) */
int nele_gap_mlp_fwd_parts(const double* part, int nparts, int B, int P, int Wout, const int* wvalid, const float* const* mlp_host, int nout,
                           float slope, float* pooled, float* h1, float* h2, float* score, void* stream);
int nele_mlp_wgrad(const float* dz, const float* x, int B, int N, int K, float* dW, float* db, void* stream);

/* torch.optim.Adam (train_nele.py:89-91) on flat buffers; step counts from 1. */
int nele_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                   int step, void* stream);
/* The same update, skipped as a whole when any element of g is not finite (the reference raises from pysiib / pyhaspi2.py:357-358
 * before such a target ever reaches the optimiser; here the step is masked on the device and counted).
 * guard: device int[2], zeroed once by the caller = {last step with a non-finite gradient, number of skipped steps}. */
int nele_adam_step_guarded(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                           int step, int* guard, void* stream);

/* ---- batched objective metrics (csrc/estoi.hip, csrc/siib.hip, csrc/haspi.hip) --------------------
 * Replace audio_util.py:120-203 read_batch_{STOI,SIIB,HASPI} (32 joblib processes over wav files) and
 * the intel.py wrappers they call.  x = clean [B][L], y = degraded (enhanced + noise, audio_util.py:139-141)
 * [B][L], float32 at 16 kHz, equal lengths (the caller truncates to the common length, intel.py:58-60).
 * raw [B] = metric value, mapped [B] = its logistic map to [0,1] (intel.py:102-106, 116-120, 136-140);
 * either may be NULL.  workspace: device scratch of at least *_workspace_bytes(B, L). */

/* intel.py:122-134 ESTOI_Wrapper[_raw]_harvard -> pystoi.stoi(x, y, 16000, extended=True) (algorithm
 * restated, oracle/estoi.py).  Fewer than 30 frames after silence removal -> raw = 1e-5 as pystoi. */
long long nele_metric_estoi_workspace_bytes(int B, int L);
int nele_metric_estoi(const float* x, const float* y, int B, int L, void* workspace, long long workspace_bytes, float* raw,
                      float* mapped, void* stream);
int nele_metric_estoi_var(const float* x, const float* y, const int* lengths, int B, int L, void* workspace, long long workspace_bytes,
                          float* raw, float* mapped, void* stream);

/* intel.py:57-100 SIIB_Wrapper[_raw]_harvard: VAD (intel.py:37-50), replication rule (intel.py:93-97) and
 * pysiib.SIIB(x, y, 16000, gauss=True) (algorithm restated, oracle/siib.py).  info [B][4] (may be NULL) =
 * {replication factor M, frames of the tiled signal, active frames, status bits: 1 M clamped, 4 active-frame
 * buffer clamped, 8 not enough active frames (reference raises; raw = NaN), 32 the utterance's covariance took the eigensolver's repair
 * path (nele_eigh_repaired: score unaffected, the call was slower)}.  The KLT eigenvectors come from
 * nele_eigh_sym_batched below. */
long long nele_metric_siib_workspace_bytes(int B, int L);
int nele_metric_siib(const float* x, const float* y, int B, int L, void* workspace, long long workspace_bytes, float* raw,
                     float* mapped, int* info, void* stream);
/* Same, split so that the caller can interleave independent work: phase 1 = wide front kernels (VAD .. covariance),
 * phase 2 = latency-bound back end (eigenvectors, projections, score) on the same workspace; phase 0 = both.
 * Alternative split by data dependence: phase 3 = everything that needs only the CLEAN signal x (VAD, active frames, x spectra /
 * masking / stacking, covariance and its eigen-decomposition - SIIB's KLT basis is the clean signal's; y may be NULL),
 * phase 4 = the rest (y spectra / masking / stacking, projections, score).  Phase 3 can run before y exists. */
int nele_metric_siib_phase(const float* x, const float* y, int B, int L, void* workspace, long long workspace_bytes, float* raw,
                           float* mapped, int* info, int phase, void* stream);
/* Clean-signal state across calls.  The reference scores the SAME clean training files in every GAN epoch (train_nele.py:35-38,119,
 * 318-340 -> audio_util.py:120-203 -> intel.py:57-100) and recomputes their half of SIIB - VAD, clean spectra, covariance, the KLT
 * eigen-decomposition (np.linalg.eigh in pysiib) - every time.  This describes what phase 3 leaves in a workspace for phase 4 as byte
 * ranges of the workspace, out[3k..3k+2] = {offset, stride, bytes}: stride > 0 = per-utterance section (utterance b's part = the first
 * `bytes` bytes at offset + b * stride), stride 0 = a table all utterances share.  A caller may copy the ranges out after phase 3 and, in
 * a later call with the same L, copy them in (any row order, any B) INSTEAD of running phase 3: phase 4 then gives bit-identical scores.
 * Returns the number of sections (max_sections >= 13) or a negative status.  Host-only: no device work. */
int nele_metric_siib_clean_sections(int B, int L, long long* out, int max_sections);
/* Same with per-utterance lengths (the frame-periodic shortcut for L % 200 == 0 is not taken then; results are identical). */
int nele_metric_siib_var(const float* x, const float* y, const int* lengths, int B, int L, void* workspace, long long workspace_bytes,
                         float* raw, float* mapped, int* info, int phase, void* stream);

/* Batched symmetric eigen-decomposition, float64, n <= 512 (np.linalg.eigh in pysiib's KLT): A [B][n][n] symmetric
 * (destroyed) -> lam [B][n] ascending, U [B][n][n] with ROW j = eigenvector j.  Householder tridiagonalisation,
 * Sturm bisection, inverse iteration, back-transformation (csrc/eigh.hip). */
long long nele_eigh_workspace_bytes(int B, int n);
int nele_eigh_sym_batched(double* A, int n, int B, double* lam, double* U, void* workspace, long long workspace_bytes,
                          void* stream);
/* Matrices of the last nele_eigh_sym_batched call on `workspace` whose cluster tridiagonalisation gave up (its workgroups were not
 * co-resident within the spin limit, e.g. another process holds the GPU) and were redone by the single-workgroup repair kernel.
 * Synchronises the device.  0 on a GPU the launch fits on. */
int nele_eigh_repaired(void* workspace, int B, int n);

/* intel.py:108-114 HASPI_Wrapper[_raw]_harvard -> pyHASPI/pyhaspi2.py:76-107 haspi_v2(x, fs, y, fs), HL = 0.
 * fs_in: any rate up to 24000 Hz (below it the signals are resampled to 24 kHz as librosa.resample = resampy kaiser_best + fix_length,
 * pyhaspi2.py:810-821; above it the reference raises NotImplementedError and so does this call).
 * dither: NULL (no IHC firing jitter) or float64 standard normals [B][2][nsub][32], nsub =
 * nele_metric_haspi_nsub(L, fs_in); row k perturbs the k-th ACTIVE sub-sampled frame exactly as the reference's
 * np.random.randn(n_active, 32) draws (pyhaspi2.py:362-365).  info [B][2] (may be NULL) = {active frames,
 * status: 1 = signal below threshold (reference raises, pyhaspi2.py:357-358; raw = NaN)}. */
long long nele_metric_haspi_workspace_bytes(int B, int L, int fs_in);
int nele_metric_haspi_nsub(int L, int fs_in);
int nele_metric_haspi(const float* x, const float* y, int B, int L, int fs_in, const double* dither, void* workspace,
                      long long workspace_bytes, float* raw, float* mapped, int* info, void* stream);
/* The same with per-utterance lengths and split by data dependence.  lengths [B] (device, may be NULL = every row has L samples):
 * samples of each utterance in the padded [B][L] buffers - the reference scores files of any length one at a time
 * (audio_util.py:134-141, intel.py:58-60), a batch carries them side by side.  phase 0 = everything; phase 3 = everything that
 * needs only the CLEAN signal x (its whole ear model, envelope filter, silence gate, group-delay shifts, cepstra, modulation
 * filtering; y, raw, mapped may be NULL); phase 4 = the degraded signal's chain + correlation + score on the same workspace
 * (x may be NULL).  Phase 0 runs 3 then 4, so the split is bit-identical to the one-shot call. */
int nele_metric_haspi_var(const float* x, const float* y, const int* lengths, int B, int L, int fs_in, const double* dither,
                          void* workspace, long long workspace_bytes, float* raw, float* mapped, int* info, int phase, void* stream);
/* The same for HASPI's reference-signal half (pyhaspi2.py:76-107: eb_EarModel / eb_EnvFilt / eb_melcor9's reference side of the clean
 * file, recomputed by the reference every epoch): byte ranges {offset, stride, bytes} of what phase 3 leaves for phase 4.  Valid for the
 * same L, fs_in, audiogram and dither rows.  max_sections >= 16. */
int nele_metric_haspi_clean_sections(int B, int L, int fs_in, long long* out, int max_sections);
/* pyhaspi2.py:362-365: the reference dithers the envelopes in EVERY haspi_v2 call with np.random.randn(n_active, 32) rows (x, then y)
 * from numpy's global generator.  This fills the `dither` argument of nele_metric_haspi* with rows that are a pure function of
 * (seed, utt_ids[b], signal, active-frame index, channel): the same on whichever rank / in whichever batch the utterance is scored
 * (SURVEY 8e).  utt_ids [B] int64 (device), out [B][2][nsub][32] float64 (device), nsub = nele_metric_haspi_nsub(L, fs_in). */
int nele_haspi_dither_rows(const long long* utt_ids, unsigned long long seed, int B, int nsub, double* out, void* stream);
/* haspi_v2(x, fx, y, fy, HL) for a hearing-impaired listener (pyhaspi2.py:76-107 with eb_LossParameters :779-807 and the HLx / HL
 * split of eb_EarModel :1155-1166).  hl6_host: HOST pointer to the audiogram at 250, 500, 1000, 2000, 4000, 6000 Hz in dB HL (NULL =
 * normal hearing); itype 0 = the reference signal is heard with normal hearing (haspi_v2, haspi), 2 = both signals with the loss
 * (hasqi_v2); itype 1 (NAL-R) raises NotImplementedError in the reference itself (eb_NALR :830-831): NELE_ERR_UNSUPPORTED. */
int nele_metric_haspi_var_hl(const float* x, const float* y, const int* lengths, int B, int L, int fs_in, const double* dither,
                             const double* hl6_host, int itype, void* workspace, long long workspace_bytes, float* raw, float* mapped,
                             int* info, int phase, void* stream);

/* pyHASPI/pyhaspi2.py:109-157 haspi(x, fx, y, fy, HL, alpha) (HASPI version 1) and :32-74 hasqi_v2(x, fx, y, fy, HL), HL = 0: same ear
 * model plus the basilar-membrane motion (pyhaspi2.py:897, :997, :1076, :1087), 16 ms raised-cosine segments (eb_EnvSmooth :674-703),
 * cepstral correlation (eb_melcor :706-751), segment cross-covariance of the BM motion (eb_BMcovary :550-657), its three-level and
 * synchrony averages (eb_3LevelCovary :416-547, eb_AveCovary2 :160-220), long-term spectral differences (eb_aveSL :1135-1152,
 * eb_SpectDiff :222-251).  One launch chain scores both.  x = reference, y = processed, [B][L] float32; lengths [B] or NULL.
 * noise != 0: eb_BMaddnoise (pyhaspi2.py:1091-1095: N(0, 10^(-75/20)) on every BM sample, numpy's global generator in the reference)
 * from a counter-based generator seeded with `seed`; 0 = none (deterministic).  alpha: logistic slope of haspi() (default -1).
 * out [B][12] float64 = {HASPI v1 Intel, CepCorr, cov3 low, mid, high, HASQI Combined, Nonlin, Linear, BMsync5, Dloud, Dslope, avecov};
 * info [B][4] (may be NULL) = {segments above threshold in eb_melcor, status, segments above threshold in the covariance stages,
 * histogram bins}; status bit 1: eb_melcor's 'Signal below threshold' (reference raises, pyhaspi2.py:723-724), bit 2: the covariance
 * stages' (:427-428; eb_AveCovary2 returns (0, 0) and hasqi_v2 fails on it), bit 4: more than 2048 histogram bins; the affected values
 * are NaN. */
long long nele_metric_haspi_quality_workspace_bytes(int B, int L, int fs_in);
int nele_metric_haspi_quality(const float* x, const float* y, const int* lengths, int B, int L, int fs_in, int noise,
                              unsigned long long seed, double alpha, void* workspace, long long workspace_bytes, double* out, int* info,
                              void* stream);
/* The same for a hearing-impaired listener (hl6_host, itype as for nele_metric_haspi_var_hl): itype 0 = haspi() (columns 0-4 of out are
 * that call's results), itype 2 = hasqi_v2() (columns 5-11): with a loss the two ear models differ, one call serves one of them. */
int nele_metric_haspi_quality_hl(const float* x, const float* y, const int* lengths, int B, int L, int fs_in, int noise,
                                 unsigned long long seed, double alpha, const double* hl6_host, int itype, void* workspace,
                                 long long workspace_bytes, double* out, int* info, void* stream);

/* ---- evaluation path (csrc/reverb.hip) ---------------------------------------------------------------------- */

/* eval_metrics.py:132,137 scipy.signal.lfilter(h, [1], x): room impulse response h [Lh] (float64) applied to x [B][L] (float32),
 * y [B][L] float64 (lfilter promotes: a = [1] is an integer array), terms added from the oldest tap to the newest. */
int nele_fir_filter(const float* x, int B, int L, const double* h, int Lh, double* y, void* stream);

/* eval_metrics.py:104,133-134,138-139,143-144 + audio_util.py:67-74: v = a (+ add); v = v / rms(v) * target_rms when target_rms > 0;
 * clip(v) (divide by 1.05, 1.10, ... while max >= 1 or min < -1).  Exactly one of a64 / a32 [B][N]; add [B][N] or NULL;
 * out (float32) and / or out64; nclip [B] (may be NULL) = number of clip divisions applied. */
int nele_norm_clip(const double* a64, const float* a32, const float* add, int B, int N, double target_rms, float* out, double* out64,
                   int* nclip, void* stream);

#ifdef __cplusplus
}
#endif
#endif
