// Batched HASPI v2, normal-hearing reference (reference intel.py:108-120 -> pyHASPI/pyhaspi2.py:76-107
// haspi_v2 and its call tree :1155-1248 eb_EarModel, :378-414 ebm_EnvFilt, :342-375 ebm_CepCoef,
// :275-339 ebm_ModFilt, :254-273 ebm_ModCorr).  float64 after the middle-ear filter, as the reference.
//
// One wave (64 lanes) per utterance for every recurrence: lane = signal*32 + channel, so the 32
// channels of a signal walk the sample axis together and each step reads / writes one coalesced
// 256-byte row of the [n][32] channel-minor buffers.  Recurrences (IIR sections, demodulator
// rotation, IHC adaptation) are true serial dependences over 1.5*L samples; everything that is
// point-wise (log / pow / sqrt, FIR filters, correlations) runs in wide kernels between them:
//   h1 rms-normalise (float32) + resampy kaiser_best 16 -> 24 kHz (float32 accumulate) + RMS restore
//   h2 middle ear IIR (serial, lanes 0..1 = the two signals)
//   h3 control filter bank: complex demodulation + 2 x 4th-order gammatone IIR -> control envelope,
//      mean square -> bandwidth adjustment (eb_BWadjust)
//   h4 signal filter bank with the adjusted bandwidths -> envelope
//   h5 point-wise OHC compression gain from the control envelope
//   h6 gain low-pass IIR (serial) * envelope
//   h7 point-wise dB SL conversion
//   h8 IHC adaptation (serial) -> envelope in dB SL
//   h9 group-delay shifts, 52-tap Hann FIR + 9:1 sub-sampling
//   h10 silence gate + ordered compaction + dither + 6 cepstral bases + mean removal
//   h11 10 modulation filters (complex demodulation, Hann FIR up to 615 taps) fused with the
//       normalised cross-correlation of reference and processed sequences
//   h12 average over bases 2-6, weighted sum, logistic map
#include "common.h"
#include <type_traits>
#include <cstdlib>

#define HP_NCH 32
#define HP_FS 24000.0
#define HP_LEVEL 65.0
#define HP_NBASIS 6
#define HP_NMOD 10
#define HP_NWIN 32769      // resampy kaiser_best half window: 64 zero crossings * 512 + 1
#define HP_NTAB 512
#define HP_SPACE 9
#define HP_NFILT 52
#define HP_NHALF 26

// Storage type of the two per-sample, per-channel envelope arrays (12.6 GB each in float64 at B = 256, and every stage between the
// filter banks and the envelope low-pass streams them through HBM).  All ARITHMETIC on them stays float64 (filter states, IHC
// adaptation, FIR sums); only the stored |u|^2 / dB values are rounded to float32: 6e-8 relative = 2.6e-7 dB on |u|^2, < 4e-6 dB on
// the dB-SL envelope (values 0..100) - against the 0.1 dB dither the reference itself adds per frame (pyhaspi2.py:362-365), and
// 1e-4 relative on the final score (measured effect: < 1e-7, tests/test_metrics_gpu.py golden + oracle comparisons).
typedef float hp_env_t;

struct HaspiWs {
    double* win;     // [HP_NWIN] resampler half window
    float* r24;      // [B][2][n24]   resampled, RMS-restored (float32 as in the reference)
    double* mid;     // [B][2][n24]   middle-ear output
    hp_env_t* ctl;   // [B][2][n24][32] control envelope |u|^2 (-> compression gain in the unfused diagnostic path)
    hp_env_t* env;   // [B][2][n24][32] signal envelope |u|^2 -> compressed, dB SL -> adapted dB SL
    double* bw;      // [B][2][32]    adjusted bandwidths (x then y)
    double* loss;    // [2][5][32]    eb_LossParameters per signal (x, y) and channel: attnOHC, BWmin, lowknee, CR, attnIHC (haspi_loss_kernel)
    double* ssp;     // [B][2][nchunk][32] control-bank sum-of-squares partials per scan chunk
    double* est;     // [B][2][nchunk][4][64] filter-bank state at the END of each scan chunk, started from a zero state (pass 1)
    double* pmat;    // [1 + B*2][32][16] 4x4 state-transition matrices M^lc per channel: entry 0 control bank, 1 + row signal bank
    double* ihe;     // [B][2][nchunk_g][2][32] IHC adaptation state at the end of each gain-pass chunk from a zero state
    double* pihc;    // [4] IHC state transition over GL_N samples
    double* rsp;     // [B][2][RS_MAXC] resampler: sum of squares of each chunk of outputs
    float* rinfo;    // [B][2][4] {rms of the input, rms of the normalised input, restore gain xRMS / yRMS, -}
    int nchunk, lc;  // scan chunks per row (of the longest row) and their length
    int lcg;         // samples per IHC-adaptation chunk: GL_N when the gain pass is its own kernel, lc when it rides in the signal-bank pass
    int fmul;        // haspi_ihc_fir_kernel walks fmul IHC chunks per thread (its 51-sample back-stepping is per thread)
    double* benv;    // [52] envelope low-pass taps (np.hanning(52) / sum), written by haspi_shift_kernel
    double* bkt;     // [10][616] modulation-filter taps per band, written by haspi_shift_kernel
    int* shift;      // [B][32]
    double* lp;      // [B][2][nlp][32]  low-passed, sub-sampled envelope: row = frame i (lp_raw = 0), or row = group g of haspi_ihc_fir9_kernel
                     //                  with frame i of channel c in row i - di(c) (lp_raw = 1; rows g < 0 are zeros) - read it through hp_lp_at
    int* act;        // [B][nsub]     compacted position of a frame among the active ones, or -1 (serial cepstrum kernel only)
    int* grank;      // [B][nsub]     rank of an active frame inside its block of CP_F frames, or -1 (silence gate)
    int* gcnt;       // [B][ngb]      active frames per block -> exclusive offsets
    double* cpsum;   // [B][2][ngb][6] block partial sums of the cepstral sequences
    double* cmean;   // [B][2][6]     sequence means (subtracted by the modulation filters on load)
    int ngb;         // blocks of CP_F sub-sampled frames per row
    int* info;       // [B][2]        {n_active, status}
    double* cep;     // [B][2][6][nsub] mean-removed cepstral sequences (only the first n_active columns)
    double* cm;      // [B][6][10]    |rho|
    double* xf;      // modulation-filtered reference sequences (clean part -> degraded part): [B][nsub][64] (lane = (basis-1)*10 + band)
                     // from the sliding kernel, [B][5][10][nsub] from the direct one
    double* cpart;   // [B][MS_MAXC][64][5] correlation sums per chunk of outputs
    const int* lens; // [B] samples per utterance at the input rate, or NULL (every row has L samples)
    float* cphi;     // quality path only (haspi_quality.h): [B][2][n24p][32] cosine of the carrier phase, BM motion / envelope
    double* sse;     // quality path only: [B][2][nchunk][32] signal-bank sum-of-squares partials per scan chunk
    int fs_in;
    int n24, nsub;   // of the longest row: buffer strides
    int nlp, lp_raw; // rows per (utterance, signal) block of lp (nsub + 4); layout of lp (see there)
    int n24p;        // n24 rounded up to a multiple of HP_CH: row stride of the per-sample buffers (chunked kernels read/write whole chunks)
};

// Per-utterance lengths inside one padded batch (the reference is batch 1 over files of any length, audio_util.py:134-141)
__device__ __forceinline__ int hp_len(const HaspiWs& ws, int b, int L) { return ws.lens ? min(ws.lens[b], L) : L; }
// Length at 24 kHz of an input of L samples at fs_in < 24 kHz (pyhaspi2.py:815 -> librosa.resample, fix=True): resampy writes
// int(L * ratio) outputs (hp_nres_of), librosa pads with zeros to ceil(L * ratio) (hp_n24_of); ratio = float(24000) / fs_in.
__host__ __device__ static inline int hp_n24_of(int L, int fs_in) { return fs_in == 24000 ? L : (int)ceil((double)L * (24000.0 / (double)fs_in)); }
__host__ __device__ static inline int hp_nres_of(int L, int fs_in) { return fs_in == 24000 ? L : (int)((double)L * (24000.0 / (double)fs_in)); }
__device__ __forceinline__ int hp_n24(const HaspiWs& ws, int b) {
    if (!ws.lens) return ws.n24;
    return min(ws.n24, hp_n24_of(ws.lens[b], ws.fs_in));
}
__device__ __forceinline__ int hp_nsub(const HaspiWs& ws, int b) { return (hp_n24(ws, b) + HP_SPACE - 1) / HP_SPACE; }
// (utterance, signal) row of a launch that covers nsig signals starting at sig0: idx = b * nsig + s
__device__ __forceinline__ int hp_row(int idx, int sig0, int nsig) { return nsig == 2 ? idx : 2 * idx + sig0; }

// Frame offset of channel ch in the group-space layout of lp: haspi_ihc_fir9_kernel emits frame i of a channel with group-delay shift sh
// on sample ul = (26 - sh) mod 9 of group g = i - di, di = (ul + sh - 26) / 9 (exact), and stores GROUPS as rows - all 32 channels of a
// row at once, full cache lines - instead of scattering 8-byte values over the rows i = g + di(ch) (measured: 3.4 x write
// amplification, and the stores, not the arithmetic, set the kernel's time).
__device__ __forceinline__ int hp_lp_di(const HaspiWs& ws, int b, int ch) {
    if (!ws.lp_raw) return 0;
    const int sh = ws.shift[(size_t)b * HP_NCH + ch];
    const int ul = (((26 - sh) % 9) + 9) % 9;
    return (ul + sh - 26) / 9;
}
// lp of frame i (< nsub), channel ch; lpb = block of one (utterance, signal), di = hp_lp_di of the channel
__device__ __forceinline__ double hp_lp_at(const double* __restrict__ lpb, int i, int ch, int di) {
    const int g = i - di;
    return g >= 0 ? lpb[(size_t)g * HP_NCH + ch] : 0.0;
}

__device__ __forceinline__ double i0_series(double x) {
    double s = 1.0, t = 1.0;
    const double q = 0.25 * x * x;
    for (int k = 1; k < 200; ++k) {
        t *= q / ((double)k * (double)k);
        s += t;
        if (t < 1e-18 * s) break;
    }
    return s;
}

// resampy sinc_window(num_zeros=64, precision=9, rolloff=0.9475937167399596) * kaiser(beta=14.769656459379492)
__global__ void haspi_win_kernel(double* __restrict__ win) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HP_NWIN) return;
    const double rolloff = 0.9475937167399596, beta = 14.769656459379492;
    const int n = HP_NWIN - 1;
    const double t = 64.0 * (double)i / (double)n;          // linspace(0, 64, n+1)
    const double a = rolloff * t;
    const double sinc = (i == 0) ? 1.0 : sinpi(a) / (M_PI * a);
    const double r = (double)i / (double)n;
    const double kais = i0_series(beta * sqrt(fmax(0.0, 1.0 - r * r))) / i0_series(beta);
    win[i] = kais * rolloff * sinc;
}

// ---- h1: rms normalisation + resampy kaiser_best 16 -> 24 kHz + RMS restore, three launches:
//   haspi_rms_kernel           one block per (utterance, signal): rms of the input, rms of the normalised input (float32, as the reference)
//   haspi_resample_kernel      one block per chunk of RS_CH outputs: the resampler proper + that chunk's output sum of squares
//   haspi_resample_gain_kernel per row: g = xRMS / yRMS from the chunk partials in chunk order; applied by the middle-ear kernels
// (one block per row walking all 125 chunks took 3.7 ms at B = 256: a latency-bound serial loop on 256 of the 1024 SIMDs)
#define RS_CH 768
#define RS_MAXC 512                 // chunks of RS_CH outputs per row at most: n24 <= 393 216 (16 s)
__global__ __launch_bounds__(256) void haspi_rms_kernel(const float* __restrict__ x, const float* __restrict__ y, int Lmax, int fs_in, HaspiWs ws,
                                                        int sig0) {
    __shared__ double red[8];
    const int b = blockIdx.x, sig = sig0 + blockIdx.y, tid = threadIdx.x, row = 2 * b + sig;
    const int L = hp_len(ws, b, Lmax), n24 = hp_n24(ws, b);
    const float* src = (sig ? y : x) + (size_t)b * Lmax;
    float* dst = ws.r24 + (size_t)row * ws.n24p;
    for (int i = n24 + tid; i < ws.n24p; i += 256) dst[i] = 0.f;      // the chunked kernels read whole chunks: defined values behind a short row
    // rms normalisation (pyhaspi2.py:81-84), float32 like the reference's arrays
    // (both sums in groups of eight loads in flight: one thread block per row, and a load-add loop paid a memory latency per element -
    //  0.18 ms per call for 33 MB; the additions stay in the same order)
    double acc = 0.0;
    {
        int i = tid;
        for (; i + 7 * 256 < L; i += 8 * 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[i + 256 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)(v[u] * v[u]);
        }
        for (; i < L; i += 256) acc += (double)(src[i] * src[i]);
    }
    acc = block_sum(acc, red);
    const float rms = sqrtf((float)acc / (float)L);
    float* ri = ws.rinfo + (size_t)row * 4;
    if (fs_in == 24000) {
        for (int i = tid; i < L; i += 256) dst[i] = src[i] / rms;
        if (tid == 0) { ri[0] = rms; ri[1] = 1.f; ri[2] = 1.f; }
        return;
    }
    double xs = 0.0;                                                   // xRMS of the normalised input (pyhaspi2.py:816)
    {
        int i = tid;
        for (; i + 7 * 256 < L; i += 8 * 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[i + 256 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float w = v[u] / rms; xs += (double)(w * w); }
        }
        for (; i < L; i += 256) { const float v = src[i] / rms; xs += (double)(v * v); }
    }
    xs = block_sum(xs, red);
    if (tid == 0) { ri[0] = rms; ri[1] = sqrtf((float)(xs / (double)L)); }
}

// resampy resample_f (ratio 1.5: scale = 1, index_step = num_table).  For this ratio the fractional position of an output takes three
// values (t mod 3), so there are only 3 x 2 tap columns: interpolated window values w[phase][wing][i] = win[offset + 512 i] + eta delta[...],
// built once per workgroup in LDS; the normalised input of the chunk is staged in LDS as float64, and a tap is two LDS reads and the
// reference's float32-rounded accumulate y[t] += w * x (resampy keeps y in the input's dtype: every tap rounds to float32 - kept, it is
// the reference's arithmetic; the accumulation order is the reference's too: left wing, then right wing).
// The tap columns use eta of the exact phases 0, 1/3, 2/3; the reference's eta = frac(t / 1.5) * 512 - offset carries the rounding of
// t / 1.5 (1e-11 at t = 1e5), i.e. a tap weight may differ by 1e-14 relative: one float32 rounding of one output in ~1e3 falls the other
// way (6e-8 relative on that sample).  Outputs whose taps leave the staged input (none for 16 kHz inputs of any length) and offsets
// outside the three phases fall back to indexing the 32769-entry window in memory.  grid (chunks, nsig, B), block 256.
__global__ __launch_bounds__(256) void haspi_resample_kernel(const float* __restrict__ x, const float* __restrict__ y, int Lmax, HaspiWs ws,
                                                             int sig0) {
    __shared__ double red[8];
    constexpr int RS_IN = RS_CH * 2 / 3 + 2 * 66 + 4;
    __shared__ double wt[3][2][66];
    __shared__ int woff[3][2];
    __shared__ double xsn[RS_IN];
    const int b = blockIdx.z, sig = sig0 + blockIdx.y, tid = threadIdx.x, row = 2 * b + sig;
    const int L = hp_len(ws, b, Lmax), n24 = hp_n24(ws, b);
    const int t0 = blockIdx.x * RS_CH;
    if (t0 >= n24) return;
    const float* src = (sig ? y : x) + (size_t)b * Lmax;
    float* dst = ws.r24 + (size_t)row * ws.n24p;
    const float rms = ws.rinfo[(size_t)row * 4];
    // resampy: sample_ratio = float(sr_new) / sr_orig, time_increment = 1 / sample_ratio (1 / 1.5 for 16 kHz inputs)
    const double time_increment = 1.0 / (24000.0 / (double)ws.fs_in);
    const bool three_phase = ws.fs_in == 16000;       // other rates (pyhaspi2.py:814: anything below 24 kHz) take the general tap loops below
    const int n_res = hp_nres_of(L, ws.fs_in);        // resampy's output count; librosa pads with zeros up to n24 = ceil(L * ratio)
    for (int e = tid; e < 3 * 2 * 66; e += 256) {
        const int ph = e / 132, wing = (e - ph * 132) / 66, i = e - ph * 132 - wing * 66;
        const double tr = (double)ph * time_increment;
        double frac = tr - (double)(int)tr;
        if (wing) frac = 1.0 - frac;
        const double index_frac = frac * HP_NTAB;
        const int offset = (int)index_frac;
        const double eta = index_frac - offset;
        const int idx = offset + i * HP_NTAB;
        double w = 0.0;
        if (idx < HP_NWIN) {
            const double v = ws.win[idx];
            w = v + eta * ((idx + 1 < HP_NWIN) ? ws.win[idx + 1] - v : 0.0);
        }
        wt[ph][wing][i] = w;
        if (i == 0) woff[ph][wing] = three_phase ? offset : -1;
    }
    double a2 = 0.0;
    const int nbase = max(0, (int)((double)t0 * time_increment) - 66);
    for (int e = tid; e < RS_IN; e += 256) xsn[e] = (nbase + e < L) ? (double)(src[nbase + e] / rms) : 0.0;
    __syncthreads();
    for (int t = t0 + tid; t < min(n24, t0 + RS_CH); t += 256) {
        if (t >= n_res) { dst[t] = 0.f; continue; }  // librosa.util.fix_length's padding
        const int ph = t % 3;
        const double time_register = (double)t * time_increment;
        const int n = (int)time_register;
        double frac = time_register - (double)n;
        double index_frac = frac * HP_NTAB;
        int offset = (int)index_frac;
        double eta = index_frac - offset;
        int i_max = (HP_NWIN - offset) / HP_NTAB;
        if (n + 1 < i_max) i_max = n + 1;
        float yv = 0.f;
        if (offset == woff[ph][0] && n - (i_max - 1) >= nbase) {
            const double* wl = wt[ph][0];
            const double* xl = xsn + (n - nbase);
#pragma unroll 4
            for (int i = 0; i < i_max; ++i) yv = (float)fma(wl[i], xl[-i], (double)yv);
        } else {
            for (int i = 0; i < i_max; ++i) {
                const int idx = offset + i * HP_NTAB;
                const double d = (idx + 1 < HP_NWIN) ? ws.win[idx + 1] - ws.win[idx] : 0.0;
                const double w = ws.win[idx] + eta * d;
                yv = (float)((double)yv + w * (double)(src[n - i] / rms));
            }
        }
        frac = 1.0 - frac;
        index_frac = frac * HP_NTAB;
        offset = (int)index_frac;
        eta = index_frac - offset;
        int k_max = (HP_NWIN - offset) / HP_NTAB;
        if (L - n - 1 < k_max) k_max = L - n - 1;
        if (offset == woff[ph][1] && n + k_max - nbase < RS_IN) {
            const double* wr = wt[ph][1];
            const double* xr = xsn + (n + 1 - nbase);
#pragma unroll 4
            for (int k = 0; k < k_max; ++k) yv = (float)fma(wr[k], xr[k], (double)yv);
        } else {
            for (int k = 0; k < k_max; ++k) {
                const int idx = offset + k * HP_NTAB;
                const double d = (idx + 1 < HP_NWIN) ? ws.win[idx + 1] - ws.win[idx] : 0.0;
                const double w = ws.win[idx] + eta * d;
                yv = (float)((double)yv + w * (double)(src[n + k + 1] / rms));
            }
        }
        dst[t] = yv;
        a2 += (double)(yv * yv);
    }
    a2 = block_sum(a2, red);
    if (tid == 0) ws.rsp[(size_t)row * RS_MAXC + blockIdx.x] = a2;
}

// ---- 16 -> 24 kHz, a thread per output TRIPLE (second session of round 3).  haspi_resample_kernel above pays two LDS reads per tap (the
// tap weight of the lane's phase and the sample) and is bound by exactly that (the LDS pipe of a CU moves one 8-byte read of a wave in 4
// cycles; the three dependent float64 operations of a tap take 12 on its SIMD, four SIMDs share the pipe).  Here thread q owns outputs
// 3q, 3q+1, 3q+2: their input positions are 2q, 2q, 2q+1, so phases 0 and 1 meet the SAME sample at every tap step and phase 2 meets the
// previous step's - one LDS read per step serves three taps - and since every lane runs the same three phases the tap weights are uniform:
// they come from a 2 x 64 x 4 table in memory by scalar loads (haspi_rs_taps_kernel, built once per call next to the window), no LDS
// read at all.  Each output's own chain is unchanged - left wing then right wing, every tap rounded to float32 like resampy's float32
// accumulator - so the samples are bit-identical to the kernel above; taps the reference does not execute at the signal's ends (n - i < 0,
// n + k + 1 >= L) meet a staged zero, and fma(w, 0, y) = y exactly.  Outputs whose float64 position arithmetic (t / 1.5 as the reference
// rounds it) does not land on the expected (n, offset) take the general loops.  grid (chunks, nsig, B), block 256, RS_CH = 768 outputs.
#define RS3_NT 64                       // taps per wing in the table: (HP_NWIN - offset) / HP_NTAB <= 64, zero beyond a phase's own count
#define HP_NWIN_AL 32776                // window length rounded up to 8 doubles: the tap table follows it in the workspace
#define RS3_TAB (2 * RS3_NT * 4 + 8)    // [wing][i][phase 0..2, pad] + the six table offsets
__global__ __launch_bounds__(256) void haspi_rs_taps_kernel(double* __restrict__ win) {
    double* tab = win + HP_NWIN_AL;
    const double time_increment = 1.0 / 1.5;
    for (int e = threadIdx.x; e < 2 * RS3_NT * 4; e += 256) {
        const int ph = e & 3, i = (e >> 2) % RS3_NT, wing = e / (4 * RS3_NT);
        double w = 0.0;
        if (ph < 3) {
            const double tr = (double)ph * time_increment;
            double frac = tr - (double)(int)tr;
            if (wing) frac = 1.0 - frac;
            const double index_frac = frac * HP_NTAB;
            const int offset = (int)index_frac;
            const double eta = index_frac - offset;
            const int idx = offset + i * HP_NTAB;
            if (i < (HP_NWIN - offset) / HP_NTAB) {                  // resampy's i_max / k_max of an interior output
                const double v = win[idx];
                w = v + eta * ((idx + 1 < HP_NWIN) ? win[idx + 1] - v : 0.0);
            }
            if (i == 0) tab[2 * RS3_NT * 4 + ph * 2 + wing] = (double)offset;
        }
        tab[e] = w;
    }
}

// one output by resampy's loops as they stand (window indexed in memory): irregular positions of the triple kernel
__device__ __noinline__ float rs_general_output(const float* __restrict__ src, float rms, int L, const double* __restrict__ win, int t, double time_increment) {
    const double time_register = (double)t * time_increment;
    const int n = (int)time_register;
    double frac = time_register - (double)n;
    double index_frac = frac * HP_NTAB;
    int offset = (int)index_frac;
    double eta = index_frac - offset;
    int i_max = (HP_NWIN - offset) / HP_NTAB;
    if (n + 1 < i_max) i_max = n + 1;
    float yv = 0.f;
    for (int i = 0; i < i_max; ++i) {
        const int idx = offset + i * HP_NTAB;
        const double d = (idx + 1 < HP_NWIN) ? win[idx + 1] - win[idx] : 0.0;
        const double w = win[idx] + eta * d;
        yv = (float)fma(w, (double)(src[n - i] / rms), (double)yv);
    }
    frac = 1.0 - frac;
    index_frac = frac * HP_NTAB;
    offset = (int)index_frac;
    eta = index_frac - offset;
    int k_max = (HP_NWIN - offset) / HP_NTAB;
    if (L - n - 1 < k_max) k_max = L - n - 1;
    for (int k = 0; k < k_max; ++k) {
        const int idx = offset + k * HP_NTAB;
        const double d = (idx + 1 < HP_NWIN) ? win[idx + 1] - win[idx] : 0.0;
        const double w = win[idx] + eta * d;
        yv = (float)fma(w, (double)(src[n + k + 1] / rms), (double)yv);
    }
    return yv;
}

__global__ __launch_bounds__(256) void haspi_resample3_kernel(const float* __restrict__ x, const float* __restrict__ y, int Lmax, HaspiWs ws,
                                                              int sig0) {
    __shared__ double red[8];
    constexpr int RS_IN = RS_CH * 2 / 3 + 2 * RS3_NT + 8;
    __shared__ double xsn[RS_IN];
    const int b = blockIdx.z, sig = sig0 + blockIdx.y, tid = threadIdx.x, row = 2 * b + sig;
    const int L = hp_len(ws, b, Lmax), n24 = hp_n24(ws, b);
    const int t0 = blockIdx.x * RS_CH;
    if (t0 >= n24) return;
    const int n_res = hp_nres_of(L, 16000);
    const float* src = (sig ? y : x) + (size_t)b * Lmax;
    float* dst = ws.r24 + (size_t)row * ws.n24p;
    const float rms = ws.rinfo[(size_t)row * 4];
    const double* __restrict__ tab = ws.win + HP_NWIN_AL;            // uniform indices below: scalar loads
    const double time_increment = 1.0 / 1.5;
    const int q0 = t0 / 3;
    const int nbase = 2 * q0 - RS3_NT - 2;                           // first staged input (may lie before the signal: zeros)
    for (int e = tid; e < RS_IN; e += 256) {
        const int idx = nbase + e;
        xsn[e] = (idx >= 0 && idx < L) ? (double)(src[idx] / rms) : 0.0;
    }
    __syncthreads();
    const int q = q0 + tid;
    bool regular = true;
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
        const double time_register = (double)(3 * q + ph) * time_increment;
        const int n = (int)time_register;
        const double frac = time_register - (double)n;
        const int offl = (int)(frac * HP_NTAB), offr = (int)((1.0 - frac) * HP_NTAB);
        regular = regular && n == 2 * q + (ph == 2) && offl == (int)tab[2 * RS3_NT * 4 + ph * 2] && offr == (int)tab[2 * RS3_NT * 4 + ph * 2 + 1];
    }
    float y0 = 0.f, y1 = 0.f, y2 = 0.f;
    if (regular) {
        const double* xc = xsn + (2 * q - nbase);
        double xb = xc[1];
#pragma unroll 8
        for (int i = 0; i < RS3_NT; ++i) {                           // left wings: x[2q - i] (phases 0, 1), x[2q + 1 - i] (phase 2)
            const double xa = xc[-i];
            y0 = (float)fma(tab[4 * i], xa, (double)y0);
            y1 = (float)fma(tab[4 * i + 1], xa, (double)y1);
            y2 = (float)fma(tab[4 * i + 2], xb, (double)y2);
            xb = xa;
        }
        const double* __restrict__ tabr = tab + 4 * RS3_NT;
        double xa = xc[1];
#pragma unroll 8
        for (int k = 0; k < RS3_NT; ++k) {                           // right wings: x[2q + 1 + k] (phases 0, 1), x[2q + 2 + k] (phase 2)
            const double xn = xc[2 + k];
            y0 = (float)fma(tabr[4 * k], xa, (double)y0);
            y1 = (float)fma(tabr[4 * k + 1], xa, (double)y1);
            y2 = (float)fma(tabr[4 * k + 2], xn, (double)y2);
            xa = xn;
        }
    } else {
        const int t = 3 * q;
        if (t < n_res) y0 = rs_general_output(src, rms, L, ws.win, t, time_increment);
        if (t + 1 < n_res) y1 = rs_general_output(src, rms, L, ws.win, t + 1, time_increment);
        if (t + 2 < n_res) y2 = rs_general_output(src, rms, L, ws.win, t + 2, time_increment);
    }
    double a2 = 0.0;
    {
        const int t = 3 * q;
        if (t < n24) { const float v = (t < n_res) ? y0 : 0.f; dst[t] = v; a2 += (double)(v * v); }       // t >= n_res: librosa's fix_length padding
        if (t + 1 < n24) { const float v = (t + 1 < n_res) ? y1 : 0.f; dst[t + 1] = v; a2 += (double)(v * v); }
        if (t + 2 < n24) { const float v = (t + 2 < n_res) ? y2 : 0.f; dst[t + 2] = v; a2 += (double)(v * v); }
    }
    a2 = block_sum(a2, red);
    if (tid == 0) ws.rsp[(size_t)row * RS_MAXC + blockIdx.x] = a2;
}

// y = (xRMS / yRMS) * y (pyhaspi2.py:816-818).  grid rows, block 64: the chunk partials are added in chunk order by one lane.
__global__ void haspi_resample_gain_kernel(HaspiWs ws, int sig0, int nsig) {
    const int row = hp_row(blockIdx.x, sig0, nsig);
    if (threadIdx.x != 0) return;
    const int n24 = hp_n24(ws, row >> 1);
    double a2 = 0.0;
    for (int c = 0; c * RS_CH < n24; ++c) a2 += ws.rsp[(size_t)row * RS_MAXC + c];
    float* ri = ws.rinfo + (size_t)row * 4;
    ri[2] = ri[1] / sqrtf((float)(a2 / (double)n24));
}

// ---- h2: middle ear (pyhaspi2.py:833-841), scipy lfilter = direct form II transposed. grid B, block 64 (lanes 0,1 active).
// All serial kernels below move samples in register chunks of HP_CH: the loads of a chunk are issued back to back (one
// memory latency per chunk instead of one per sample), then the recurrence runs out of registers.
#define HP_CH 32
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(64) void haspi_midear_kernel(HaspiWs ws, int sig0, int nsig) {
    const int b = blockIdx.x, sig = sig0 + threadIdx.x;
    if ((int)threadIdx.x >= nsig) return;
    const float* src = ws.r24 + ((size_t)b * 2 + sig) * ws.n24p;
    double* dst = ws.mid + ((size_t)b * 2 + sig) * ws.n24p;
    const double b0 = 0.434173751206302, b1 = 0.434173751206302, a1 = -0.131652497587396;
    const double c0 = 0.937260390269893, c1 = -1.874520780539785, c2 = 0.937260390269893, d1 = -1.870580640735279, d2 = 0.878460920344291;
    const float g = ws.rinfo[((size_t)b * 2 + sig) * 4 + 2];          // y = (xRMS / yRMS) * y, float32 like the reference's arrays
    double z = 0.0, w0 = 0.0, w1 = 0.0;
    for (int n0 = 0; n0 < ws.n24; n0 += HP_CH) {
        float xin[HP_CH];
#pragma unroll
        for (int u = 0; u < HP_CH; ++u) xin[u] = g * src[n0 + u];    // buffers are padded to whole chunks: no per-element guards
        double yo[HP_CH];
#pragma unroll
        for (int u = 0; u < HP_CH; ++u) {
            const double x = (double)xin[u];
            const double y1 = b0 * x + z;
            z = b1 * x - a1 * y1;
            const double y2 = c0 * y1 + w0;
            w0 = c1 * y1 - d1 * y2 + w1;
            w1 = c2 * y1 - d2 * y2;
            yo[u] = y2;
        }
#pragma unroll
        for (int u = 0; u < HP_CH; ++u) dst[n0 + u] = yo[u];
    }
}
#endif  // NELE_AB

// The same filter, parallel over chunks: the slowest pole of the cascade has modulus 0.937 (high-pass section) and 0.937^1024 = 2e-29,
// so a thread that starts 1024 samples before its 2048-sample chunk from a zero state reproduces the serial filter to well below the
// rounding of the samples (two active lanes per utterance made the serial kernel 7 ms at B = 256).  grid (ceil(chunks / 64), 2 B).
#define ME_N 2048
#define ME_W 1024
__global__ __launch_bounds__(64) void haspi_midear_par_kernel(HaspiWs ws, int sig0, int nsig) {
    const int row = hp_row(blockIdx.y, sig0, nsig), n0 = (blockIdx.x * 64 + threadIdx.x) * ME_N;
    if (n0 >= ws.n24p) return;
    const int n1 = min(n0 + ME_N, ws.n24p);
    const float* src = ws.r24 + (size_t)row * ws.n24p;
    double* dst = ws.mid + (size_t)row * ws.n24p;
    const double b0 = 0.434173751206302, b1 = 0.434173751206302, a1 = -0.131652497587396;
    const double c0 = 0.937260390269893, c1 = -1.874520780539785, c2 = 0.937260390269893, d1 = -1.870580640735279, d2 = 0.878460920344291;
    const float g = ws.rinfo[(size_t)row * 4 + 2];                    // y = (xRMS / yRMS) * y, float32 like the reference's arrays
    double z = 0.0, w0 = 0.0, w1 = 0.0;
    for (int n = max(0, n0 - ME_W); n < n1; n += HP_CH) {
        float xin[HP_CH];
#pragma unroll
        for (int u = 0; u < HP_CH / 4; ++u) {
            const float4 v = *reinterpret_cast<const float4*>(src + n + 4 * u);
            xin[4 * u] = g * v.x; xin[4 * u + 1] = g * v.y; xin[4 * u + 2] = g * v.z; xin[4 * u + 3] = g * v.w;
        }
        double yo[HP_CH];
#pragma unroll
        for (int u = 0; u < HP_CH; ++u) {
            const double x = (double)xin[u];
            const double y1 = b0 * x + z;
            z = b1 * x - a1 * y1;
            const double y2 = c0 * y1 + w0;
            w0 = c1 * y1 - d1 * y2 + w1;
            w1 = c2 * y1 - d2 * y2;
            yo[u] = y2;
        }
        if (n >= n0) {
#pragma unroll
            for (int u = 0; u < HP_CH / 2; ++u) *reinterpret_cast<double2*>(dst + n + 2 * u) = make_double2(yo[2 * u], yo[2 * u + 1]);
        }
    }
}

struct GtCoef { double a1, a2, a3, a4, a5, gain; };

__device__ __forceinline__ double hp_cfreq(int ch) {
    // pyhaspi2.py:753-777
    const double lowFreq = 80.0, highFreq = 8000.0, EarQ = 9.26449, minBW = 24.7;
    const int k = HP_NCH - 1 - ch;  // flipped order: ch 31 is highFreq
    if (k == 0) return highFreq;
    return -(EarQ * minBW) + exp((double)k * (-log(highFreq + EarQ * minBW) + log(lowFreq + EarQ * minBW)) / (double)(HP_NCH - 1)) *
                                 (highFreq + EarQ * minBW);
}

__device__ __forceinline__ GtCoef hp_gt(double BW, double cf) {
    const double ERB = 24.7 + cf / 9.26449;
    const double tpt = 2.0 * M_PI / HP_FS;
    const double a = exp(-(BW * tpt * ERB * 1.019));
    GtCoef c;
    c.a1 = 4.0 * a; c.a2 = -6.0 * a * a; c.a3 = 4.0 * a * a * a; c.a4 = -a * a * a * a; c.a5 = 4.0 * a * a;
    c.gain = 2.0 * (1 - c.a1 - c.a2 - c.a3 - c.a4) / (1 + c.a1 + c.a5);
    return c;
}

// Control bandwidth of channel ch for HL = 100 dB (pyhaspi2.py:779-807 with loss = 100 everywhere)
__device__ __forceinline__ double hp_bw1(int ch) {
    const double CR = 1.25 + 2.25 * (double)ch / (double)(HP_NCH - 1);
    const double maxOHC = 70.0 * (1.0 - 1.0 / CR), thrOHC = 1.25 * maxOHC;
    const double attnOHC = (100.0 < thrOHC) ? 80.0 : 0.8 * thrOHC;
    const double r = attnOHC / 50.0;
    return 1.0 + r + 2.0 * r * r * r * r * r * r;
}

// eb_LossParameters (pyhaspi2.py:779-807) of an audiogram HL[6] at (250, 500, 1000, 2000, 4000, 6000) Hz, per signal and channel.
// haspi_v2 / haspi (itype 0): the reference signal x is heard with normal hearing (HLx = 0 HL, pyhaspi2.py:1162-1165), the processed
// signal y with HL; hasqi_v2 (itype 2): both with HL.  HL = 0: attnOHC = attnIHC = 0, BWmin = 1, lowknee = 30, CR = 1.25 + 2.25 ch / 31.
struct HpHL { double x[6], y[6]; };
struct HpLoss { double attnOHC, BWmin, lowknee, CR, attnIHC; };
__device__ __forceinline__ HpLoss hp_loss(const HaspiWs& ws, int sig, int ch) {
    const double* p = ws.loss + (size_t)sig * 5 * HP_NCH + ch;
    HpLoss l;
    l.attnOHC = p[0]; l.BWmin = p[HP_NCH]; l.lowknee = p[2 * HP_NCH]; l.CR = p[3 * HP_NCH]; l.attnIHC = p[4 * HP_NCH];
    return l;
}
__global__ __launch_bounds__(64) void haspi_loss_kernel(HaspiWs ws, HpHL hl) {
    const int sig = threadIdx.x >> 5, ch = threadIdx.x & 31;
    const double* HL = sig ? hl.y : hl.x;
    // np.interp(cfreq, [cfreq[0], 250 .. 6000, cfreq[-1]], [HL[0], HL[0..5], HL[5]]), negative losses clipped to 0
    const double aud[6] = {250.0, 500.0, 1000.0, 2000.0, 4000.0, 6000.0};
    const double cf = hp_cfreq(ch);
    double loss;
    if (cf <= aud[0]) loss = HL[0];
    else if (cf >= aud[5]) loss = HL[5];
    else {
        int k = 0;
        while (k < 4 && cf >= aud[k + 1]) ++k;
        const double slope = (HL[k + 1] - HL[k]) / (aud[k + 1] - aud[k]);
        loss = slope * (cf - aud[k]) + HL[k];                               // numpy's interp formula
    }
    if (loss < 0.0) loss = 0.0;
    const double CR0 = 1.25 + 2.25 * (double)ch / (double)(HP_NCH - 1);
    const double maxOHC = 70.0 * (1.0 - (1.0 / CR0)), thrOHC = 1.25 * maxOHC;
    double attnOHC, attnIHC;
    if (loss < thrOHC) { attnOHC = 0.8 * loss; attnIHC = 0.2 * loss; }
    else { attnOHC = 0.8 * thrOHC; attnIHC = 0.2 * thrOHC + (loss - thrOHC); }
    const double r = attnOHC / 50.0;
    const double BW = 1.0 + r + 2.0 * (r * r * r * r * r * r);
    const double lowknee = attnOHC + 30.0, upamp = 30.0 + 70.0 / CR0;
    const double CR = (100.0 - lowknee) / (upamp + attnOHC - lowknee);
    double* p = ws.loss + (size_t)sig * 5 * HP_NCH + ch;
    p[0] = attnOHC; p[HP_NCH] = BW; p[2 * HP_NCH] = lowknee; p[3 * HP_NCH] = CR; p[4 * HP_NCH] = attnIHC;
}

// Gammatone envelope: demodulate by the rotation recurrence (eb_CosSinCF), filter real and imaginary parts with
// lfilter([1,a1,a5],[1,-a1,-a2,-a3,-a4]) (DF2T), envelope = gain*|u|.  One wave per (utterance, signal): lane = part*32 +
// channel, part 0 filters x*cos, part 1 filters x*sin; the two halves meet through one cross-lane swap per sample.
// Returns (on the part-0 lanes) the sum of squares of the envelope; out[n*32 + ch] is written by the part-0 lanes.
// The stored value is the SQUARED magnitude |u|^2 and the return value its plain sum: gain * sqrt(.) is applied by the point-wise
// consumers (haspi_gain_kernel / haspi_sl_kernel, wide and memory-bound), where it is free; inside this loop - one wave per
// (utterance, signal), issue-bound - the float64 square root was half of the instructions of a sample.
__device__ __forceinline__ void hp_rotate(double& cold, double& sold, double cn, double sn);
__device__ __forceinline__ double hp_gammatone_wave(const double* __restrict__ xin, int n24, const GtCoef c, double cf, int part,
                                                    hp_env_t* __restrict__ out) {
    const double tpt = 2.0 * M_PI / HP_FS;
    const double cn = cos(tpt * cf), sn = sin(tpt * cf);
    double cold = 1.0, sold = 0.0;
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    double ss = 0.0;
    for (int n0 = 0; n0 < n24; n0 += HP_CH) {
        double xc[HP_CH];
#pragma unroll
        for (int u = 0; u < HP_CH; ++u) xc[u] = xin[n0 + u];
        double eo[HP_CH];
#pragma unroll
        for (int u = 0; u < HP_CH; ++u) {
            if (n0 + u > 0) hp_rotate(cold, sold, cn, sn);
            const double xr = xc[u] * (part ? sold : cold);
            const double yr = xr + r0;
            r0 = c.a1 * xr + c.a1 * yr + r1;
            r1 = c.a5 * xr + c.a2 * yr + r2;
            r2 = c.a3 * yr + r3;
            r3 = c.a4 * yr;
            const double yo = lane_xor32(yr);
            const double e2 = yr * yr + yo * yo;
            eo[u] = e2;
            ss += (n0 + u < n24) ? e2 : 0.0;
        }
        if (part == 0) {
#pragma unroll
            for (int u = 0; u < HP_CH; ++u) out[(size_t)(n0 + u) * HP_NCH] = (hp_env_t)eo[u];
        }
    }
    return ss;
}

// The same banks as an EXACT parallel scan over chunks of lc samples (a few thousand waves instead of one per (utterance, signal)).
// The filter is linear: r[n+1] = M r[n] + q xr[n] with the 4 DF2T states r.  Pass 1 runs every chunk from a ZERO state and keeps
// only the end state e_c; the true state at the start of chunk c is R_c = sum_j M^(lc (c-1-j)) e_j, evaluated by Horner's rule with the
// per-channel matrix P = M^lc (haspi_pmat_kernel: the homogeneous recurrence itself, iterated lc times on the four unit vectors -
// no matrix powers, M is a Jordan block and squaring it cancels catastrophically); pass 2 reruns the chunk from R_c and writes the
// envelope.  Same arithmetic per sample as the serial kernel; the only difference is the rounding of R_c (1e-16 relative).
// The demodulator state at a chunk start (eb_CosSinCF's rotation recurrence, pyhaspi2.py:855-860) is evaluated directly: the
// recurrence's own accumulated rounding is ~1e-11 of a radian after 96 000 samples, far below anything the score can see.
// (Round 1 ran the chunks from a zero state 8192 samples early instead: 4 chunks at most before the warm-up dominated.)
#define GS_RC 16                   // samples per register chunk
#define GS_LC 1536                 // default scan-chunk length
#define GS_MAXC 160                // scan chunks per row at most (the host lengthens the chunks of longer signals)
__device__ __forceinline__ void hp_rotate(double& cold, double& sold, double cn, double sn) {
    const double arg = fma(sold, sn, cold * cn);
    sold = fma(sold, cn, -(cold * sn));
    cold = arg;
}
// One sample of a branch of the 4th-order gammatone section, y = x + r0 computed by the caller: the serial kernel's state update
// (r0 = a1 x + a1 y + r1, r1 = a5 x + a2 y + r2, r2 = a3 y + r3, r3 = a4 y; pyhaspi2.py:905-912 through scipy's lfilter) with the two
// three-term sums as multiply-add chains - 6 operations instead of 8, results equal to rounding (1e-16 relative per step).
__device__ __forceinline__ void hp_gt_step(const GtCoef& c, double x, double y, double& r0, double& r1, double& r2, double& r3) {
    r0 = fma(c.a1, x + y, r1);
    r1 = fma(c.a5, x, fma(c.a2, y, r2));
    r2 = fma(c.a3, y, r3);
    r3 = c.a4 * y;
}
// P = M^lc per channel.  grid (1 + rows) for the signal bank (blockIdx.x = 1 + launch row) / 1 for the control bank, block 128:
// thread = (unit vector k, channel).
__global__ __launch_bounds__(128) void haspi_pmat_kernel(HaspiWs ws, int lc, int signal, int sig0, int nsig) {
    const int ch = threadIdx.x & 31, k = threadIdx.x >> 5;
    const int row = signal ? hp_row(blockIdx.x, sig0, nsig) : 0;
    const double BW = signal ? ws.bw[(size_t)row * HP_NCH + ch] : hp_bw1(ch);
    const GtCoef c = hp_gt(BW, hp_cfreq(ch));
    double r0 = (k == 0), r1 = (k == 1), r2 = (k == 2), r3 = (k == 3);
    for (int n = 0; n < lc; ++n) {                       // the serial kernel's update with xr = 0
        const double yr = r0;
        r0 = c.a1 * yr + r1;
        r1 = c.a2 * yr + r2;
        r2 = c.a3 * yr + r3;
        r3 = c.a4 * yr;
    }
    double* P = ws.pmat + ((size_t)(signal ? 1 + row : 0) * HP_NCH + ch) * 16;
    P[0 * 4 + k] = r0; P[1 * 4 + k] = r1; P[2 * 4 + k] = r2; P[3 * 4 + k] = r3;     // column k of M^lc
}

// End states of pass 1 -> start states of pass 2, in place: R_0 = 0, R_(j+1) = P R_j + e_j.  grid rows, block 64 (lane = part * 32 +
// channel); the end states of a group of chunks are loaded together ahead of the (serial) Horner steps.
template <bool SIGNAL>
__global__ __launch_bounds__(64) void haspi_bank_prefix_kernel(HaspiWs ws, int sig0, int nsig) {
    const int row = hp_row(blockIdx.x, sig0, nsig), lane = threadIdx.x, ch = lane & 31;
    const int n24 = hp_n24(ws, row >> 1);
    const int nch = ((n24 + GS_RC - 1) / GS_RC * GS_RC + ws.lc - 1) / ws.lc;          // chunks pass 1 produced for this row
    const double* P = ws.pmat + ((size_t)(SIGNAL ? 1 + row : 0) * HP_NCH + ch) * 16;
    double Pm[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) Pm[q] = P[q];
    double* est = ws.est + (size_t)row * ws.nchunk * 256 + lane;
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    constexpr int G = 8;
    for (int j0 = 0; j0 < nch; j0 += G) {
        double e[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int j = min(j0 + g, nch - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) e[g][q] = est[(size_t)j * 256 + 64 * q];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (j0 + g < nch) {
                double* o = est + (size_t)(j0 + g) * 256;
                o[0] = r0; o[64] = r1; o[128] = r2; o[192] = r3;
                const double t0 = ((Pm[0] * r0 + Pm[1] * r1) + (Pm[2] * r2 + Pm[3] * r3)) + e[g][0];
                const double t1 = ((Pm[4] * r0 + Pm[5] * r1) + (Pm[6] * r2 + Pm[7] * r3)) + e[g][1];
                const double t2 = ((Pm[8] * r0 + Pm[9] * r1) + (Pm[10] * r2 + Pm[11] * r3)) + e[g][2];
                const double t3 = ((Pm[12] * r0 + Pm[13] * r1) + (Pm[14] * r2 + Pm[15] * r3)) + e[g][3];
                r0 = t0; r1 = t1; r2 = t2; r3 = t3;
            }
        }
    }
}

#define GL_N 2048
#define GL_W 256
#define GL_U 8
// log2 / exp2 on the float32 transcendental unit (v_log_f32 / v_exp_f32, 1 ulp): the dB values they produce are accurate to 1e-6 dB,
// like the float32 storage of the envelopes (see hp_env_t); the float64 log10 / exp / sqrt calls they replace were 90 % of this
// kernel's instructions and kept it compute-bound at 0.3 of the HBM rate.
__device__ __forceinline__ float hp_log2f(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float hp_exp2f(float x) { return __builtin_amdgcn_exp2f(x); }
struct IhcC { double R2, R12C1, R23C2, a11, a12, a21, a22, denom, R1inv, c10, c11, c12, c20, c21, c22; };
__device__ __forceinline__ IhcC hp_ihc_consts() {
    const double delta = 2.0;
    const double tau1 = 0.001 * 2, tau2 = 0.001 * 60;
    const double T = 1 / HP_FS;
    const double R1 = 1 / delta, R2 = 0.5 * (1 - R1), R3 = R2;
    const double C1 = tau1 * (R1 + R2) / (R1 * R2);
    const double C2 = tau2 / ((R1 + R2) * R3);
    IhcC k;
    k.a11 = R1 + R2 + R1 * R2 * (C1 / T); k.a12 = -R1; k.a21 = -R3; k.a22 = R2 + R3 + R2 * R3 * (C2 / T);
    k.denom = 1.0 / (k.a11 * k.a22 - k.a21 * k.a12);
    k.R1inv = 1.0 / R1; k.R12C1 = R1 * R2 * (C1 / T); k.R23C2 = R2 * R3 * (C2 / T); k.R2 = R2;
    // the step of pyhaspi2.py:1061-1070 (b1 = R2 V0 + R12C1 V1, b2 = R23C2 V2, V = denom * adj(A) b) with its constants multiplied out:
    // V1' = c10 V0 + c11 V1 + c12 V2, V2' = c20 V0 + c21 V1 + c22 V2 - six operations on the serial chain instead of nine
    k.c10 = k.denom * k.a22 * k.R2;  k.c11 = k.denom * k.a22 * k.R12C1;  k.c12 = -k.denom * k.a12 * k.R23C2;
    k.c20 = -k.denom * k.a21 * k.R2; k.c21 = -k.denom * k.a21 * k.R12C1; k.c22 = k.denom * k.a11 * k.R23C2;
    return k;
}
__device__ __forceinline__ void hp_ihc_step(const IhcC& k, double V0, double& V1, double& V2) {
    const double t1 = fma(k.c12, V2, fma(k.c11, V1, V0 * k.c10));
    const double t2 = fma(k.c22, V2, fma(k.c21, V1, V0 * k.c20));
    V1 = t1; V2 = t2;
}

// grid (ceil(chunks / 2), nsig, B), block 64: lane = (chunk parity) * 32 + channel.  A lane runs BOTH demodulated branches (x cos and
// x sin) of its channel: the rotation recurrence is computed once per channel instead of once per branch, and the envelope
// |u|^2 = yr^2 + yi^2 needs no cross-lane exchange (the serial kernel's lane = branch * 32 + channel layout spent half of its issue
// slots on selects, register moves and v_permlane32_swap: every VALU instruction of a wave64 costs 4 cycles on the 16-wide SIMDs,
// whatever its width).  Same arithmetic per branch as the serial kernel.
// BM (quality path, pass 2 of the signal bank): also the basilar-membrane motion u_r cos + u_i sin (pyhaspi2.py:897) as its ratio to the
// envelope |u| - the cosine of the carrier phase, which is all the later stages need (see haspi_quality.h) - and the envelope's sum of squares.
// GAIN (pass 2 of the signal bank, training path): the compression gain (pyhaspi2.py:982-997), its low-pass, the dB-SL conversion
// (:1080-1088) and pass 1 of the IHC adaptation ride along - the same arithmetic as haspi_gain_lp_sl_kernel on the same float32-rounded
// |u|^2, so the result is bit-identical, but the envelope goes to memory once (as dB SL) instead of |u|^2 out, |u|^2 + control in, dB SL
// out: 8.7 instead of 16 bytes per (sample, channel) over the two kernels.  The IHC chunks are the scan chunks then (ws.lcg = ws.lc).
template <bool SIGNAL, bool PASS2, bool BM = false, bool GAIN = false>
__global__ __launch_bounds__(64) void haspi_bank_scan_kernel(HaspiWs ws, int sig0) {
    const int b = blockIdx.z, sig = sig0 + blockIdx.y, lane = threadIdx.x, ch = lane & 31;
    const int chunk = 2 * blockIdx.x + (lane >> 5);
    const int row = 2 * b + sig, lc = ws.lc;
    const int n24 = hp_n24(ws, b);
    const int n0 = chunk * lc, n1 = min(n0 + lc, (n24 + GS_RC - 1) / GS_RC * GS_RC);
    if (n0 >= n1) return;                                  // chunk behind the end of a short row: nothing reads its state
    const double cf = hp_cfreq(ch);
    const GtCoef c = hp_gt(SIGNAL ? ws.bw[(size_t)row * HP_NCH + ch] : hp_bw1(ch), cf);
    const double tpt = 2.0 * M_PI / HP_FS;
    const double cn = cos(tpt * cf), sn = sin(tpt * cf);
    double* est = ws.est + ((size_t)row * ws.nchunk + chunk) * 256 + ch;       // [4][branch * 32 + channel]
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0, i0 = 0, i1 = 0, i2 = 0, i3 = 0;
    if (PASS2) {                                          // true state at the chunk start (haspi_bank_prefix_kernel)
        r0 = est[0]; r1 = est[64]; r2 = est[128]; r3 = est[192];
        i0 = est[32]; i1 = est[96]; i2 = est[160]; i3 = est[224];
    }
    // demodulator state one step BEFORE sample n0, so that the rotation below is unconditional: R^(n0 - 1) (1, 0); for n0 = 0 that
    // is R^(-1) (1, 0) = (cos, +sin), which the first rotation turns into (1, 0) to rounding
    double cold, sold;
    {
        const double ang = tpt * cf * (double)(n0 - 1);
        cold = cos(ang);
        sold = -sin(ang);
    }
    const double* xin = ws.mid + (size_t)row * ws.n24p;
    hp_env_t* out = (SIGNAL ? ws.env : ws.ctl) + ((size_t)row * ws.n24p) * HP_NCH + ch;
    float* cph = BM ? ws.cphi + ((size_t)row * ws.n24p) * HP_NCH + ch : nullptr;
    double ss = 0.0;
    // ---- GAIN state (see haspi_gain_lp_sl_kernel for the derivation of the constants)
    const hp_env_t* ctl = ws.ctl + (size_t)row * ws.n24p * HP_NCH + ch;
    const float TEN_LOG10_2 = 3.0102999566398120f, LOG2_10_OVER_20 = 0.16609640474436813f;
    float slope = 0.f, c_off = 0.f, s_off = 0.f;
    double gz = 0.0, V1 = 0.0, V2 = 0.0;
    const double gb0 = 0.095107983402496, ga1 = -0.809784033195007;
    const IhcC ik = hp_ihc_consts();
    float knee = 30.0f, attn = 0.f;
    if (GAIN) {
        const HpLoss ls = hp_loss(ws, sig, ch);
        slope = (float)(1.0 - (1.0 / ls.CR));
        knee = (float)ls.lowknee; attn = (float)ls.attnOHC;
        c_off = (float)(HP_LEVEL + 20.0 * log10(hp_gt(hp_bw1(ch), cf).gain));
        s_off = (float)(HP_LEVEL - ls.attnIHC + 20.0 * log10(c.gain));
        for (int n = max(0, n0 - GL_W); n < n0; n += GL_U) {    // low-pass warm-up on the control envelope alone (0.81^256 = 4e-24)
            float gc[GL_U];
#pragma unroll
            for (int u = 0; u < GL_U; ++u) gc[u] = ctl[(size_t)(n + u) * HP_NCH];
#pragma unroll
            for (int u = 0; u < GL_U; ++u) {
                float le = c_off + TEN_LOG10_2 * hp_log2f(gc[u]);
                le = fminf(fmaxf(le, knee), 100.0f);
                const double gx = (double)hp_exp2f((-attn - (le - knee) * slope) * LOG2_10_OVER_20);
                const double y = gb0 * gx + gz;
                gz = gb0 * gx - ga1 * y;
            }
        }
    }
    for (int nb = n0; nb < n1; nb += GS_RC) {
        double xc[GS_RC];
        const hp_env_t* ctl_nb = ctl + (size_t)nb * HP_NCH;              // one 64-bit address per group, immediates per sample
        hp_env_t* out_nb = out + (size_t)nb * HP_NCH;
        float* cph_nb = BM ? cph + (size_t)nb * HP_NCH : nullptr;
#pragma unroll
        for (int u = 0; u < GS_RC / 2; ++u) {
            const double2 v = *reinterpret_cast<const double2*>(xin + nb + 2 * u);
            xc[2 * u] = v.x; xc[2 * u + 1] = v.y;
        }
        float eo[GS_RC], co[BM ? GS_RC : 1];
        auto bank = [&](auto tail_tag) {                    // TAIL: the group that holds the row's end (only its samples < n24 count in ss)
            constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
            for (int u = 0; u < GS_RC; ++u) {
                hp_rotate(cold, sold, cn, sn);
                const double xr = xc[u] * cold, xi = xc[u] * sold;
                const double yr = xr + r0, yi = xi + i0;
                hp_gt_step(c, xr, yr, r0, r1, r2, r3);
                hp_gt_step(c, xi, yi, i0, i1, i2, i3);
                if (PASS2) {
                    const double e2 = yr * yr + yi * yi;
                    eo[u] = (float)e2;
                    if (!SIGNAL || BM) ss += (!TAIL || nb + u < n24) ? e2 : 0.0;
                    if (BM) co[u] = e2 > 0.0 ? (float)((yr * cold + yi * sold) / sqrt(e2)) : 0.f;
                }
            }
        };
        if (PASS2 && (!SIGNAL || BM) && nb + GS_RC > n24) bank(std::true_type{}); else bank(std::false_type{});
        if (GAIN) {
            float gc[GS_RC];
#pragma unroll
            for (int u = 0; u < GS_RC; ++u) gc[u] = ctl_nb[u * HP_NCH];
#pragma unroll
            for (int u = 0; u < GS_RC; ++u) {
                float le = c_off + TEN_LOG10_2 * hp_log2f(gc[u]);
                le = fminf(fmaxf(le, knee), 100.0f);
                const double gx = (double)hp_exp2f((-attn - (le - knee) * slope) * LOG2_10_OVER_20);
                const double yl = gb0 * gx + gz;
                gz = gb0 * gx - ga1 * yl;
                const float g = (float)yl;
                float y = s_off + TEN_LOG10_2 * hp_log2f(g * g * eo[u]);
                y = y > 0.0f ? y : 0.0f;
                eo[u] = y;
                hp_ihc_step(ik, (double)y, V1, V2);
            }
        }
        if (PASS2) {
#pragma unroll
            for (int u = 0; u < GS_RC; ++u) out_nb[u * HP_NCH] = eo[u];
            if (BM) {
#pragma unroll
                for (int u = 0; u < GS_RC; ++u) cph_nb[u * HP_NCH] = co[u];
            }
        }
    }
    if (!PASS2) {
        est[0] = r0; est[64] = r1; est[128] = r2; est[192] = r3;
        est[32] = i0; est[96] = i1; est[160] = i2; est[224] = i3;
    } else if (!SIGNAL) {
        ws.ssp[((size_t)row * ws.nchunk + chunk) * HP_NCH + ch] = ss;
    } else if (BM) {
        ws.sse[((size_t)row * ws.nchunk + chunk) * HP_NCH + ch] = ss;
    }
    if (GAIN) {
        const int ncg = (ws.n24p + ws.lcg - 1) / ws.lcg;
        double* ihe = ws.ihe + ((size_t)row * ncg + chunk) * 64 + ch;
        ihe[0] = V1; ihe[32] = V2;
    }
}

// Pass 1 with the chunk's HEAD skipped (round 3).  The end state of a chunk run from a zero state depends on its last W samples only: the
// bank's impulse response decays like n^3 a^n, a = exp(-BW 2 pi / 24000 ERB 1.019), so everything further back than
//     W(a):  n^3 a^n < 1e-22 of the response's peak
// is below the float64 rounding of the state (the same argument as for the middle-ear and gain low-pass warm-ups, 1e-20 there).  W is 60
// samples at 8 kHz and 5000 at 80 Hz: with the lane = channel layout of haspi_bank_scan_kernel a wave takes as long as its slowest
// channel, so this kernel turns the mapping around - a WAVE owns one channel and its 64 lanes own 64 chunks of one row: every lane runs
// the same W steps, ending at its own chunk end (samples before a short chunk's start enter as zeros).  Average work: 0.48 of a chunk for
// the control bank, 0.55-0.69 for the signal bank (lc = 1536).
// Input: the 64 lanes read 64 different chunks, and every channel reads the whole signal.  One wave per channel with its own loads ran at
// the L2 -> L1 rate (32 x the signal per launch: 0.8 ms whatever W), so a workgroup is 8 channels (neighbours: similar W) and stages 16
// samples of each of the 64 chunks ONCE for all of them: 512 threads x 16 bytes = 8 lines of 128 bytes per load instruction, double
// buffered in LDS, one barrier per 16 samples; a wave joins the walk when the distance to the chunk end drops to its own W.
// Same arithmetic per sample as pass 1 of the scan kernel; end states differ from a full-chunk run by < 1e-20 relative.
// grid 8 * 4 * ceil(nchunk / 64) * ceil(rows / 8) (1-D, see below), block 512 = 8 channels.  full != 0: W = lc for every channel (diagnostic).
template <bool SIGNAL>
__global__ __launch_bounds__(512) void haspi_bank_tail_kernel(HaspiWs ws, int sig0, int nsig, int full, int nrows) {
    __shared__ double stage[2][64][GS_RC + 1];
    __shared__ int wred[8];
    // 1-D grid.  Workgroup ids go round-robin over the 8 XCDs; with (channel group, row) in the natural order every XCD would receive
    // ONE channel group - and the XCDs with the low channels (full-length walks) would finish long after the others.  Here the id is
    // decoded as (xcd = id & 7, slot = id >> 3): channel group = slot & 3, so every XCD walks through all four groups.
    const int id = (int)blockIdx.x, slot = id >> 3, cgrp = slot & 3, rest = slot >> 2;
    const int ncg = (ws.nchunk + 63) / 64;
    const int ridx = (rest / ncg) * 8 + (id & 7);
    if (ridx >= nrows) return;                               // (whole workgroup: no barrier is skipped by a part of it)
    const int row = hp_row(ridx, sig0, nsig), b = row >> 1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ch = 8 * cgrp + wv;
    const int c0 = 64 * (rest % ncg), chunk = c0 + lane, lc = ws.lc;
    const int n24 = hp_n24(ws, b), nend = (n24 + GS_RC - 1) / GS_RC * GS_RC;
    const int n0 = chunk * lc, n1 = min(n0 + lc, nend);
    const bool live = n0 < n1;                              // chunks behind the end of a short row: nothing reads their state
    const double cf = hp_cfreq(ch);
    const double BW = SIGNAL ? ws.bw[(size_t)row * HP_NCH + ch] : hp_bw1(ch);
    const GtCoef c = hp_gt(BW, cf);
    const double tpt = 2.0 * M_PI / HP_FS;
    // decay length: smallest multiple of 16 with 3 ln n + n ln a < ln(1e-22) + (peak of 3 ln n + n ln a)
    int W;
    {
        const double la = BW * tpt * (24.7 + cf / 9.26449) * 1.019;        // -ln a
        const double npk = fmax(3.0 / la, 1.0);                               // the envelope n^3 a^n peaks at n = 3 / (-ln a)
        const double peak = 3.0 * log(npk) - la * npk;
        double nn = 60.0 / la;
        for (int it = 0; it < 6; ++it) nn = (50.66 - peak + 3.0 * log(nn)) / la;   // 50.66 = -ln 1e-22
        W = ((int)nn + 1 + GS_RC) / GS_RC * GS_RC;
        W = min(W, lc);
        if (full == 1) W = lc;
        if (full == 2) W = GS_RC;                            // (diagnostic: prologue cost)
    }
    if (lane == 0) wred[wv] = W;
    __syncthreads();
    int Wmax = wred[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) Wmax = max(Wmax, wred[q]);
    const double cn = cos(tpt * cf), sn = sin(tpt * cf);
    double cold, sold;                                      // demodulator one step before this lane's first sample n1 - W
    {
        const double ang = tpt * cf * (double)(n1 - W - 1);
        cold = cos(ang);
        sold = -sin(ang);
    }
    const double* xin = ws.mid + (size_t)row * ws.n24p;
    // staging: thread -> (chunk cc = tid >> 3, samples 2 (tid & 7) .. + 1) of a group; sample index = cn1 - Wmax + t + 2 part
    const int cc = tid >> 3, part = tid & 7;
    const int cn0 = (c0 + cc) * lc, cn1 = min(cn0 + lc, nend);
    const double* psrc = xin + (cn1 - Wmax + 2 * part);
    const int pfirst = (cn0 < cn1) ? cn0 - (cn1 - Wmax + 2 * part) : 0x7fffffff;       // the sample lies inside its chunk  <=>  t >= pfirst
    double2 pv = (0 >= pfirst) ? *reinterpret_cast<const double2*>(psrc) : make_double2(0.0, 0.0);
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0, i0 = 0, i1 = 0, i2 = 0, i3 = 0;
    const int tjoin = Wmax - W;                             // this wave's first group
    for (int t = 0, it = 0; t < Wmax; t += GS_RC, ++it) {
        double (*st)[GS_RC + 1] = stage[it & 1];
        st[cc][2 * part] = pv.x; st[cc][2 * part + 1] = pv.y;
        __syncthreads();                                    // group t is complete; the other buffer (group t - 16) is free again after it
        if (t + GS_RC < Wmax) pv = (t + GS_RC >= pfirst) ? *reinterpret_cast<const double2*>(psrc + t + GS_RC) : make_double2(0.0, 0.0);
        if (t >= tjoin) {
            double xc[GS_RC];
#pragma unroll
            for (int u = 0; u < GS_RC; ++u) xc[u] = st[lane][u];
#pragma unroll
            for (int u = 0; u < GS_RC; ++u) {
                hp_rotate(cold, sold, cn, sn);
                const double xr = xc[u] * cold, xi = xc[u] * sold;
                const double yr = xr + r0, yi = xi + i0;
                hp_gt_step(c, xr, yr, r0, r1, r2, r3);
                hp_gt_step(c, xi, yi, i0, i1, i2, i3);
            }
        }
    }
    if (live) {
        double* est = ws.est + ((size_t)row * ws.nchunk + chunk) * 256 + ch;       // [4][branch * 32 + channel]
        est[0] = r0; est[64] = r1; est[128] = r2; est[192] = r3;
        est[32] = i0; est[96] = i1; est[160] = i2; est[224] = i3;
    }
}
// eb_BWadjust from the chunk partials (added in chunk order).  grid rows, block 32
__global__ void haspi_bw_kernel(HaspiWs ws, int sig0, int nsig) {
    const int row = hp_row(blockIdx.x, sig0, nsig), ch = threadIdx.x;
    const double bw1 = hp_bw1(ch);
    const GtCoef cc = hp_gt(bw1, hp_cfreq(ch));
    const int n24 = hp_n24(ws, row >> 1);
    const int nch = (n24 + ws.lc - 1) / ws.lc;
    double ss = 0.0;
    for (int c = 0; c < nch; ++c) ss += ws.ssp[((size_t)row * ws.nchunk + c) * HP_NCH + ch];
    ss *= cc.gain * cc.gain;
    const double cdB = 20.0 * log10(sqrt(ss / (double)n24)) + HP_LEVEL;
    const double bwmin = hp_loss(ws, row & 1, ch).BWmin;                     // eb_BWadjust (pyhaspi2.py:971-980)
    double BW;
    if (cdB < 50.0) BW = bwmin;
    else if (cdB > 100.0) BW = bw1;
    else BW = bwmin + ((cdB - 50.0) / 50.0) * (bw1 - bwmin);
    ws.bw[(size_t)row * HP_NCH + ch] = BW;
}

// ---- h3: control bank + bandwidth adjustment. grid (2, B), block 64
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(64) void haspi_control_kernel(HaspiWs ws, int sig0) {
    const int b = blockIdx.y, sig = sig0 + blockIdx.x, lane = threadIdx.x, part = lane >> 5, ch = lane & 31;
    const double cf = hp_cfreq(ch), bw1 = hp_bw1(ch);
    const double* xin = ws.mid + ((size_t)b * 2 + sig) * ws.n24p;
    hp_env_t* out = ws.ctl + (((size_t)b * 2 + sig) * ws.n24p) * HP_NCH + ch;
    const GtCoef cc = hp_gt(bw1, cf);
    const int n24 = hp_n24(ws, b);
    const double ss = (cc.gain * cc.gain) * hp_gammatone_wave(xin, n24, cc, cf, part, out);   // sum of (gain |u|)^2
    if (part == 0) {
        // eb_BWadjust (pyhaspi2.py:971-980)
        const double cdB = 20.0 * log10(sqrt(ss / (double)n24)) + HP_LEVEL;
        const double bwmin = hp_loss(ws, sig, ch).BWmin;
        double BW;
        if (cdB < 50.0) BW = bwmin;
        else if (cdB > 100.0) BW = bw1;
        else BW = bwmin + ((cdB - 50.0) / 50.0) * (bw1 - bwmin);
        ws.bw[((size_t)b * 2 + sig) * HP_NCH + ch] = BW;
    }
}
#endif  // NELE_AB

// ---- h4: signal bank. grid (2, B), block 64
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(64) void haspi_signal_kernel(HaspiWs ws, int sig0) {
    const int b = blockIdx.y, sig = sig0 + blockIdx.x, lane = threadIdx.x, part = lane >> 5, ch = lane & 31;
    const double cf = hp_cfreq(ch);
    const double BW = ws.bw[((size_t)b * 2 + sig) * HP_NCH + ch];
    const double* xin = ws.mid + ((size_t)b * 2 + sig) * ws.n24p;
    hp_env_t* out = ws.env + (((size_t)b * 2 + sig) * ws.n24p) * HP_NCH + ch;
    (void)hp_gammatone_wave(xin, hp_n24(ws, b), hp_gt(BW, cf), cf, part, out);
}
#endif  // NELE_AB

// ---- h5: compression gain from the control envelope (pyhaspi2.py:982-991), point-wise, in place on ctl
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ void haspi_gain_kernel(HaspiWs ws, size_t per_row, int sig0, int nsig) {
    // the grid stride is a multiple of 32, so a thread stays on one channel: its control-filter gain is computed once
    const int ch = (int)(threadIdx.x & 31);
    const double cgain = hp_gt(hp_bw1(ch), hp_cfreq(ch)).gain;
    const HpLoss ls = hp_loss(ws, hp_row(blockIdx.y, sig0, nsig) & 1, ch);
    const double CR = ls.CR;
    const size_t base = (size_t)hp_row(blockIdx.y, sig0, nsig) * per_row;
    for (size_t i = base + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < base + per_row; i += (size_t)gridDim.x * blockDim.x) {
        double le = fmax(cgain * sqrt((double)ws.ctl[i]), 1.0e-30);  // control envelope = gain |u| (ctl holds |u|^2)
        le = HP_LEVEL + 20.0 * log10(le);
        le = fmin(fmax(le, ls.lowknee), 100.0);
        const double g = -ls.attnOHC - (le - ls.lowknee) * (1.0 - (1.0 / CR));
        ws.ctl[i] = (hp_env_t)exp(g * (2.302585092994046 / 20.0));   // 10^(g/20)
    }
}
#endif  // NELE_AB

// ---- h6: gain low-pass lfilter([b,b],[1,a]) (pyhaspi2.py:992-995), serial, in place on ctl (one stream per lane). grid B, block 64
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(64) void haspi_gainlp_kernel(HaspiWs ws, int sig0, int nsig, int nrows) {
    const int lane = threadIdx.x, idx = 2 * blockIdx.x + (lane >> 5);
    if (idx >= nrows) return;
    hp_env_t* g = ws.ctl + ((size_t)hp_row(idx, sig0, nsig) * ws.n24p) * HP_NCH + (lane & 31);
    const double b0 = 0.095107983402496, a1 = -0.809784033195007;
    double z = 0.0;
    for (int n0 = 0; n0 < ws.n24; n0 += HP_CH) {
        double gx[HP_CH];
#pragma unroll
        for (int u = 0; u < HP_CH; ++u) gx[u] = (double)g[(size_t)(n0 + u) * HP_NCH];
#pragma unroll
        for (int u = 0; u < HP_CH; ++u) {
            const double y = b0 * gx[u] + z;
            z = b0 * gx[u] - a1 * y;
            gx[u] = y;
        }
#pragma unroll
        for (int u = 0; u < HP_CH; ++u) g[(size_t)(n0 + u) * HP_NCH] = (hp_env_t)gx[u];
    }
}
#endif  // NELE_AB

// ---- h7: compressed envelope = filtered gain * envelope (pyhaspi2.py:997) and eb_EnvSL2 (pyhaspi2.py:1080-1088), point-wise
// grid (blocks, 2 B): blockIdx.y = (utterance, signal), whose adjusted bandwidth fixes the signal filter's gain per channel
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ void haspi_sl_kernel(HaspiWs ws, size_t per_row, int sig0, int nsig) {
    const int ch = (int)(threadIdx.x & 31), row = hp_row(blockIdx.y, sig0, nsig);
    const double sgain = hp_gt(ws.bw[(size_t)row * HP_NCH + ch], hp_cfreq(ch)).gain;
    const double attnIHC = hp_loss(ws, row & 1, ch).attnIHC;
    const size_t base = (size_t)row * per_row;
    for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < per_row; r += (size_t)gridDim.x * blockDim.x) {
        const size_t i = base + r;
        const double c = (double)ws.ctl[i] * (sgain * sqrt((double)ws.env[i]));      // signal envelope = gain |u| (env holds |u|^2)
        const double y = HP_LEVEL - attnIHC + 20.0 * log10(c + 1.0e-30);
        ws.env[i] = (hp_env_t)(y < 0.0 ? 0.0 : y);
    }
}
#endif  // NELE_AB

// ---- h5-h7 fused: compression gain (point-wise) -> gain low-pass lfilter([b,b],[1,a]) -> compressed envelope in dB SL, one pass.
// The three separate kernels moved 12.6 GB arrays (B = 256) through HBM seven times for 22 ms.  The low-pass is a first-order
// recursion with pole 0.81: its state forgets the past as 0.81^n, so a thread can start 256 samples before its chunk from a zero
// state - the neglected history is below 0.81^256 = 4e-24 of the signal, far under the float64 rounding of the values themselves -
// which makes the recursion parallel over chunks.  Thread = (chunk of 2048 samples, channel); block = 8 chunks x 32 channels;
// grid (ceil(n24p / 16384), 2 B).  ctl (|u|^2 of the control bank) is only read, env (|u|^2 of the signal bank) is rewritten in place.

__global__ __launch_bounds__(256) void haspi_gain_lp_sl_kernel(HaspiWs ws, int sig0, int nsig) {
    const int ch = threadIdx.x & 31, row = hp_row(blockIdx.y, sig0, nsig);
    const int n0 = (blockIdx.x * 8 + (threadIdx.x >> 5)) * ws.lcg;
    const int n24r = (hp_n24(ws, row >> 1) + HP_CH - 1) / HP_CH * HP_CH;   // whole register chunks of this row
    if (n0 >= n24r) return;
    const int n1 = min(n0 + ws.lcg, n24r);
    const double cgain = hp_gt(hp_bw1(ch), hp_cfreq(ch)).gain;
    const double sgain = hp_gt(ws.bw[(size_t)row * HP_NCH + ch], hp_cfreq(ch)).gain;
    const HpLoss ls = hp_loss(ws, row & 1, ch);
    const float slope = (float)(1.0 - (1.0 / ls.CR)), knee = (float)ls.lowknee, attn = (float)ls.attnOHC;
    // 20 log10(gain sqrt(c)) = 20 log10(gain) + 10 log10(c) = 20 log10(gain) + (10 log10 2) log2(c)
    const float TEN_LOG10_2 = 3.0102999566398120f, LOG2_10_OVER_20 = 0.16609640474436813f;
    const float c_off = (float)(HP_LEVEL + 20.0 * log10(cgain)), s_off = (float)(HP_LEVEL - ls.attnIHC + 20.0 * log10(sgain));
    const double b0 = 0.095107983402496, a1 = -0.809784033195007;
    const hp_env_t* ctl = ws.ctl + (size_t)row * ws.n24p * HP_NCH + ch;
    hp_env_t* env = ws.env + (size_t)row * ws.n24p * HP_NCH + ch;
    double z = 0.0;
    const IhcC ik = hp_ihc_consts();
    double V1 = 0.0, V2 = 0.0;                                 // IHC adaptation pass 1: this chunk from a zero state
    for (int n = max(0, n0 - GL_W); n < n1; n += GL_U) {       // n0, GL_W and n24p are multiples of GL_U
        float gc[GL_U], ev[GL_U];
        double gx[GL_U];
        const bool live = n >= n0;
#pragma unroll
        for (int u = 0; u < GL_U; ++u) gc[u] = ctl[(size_t)(n + u) * HP_NCH];
        if (live) {
#pragma unroll
            for (int u = 0; u < GL_U; ++u) ev[u] = env[(size_t)(n + u) * HP_NCH];
        }
#pragma unroll
        for (int u = 0; u < GL_U; ++u) {                       // pyhaspi2.py:982-991 (the 1e-30 floor lies far below the 30 dB clamp)
            float le = c_off + TEN_LOG10_2 * hp_log2f(gc[u]);
            le = fminf(fmaxf(le, knee), 100.0f);
            gx[u] = (double)hp_exp2f((-attn - (le - knee) * slope) * LOG2_10_OVER_20);    // 10^(g/20)
        }
#pragma unroll
        for (int u = 0; u < GL_U; ++u) {                       // pyhaspi2.py:992-995
            const double y = b0 * gx[u] + z;
            z = b0 * gx[u] - a1 * y;
            gx[u] = y;
        }
        if (live) {
#pragma unroll
            for (int u = 0; u < GL_U; ++u) {                   // pyhaspi2.py:997, 1080-1088: 20 log10(g sgain sqrt(e) + 1e-30), clamped at 0
                // (the 1e-30 term only matters where the result is far below the clamp)
                const float g = (float)gx[u];
                float y = s_off + TEN_LOG10_2 * hp_log2f(g * g * ev[u]);
                y = y > 0.0f ? y : 0.0f;
                env[(size_t)(n + u) * HP_NCH] = y;
                hp_ihc_step(ik, (double)y, V1, V2);            // on the stored (float32) value: pass 2 reads exactly that
            }
        }
    }
    const int ncg = (ws.n24p + ws.lcg - 1) / ws.lcg;
    double* ihe = ws.ihe + ((size_t)row * ncg + (n0 / ws.lcg)) * 64 + ch;
    ihe[0] = V1; ihe[32] = V2;
}

// eb_IHCadapt (pyhaspi2.py:1028-1078): a LINEAR two-state recurrence driven by the dB-SL envelope; the max(., 0) on the output does
// not feed back (pyhaspi2.py:1072), so the state is a linear scan like the filter banks': pass 1 (inside the gain pass, which walks
// every chunk of GL_N samples in order anyway) runs the recurrence from a zero state and keeps the end state, pass 2
// (haspi_ihc_scan_kernel) starts every chunk from the Horner-combined true state.  P over GL_N samples comes from iterating the
// homogeneous recurrence itself (haspi_shift_kernel).
// IHC end states of pass 1 (inside the gain pass) -> start states of pass 2, in place.  grid rows, block 32 (channel).
__global__ __launch_bounds__(32) void haspi_ihc_prefix_kernel(HaspiWs ws, int sig0, int nsig) {
    const int row = hp_row(blockIdx.x, sig0, nsig), ch = threadIdx.x;
    const int n24 = hp_n24(ws, row >> 1), ncg = (ws.n24p + ws.lcg - 1) / ws.lcg;
    const int nch = ((n24 + HP_CH - 1) / HP_CH * HP_CH + ws.lcg - 1) / ws.lcg;            // chunks the gain pass produced for this row
    double* ihe = ws.ihe + ((size_t)row * ncg) * 64 + ch;
    const double p00 = ws.pihc[0], p01 = ws.pihc[1], p10 = ws.pihc[2], p11 = ws.pihc[3];
    double V1 = 0.0, V2 = 0.0;
    constexpr int G = 8;
    for (int j0 = 0; j0 < nch; j0 += G) {
        double e1[G], e2[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { const int j = min(j0 + g, nch - 1); e1[g] = ihe[(size_t)j * 64]; e2[g] = ihe[(size_t)j * 64 + 32]; }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (j0 + g < nch) {
                ihe[(size_t)(j0 + g) * 64] = V1; ihe[(size_t)(j0 + g) * 64 + 32] = V2;
                const double t1 = (p00 * V1 + p01 * V2) + e1[g], t2 = (p10 * V1 + p11 * V2) + e2[g];
                V1 = t1; V2 = t2;
            }
        }
    }
}

// IHC pass 2 FUSED with ebm_EnvFilt (pyhaspi2.py:378-414: Hann(52)/sum FIR at "same" alignment, every 9th sample, after the
// group-delay shift).  The adapted envelope never goes back to memory: a thread (chunk of GL_N samples, channel) runs the IHC
// recurrence and feeds its output - rounded to float32 exactly as the stored envelope of the unfused path was - into the FIR in
// SLIDING form: np.hanning(52)[k] = 0.5 - 0.5 cos(2 pi k / 51) has zero end taps, so the window is one period L = 51 of
//     S_m[n] = e^{j m phi} S_m[n - 1] + o[n] - o[n - 51],      y[n] = (0.5 S_0 - 0.5 Re S_1)[n] / 25.5,
// 6 operations per sample instead of 52 / 9 multiply-adds + tap look-ups; o[n - 51] comes from a 64-deep float32 ring in LDS.
// The group-delay shift s of the channel only relabels the lane's own time axis: output i is emitted when its window ends,
// n = 9 i + 26 - s, and samples with n + s >= n24 count as zero.  A chunk needs o[n] from 51 samples before its start: the IHC
// recurrence is invertible (its inverse grows by 1.02 per step), so the thread steps its Horner-combined state 51 samples BACK and
// reruns them forward.  Every output is written by exactly one thread (the one whose chunk holds its window end): no atomics.
// The separate envelope-filter kernel below gathered 4 bytes per lane from 32 different rows (the shifts differ per channel): 64
// cache lines per load instruction, 4.2 ms per call at B = 256.
// grid (ceil(chunks / 4), rows), block 128 = 4 chunks x 32 channels.
#define IF_L 51
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(128) void haspi_ihc_fir_kernel(HaspiWs ws, int sig0, int nsig) {
    __shared__ float ring[64][128];
    const int tid = threadIdx.x, ch = tid & 31, row = hp_row(blockIdx.y, sig0, nsig), b = row >> 1;
    const int lcf = ws.lcg * ws.fmul;                          // samples per thread
    const int chunk = blockIdx.x * 4 + (tid >> 5), n0 = chunk * lcf;
    const int n24 = hp_n24(ws, b), nsub = hp_nsub(ws, b);
    const int ncg = (ws.n24p + ws.lcg - 1) / ws.lcg;
    const int last = (n24 - 1) / lcf;                       // chunk that holds the row's last sample
#pragma unroll
    for (int q = 0; q < 64; ++q) ring[q][tid] = 0.f;          // own column only: no barrier needed
    if (chunk > last) return;
    const int sh = ws.shift[(size_t)b * HP_NCH + ch];
    // this chunk emits the outputs whose window ends in [n0, n1); the last chunk also those that end behind the row's end
    const int n1 = (chunk == last) ? n24 + 9 + 26 : n0 + lcf;
    const IhcC k = hp_ihc_consts();
    const double* ihe = ws.ihe + ((size_t)row * ncg + (size_t)chunk * ws.fmul) * 64 + ch;
    double V1 = ihe[0], V2 = ihe[32];                        // true state at the chunk start (haspi_ihc_prefix_kernel)
    const hp_env_t* e = ws.env + ((size_t)row * ws.n24p) * HP_NCH + ch;
    int nstart = n0;
    if (chunk > 0) {                                         // state at n0 -> state at n0 - 51 (inverse of hp_ihc_step)
        nstart = n0 - IF_L;
        float hb[IF_L];                                      // all 51 loads first, then the (serial) backward steps
#pragma unroll
        for (int q = 0; q < IF_L; ++q) hb[q] = e[(size_t)(n0 - 1 - q) * HP_NCH];
        const double iR12 = 1.0 / k.R12C1, iR23 = 1.0 / k.R23C2;
#pragma unroll
        for (int q = 0; q < IF_L; ++q) {
            const double V0 = (double)hb[q];
            const double b1 = k.a11 * V1 + k.a12 * V2, b2 = k.a21 * V1 + k.a22 * V2;
            V1 = (b1 - V0 * k.R2) * iR12;
            V2 = b2 * iR23;
        }
    }
    double* lp = ws.lp + ((size_t)row * ws.nlp) * HP_NCH + ch;
    if (chunk == 0) {                                        // outputs whose whole window lies before the lane's first sample
        for (int i = 0; i < nsub && 9 * i + 26 - sh < 0; ++i) lp[(size_t)i * HP_NCH] = 0.0;
    }
    const double phi = 2.0 * M_PI / (double)IF_L, rc = cos(phi), rs = sin(phi);
    double s0 = 0.0, s1r = 0.0, s1i = 0.0;
    // n + s - 26 = 9 icur + ph with 0 <= ph < 9 (floor division): output icur is due at the sample where ph == 0
    int icur, ph;
    {
        const int t = nstart + sh - 26;
        icur = (t >= 0) ? t / 9 : -((-t + 8) / 9);
        ph = t - 9 * icur;
    }
    float nx[GL_U];                                          // the next group's samples are in flight while this group is processed
#pragma unroll
    for (int u = 0; u < GL_U; ++u) nx[u] = e[(size_t)min(nstart + u, n24 - 1) * HP_NCH];
    for (int nb = nstart; nb < n1; nb += GL_U) {
        float ex[GL_U], olds[GL_U], news[GL_U];
#pragma unroll
        for (int u = 0; u < GL_U; ++u) ex[u] = nx[u];
#pragma unroll
        for (int u = 0; u < GL_U; ++u) nx[u] = e[(size_t)min(nb + GL_U + u, n24 - 1) * HP_NCH];
        // the samples leaving the window during this group (slots n - 51 .. n - 44) are all older than anything the group writes: one
        // LDS round trip per group instead of one per sample on the serial chain
#pragma unroll
        for (int u = 0; u < GL_U; ++u) olds[u] = ring[(nb + u - IF_L) & 63][tid];
        double yv = 0.0;
        int yi = -1;                                         // at most one output falls into 8 consecutive samples (they are 9 apart)
#pragma unroll
        for (int u = 0; u < GL_U; ++u) {
            const int n = nb + u;
            float o = 0.f;
            if (n + sh < n24) {                              // (n < n24 follows; beyond it the shifted envelope is zero)
                const double V0 = (double)ex[u];
                hp_ihc_step(k, V0, V1, V2);
                const double out = (V0 - V1) * k.R1inv;
                o = (float)(out < 0.0 ? 0.0 : out);
            }
            news[u] = o;
            const double dx = (double)o - (double)olds[u];
            s0 += dx;
            { const double nr = (rc * s1r - rs * s1i) + dx; s1i = rc * s1i + rs * s1r; s1r = nr; }
            const bool due = (ph == 0) && n >= n0 && n < n1;
            yv = due ? (0.5 * s0 - 0.5 * s1r) * (1.0 / 25.5) : yv;
            yi = due ? icur : yi;
            const bool wrap = ph == 8;
            ph = wrap ? 0 : ph + 1;
            icur += wrap ? 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < GL_U; ++u) ring[(nb + u) & 63][tid] = news[u];
        if (yi >= 0 && yi < nsub) lp[(size_t)yi * HP_NCH] = yv;
    }
}
#endif  // NELE_AB

// The same pass in groups of NINE samples.  Outputs are 9 samples apart, so with groups that start on multiples of 9 a lane's output
// always falls on the SAME sample u_lane = (26 - shift) mod 9 of a group: the phase counters, bounds tests and output selects of the
// kernel above (62 instructions per sample, issue-bound) shrink to one compare-select of s0 - Re s1 per sample (groups that reach
// n + shift >= n24 take a predicated copy of the body).  The time axis stays common to the wave - per-lane axes aligned to each
// channel's own phase were measured: one cache line per LANE and load, bound by the L1 tag rate.
// The window's trailing edge o[n - 51] lies 5 groups + 6 samples back, so the delay line is 6 groups = 54 values updated in place
// (sample u of a group reads position u + 3 of the oldest group, or u - 6 of the next one, before position u is overwritten) - and with
// the loop unrolled over those 6 groups every position is a compile-time index: the delay line lives in 54 REGISTERS.  (In LDS, 252
// bytes per thread, it capped the kernel at 2.5 waves per SIMD - 1.6 on average with the last round of workgroups - and the waves
// spent half their time waiting: 1.43 ms per call at B = 256 against 0.63 ms of instruction issue.)
// Warm-up: the first group starts 51 .. 59 samples before the chunk (the IHC state is stepped back that far) with a zero delay line -
// by the chunk's first output the window is complete.
#define IF_G 9
#define IF_SLOTS 6
__global__ __launch_bounds__(128) void haspi_ihc_fir9_kernel(HaspiWs ws, int sig0, int nsig) {
    const int tid = threadIdx.x, ch = tid & 31, row = hp_row(blockIdx.y, sig0, nsig), b = row >> 1;
    const int lcf = ws.lcg * ws.fmul;                          // samples per thread
    const int chunk = blockIdx.x * 4 + (tid >> 5), n0 = chunk * lcf;
    const int n24 = hp_n24(ws, b), nsub = hp_nsub(ws, b);
    const int ncg = (ws.n24p + ws.lcg - 1) / ws.lcg;
    const int last = (n24 - 1) / lcf;                          // chunk that holds the row's last sample
    if (chunk > last) return;
    const int sh = ws.shift[(size_t)b * HP_NCH + ch];
    // this chunk emits the outputs whose window ends in [n0, n1); the last chunk also those that end behind the row's end
    const int n1 = (chunk == last) ? n24 + 9 + 26 : n0 + lcf;
    const IhcC k = hp_ihc_consts();
    const double* ihe = ws.ihe + ((size_t)row * ncg + (size_t)chunk * ws.fmul) * 64 + ch;
    double V1 = ihe[0], V2 = ihe[32];                          // true state at the chunk start (haspi_ihc_prefix_kernel)
    const hp_env_t* e = ws.env + ((size_t)row * ws.n24p) * HP_NCH + ch;
    // output i is due at n = 9 i + 26 - sh = 9 g + ul in group g: i = g + di (hp_lp_di)
    const int ul = (((26 - sh) % 9) + 9) % 9;
    int nG = (chunk > 0) ? (n0 - IF_L) / 9 * 9 : 0;            // first group: 51 .. 59 samples before the chunk
    if (chunk > 0) {                                           // state at n0 -> state at nG (inverse of hp_ihc_step; it grows by 1.02 per step)
        const int cnt = n0 - nG;
        float hb[IF_L + IF_G - 1];                             // all loads first, then the (serial) backward steps
#pragma unroll
        for (int q = 0; q < IF_L + IF_G - 1; ++q) hb[q] = e[(size_t)(n0 - 1 - q) * HP_NCH];
        const double idet = 1.0 / (k.c11 * k.c22 - k.c12 * k.c21);
#pragma unroll
        for (int q = 0; q < IF_L + IF_G - 1; ++q) {
            const double V0 = (double)hb[q];
            const double w1 = V1 - k.c10 * V0, w2 = V2 - k.c20 * V0;
            const double o1 = (k.c22 * w1 - k.c12 * w2) * idet, o2 = (k.c11 * w2 - k.c21 * w1) * idet;
            V1 = q < cnt ? o1 : V1;
            V2 = q < cnt ? o2 : V2;
        }
    }
    // group-space rows (ws.lp_raw, see hp_lp_di): row g holds every channel's output of group g - the consumers read frame i of channel c
    // from row i - di(c) and take rows g < 0 (windows that end before the lane's first sample) as zeros
    double* lp = ws.lp + ((size_t)row * ws.nlp) * HP_NCH + ch;
    const double phi = 2.0 * M_PI / (double)IF_L, rc = cos(phi), rs = sin(phi);
    double s0 = 0.0, s1r = 0.0, s1i = 0.0;
    const hp_env_t* ep = e + (size_t)nG * HP_NCH;              // (the prefetch runs a few rows past the row's end: still inside the workspace,
                                                               //  and what it reads there is replaced before use)
    float nx[IF_G];                                            // the next group's samples are in flight while this group is processed
#pragma unroll
    for (int u = 0; u < IF_G; ++u) nx[u] = ep[u * HP_NCH];
    float dl[IF_SLOTS * IF_G];                                 // the delay line: every index below is a compile-time constant
#pragma unroll
    for (int q = 0; q < IF_SLOTS * IF_G; ++q) dl[q] = 0.f;
    int i = nG / 9;                                            // group = row of lp
    auto group = [&](auto slot_tag, auto edge_tag) {
        constexpr int S = decltype(slot_tag)::value, S1 = (S + 1) % IF_SLOTS;
        constexpr bool EDGE = decltype(edge_tag)::value;
        float ex[IF_G];
#pragma unroll
        for (int u = 0; u < IF_G; ++u) ex[u] = nx[u];
#pragma unroll
        for (int u = 0; u < IF_G; ++u) nx[u] = ep[(IF_G + u) * HP_NCH];
        double z = 0.0;
#pragma unroll
        for (int u = 0; u < IF_G; ++u) {
            const double V0 = (double)ex[u];
            hp_ihc_step(k, V0, V1, V2);
            float o = (float)fmax((V0 - V1) * k.R1inv, 0.0);
            if (EDGE) o = (nG + u + sh < n24) ? o : 0.f;       // beyond the row's end the shifted envelope is zero
            const float old = (u < 6) ? dl[S * IF_G + u + 3] : dl[S1 * IF_G + u - 6];     // o[n - 51]
            dl[S * IF_G + u] = o;
            const double dx = (double)o - (double)old;
            s0 += dx;
            { const double nr = fma(-rs, s1i, fma(rc, s1r, dx)); s1i = fma(rc, s1i, rs * s1r); s1r = nr; }
            z = (u == ul) ? s0 - s1r : z;
        }
        const int nE = nG + ul;
        if (nE >= n0 && nE < n1 && i < ws.nlp) lp[(size_t)i * HP_NCH] = (0.5 * z) * (1.0 / 25.5);
        nG += IF_G; ep += IF_G * HP_NCH; ++i;
    };
#define IF_GROUP(S)                                                                                              \
    if (nG >= n1) break;                                                                                         \
    if (nG + (IF_G - 1) + sh >= n24) group(std::integral_constant<int, S>{}, std::true_type{});                  \
    else group(std::integral_constant<int, S>{}, std::false_type{});
    for (;;) { IF_GROUP(0) IF_GROUP(1) IF_GROUP(2) IF_GROUP(3) IF_GROUP(4) IF_GROUP(5) }
#undef IF_GROUP
}

// ---- h8: eb_IHCadapt (pyhaspi2.py:1028-1078), serial, in place on env. grid B, block 64
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(64) void haspi_ihc_kernel(HaspiWs ws, int sig0, int nsig, int nrows) {
    const int lane = threadIdx.x, idx = 2 * blockIdx.x + (lane >> 5);
    if (idx >= nrows) return;
    const int row = hp_row(idx, sig0, nsig), n24 = hp_n24(ws, row >> 1);
    hp_env_t* e = ws.env + ((size_t)row * ws.n24p) * HP_NCH + (lane & 31);
    const IhcC k = hp_ihc_consts();
    double V1 = 0.0, V2 = 0.0;
    for (int n0 = 0; n0 < n24; n0 += HP_CH) {
        double ex[HP_CH];
#pragma unroll
        for (int u = 0; u < HP_CH; ++u) ex[u] = (double)e[(size_t)(n0 + u) * HP_NCH];
#pragma unroll
        for (int u = 0; u < HP_CH; ++u) {
            const double V0 = ex[u];
            hp_ihc_step(k, V0, V1, V2);
            const double out = (V0 - V1) * k.R1inv;
            ex[u] = out < 0.0 ? 0.0 : out;
        }
#pragma unroll
        for (int u = 0; u < HP_CH; ++u) e[(size_t)(n0 + u) * HP_NCH] = (hp_env_t)ex[u];
    }
}
#endif  // NELE_AB

// ---- h9a: group-delay shifts from BWx (pyhaspi2.py:1098-1131; both envelopes use BWx, :1239-1240). grid B, block 64
__global__ __launch_bounds__(64) void haspi_shift_kernel(HaspiWs ws) {
    const int b = blockIdx.x, ch = threadIdx.x;
    double gd = 0.0;
    if (ch < HP_NCH) {
        const GtCoef c = hp_gt(ws.bw[(size_t)b * 2 * HP_NCH + ch], hp_cfreq(ch));
        const double a = 0.25 * c.a1;
        gd = rint((c.a1 + 2.0 * c.a5) / (1.0 + c.a1 + c.a5) + 4.0 * a / (1.0 - a));   // group delay at w = 0, np.round
    }
    double mn = (ch < HP_NCH) ? gd : 1e300, mx = (ch < HP_NCH) ? gd : -1e300;
    for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o, 64)); mx = fmax(mx, __shfl_xor(mx, o, 64)); }
    if (ch < HP_NCH) ws.shift[(size_t)b * HP_NCH + ch] = (int)((mx - mn) - (gd - mn));
    if (b == 0 && ch < 2) {                              // IHC state transition over one gain-pass chunk: column ch of the 2x2 matrix
        const IhcC k = hp_ihc_consts();
        double V1 = (ch == 0), V2 = (ch == 1);
        for (int n = 0; n < ws.lcg; ++n) hp_ihc_step(k, 0.0, V1, V2);
        ws.pihc[ch] = V1; ws.pihc[2 + ch] = V2;
    }
    if (b == 0 && ch < HP_NFILT) ws.benv[ch] = (0.5 - 0.5 * cospi(2.0 * (double)ch / 51.0)) / 25.5;   // np.hanning(52) / sum, for haspi_envfilt_kernel
    if (b == 0) {                                        // modulation-filter taps (np.hanning(nfir + 1) / sum), for haspi_mod_kernel
        const int nfirs[10] = {614, 614, 614, 384, 244, 152, 96, 60, 38, 24};
        for (int k = 0; k < 10; ++k)
            for (int i = ch; i <= nfirs[k]; i += 64) ws.bkt[k * 616 + i] = (0.5 - 0.5 * cospi(2.0 * (double)i / (double)nfirs[k])) / (0.5 * (double)nfirs[k]);
    }
}

// ---- h9b: ebm_EnvFilt (pyhaspi2.py:378-414): Hann(52)/sum FIR, "same" alignment (nhalf = 26), every 9th sample.
// grid (ceil(nsub/EF_SUB), B, 2), block 256: the EF_SUB*9 + 52 input samples of a block (per channel, group-delay shift
// applied while loading) are staged in LDS once; thread = (sub-frame, channel).
#define EF_SUB 16
#define EF_SPAN (EF_SUB * HP_SPACE + HP_NFILT)
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(256) void haspi_envfilt_kernel(HaspiWs ws, int sig0) {
    __shared__ double xs[EF_SPAN][HP_NCH + 1];
    const int b = blockIdx.y, sig = sig0 + blockIdx.z, tid = threadIdx.x, ch = tid & 31;
    const int i0 = blockIdx.x * EF_SUB;
    const int n24 = hp_n24(ws, b), nsub = hp_nsub(ws, b);
    if (i0 >= nsub) return;
    const double* __restrict__ benv = ws.benv;           // uniform index -> scalar loads: the taps cost no LDS read (the loop was bound by LDS issue)
    const int s = ws.shift[(size_t)b * HP_NCH + ch];
    const hp_env_t* e = ws.env + (((size_t)b * 2 + sig) * ws.n24p) * HP_NCH + ch;
    // LDS row q <-> shifted-envelope index m = 9*i0 + 26 - 51 + q
    const int mbase = HP_SPACE * i0 + HP_NHALF - (HP_NFILT - 1);
    {   // all loads of the thread are issued before the first LDS store (unconditional, clamped index, then select)
        constexpr int NQ = (EF_SPAN + 7) / 8;
        double val[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = (tid >> 5) + 8 * u;
            const int src = mbase + q - s;
            val[u] = (double)e[(size_t)min(max(src, 0), n24 - 1) * HP_NCH];
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = (tid >> 5) + 8 * u;
            const int m = mbase + q, src = m - s;
            if (q < EF_SPAN) xs[q][ch] = (m >= 0 && m < n24 && src >= 0) ? val[u] : 0.0;
        }
    }
    __syncthreads();
    for (int li = tid >> 5; li < EF_SUB; li += 8) {
        const int i = i0 + li;
        if (i >= nsub) break;
        double acc = 0.0;
        // out[i] = sum_k benv[k] * x[9 i + 26 - k]  ->  LDS row (9 li + 51 - k)
#pragma unroll 4
        for (int k = 0; k < HP_NFILT; ++k) acc += benv[k] * xs[HP_SPACE * li + (HP_NFILT - 1) - k][ch];
        ws.lp[(((size_t)b * 2 + sig) * ws.nlp + i) * HP_NCH + ch] = acc;
    }
}
#endif  // NELE_AB

// ---- h10: ebm_CepCoef (pyhaspi2.py:342-375), parallel over sub-sampled frames (one block per utterance walking 10 667 frames - 32
// float64 pow() each for the silence gate - took 0.75 ms alone and 4.8 ms beside the convolutions):
//   haspi_gate_kernel      silence gate on the REFERENCE envelope, 256 frames per block: flag + rank inside the block + block count
//   haspi_gate_scan_kernel block offsets (ordered compaction), n_active, status
//   haspi_cepstra_kernel   cepstral coefficients of the active frames (+ dither) at their compacted position, block partial sums
//   haspi_cepmean_kernel   sequence means from the partials in block order; the consumers (modulation filters) subtract them on load
#define CP_F 128
// the block's [CP_F frames][32 channels] tile of the low-passed envelope, loaded with coalesced reads (a thread that walks its own
// 256-byte row touches 64 cache lines per load instruction) and padded against bank conflicts
__device__ __forceinline__ void hp_stage_lp(const HaspiWs& ws, int b, const double* __restrict__ lp, int i0, int nsub, double (*tile)[HP_NCH + 1]) {
    const int nfr = min(CP_F, nsub - i0);
    const int di = hp_lp_di(ws, b, threadIdx.x & 31);      // (CP_F is a multiple of 32: a thread keeps its channel)
    for (int e = threadIdx.x; e < CP_F * HP_NCH; e += CP_F) tile[e >> 5][e & 31] = ((e >> 5) < nfr) ? hp_lp_at(lp, i0 + (e >> 5), e & 31, di) : 0.0;
}
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(CP_F) void haspi_gate_kernel(HaspiWs ws) {
    __shared__ double tile[CP_F][HP_NCH + 1];
    __shared__ int scan[CP_F];
    const int b = blockIdx.y, tid = threadIdx.x, i0 = blockIdx.x * CP_F, i = i0 + tid;
    const int nsub = hp_nsub(ws, b);
    int k = 0;
    if (i0 < nsub) {
        hp_stage_lp(ws, b, ws.lp + ((size_t)b * 2) * ws.nlp * HP_NCH, i0, nsub, tile);
        __syncthreads();
        if (i < nsub) {                                    // 20 log10(mean_k 10^(x/20)) > 2.5
            double s = 0.0;
            for (int c = 0; c < HP_NCH; ++c) s += exp10(tile[tid][c] / 20.0);      // (10^x: half the instructions of the general pow)
            k = (20.0 * log10(s / (double)HP_NCH) > 2.5) ? 1 : 0;
        }
    }
    scan[tid] = k;
    __syncthreads();
    for (int o = 1; o < CP_F; o <<= 1) {
        const int v = (tid >= o) ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    if (i < ws.nsub) ws.grank[(size_t)b * ws.nsub + i] = k ? scan[tid] - 1 : -1;
    if (tid == CP_F - 1) ws.gcnt[(size_t)b * ws.ngb + blockIdx.x] = scan[CP_F - 1];
}
#endif  // NELE_AB
// grid B, block 64 (one lane works: at most a few dozen blocks)
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ void haspi_gate_scan_kernel(HaspiWs ws) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int nb = (hp_nsub(ws, b) + CP_F - 1) / CP_F;
    int* cnt = ws.gcnt + (size_t)b * ws.ngb;
    int tot = 0;
    for (int q = 0; q < nb; ++q) { const int c = cnt[q]; cnt[q] = tot; tot += c; }      // counts -> exclusive offsets
    ws.info[2 * b] = tot;
    ws.info[2 * b + 1] = (tot <= 1) ? 1 : 0;
}
#endif  // NELE_AB
// grid (blocks of CP_F frames, B, nsig), block CP_F
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(CP_F) void haspi_cepstra_kernel(HaspiWs ws, const double* __restrict__ dither, double thr_nerve, int sig0) {
    __shared__ double tile[CP_F][HP_NCH + 1];
    __shared__ double cepm[HP_NCH][HP_NBASIS];
    __shared__ double part[CP_F / 64][HP_NBASIS];
    const int b = blockIdx.y, sig = sig0 + blockIdx.z, tid = threadIdx.x, i0 = blockIdx.x * CP_F, i = i0 + tid;
    const int nsub = hp_nsub(ws, b);
    if (ws.info[2 * b + 1] || i0 >= nsub) return;
    if (tid < HP_NBASIS) {
        double nn = 0.0;
        for (int k = 0; k < HP_NCH; ++k) { const double v = cos((double)tid * M_PI * (double)k / (double)(HP_NCH - 1)); nn += v * v; }
        nn = sqrt(nn);
        for (int k = 0; k < HP_NCH; ++k) cepm[k][tid] = cos((double)tid * M_PI * (double)k / (double)(HP_NCH - 1)) / nn;
    }
    hp_stage_lp(ws, b, ws.lp + (((size_t)b * 2 + sig) * ws.nlp) * HP_NCH, i0, nsub, tile);
    __syncthreads();
    const int rank = (i < ws.nsub) ? ws.grank[(size_t)b * ws.nsub + i] : -1;
    const double* dz = dither ? dither + (((size_t)b * 2 + sig) * ws.nsub) * HP_NCH : nullptr;
    double* cep = ws.cep + (((size_t)b * 2 + sig) * HP_NBASIS) * ws.nsub;
    double c6[HP_NBASIS] = {0, 0, 0, 0, 0, 0};
    if (rank >= 0) {
        const int k = ws.gcnt[(size_t)b * ws.ngb + blockIdx.x] + rank;          // compacted position of this frame
        for (int c = 0; c < HP_NCH; ++c) {
            double v = tile[tid][c];
            if (dz) v += thr_nerve * dz[(size_t)k * HP_NCH + c];
#pragma unroll
            for (int q = 0; q < HP_NBASIS; ++q) c6[q] += v * cepm[c][q];
        }
#pragma unroll
        for (int q = 0; q < HP_NBASIS; ++q) cep[(size_t)q * ws.nsub + k] = c6[q];
    }
    // block partial sums of the six sequences (fixed order: wave butterflies, then the waves in order)
#pragma unroll
    for (int q = 0; q < HP_NBASIS; ++q) {
        const double t = wave_sum(c6[q]);
        if ((tid & 63) == 0) part[tid >> 6][q] = t;
    }
    __syncthreads();
    if (tid < HP_NBASIS) {
        double t = 0.0;
        for (int w = 0; w < CP_F / 64; ++w) t += part[w][tid];
        ws.cpsum[((((size_t)b * 2 + sig) * ws.ngb) + blockIdx.x) * HP_NBASIS + tid] = t;
    }
}
#endif  // NELE_AB
// grid (B, nsig), block 64
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ void haspi_cepmean_kernel(HaspiWs ws, int sig0) {
    const int b = blockIdx.x, sig = sig0 + blockIdx.y, q = threadIdx.x;
    if (q >= HP_NBASIS || ws.info[2 * b + 1]) return;
    const int nb = (hp_nsub(ws, b) + CP_F - 1) / CP_F;
    double t = 0.0;
    for (int g = 0; g < nb; ++g) t += ws.cpsum[((((size_t)b * 2 + sig) * ws.ngb) + g) * HP_NBASIS + q];
    ws.cmean[((size_t)b * 2 + sig) * HP_NBASIS + q] = t / (double)ws.info[2 * b];
}
#endif  // NELE_AB

// ---- h10, one block per utterance (round-2 start; kept behind NELE_HASPI_CEP_SERIAL=1): it takes 0.75 ms alone against 0.15 ms for the
// parallel kernels above, and 4.8 ms inside a step - but see the note at its launch site.
// gate != 0: silence gate on the REFERENCE envelope + ordered compaction of the active frames (needs x only);
// then the cepstral sequences of signals sig0 .. sig0+nsig-1 over those frames.
// Frames walk through LDS in tiles of 256 frames x 16 channels (two halves per tile): in the group-space layout of lp a frame's channels
// sit in up to ~40 different rows, and a thread that walked "its" frame through global memory alone touched a new cache line on nearly
// every channel (1.1 ms per call at B = 256 against 0.84 in frame-space rows; staged: the 16 lanes of a frame read neighbouring rows).
// Barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory queue (the fences of a workgroup-scope barrier on
// this target: s_waitcnt vmcnt(0)), which makes every load that was issued ahead of it - the next tile's prefetch - wait right there.
// Use where the threads exchange data through LDS alone; global data that changes hands between threads still needs __syncthreads().
__device__ __forceinline__ void hp_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__global__ __launch_bounds__(256) void haspi_cep_kernel(HaspiWs ws, const double* __restrict__ dither, double thr_nerve, int gate, int sig0,
                                                        int nsig) {
    __shared__ double cepm[HP_NCH][HP_NBASIS];
    __shared__ double tile[256][17];
    __shared__ int scan[256];
    __shared__ int base;
    __shared__ int dis[HP_NCH];
    __shared__ double red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nsub = hp_nsub(ws, b);
    if (tid >= 64 && tid < 64 + HP_NCH) dis[tid - 64] = hp_lp_di(ws, b, tid - 64);
    if (tid < HP_NBASIS) {
        double nn = 0.0;
        for (int k = 0; k < HP_NCH; ++k) { const double v = cos((double)tid * M_PI * (double)k / (double)(HP_NCH - 1)); nn += v * v; }
        nn = sqrt(nn);
        for (int k = 0; k < HP_NCH; ++k) cepm[k][tid] = cos((double)tid * M_PI * (double)k / (double)(HP_NCH - 1)) / nn;
    }
    if (tid == 0) base = 0;
    __syncthreads();
    const double* xlp = ws.lp + ((size_t)b * 2) * ws.nlp * HP_NCH;
    const double* ylp = xlp + (size_t)ws.nlp * HP_NCH;
    int* act = ws.act + (size_t)b * ws.nsub;             // compacted position of frame i among the active frames, or -1
    // Half-tiles: channels 16 h .. 16 h + 15 of frames i0 .. i0 + 255, numbered t = 2 (i0 / 256) + h.  A thread fetches 16 elements of a
    // half-tile (all loads in flight together) into registers while the block works on the previous one.
    const int cth = tid & 15, fth = tid >> 4;                    // element j of a thread: frame fth + 16 j, channel cth of the half
    double pf[16];
    auto fetch = [&](const double* lpb, int t) {
        const int i0 = (t >> 1) * 256, c = 16 * (t & 1) + cth, d = dis[c];
#pragma unroll
        for (int j = 0; j < 16; ++j) pf[j] = (i0 + fth + 16 * j < nsub) ? hp_lp_at(lpb, i0 + fth + 16 * j, c, d) : 0.0;
    };
    auto put = [&]() {                                          // registers -> tile (LDS barriers on both sides)
        hp_lds_barrier();
#pragma unroll
        for (int j = 0; j < 16; ++j) tile[fth + 16 * j][cth] = pf[j];
        hp_lds_barrier();
    };
    const int ntile2 = 2 * ((nsub + 255) / 256);
    int na;
    if (gate) {
        // silence gate on the reference: 20 log10(mean_k 10^(x/20)) > 2.5
        fetch(xlp, 0);
        double sm = 0.0;
        for (int t = 0; t < ntile2; ++t) {
            const int i = (t >> 1) * 256 + tid;
            put();
            if (t + 1 < ntile2) fetch(xlp, t + 1);
            if (i < nsub) for (int c = 0; c < 16; ++c) sm += exp10(tile[tid][c] / 20.0);     // (10^x: half the instructions of the general pow)
            if (t & 1) {
                const int k = (i < nsub && 20.0 * log10(sm / (double)HP_NCH) > 2.5) ? 1 : 0;
                sm = 0.0;
                scan[tid] = k;
                hp_lds_barrier();
                for (int o = 1; o < 256; o <<= 1) {
                    const int v = (tid >= o) ? scan[tid - o] : 0;
                    hp_lds_barrier();
                    scan[tid] += v;
                    hp_lds_barrier();
                }
                if (i < nsub) act[i] = k ? base + scan[tid] - 1 : -1;      // read back by the same thread only (the cepstrum loop below)
                hp_lds_barrier();
                if (tid == 255) base += scan[255];
                hp_lds_barrier();
            }
        }
        na = base;
        if (tid == 0) { ws.info[2 * b] = na; ws.info[2 * b + 1] = (na <= 1) ? 1 : 0; }
    } else {
        na = ws.info[2 * b];
    }
    if (na <= 1) return;
    // cepstra of the active frames (+ dither), then remove the mean of each sequence
    for (int sig = sig0; sig < sig0 + nsig; ++sig) {
        const double* lp = sig ? ylp : xlp;
        const double* dz = dither ? dither + (((size_t)b * 2 + sig) * ws.nsub) * HP_NCH : nullptr;
        double* cep = ws.cep + (((size_t)b * 2 + sig) * HP_NBASIS) * ws.nsub;
        double sums[HP_NBASIS] = {0, 0, 0, 0, 0, 0};
        double c6[HP_NBASIS] = {0, 0, 0, 0, 0, 0};
        fetch(lp, 0);
        for (int t = 0; t < ntile2; ++t) {
            const int i = (t >> 1) * 256 + tid, h = t & 1;
            const int k = (i < nsub) ? act[i] : -1;
            put();
            if (t + 1 < ntile2) fetch(lp, t + 1);
            if (k >= 0) {
                for (int c = 0; c < 16; ++c) {
                    double v = tile[tid][c];
                    if (dz) v += thr_nerve * dz[(size_t)k * HP_NCH + 16 * h + c];
#pragma unroll
                    for (int q = 0; q < HP_NBASIS; ++q) c6[q] += v * cepm[16 * h + c][q];
                }
            }
            if (h) {
                if (k >= 0) {
#pragma unroll
                    for (int q = 0; q < HP_NBASIS; ++q) { cep[(size_t)q * ws.nsub + k] = c6[q]; sums[q] += c6[q]; }
                }
#pragma unroll
                for (int q = 0; q < HP_NBASIS; ++q) c6[q] = 0.0;
            }
        }
        for (int q = 0; q < HP_NBASIS; ++q) {
            const double mu = block_sum(sums[q], red) / (double)na;
            __syncthreads();
            for (int k = tid; k < na; k += 256) cep[(size_t)q * ws.nsub + k] -= mu;
            if (tid == 0) ws.cmean[((size_t)b * 2 + sig) * HP_NBASIS + q] = 0.0;    // already removed (the consumers subtract cmean)
            __syncthreads();
        }
    }
}

// ---- h11: ebm_ModFilt + ebm_ModCorr for one (modulation band, basis, utterance). grid (10, 5, B), block 256
__constant__ double c_modcf[HP_NMOD] = {2, 6, 10, 16, 25, 40, 64, 100, 160, 256};
__constant__ int c_modnfir[HP_NMOD] = {614, 614, 614, 384, 244, 152, 96, 60, 38, 24};
#define MS_MAXC 128           // chunks of MS_TC outputs per utterance at most (nsub <= 131 072)
#define HP_TILE 1024          // outputs per tile: 256 threads x 4 consecutive outputs (sliding register window over the taps)
#define HP_MAXFIR 614
// LDS position of sequence element e: a thread reads elements 4 tid + c, so the four residues mod 4 live in four sub-arrays and a
// wave's reads are consecutive (the plain layout put a wave's 64 reads on 8 banks)
#define MF_L4 ((HP_TILE + HP_MAXFIR + 4 + 3) / 4 + 1)
#define MF_POS(e) ((((e) & 3) * MF_L4) + ((e) >> 2))

// SIG = 0 (clean part): filter the reference's cepstral sequence and keep the filtered sequence xf [b][basis-1][band][t];
// SIG = 1 (degraded part): filter the processed signal's sequence and correlate it with the stored xf (ebm_ModCorr).
// Each pass stages (v cos, v sin) of ONE signal: half of the LDS and of the per-tap work of a joint pass; the reference half runs
// before the enhanced signal exists (GanTrainer overlaps it with the G-step).
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
template <int SIG>
__global__ __launch_bounds__(256) void haspi_mod_direct_kernel(HaspiWs ws) {
    __shared__ __attribute__((aligned(16))) double2 sq[4 * MF_L4];   // (v cos, v sin) of one sequence element: one 16-byte read per tap
    __shared__ double red[8];
    const int k = blockIdx.x, basis = blockIdx.y + 1, b = blockIdx.z, tid = threadIdx.x;
    const int na = ws.info[2 * b];
    if (ws.info[2 * b + 1]) return;
    const int nfir = c_modnfir[k], nh = nfir / 2;
    // np.hanning(nfir+1) / sum ; sum of a symmetric Hann window of M points = (M-1)/2
    const double* __restrict__ bk = ws.bkt + k * 616;    // uniform index in the tap loop -> scalar loads
    const double* vc = ws.cep + (((size_t)b * 2 + SIG) * HP_NBASIS + basis) * ws.nsub;
    const double vmu = ws.cmean[((size_t)b * 2 + SIG) * HP_NBASIS + basis];          // the sequence's mean (ebm_CepCoef removes it)
    double* xf = ws.xf + (((size_t)b * (HP_NBASIS - 1) + (basis - 1)) * HP_NMOD + k) * ws.nsub;
    const double cf = c_modcf[k];
    const double SQ2 = 1.4142135623730951;
    double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    for (int t0 = 0; t0 < na; t0 += HP_TILE) {
        __syncthreads();
        // demodulated inputs for outputs t0 .. t0+1023: input index j = t + nh - i, i = 0..nfir -> j in [t0 - nh, t0 + 1023 + nh]
        for (int e = tid; e < HP_TILE + nfir; e += 256) {
            const int j = t0 - nh + e;
            const int jc = min(max(j, 0), na - 1);
            double v = vc[jc] - vmu, c = 1.0, s = 0.0;
            if (!(j >= 0 && j < na)) v = 0.0;
            if (k > 0) {
                // sqrt(2) cos(pi n cf / fNyq), n = j + 1, fNyq = 1280
                const double ang = M_PI * (double)(j + 1) * cf / 1280.0;
                c = SQ2 * cos(ang);
                s = SQ2 * sin(ang);
            }
            sq[MF_POS(e)] = make_double2(v * c, v * s);
        }
        __syncthreads();
        // thread -> outputs t = t0 + 4 tid + q, q = 0..3:  u[t] = sum_i b[i] z[t + nh - i]  (LDS index 4 tid + q + nfir - i)
        const int tb = t0 + 4 * tid;
        if (tb < na) {
            double ur[4] = {0, 0, 0, 0}, ui[4] = {0, 0, 0, 0};
            const int e0 = 4 * tid + nfir;
            // window registers hold z[e0 - i + q] for q = 0..3
            double wc[4], wsn[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const double2 v2 = sq[MF_POS(e0 + q)]; wc[q] = v2.x; wsn[q] = v2.y; }
#pragma unroll 4                                               // the register window then rotates by renaming instead of moves per tap
            for (int i = 0; i <= nfir; ++i) {
                const double w = bk[i];
#pragma unroll
                for (int q = 0; q < 4; ++q) { ur[q] += w * wc[q]; ui[q] -= w * wsn[q]; }
                // slide: next tap reads one element lower
#pragma unroll
                for (int q = 3; q > 0; --q) { wc[q] = wc[q - 1]; wsn[q] = wsn[q - 1]; }
                const int en = e0 - i - 1;
                if (en >= 0) { const double2 v2 = sq[MF_POS(en)]; wc[0] = v2.x; wsn[0] = v2.y; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = tb + q;
                if (t < na) {
                    double c = 1.0, s = 0.0;
                    if (k > 0) {
                        const double ang = M_PI * (double)(t + 1) * cf / 1280.0;
                        c = SQ2 * cos(ang);
                        s = SQ2 * sin(ang);
                    }
                    const double f = ur[q] * c - ui[q] * s;
                    if (SIG == 0) {
                        xf[t] = f;
                    } else {
                        const double xv = xf[t];
                        sx += xv; sy += f; sxx += xv * xv; syy += f * f; sxy += xv * f;
                    }
                }
            }
        }
    }
    if (SIG == 0) return;
    sx = block_sum(sx, red); sy = block_sum(sy, red); sxx = block_sum(sxx, red); syy = block_sum(syy, red); sxy = block_sum(sxy, red);
    if (tid == 0) {
        const double n = (double)na;
        const double xsum = sxx - sx * sx / n, ysum = syy - sy * sy / n, xy = sxy - sx * sy / n;
        double cm = 0.0;
        if (!(xsum < 1.0e-30 || ysum < 1.0e-30)) cm = fabs(xy) / sqrt(xsum * ysum);
        ws.cm[((size_t)b * HP_NBASIS + basis) * HP_NMOD + k] = cm;
    }
}
#endif  // NELE_AB

// ---- h11, sliding form (the default).  The modulation filters are Hann windows, b[i] = (0.5 - 0.5 cos(2 pi i / L)) / (L / 2),
// i = 0..L (L = nfir; both end taps are zero), so the FIR is a combination of three sliding sums over exactly one period L:
//     S_m[tau] = sum_{i=0}^{L-1} e^{j m phi i} z[tau - i]      (m = 0, +1, -1; phi = 2 pi / L)
//     S_m[tau] = e^{j m phi} S_m[tau - 1] + z[tau] - z[tau - L]            (e^{j m phi L} = 1)
//     u[t]     = (2 / L) (0.5 S_0 - 0.25 S_+1 - 0.25 S_-1)[t + nh]
// i.e. a few dozen float64 operations per output instead of 2 (nfir + 1) multiply-adds (nfir up to 614): the direct form above was 14 ms
// of every B = 256 step.  (The kernel runs the sums in the demodulator's rotating frame, see the constants below.)  The rotations have modulus one, so rounding accumulates linearly: < 1e-12 relative over a chunk (A/B-tested
// against the direct kernel, NELE_HASPI_MOD_DIRECT=1).  One wave per (utterance, chunk of MS_TC outputs): lane = basis * 10 + band
// (50 of 64 lanes), every chunk warms its sums up over the L samples before it (recurrence without the subtraction).  A lane owns
// one (basis, band) pair, so the correlation sums of ebm_ModCorr stay in its registers; chunk partials are combined in chunk order.
// SIG = 0 stores the filtered reference sequence xf [b][t][64]; SIG = 1 filters the processed signal and correlates it with xf.
#ifndef MS_TC
#define MS_TC 1024
#endif
#ifndef MS_PAD
#define MS_PAD 0             // extra LDS (doubles) per workgroup: caps the workgroups per CU (A/B builds)
#endif
#define MS_R 640
template <int SIG>
__global__ __launch_bounds__(64) void haspi_mod_slide_kernel(HaspiWs ws) {
    __shared__ double ring[HP_NBASIS - 1][MS_R + 8 + MS_PAD / (HP_NBASIS - 1)];
    const int b = blockIdx.y, chunk = blockIdx.x, lane = threadIdx.x;
    const int na = ws.info[2 * b];
    if (ws.info[2 * b + 1]) return;
    const int t0 = chunk * MS_TC, t1 = min(t0 + MS_TC, na);
    if (t0 >= na) return;
    const bool act = lane < (HP_NBASIS - 1) * HP_NMOD;
    const int basis = act ? 1 + lane / HP_NMOD : 1, k = act ? lane % HP_NMOD : 0;
    const int L = c_modnfir[k], nh = L / 2;
    const double theta = (k > 0) ? M_PI * c_modcf[k] / 1280.0 : 0.0;     // pi cf / fNyq
    const double phi = 2.0 * M_PI / (double)L;
    // In the frame that rotates with the demodulator, T_m = conj(E) S_m (E = e^{-j theta (tau + 1)} the demodulator phase, |E| = 1), the
    // three sliding sums of the header become recurrences with CONSTANT complex factors and a real-valued drive,
    //     T_m[tau] = e^{j (theta + m phi)} T_m[tau - 1] + v[tau] - e^{j theta L} v[tau - L],
    //     f[t]     = scale Re( e^{-j theta nh} (0.5 T_0 - 0.25 T_+1 - 0.25 T_-1)[t + nh] ):
    // no demodulator / remodulator phases to carry along (20 float64 operations per step instead of 45).
    const double rc0 = cos(theta), rs0 = sin(theta);
    const double rcp = cos(theta + phi), rsp = sin(theta + phi), rcm = cos(theta - phi), rsm = sin(theta - phi);
    const double k1c = cos(theta * (double)L), k1s = sin(theta * (double)L);     // e^{+j theta L}: phase of the sample leaving the window
    const double k2c = cos(theta * (double)nh), k2s = -sin(theta * (double)nh);  // e^{-j theta nh}
    const double scale = ((k > 0) ? 2.0 : 1.0) * (2.0 / (double)L);      // sqrt(2) of the demodulator and of the remodulator
    const double A0 = scale * 0.5 * k2c, B0 = -scale * 0.5 * k2s, A1 = -scale * 0.25 * k2c, B1 = scale * 0.25 * k2s;
    const double* v = ws.cep + (((size_t)b * 2 + SIG) * HP_NBASIS + basis) * ws.nsub;
    const double vmu = ws.cmean[((size_t)b * 2 + SIG) * HP_NBASIS + basis];          // the sequence's mean (ebm_CepCoef removes it)
    double* xf = ws.xf + ((size_t)b * ws.nsub) * 64 + lane;
    // common output index tt = t0 - LMAX + step; this lane's newest input is tau = tt + nh
    constexpr int LMAX = HP_MAXFIR;
    double s0r = 0, s0i = 0, spr = 0, spi = 0, smr = 0, smi = 0;
    double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    // The five cepstral sequences of the utterance pass through a ring in LDS.  At step tt a lane reads its sequence at tt + nh (entering
    // the window) and tt + nh - L (leaving it): 80 different streams per wave, and straight from memory every load instruction touched
    // ~50 cache lines (PMC: 434 M L1 look-ups, 36 % of them misses, 4.7 GB fetched per launch for 0.26 GB of sequences - the kernel ran
    // at the speed of the L1 / L2 path, 1.8 ms for 0.5 ms of instructions).  The ring holds the common window [tt - 307, tt + 307 + 8]:
    // MS_R = 640 elements per sequence (+ 8 mirrored at the end, so that a group's 8 consecutive reads need no wrap), filled by 40 lanes
    // with ONE coalesced load per group of 8 steps, fetched a group ahead; the mean is removed on the way in.
    constexpr int MS_U = 8, NHM = LMAX / 2;
    const int base0 = t0 - LMAX - NHM;                                   // sequence index at ring position 0
    const int sl = lane >> 3, jl = lane & 7;                             // filler lanes: sequence (basis - 1), element of the group
    const double* vfill = ws.cep + (((size_t)b * 2 + SIG) * HP_NBASIS + 1 + min(sl, HP_NBASIS - 2)) * ws.nsub;
    const double mufill = ws.cmean[((size_t)b * 2 + SIG) * HP_NBASIS + 1 + min(sl, HP_NBASIS - 2)];
    {                                                                    // initial fill: offsets 0 .. 2 NHM + MS_U - 1 (the window of the first group)
        // all of a sequence's loads are issued before the first value is used: the one-element-per-iteration loop paid a memory latency
        // per iteration, 49 in a row - a sixth of a wave's whole run time
        constexpr int W0 = 2 * NHM + MS_U, NR = (W0 + 63) / 64;
#pragma unroll
        for (int q = 0; q < HP_NBASIS - 1; ++q) {
            const double* vq = ws.cep + (((size_t)b * 2 + SIG) * HP_NBASIS + 1 + q) * ws.nsub;
            const double muq = ws.cmean[((size_t)b * 2 + SIG) * HP_NBASIS + 1 + q];
            double tmp[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) tmp[r] = vq[min(max(base0 + min(lane + 64 * r, W0 - 1), 0), na - 1)];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int off = lane + 64 * r;
                if (off < W0) {
                    const double val = tmp[r] - muq;
                    ring[q][off] = val;
                    if (off < MS_U) ring[q][MS_R + off] = val;
                }
            }
        }
    }
    int fo = 2 * NHM + MS_U;                                             // ring offset (not yet wrapped) of the next group to fill
    // the elements of the next three groups are in flight / in registers (a lone wave per SIMD has nothing else to cover a memory latency)
    // (the RAW element is fetched; its mean comes off when it enters the ring three groups later - subtracting at fetch time made every
    //  group wait for the load it had just issued, which was half of this kernel's time)
    // Four fetch slots, one per group of the four-fold unrolled loop below: a slot is refilled right after it is consumed and is never copied
    // (rotating three values through register moves made each move wait for a load issued one group earlier).
    // (every lane loads - lanes 40..63 a duplicate of sequence 4 - so that the loop holds no branch around a memory instruction)
    auto fetch = [&](int ahead) { return vfill[min(max(base0 + fo + MS_U * ahead + jl, 0), na - 1)]; };
    double fs0 = fetch(0), fs1 = fetch(1), fs2 = fetch(2), fs3 = fetch(3);
    // SIG = 1: the stored reference outputs of a group are loaded one group ahead into one of two register sets that swap roles (the loop
    // below is unrolled by two groups: a set is never copied, so nothing waits for a load before its values are used); the warm-up groups
    // (tb + MS_U <= t0) use none and load none.
    double xa[MS_U], xb[MS_U];
    auto xload = [&](double (&xs)[MS_U], int tbn, bool always) {
        if (SIG == 1 && (always || (tbn + MS_U > t0 && tbn < t1))) {
#pragma unroll
            for (int u = 0; u < MS_U; ++u) xs[u] = xf[(size_t)min(max(tbn + u, 0), na - 1) * 64];
        }
    };
    xload(xa, t0 - LMAX, false);
    // ring positions of this lane's two reads at the first step: (tau - base0) mod MS_R, (tau - L - base0) mod MS_R
    int pn = (NHM + nh) % MS_R, po = (NHM + nh - L + MS_R) % MS_R;
    const double* rq = &ring[basis - 1][0];
    __syncthreads();
    // STEADY: -1 = the group decides its kind itself (chunk edges); 0 / 1 = a WARM / MAIN group inside a run of four such groups - no
    // branch anywhere in the group, so the compiler can count the memory operations in flight and wait for exactly the four-groups-old
    // fetch slot (s_waitcnt vmcnt(n)) instead of draining the queue - stores included - at every branch join.
    auto group = [&](auto steady_tag, int tb, double (&xvv)[MS_U], double (&xnext)[MS_U], double& fslot) {
        constexpr int STEADY = decltype(steady_tag)::value;
        double vnv[MS_U], vov[MS_U];
#pragma unroll
        for (int u = 0; u < MS_U; ++u) {
            vnv[u] = rq[pn + u];
            vov[u] = rq[po + u];
        }
        if (STEADY == 1) xload(xnext, tb + MS_U, true);
        else if (STEADY == -1) xload(xnext, tb + MS_U, false);
        pn = pn + MS_U >= MS_R ? pn + MS_U - MS_R : pn + MS_U;
        po = po + MS_U >= MS_R ? po + MS_U - MS_R : po + MS_U;
        // Three copies of the group's body: WARM (all 8 steps before t0: nothing leaves the sums, no output), MAIN (all 8 steps inside
        // the chunk and both window ends inside the sequence for every lane: no tests at all) and the general one - the per-step
        // validity tests were half of the loop's instructions.
        auto body = [&](auto kind_tag) {
            constexpr int KIND = decltype(kind_tag)::value;                  // 0 WARM, 1 MAIN, 2 general
#pragma unroll
            for (int u = 0; u < MS_U; ++u) {
                const int tt = tb + u, tau = tt + nh, to = tau - L;
                // before this lane's warm-up window (tt < t0 - L) nothing enters the sums; during it (tt < t0) nothing leaves them
                const double vn = (KIND == 1 || (tt >= t0 - L && tau >= 0 && tau < na)) ? vnv[u] : 0.0;
                const double vo = (KIND == 1) ? vov[u] : (KIND == 0) ? 0.0 : (tt >= t0 && to >= 0 && to < na) ? vov[u] : 0.0;
                const double wr = fma(-vo, k1c, vn), wi = -vo * k1s;                     // v[tau] - e^{j theta L} v[tau - L]
                { const double nr = fma(rc0, s0r, fma(-rs0, s0i, wr)); s0i = fma(rc0, s0i, fma(rs0, s0r, wi)); s0r = nr; }
                { const double nr = fma(rcp, spr, fma(-rsp, spi, wr)); spi = fma(rcp, spi, fma(rsp, spr, wi)); spr = nr; }
                { const double nr = fma(rcm, smr, fma(-rsm, smi, wr)); smi = fma(rcm, smi, fma(rsm, smr, wi)); smr = nr; }
                if (KIND == 0) continue;
                const bool live = KIND == 1 || (tt >= t0 && tt < t1);
                const double f = fma(A0, s0r, fma(B0, s0i, fma(A1, spr + smr, B1 * (spi + smi))));
                if (SIG == 0) {
                    if (live) xf[(size_t)tt * 64] = f;
                } else if (live) {
                    const double xv = xvv[u];
                    sx += xv; sy += f; sxx += xv * xv; syy += f * f; sxy += xv * f;
                }
            }
        };
        if (STEADY == 0) body(std::integral_constant<int, 0>{});
        else if (STEADY == 1) body(std::integral_constant<int, 1>{});
        else if (tb + MS_U <= t0) body(std::integral_constant<int, 0>{});
        else if (tb >= t0 && tb + MS_U <= t1 && tb - NHM >= 0 && tb + MS_U + NHM <= na) body(std::integral_constant<int, 1>{});
        else body(std::integral_constant<int, 2>{});
        // the next group's 8 new elements per sequence replace the 8 oldest (no lane reads them any more).  The workgroup is ONE wave: its
        // LDS operations execute in issue order, so ordering them is all a barrier has to do here - __syncthreads() also drains the
        // vector-memory queue (the fence of a workgroup-scope barrier), i.e. waited for every store of the group.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (sl < HP_NBASIS - 1) {
            const int pos = (fo + jl) % MS_R;
            const double fin = fslot - mufill;
            ring[sl][pos] = fin;
            if (pos < MS_U) ring[sl][MS_R + pos] = fin;
        }
        fslot = fetch(4);
        fo += MS_U;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    const std::integral_constant<int, -1> edge{};
    const std::integral_constant<int, 0> warm{};
    const std::integral_constant<int, 1> mainr{};
    for (int tb = t0 - LMAX; tb < t1;) {
        if (tb + 5 * MS_U <= t0) {                                       // four warm-up groups, and the group after them is one too (no reference loads)
            do {
                group(warm, tb, xa, xb, fs0); group(warm, tb + MS_U, xb, xa, fs1); group(warm, tb + 2 * MS_U, xa, xb, fs2); group(warm, tb + 3 * MS_U, xb, xa, fs3);
                tb += 4 * MS_U;
            } while (tb + 5 * MS_U <= t0);
        } else if (tb >= t0 && tb + 4 * MS_U <= t1 && tb - NHM >= 0 && tb + 4 * MS_U + NHM <= na) {
            do {
                group(mainr, tb, xa, xb, fs0); group(mainr, tb + MS_U, xb, xa, fs1); group(mainr, tb + 2 * MS_U, xa, xb, fs2); group(mainr, tb + 3 * MS_U, xb, xa, fs3);
                tb += 4 * MS_U;
            } while (tb + 4 * MS_U <= t1 && tb + 4 * MS_U + NHM <= na);
        } else {
            group(edge, tb, xa, xb, fs0);
            if (tb + MS_U < t1) group(edge, tb + MS_U, xb, xa, fs1);
            if (tb + 2 * MS_U < t1) group(edge, tb + 2 * MS_U, xa, xb, fs2);
            if (tb + 3 * MS_U < t1) group(edge, tb + 3 * MS_U, xb, xa, fs3);
            tb += 4 * MS_U;
        }
    }
    if (SIG == 1 && act) {
        double* cp = ws.cpart + (((size_t)b * MS_MAXC + chunk) * 64 + lane) * 5;
        cp[0] = sx; cp[1] = sy; cp[2] = sxx; cp[3] = syy; cp[4] = sxy;
    }
}
// chunk partials -> |rho| per (basis, band).  grid B, block 64
__global__ void haspi_modcorr_kernel(HaspiWs ws) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (ws.info[2 * b + 1] || lane >= (HP_NBASIS - 1) * HP_NMOD) return;
    const int na = ws.info[2 * b];
    double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    for (int c = 0; c * MS_TC < na; ++c) {
        const double* cp = ws.cpart + (((size_t)b * MS_MAXC + c) * 64 + lane) * 5;
        sx += cp[0]; sy += cp[1]; sxx += cp[2]; syy += cp[3]; sxy += cp[4];
    }
    const double n = (double)na;
    const double xsum = sxx - sx * sx / n, ysum = syy - sy * sy / n, xy = sxy - sx * sy / n;
    double cm = 0.0;
    if (!(xsum < 1.0e-30 || ysum < 1.0e-30)) cm = fabs(xy) / sqrt(xsum * ysum);
    ws.cm[((size_t)b * HP_NBASIS + 1 + lane / HP_NMOD) * HP_NMOD + lane % HP_NMOD] = cm;
}

__global__ void haspi_final_kernel(HaspiWs ws, float* __restrict__ raw, float* __restrict__ mapped, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double w[HP_NMOD] = {1.361, 1.521, 1.164, 0.492, 0.436, 0.690, 1.142, 0.816, 1.576, 2.269};
    double v = 0.0;
    if (ws.info[2 * b + 1]) {
        v = nan("");   // reference raises 'Signal below threshold'
    } else {
        for (int k = 0; k < HP_NMOD; ++k) {
            double a = 0.0;
            for (int j = 1; j < HP_NBASIS; ++j) a += ws.cm[((size_t)b * HP_NBASIS + j) * HP_NMOD + k];
            v += w[k] * (a / 5.0);
        }
    }
    if (raw) raw[b] = (float)v;
    if (mapped) mapped[b] = (float)(1.0 / (1.0 + exp(-0.95 * (v - 2.8))));
}

// ------------------------------------------------------------------------------------------ C ABI
static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t haspi_layout(int B, int L, int fs_in, HaspiWs* w, char* base) {
    const int n24 = hp_n24_of(L, fs_in);
    const int nsub = (n24 + HP_SPACE - 1) / HP_SPACE;
    const int n24p = (n24 + 31) / 32 * 32;
    size_t o = 0;
#define TAKE(field, type, count) do { if (w) w->field = (type*)(base + o); o += al(sizeof(type) * (size_t)(count)); } while (0)
    TAKE(win, double, HP_NWIN_AL + RS3_TAB);           // resampler half window + the 16 kHz tap table
    TAKE(r24, float, (size_t)B * 2 * n24p);
    TAKE(mid, double, (size_t)B * 2 * n24p);
    TAKE(ctl, hp_env_t, (size_t)B * 2 * n24p * HP_NCH);
    TAKE(env, hp_env_t, (size_t)B * 2 * n24p * HP_NCH);
    TAKE(bw, double, (size_t)B * 2 * HP_NCH);
    TAKE(loss, double, 2 * 5 * HP_NCH);
    int lc = GS_LC;
    { const int lcenv = NELE_SWITCH_INT("NELE_HASPI_LC", 0); if (lcenv >= GS_LC) lc = lcenv / GS_LC * GS_LC; }
    while ((n24p + lc - 1) / lc > GS_MAXC) lc += GS_LC;
    const int nchunk = (n24p + lc - 1) / lc, ncg = (n24p + (lc < GL_N ? lc : GL_N) - 1) / (lc < GL_N ? lc : GL_N);
    TAKE(ssp, double, (size_t)B * 2 * nchunk * HP_NCH);
    TAKE(est, double, (size_t)B * 2 * nchunk * 256);
    TAKE(pmat, double, (size_t)(1 + B * 2) * HP_NCH * 16);
    TAKE(ihe, double, (size_t)B * 2 * ncg * 64);
    TAKE(pihc, double, 4);
    TAKE(rsp, double, (size_t)B * 2 * RS_MAXC);
    TAKE(rinfo, float, (size_t)B * 2 * 4);
    TAKE(benv, double, 64);
    TAKE(bkt, double, 10 * 616);
    TAKE(shift, int, (size_t)B * HP_NCH);
    TAKE(lp, double, (size_t)B * 2 * (nsub + 4) * HP_NCH);
    const int ngb = (nsub + CP_F - 1) / CP_F;
    TAKE(act, int, (size_t)B * nsub);
    TAKE(grank, int, (size_t)B * nsub);
    TAKE(gcnt, int, (size_t)B * ngb);
    TAKE(cpsum, double, (size_t)B * 2 * ngb * HP_NBASIS);
    TAKE(cmean, double, (size_t)B * 2 * HP_NBASIS);
    TAKE(info, int, (size_t)B * 2);
    TAKE(cep, double, (size_t)B * 2 * HP_NBASIS * nsub);
    TAKE(cm, double, (size_t)B * HP_NBASIS * HP_NMOD);
    TAKE(xf, double, (size_t)B * 64 * nsub);
    TAKE(cpart, double, (size_t)B * MS_MAXC * 64 * 5);
#undef TAKE
    if (w) { w->n24 = n24; w->nsub = nsub; w->n24p = n24p; w->fs_in = fs_in; w->lens = nullptr; w->nchunk = nchunk; w->lc = lc; w->ngb = ngb;
             w->cphi = nullptr; w->sse = nullptr; w->lcg = GL_N; w->fmul = 1; w->nlp = nsub + 4; w->lp_raw = 0; }
    return o;
}

extern "C" long long nele_metric_haspi_workspace_bytes(int B, int L, int fs_in) { return (long long)haspi_layout(B, L, fs_in, nullptr, nullptr); }

// What phase 3 (the reference-signal half: ear model, envelope filter, silence gate, group-delay shifts, cepstra, modulation filters of
// the CLEAN signal) leaves in a workspace for phase 4, as byte ranges {offset, stride, bytes} exactly like
// nele_metric_siib_clean_sections (stride 0 = a table shared by all utterances: resampler window, control-bank transition matrices, IHC
// transition, envelope and modulation-filter taps).  A pure function of the clean waveform, the padded length L, fs_in, the audiogram and
// the reference's dither rows: a loop that scores the same clean files every epoch may keep it and skip phase 3 (same L, fs_in, HL,
// dither; any row order, any B).  Returns the number of sections (<= max_sections) or a negative status.
extern "C" int nele_metric_haspi_clean_sections(int B, int L, int fs_in, long long* out, int max_sections) {
    NELE_CHECK_ARG(B > 0 && out && max_sections >= 16, "nele_metric_haspi_clean_sections: bad arguments (16 sections)");
    NELE_CHECK_ARG(fs_in >= 1000 && fs_in <= 24000 && L >= 2400, "nele_metric_haspi_clean_sections: bad signal geometry");
    HaspiWs w;
    char* base = reinterpret_cast<char*>((uintptr_t)1 << 20);
    haspi_layout(B, L, fs_in, &w, base);
    int k = 0;
    auto put = [&](const void* p, long long stride, long long bytes) {
        out[3 * k] = (long long)(reinterpret_cast<const char*>(p) - base); out[3 * k + 1] = stride; out[3 * k + 2] = bytes; ++k;
    };
    const size_t ns = (size_t)w.nsub;
    put(w.win, 0, sizeof(double) * (HP_NWIN_AL + RS3_TAB));
    put(w.pmat, 0, sizeof(double) * HP_NCH * 16);                                   // entry 0: the control bank (per channel only)
    put(w.pihc, 0, sizeof(double) * 4);
    put(w.benv, 0, sizeof(double) * 64);
    put(w.bkt, 0, sizeof(double) * 10 * 616);
    put(w.bw, sizeof(double) * 2 * HP_NCH, sizeof(double) * 2 * HP_NCH);
    put(w.rinfo, sizeof(float) * 2 * 4, sizeof(float) * 2 * 4);
    put(w.shift, sizeof(int) * HP_NCH, sizeof(int) * HP_NCH);
    put(w.act, sizeof(int) * ns, sizeof(int) * ns);
    put(w.grank, sizeof(int) * ns, sizeof(int) * ns);
    put(w.gcnt, sizeof(int) * (size_t)w.ngb, sizeof(int) * (size_t)w.ngb);
    put(w.cpsum, sizeof(double) * 2 * (size_t)w.ngb * HP_NBASIS, sizeof(double) * (size_t)w.ngb * HP_NBASIS);   // clean half
    put(w.cmean, sizeof(double) * 2 * HP_NBASIS, sizeof(double) * 2 * HP_NBASIS);
    put(w.info, sizeof(int) * 2, sizeof(int) * 2);
    put(w.cep, sizeof(double) * 2 * HP_NBASIS * ns, sizeof(double) * HP_NBASIS * ns);                             // clean half
    put(w.xf, sizeof(double) * 64 * ns, sizeof(double) * 64 * ns);
    return k;
}

extern "C" int nele_metric_haspi_nsub(int L, int fs_in) {
    const int n24 = hp_n24_of(L, fs_in);
    return (n24 + HP_SPACE - 1) / HP_SPACE;
}

// A/B switches of the chain, read once
struct HaspiFlags { int bank_gain, par_iir, fused_gain, fir9; };
static const HaspiFlags& haspi_flags() {
    static HaspiFlags f = [] {
        HaspiFlags v;
        v.bank_gain = NELE_SWITCH_INT("NELE_HASPI_BANK_GAIN", 1) != 0;   // =0: the gain pass as its own kernel
        v.par_iir = NELE_SWITCH_INT("NELE_HASPI_PAR_IIR", 1) != 0;       // =0: the serial recurrence kernels
        v.fused_gain = NELE_SWITCH_INT("NELE_HASPI_FUSED_GAIN", 1) != 0;
        v.fir9 = NELE_SWITCH_INT("NELE_HASPI_FIR9", 1) != 0;             // =0: the 8-sample-group IHC + envelope-filter kernel with per-sample phase counters
        return v;
    }();
    return f;
}
// lp in group-space rows (HaspiWs::lp_raw): whenever haspi_ihc_fir9_kernel produces it
static int haspi_lp_raw() { const HaspiFlags& f = haspi_flags(); return f.par_iir && f.fused_gain && f.fir9; }

// resampler window + the 16 kHz tap table behind it (kept in the workspace: the split calls' later phases reuse them)
static void haspi_build_window(const HaspiWs& ws, hipStream_t s) {
    hipLaunchKernelGGL(haspi_win_kernel, dim3((HP_NWIN + 255) / 256), dim3(256), 0, s, ws.win);
    hipLaunchKernelGGL(haspi_rs_taps_kernel, dim3(1), dim3(256), 0, s, ws.win);
}

// The ear model + envelope chain of signals sig0 .. sig0+nsig-1 (h1 .. h9 of the header comment).
static void haspi_chain(const float* x, const float* y, int B, int L, int fs_in, const HaspiWs& ws_in, int sig0, int nsig, hipStream_t s,
                        bool quality = false) {
    const int rows = B * nsig;
    HaspiWs ws = ws_in;
    const HaspiFlags& fl = haspi_flags();
    const int bank_gain = fl.bank_gain;
    int par_iir = fl.par_iir, fused_gain = fl.fused_gain;
    if (quality) par_iir = fused_gain = 1;                     // the quality path exists for the scan kernels only
    const bool in_bank = bank_gain && par_iir && fused_gain && !quality;     // gain pass inside pass 2 of the signal bank
    ws.lcg = in_bank ? ws.lc : GL_N;
    ws.fmul = 1;
    hipLaunchKernelGGL(haspi_rms_kernel, dim3(B, nsig), dim3(256), 0, s, x, y, L, fs_in, ws, sig0);
    if (fs_in != 24000) {
        const int rs3 = NELE_SWITCH_INT("NELE_HASPI_RS3", 1);                               // NELE_HASPI_RS3=0: the output-per-thread kernel at 16 kHz too (A/B diagnostic)
        if (fs_in == 16000 && rs3)
            hipLaunchKernelGGL(haspi_resample3_kernel, dim3((ws.n24 + RS_CH - 1) / RS_CH, nsig, B), dim3(256), 0, s, x, y, L, ws, sig0);
        else
            hipLaunchKernelGGL(haspi_resample_kernel, dim3((ws.n24 + RS_CH - 1) / RS_CH, nsig, B), dim3(256), 0, s, x, y, L, ws, sig0);
        hipLaunchKernelGGL(haspi_resample_gain_kernel, dim3(rows), dim3(64), 0, s, ws, sig0, nsig);
    }
    if (par_iir) hipLaunchKernelGGL(haspi_midear_par_kernel, dim3(((ws.n24p + ME_N - 1) / ME_N + 63) / 64, rows), dim3(64), 0, s, ws, sig0, nsig);
    else { NELE_AB_ONLY(hipLaunchKernelGGL(haspi_midear_kernel, dim3(B), dim3(64), 0, s, ws, sig0, nsig);) }
    if (par_iir) {
        if (sig0 == 0) hipLaunchKernelGGL(haspi_pmat_kernel, dim3(1), dim3(128), 0, s, ws, ws.lc, 0, 0, 1);   // control bank: per channel only
        const int tail1 = NELE_SWITCH_INT("NELE_HASPI_TAIL1", 1);                             // NELE_HASPI_TAIL1=0: pass 1 over whole chunks (A/B diagnostic)
        if (tail1) hipLaunchKernelGGL(haspi_bank_tail_kernel<false>, dim3(8 * 4 * ((ws.nchunk + 63) / 64) * ((rows + 7) / 8)), dim3(512), 0, s, ws, sig0, nsig, tail1 >= 2 ? tail1 - 1 : 0, rows);
        else { NELE_AB_ONLY(hipLaunchKernelGGL((haspi_bank_scan_kernel<false, false>), dim3((ws.nchunk + 1) / 2, nsig, B), dim3(64), 0, s, ws, sig0);) }
        hipLaunchKernelGGL(haspi_bank_prefix_kernel<false>, dim3(rows), dim3(64), 0, s, ws, sig0, nsig);
        hipLaunchKernelGGL((haspi_bank_scan_kernel<false, true>), dim3((ws.nchunk + 1) / 2, nsig, B), dim3(64), 0, s, ws, sig0);
        hipLaunchKernelGGL(haspi_bw_kernel, dim3(rows), dim3(32), 0, s, ws, sig0, nsig);
        hipLaunchKernelGGL(haspi_pmat_kernel, dim3(rows), dim3(128), 0, s, ws, ws.lc, 1, sig0, nsig);
        if (tail1) hipLaunchKernelGGL(haspi_bank_tail_kernel<true>, dim3(8 * 4 * ((ws.nchunk + 63) / 64) * ((rows + 7) / 8)), dim3(512), 0, s, ws, sig0, nsig, tail1 >= 2 ? tail1 - 1 : 0, rows);
        else { NELE_AB_ONLY(hipLaunchKernelGGL((haspi_bank_scan_kernel<true, false>), dim3((ws.nchunk + 1) / 2, nsig, B), dim3(64), 0, s, ws, sig0);) }
        hipLaunchKernelGGL(haspi_bank_prefix_kernel<true>, dim3(rows), dim3(64), 0, s, ws, sig0, nsig);
        if (quality) hipLaunchKernelGGL((haspi_bank_scan_kernel<true, true, true>), dim3((ws.nchunk + 1) / 2, nsig, B), dim3(64), 0, s, ws, sig0);
        else if (in_bank)
            NELE_PROF("haspi_bank_gain_kernel", s,
                      hipLaunchKernelGGL((haspi_bank_scan_kernel<true, true, false, true>), dim3((ws.nchunk + 1) / 2, nsig, B), dim3(64), 0, s, ws, sig0));
        else { NELE_AB_ONLY(hipLaunchKernelGGL((haspi_bank_scan_kernel<true, true>), dim3((ws.nchunk + 1) / 2, nsig, B), dim3(64), 0, s, ws, sig0);) }
    } else {
        NELE_AB_ONLY(hipLaunchKernelGGL(haspi_control_kernel, dim3(nsig, B), dim3(64), 0, s, ws, sig0);
                     hipLaunchKernelGGL(haspi_signal_kernel, dim3(nsig, B), dim3(64), 0, s, ws, sig0);)
    }
    if (sig0 == 0) hipLaunchKernelGGL(haspi_shift_kernel, dim3(B), dim3(64), 0, s, ws);       // group-delay shifts come from BWx alone (+ constant tables)
    if (fused_gain && par_iir) {
        if (!in_bank)
            NELE_PROF("haspi_gain_lp_sl_kernel", s,
                      hipLaunchKernelGGL(haspi_gain_lp_sl_kernel, dim3((ws.n24p + 8 * GL_N - 1) / (8 * GL_N), rows), dim3(256), 0, s, ws, sig0, nsig));
        hipLaunchKernelGGL(haspi_ihc_prefix_kernel, dim3(rows), dim3(32), 0, s, ws, sig0, nsig);
        if (quality) return;                                   // haspi_quality.h goes on from the dB-SL envelope + IHC start states
        const dim3 fgrid((ws.n24p + 4 * ws.lcg * ws.fmul - 1) / (4 * ws.lcg * ws.fmul), rows);
        if (ws.lp_raw) hipLaunchKernelGGL(haspi_ihc_fir9_kernel, fgrid, dim3(128), 0, s, ws, sig0, nsig);
        else { NELE_AB_ONLY(hipLaunchKernelGGL(haspi_ihc_fir_kernel, fgrid, dim3(128), 0, s, ws, sig0, nsig);) }
        return;                                                // the envelope filter is part of it
    }
#ifdef NELE_AB
    else {                                                     // the serial passes of the first version (A/B switch)
        if (fused_gain) {
            hipLaunchKernelGGL(haspi_gain_lp_sl_kernel, dim3((ws.n24p + 8 * GL_N - 1) / (8 * GL_N), rows), dim3(256), 0, s, ws, sig0, nsig);
        } else {
            const size_t per_row = (size_t)ws.n24p * HP_NCH;
            const unsigned bx = (unsigned)((per_row + 255) / 256 < 256 ? (per_row + 255) / 256 : 256);
            hipLaunchKernelGGL(haspi_gain_kernel, dim3(bx, rows), dim3(256), 0, s, ws, per_row, sig0, nsig);
            hipLaunchKernelGGL(haspi_gainlp_kernel, dim3((rows + 1) / 2), dim3(64), 0, s, ws, sig0, nsig, rows);
            hipLaunchKernelGGL(haspi_sl_kernel, dim3(bx, rows), dim3(256), 0, s, ws, per_row, sig0, nsig);
        }
        hipLaunchKernelGGL(haspi_ihc_kernel, dim3((rows + 1) / 2), dim3(64), 0, s, ws, sig0, nsig, rows);
    }
    hipLaunchKernelGGL(haspi_envfilt_kernel, dim3((ws.nsub + EF_SUB - 1) / EF_SUB, B, nsig), dim3(256), 0, s, ws, sig0);
#endif  // NELE_AB
}

// dither: NULL (no dither: deterministic) or standard normals [B][2][nsub][32]; row k perturbs the k-th ACTIVE
// frame (the reference draws randn(n_active, 32) for x, then for y: pyhaspi2.py:362-365).
// phase 0: everything.  Split by data dependence (the whole reference-signal chain - ear model, envelope filter, silence gate, group
// delays, cepstra, modulation filters - needs the clean signal only): phase 3 = clean part (y may be NULL), phase 4 = degraded part on
// the same workspace (x may be NULL).  Phase 0 runs exactly these two parts back to back, so the split is bit-identical by construction.
// hl6 (HOST pointer, 6 doubles or NULL = normal hearing): audiogram of the listener at 250, 500, 1000, 2000, 4000, 6000 Hz in dB HL;
// itype 0: the reference signal is heard with normal hearing, the processed one with the loss (haspi_v2 / haspi, pyhaspi2.py:76-157);
// itype 2: both with the loss (hasqi_v2, pyhaspi2.py:32-74).  itype 1 (NAL-R equalisation) raises NotImplementedError in the reference
// itself (eb_NALR, pyhaspi2.py:830-831) and is refused here.
static int haspi_hl_table(const double* hl6, int itype, HpHL* out) {
    if (itype != 0 && itype != 2)
        return nele_set_error(NELE_ERR_UNSUPPORTED, "HASPI ear model: itype %d (NAL-R) is not implemented by the reference either (eb_NALR raises)", itype);
    for (int k = 0; k < 6; ++k) {
        const double v = hl6 ? hl6[k] : 0.0;
        out->y[k] = v;
        out->x[k] = itype == 0 ? 0.0 : v;
    }
    return NELE_OK;
}

static int haspi_var_impl(const float* x, const float* y, const int* lengths, int B, int L, int fs_in, const double* dither, const double* hl6, int itype,
                          void* workspace, long long workspace_bytes, float* raw, float* mapped, int* info_out, int phase, void* stream);

extern "C" int nele_metric_haspi_var(const float* x, const float* y, const int* lengths, int B, int L, int fs_in, const double* dither,
                                     void* workspace, long long workspace_bytes, float* raw, float* mapped, int* info_out, int phase,
                                     void* stream) {
    return haspi_var_impl(x, y, lengths, B, L, fs_in, dither, nullptr, 0, workspace, workspace_bytes, raw, mapped, info_out, phase, stream);
}

// haspi_v2(x, fx, y, fy, HL) for a hearing-impaired listener (pyhaspi2.py:76-107, 779-807, 1155-1166)
extern "C" int nele_metric_haspi_var_hl(const float* x, const float* y, const int* lengths, int B, int L, int fs_in, const double* dither,
                                        const double* hl6_host, int itype, void* workspace, long long workspace_bytes, float* raw, float* mapped,
                                        int* info_out, int phase, void* stream) {
    return haspi_var_impl(x, y, lengths, B, L, fs_in, dither, hl6_host, itype, workspace, workspace_bytes, raw, mapped, info_out, phase, stream);
}

static int haspi_var_impl(const float* x, const float* y, const int* lengths, int B, int L, int fs_in, const double* dither, const double* hl6, int itype,
                          void* workspace, long long workspace_bytes, float* raw, float* mapped, int* info_out, int phase, void* stream) {
    NELE_CHECK_ARG(workspace && B > 0 && (phase == 0 || phase == 3 || phase == 4), "nele_metric_haspi: bad arguments");
    HpHL hl;
    { const int st_ = haspi_hl_table(hl6, itype, &hl); if (st_) return st_; }
    NELE_CHECK_ARG((x || phase == 4) && (y || phase == 3) && (raw || mapped || phase == 3), "nele_metric_haspi: missing signal / output for phase %d", phase);
    // pyhaspi2.py:810-821: 24 kHz passes through, lower rates are resampled, higher ones raise NotImplementedError in the reference
    NELE_CHECK_ARG(fs_in >= 1000 && fs_in <= 24000, "nele_metric_haspi: fs must be in [1000, 24000] Hz (got %d; the reference has no downsampler)", fs_in);
    if (L < 2400) return nele_set_error(NELE_ERR_SIGNAL, "nele_metric_haspi: L=%d too short", L);
    if (workspace_bytes < nele_metric_haspi_workspace_bytes(B, L, fs_in))
        return nele_set_error(NELE_ERR_WORKSPACE, "nele_metric_haspi: workspace too small");
    HaspiWs ws;
    haspi_layout(B, L, fs_in, &ws, (char*)workspace);
    ws.lens = lengths;
    ws.lp_raw = haspi_lp_raw();
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(haspi_loss_kernel, dim3(1), dim3(64), 0, s, ws, hl);
    // Cepstrum stage: one block per utterance (default) or the frame-parallel kernels (NELE_HASPI_CEP_SERIAL=0).  Alone on the GPU the
    // parallel version takes 0.15 ms against 0.75 ms per call; inside the B = 256 step (A/B in one run, three repetitions each,
    // tools/ab.sh) the step is 76.0 ms with the serial kernel and 78.2 ms with the parallel one: the GPU is saturated by the step's own
    // streams, and a side-stream kernel that bursts over every CU (87 M float64 pow() in the silence gate) takes more from the main
    // chain than its own chain gains.  Splitting HASPI into row groups (a narrower footprint throughout) did not help (78-79 ms).
    const int cep_serial = NELE_SWITCH_INT("NELE_HASPI_CEP_SERIAL", 1);
    const int mod_direct = NELE_SWITCH_INT("NELE_HASPI_MOD_DIRECT", 0);                            // NELE_HASPI_MOD_DIRECT=1: direct-form modulation FIR (A/B diagnostic)
    NELE_CHECK_ARG((ws.nsub + MS_TC - 1) / MS_TC <= MS_MAXC, "nele_metric_haspi: signal too long (%d sub-sampled frames)", ws.nsub);
    if (phase == 0 || phase == 3) {
        if (fs_in != 24000) haspi_build_window(ws, s);
        haspi_chain(x, y, B, L, fs_in, ws, 0, 1, s);
        if (cep_serial) {
            hipLaunchKernelGGL(haspi_cep_kernel, dim3(B), dim3(256), 0, s, ws, dither, 0.1, 1, 0, 1);
        } else {
            NELE_AB_ONLY(hipLaunchKernelGGL(haspi_gate_kernel, dim3(ws.ngb, B), dim3(CP_F), 0, s, ws);
                         hipLaunchKernelGGL(haspi_gate_scan_kernel, dim3(B), dim3(64), 0, s, ws);
                         hipLaunchKernelGGL(haspi_cepstra_kernel, dim3(ws.ngb, B, 1), dim3(CP_F), 0, s, ws, dither, 0.1, 0);
                         hipLaunchKernelGGL(haspi_cepmean_kernel, dim3(B, 1), dim3(64), 0, s, ws, 0);)
        }
        if (mod_direct) { NELE_AB_ONLY(hipLaunchKernelGGL(haspi_mod_direct_kernel<0>, dim3(HP_NMOD, HP_NBASIS - 1, B), dim3(256), 0, s, ws);) }
        else hipLaunchKernelGGL(haspi_mod_slide_kernel<0>, dim3((ws.nsub + MS_TC - 1) / MS_TC, B), dim3(64), 0, s, ws);
    }
    if (phase == 0 || phase == 4) {
        haspi_chain(x, y, B, L, fs_in, ws, 1, 1, s);
        if (cep_serial) {
            hipLaunchKernelGGL(haspi_cep_kernel, dim3(B), dim3(256), 0, s, ws, dither, 0.1, 0, 1, 1);
        } else {
            NELE_AB_ONLY(hipLaunchKernelGGL(haspi_cepstra_kernel, dim3(ws.ngb, B, 1), dim3(CP_F), 0, s, ws, dither, 0.1, 1);
                         hipLaunchKernelGGL(haspi_cepmean_kernel, dim3(B, 1), dim3(64), 0, s, ws, 1);)
        }
        if (mod_direct) { NELE_AB_ONLY(hipLaunchKernelGGL(haspi_mod_direct_kernel<1>, dim3(HP_NMOD, HP_NBASIS - 1, B), dim3(256), 0, s, ws);) }
        else {
            hipLaunchKernelGGL(haspi_mod_slide_kernel<1>, dim3((ws.nsub + MS_TC - 1) / MS_TC, B), dim3(64), 0, s, ws);
            hipLaunchKernelGGL(haspi_modcorr_kernel, dim3(B), dim3(64), 0, s, ws);
        }
        hipLaunchKernelGGL(haspi_final_kernel, dim3((B + 63) / 64), dim3(64), 0, s, ws, raw, mapped, B);
    }
    if (info_out) (void)hipMemcpyAsync(info_out, ws.info, sizeof(int) * 2 * (size_t)B, hipMemcpyDeviceToDevice, s);
    NELE_CHECK_LAUNCH("nele_metric_haspi");
    return NELE_OK;
}

extern "C" int nele_metric_haspi(const float* x, const float* y, int B, int L, int fs_in, const double* dither, void* workspace,
                                 long long workspace_bytes, float* raw, float* mapped, int* info_out, void* stream) {
    NELE_CHECK_ARG(x && y && (raw || mapped), "nele_metric_haspi: bad arguments");
    return nele_metric_haspi_var(x, y, nullptr, B, L, fs_in, dither, workspace, workspace_bytes, raw, mapped, info_out, 0, stream);
}

#include "haspi_quality.h"

// ---- per-utterance dither rows (pyhaspi2.py:362-365: `thrNerve * np.random.randn(n_active, 32)` for x, then for y, on every call).
// The reference draws from numpy's global generator, i.e. the draws depend on the order in which a process happens to score its files.
// Here a row is a pure function of (seed, utterance id, signal, active-frame index, channel) - the same whichever rank scores the
// utterance and whatever else shares its batch ("identical RNG seeds for HASPI dither per utterance id, not per rank").
// splitmix64 counter hash -> two 53-bit uniforms -> Box-Muller in float64.  out [B][2][nsub][32] (the `dither` argument of
// nele_metric_haspi*).  oracle/haspi.py:dither_rows is the numpy statement of the same function.
__global__ __launch_bounds__(256) void haspi_dither_rows_kernel(const long long* __restrict__ ids, unsigned long long seed, int nsub,
                                                                double* __restrict__ out) {
    const int b = blockIdx.z, sig = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;                  // k * 32 + ch
    if (e >= nsub * HP_NCH) return;
    const unsigned long long key = hq_mix(seed ^ hq_mix((unsigned long long)ids[b]));
    const unsigned long long idx = ((unsigned long long)sig << 40) | (unsigned long long)e;
    const unsigned long long r1 = hq_mix(key ^ hq_mix(2ull * idx)), r2 = hq_mix(key ^ hq_mix(2ull * idx + 1ull));
    const double u1 = ((double)(r1 >> 11) + 1.0) * 0x1.0p-53;      // (0, 1]
    const double u2 = (double)(r2 >> 11) * 0x1.0p-53;              // [0, 1)
    out[(((size_t)b * 2 + sig) * nsub) * HP_NCH + e] = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}

extern "C" int nele_haspi_dither_rows(const long long* utt_ids, unsigned long long seed, int B, int nsub, double* out, void* stream) {
    NELE_CHECK_ARG(utt_ids && out && B > 0 && nsub > 0, "nele_haspi_dither_rows: bad arguments");
    hipLaunchKernelGGL(haspi_dither_rows_kernel, dim3((nsub * HP_NCH + 255) / 256, 2, B), dim3(256), 0, as_stream(stream), utt_ids, seed, nsub, out);
    NELE_CHECK_LAUNCH("nele_haspi_dither_rows");
    return NELE_OK;
}
