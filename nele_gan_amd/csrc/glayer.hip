// Generator layers on bf16 activations (bf16 mode of model.Generator_Conv1D_cLN): one launch per layer for
//   causal Conv1d (model.py:10-40, 49-77)  ->  + bias  ->  cumulative layer norm (model.py:168-205)  ->  LeakyReLU(0.3) (model.py:88-91)
// and, with MODE 1, the plain convolution of the data gradient (same kernel over the END-padded output gradient with flipped weights).
//
// Round 5.  Until now a layer was conv1d_tile16_kernel (float32 activations converted while staged, 128 positions x 64 channels per
// workgroup pass, every workgroup streaming the layer's whole weight matrix for 128 positions: 32 B / clock / CU of L2 traffic at full
// MFMA rate, MFMA-busy 0.18) + cln_stats_kernel + cln_fwd_kernel (the raw convolution written, read twice, and the activation written
// again, all float32).  Built on the discriminator's recipe (csrc/conv16.hip):
//   * activations are bfloat16 IN MEMORY ([B][T + K - 1][C], time left-padded with K - 1 zero rows = Chomp1d): the convolution rounded
//     them to bf16 while staging anyway, so the MFMA operands are the same numbers;
//   * a workgroup (512 threads, one per CU) owns 256 CONSECUTIVE FRAMES x ALL output channels of one utterance: 16 x 16 MFMA tiles of
//     16 x 16 on 8 waves (wave tile 64 frames x 128 channels, or 32 x 64 for the 64-channel layer) - 128 float32 accumulators per lane;
//     the weight stream is fetched once per 256 frames (16 B / clock / CU);
//   * the reduction runs channel-slice-major: k-step (slice of 64 input channels, tap, half) - only the current 64-channel slice of the
//     256 + K - 1 input frames is resident (33 KB, double buffered), not the whole strip (134 KB), which leaves room for a two-slot weight
//     ring of 2 k-steps (2 x 32 KB); both arrive by global -> LDS DMA (global_load_lds_dwordx4), one barrier per 2 k-steps;
//   * operands swapped (weights = MFMA A, frames = MFMA B) with the output channels permuted inside the weight fragments: a lane ends
//     up with 32 CONSECUTIVE channels of one frame - bias, statistics, normalisation, LeakyReLU and the bf16 stores happen in registers;
//   * the cumulative statistics (sum and sum of squares over all channels and all frames so far, float64) are reduced in the epilogue:
//     per frame over the lane's channels, across the 4 lane groups (v_permlane swaps), across the wave columns (LDS), then a 256-frame
//     scan; utterances longer than 256 frames chain their strips through a 16-byte tagged slot per strip (strip s waits for the totals
//     of strip s - 1, which was dispatched before it and is running or done: no deadlock under in-order dispatch).
// The float32 raw convolution Y and the per-frame mean / 1/std are written only when the caller asks (training: the backward pass
// needs them); evaluation (inference.py:79-117) writes the next layer's bf16 input and nothing else: 2 x B x T x C bytes per layer.
#include "conv_common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

#define GL_TP 256          // frames per workgroup
#define GL_NPOS 264        // frames of a resident slice: GL_TP + K - 1 rounded up to whole 1 KB DMA pieces (8 frames each); K <= 9
#define GL_CS 64           // input channels per slice
#define GL_SB 2            // k-steps per weight chunk
#define GL_SLICE (GL_NPOS * GL_CS)    // elements per slice slot

typedef unsigned gl_u32x4 __attribute__((ext_vector_type(4)));

struct GLayerArgs {
    const __bf16* A;       // [B][T + K - 1][Cin]
    const __bf16* Wfrag;   // [nsteps + GL_SB][N / 16][64][8], see glayer_frag_kernel
    const float* bias;     // conv bias [N]                         (MODE 0)
    const float* gain;     // cLN gain0 [N]
    const float* beta;     // cLN bias0 [N]
    float* Y;              // [B][T][N] raw convolution + bias, or null
    float* mean;           // [B][T] or null
    float* rstd;           // [B][T] or null
    __bf16* out16;         // MODE 0: [B][T + padn][N], rows padn .. written; or null
    float* out32;          // MODE 0: the same rows as float32 ([B][T + padn][N]), or null; MODE 1: [B][T][N] convolution result
    gl_u32x4* carry;       // [B][nstrips][2]: 16-byte slots {lo, token, hi, token} of the running sums S and Q behind each strip (nstrips > 1)
    unsigned token;
    int B, T, Cin, N, K, padn, nstrips, nsteps;
    float slope;
};

__device__ __forceinline__ void gl_dma(const __bf16* gsrc_lane, __bf16* lds_piece) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane, (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
}

// WN wave columns x (8 / WN) wave rows; a wave owns NPW frame tiles x TNW channel tiles.  N = 16 WN TNW, GL_TP = 16 NPW 8 / WN.
template <int WN, int TNW, int NPW, int MODE>
__global__ __launch_bounds__(512, 1) void glayer16_kernel(GLayerArgs p) {
    static_assert(NPW * (8 / WN) * 16 == GL_TP, "tile");
    static_assert(TNW % 2 == 0, "channel tiles are processed in two halves");
    constexpr int TNH = TNW / 2;
    constexpr int NT = WN * TNW;                       // channel tiles of the layer
    constexpr int WSTEP = NT * 512;                    // weight elements per k-step
    constexpr int CH = GL_SB * WSTEP;                  // per chunk
    constexpr int WPIECES = GL_SB * NT;                // 1 KB pieces per chunk
    constexpr int WPW = (WPIECES + 7) / 8;             // per wave
    extern __shared__ __attribute__((aligned(16))) __bf16 gl_lds[];   // slices [2][GL_SLICE], weight ring [2][CH]; the epilogue's statistics alias the ring
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wp = wave / WN;
    const int b = blockIdx.x / p.nstrips, strip = blockIdx.x - b * p.nstrips;
    const int t0 = strip * GL_TP;
    const int Cin = p.Cin, K = p.K;
    const int rows_in = p.T + K - 1;
    __bf16* wring = gl_lds + 2 * GL_SLICE;

    // ---- DMA sources.  Slice piece k covers slot elements [512 k, 512 k + 512) = frames 8 k .. 8 k + 7; lane l supplies the 16-byte unit
    // (l & 7) of frame 8 k + (l >> 3), which holds the frame's unit (l & 7) ^ (frame & 7) (XOR swizzle: conflict-free 16-byte fragment
    // reads at a 128-byte frame stride).  Frames beyond the input re-read the last row: they only feed frames >= T, which are never stored.
    int ssrc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int q = 8 * (wave + 8 * t) + (lane >> 3), u = (lane & 7) ^ (q & 7);
        ssrc[t] = min(t0 + q, rows_in - 1) * Cin + 8 * u;
    }
    const __bf16* abase = p.A + (size_t)b * rows_in * Cin;
    auto slice_dma = [&](int cg) {
        const __bf16* src = abase + cg * GL_CS;
        __bf16* dst = gl_lds + (cg & 1) * GL_SLICE;
#pragma unroll
        for (int t = 0; t < 5; ++t)
            if (wave + 8 * t < GL_NPOS / 8) gl_dma(src + ssrc[t], dst + 512 * (wave + 8 * t));
    };
    const __bf16* wsrc = p.Wfrag + 512 * wave + 8 * lane;
    auto w_dma = [&](int c) {
        const __bf16* src = wsrc + (size_t)c * CH;
        __bf16* dst = wring + (c & 1) * CH + 512 * wave;
#pragma unroll
        for (int q = 0; q < WPW; ++q)
            if (WPIECES % 8 == 0 || wave + 8 * q < WPIECES) gl_dma(src + 4096 * q, dst + 4096 * q);
    };
    slice_dma(0);
    w_dma(0);

    f32x4 acc[NPW][TNW];
#pragma unroll
    for (int k = 0; k < NPW; ++k)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[k][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // frame-fragment address of this lane inside a slice slot: frame q = p0 + li + tap, unit (4 half + lg) ^ (q & 7)
    const int pbase = (wp * NPW) * 16 + li;
    const int nchunk = p.nsteps / GL_SB;
    const int ncg = Cin / GL_CS;
    __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0)
    __syncthreads();

    int tap = 0, cg = 0;                               // of the next step to run (steps come in (half 0, half 1) pairs: GL_SB = 2 = one tap)
    const __bf16* wl = wring + (size_t)wn * TNW * 512 + lane * 8;
    for (int c = 0; c < nchunk; ++c) {
        bool slice_issued = false;
        if (c + 1 < nchunk) {
            w_dma(c + 1);
            if (tap == 0 && cg + 1 < ncg) { slice_dma(cg + 1); slice_issued = true; }
        }
        const __bf16* sl = gl_lds + (cg & 1) * GL_SLICE;
        auto ldp = [&](int half, bf16x8 (&pf)[NPW]) {
#pragma unroll
            for (int k = 0; k < NPW; ++k) {
                const int q = pbase + 16 * k + tap;
                pf[k] = *reinterpret_cast<const bf16x8*>(sl + q * GL_CS + (((4 * half + lg) ^ (q & 7)) << 3));
            }
        };
        auto ldw = [&](int u, int jh, bf16x8 (&wf)[TNH]) {
#pragma unroll
            for (int j = 0; j < TNH; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(wl + u * WSTEP + (jh * TNH + j) * 512);
        };
        auto mm = [&](const bf16x8 (&pf)[NPW], const bf16x8 (&wf)[TNH], int jh) {
#pragma unroll
            for (int j = 0; j < TNH; ++j)
#pragma unroll
                for (int k = 0; k < NPW; ++k)
                    acc[k][jh * TNH + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], pf[k], acc[k][jh * TNH + j], 0, 0, 0);
        };
        // two k-steps (the two 32-channel halves of one tap) in four half-steps; the fragments of the next half-step are requested
        // before the MFMAs of the current one (sched_barrier keeps the order; the compiler places counted lgkmcnt waits)
        bf16x8 p0[NPW], p1[NPW], wa[TNH], wb[TNH];
        ldp(0, p0);
        ldw(0, 0, wa);
        __builtin_amdgcn_sched_barrier(0);
        ldw(0, 1, wb);
        mm(p0, wa, 0);
        __builtin_amdgcn_sched_barrier(0);
        ldp(1, p1);
        ldw(1, 0, wa);
        mm(p0, wb, 1);
        __builtin_amdgcn_sched_barrier(0);
        ldw(1, 1, wb);
        mm(p1, wa, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(p1, wb, 1);
        __builtin_amdgcn_sched_barrier(0);
        wl += (c & 1) ? -CH : CH;
        if (++tap == K) { tap = 0; ++cg; }
        if (c + 1 < nchunk) {
            // this wave's weight pieces of chunk c + 1 have landed; a slice issued in this chunk (younger than them) may still be in flight:
            // it is needed K chunks from now and is covered by the next chunk's vmcnt(0)
            if (slice_issued && K >= 2) {
                if (wave == 0) __builtin_amdgcn_s_waitcnt(0x0f70 | 5);        // vmcnt(5): wave 0 moves 5 slice pieces, the others 4
                else __builtin_amdgcn_s_waitcnt(0x0f70 | 4);
            } else {
                __builtin_amdgcn_s_waitcnt(0x0f70);
            }
            __syncthreads();
        }
    }

    // ---- epilogue.  Lane (li, lg) holds channels n0 .. n0 + 4 TNW - 1 of frame li of each of its NPW frame tiles.
    const int n0 = wn * 16 * TNW + 4 * TNW * lg;
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int t = t0 + pbase + 16 * k;
            if (t >= p.T) continue;
            float* op = p.out32 + ((size_t)b * p.T + t) * p.N + n0;
#pragma unroll
            for (int j = 0; j < TNW; ++j) *reinterpret_cast<float4*>(op + 4 * j) = make_float4(acc[k][j][0], acc[k][j][1], acc[k][j][2], acc[k][j][3]);
        }
        return;
    }
    __syncthreads();                                   // every wave is done with the weight ring: the statistics live there now
    double* psum = reinterpret_cast<double*>(wring);   // [WN][GL_TP][2]
    double* wtot = psum + WN * GL_TP * 2;              // [4][2] wave totals of the scan
    double* cin = wtot + 8;                            // [2] carry of the strips before this one
    float* smean = reinterpret_cast<float*>(cin + 2);  // [GL_TP]
    float* srstd = smean + GL_TP;                      // [GL_TP]
    {
        float bv[4 * TNW];
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const float4 t4 = *reinterpret_cast<const float4*>(p.bias + n0 + 4 * j);
            bv[4 * j] = t4.x; bv[4 * j + 1] = t4.y; bv[4 * j + 2] = t4.z; bv[4 * j + 3] = t4.w;
        }
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int pos = pbase + 16 * k, t = t0 + pos;
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[k][j][r] + bv[4 * j + r];
                    acc[k][j][r] = v;
                    s += (double)v;
                    q += (double)v * (double)v;
                }
            if (p.Y && t < p.T) {
                float* yp = p.Y + ((size_t)b * p.T + t) * p.N + n0;
#pragma unroll
                for (int j = 0; j < TNW; ++j) *reinterpret_cast<float4*>(yp + 4 * j) = make_float4(acc[k][j][0], acc[k][j][1], acc[k][j][2], acc[k][j][3]);
            }
            s += lane_xor16(s); q += lane_xor16(q);
            s += lane_xor32(s); q += lane_xor32(q);
            if (lg == 0) { psum[(wn * GL_TP + pos) * 2] = s; psum[(wn * GL_TP + pos) * 2 + 1] = q; }
        }
    }
    __syncthreads();
    // ---- cumulative sums over the strip's frames (threads 0 .. 255 = frames), then the carry of the earlier strips
    double S = 0.0, Q = 0.0;
    if (tid < GL_TP) {
#pragma unroll
        for (int w = 0; w < WN; ++w) { S += psum[(w * GL_TP + tid) * 2]; Q += psum[(w * GL_TP + tid) * 2 + 1]; }
        if (t0 + tid >= p.T) { S = 0.0; Q = 0.0; }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double us = __shfl_up(S, o, 64), uq = __shfl_up(Q, o, 64);
            if (lane >= o) { S += us; Q += uq; }
        }
        if (lane == 63) { wtot[2 * wave] = S; wtot[2 * wave + 1] = Q; }
    }
    if (p.nstrips > 1 && tid == 511) {
        double cs = 0.0, cq = 0.0;
        if (strip > 0) {
            // totals of strips 0 .. strip - 1: two 16-byte slots {lo, token, hi, token}, each written with ONE sc1 store by that strip (a
            // 16-byte store is atomic: a matching tag on both halves means the value is complete; no fences)
            const gl_u32x4* slot = p.carry + ((size_t)b * p.nstrips + strip - 1) * 2;
            gl_u32x4 x, y;
            for (;;) {
                asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(x), "=&v"(y) : "v"(slot), "v"(slot + 1) : "memory");
                if (x.y == p.token && x.w == p.token && y.y == p.token && y.w == p.token) break;
                __builtin_amdgcn_s_sleep(2);
            }
            cs = __longlong_as_double((long long)(((unsigned long long)x.z << 32) | x.x));
            cq = __longlong_as_double((long long)(((unsigned long long)y.z << 32) | y.x));
        }
        cin[0] = cs; cin[1] = cq;
    }
    __syncthreads();
    if (tid < GL_TP) {
        for (int w = 0; w < wave; ++w) { S += wtot[2 * w]; Q += wtot[2 * w + 1]; }
        if (p.nstrips > 1) {
            S += cin[0]; Q += cin[1];
            if (tid == GL_TP - 1 && strip + 1 < p.nstrips) {
                gl_u32x4* slot = p.carry + ((size_t)b * p.nstrips + strip) * 2;
                const unsigned long long us = (unsigned long long)__double_as_longlong(S), uq = (unsigned long long)__double_as_longlong(Q);
                gl_u32x4 x, y;
                x.x = (unsigned)us; x.y = p.token; x.z = (unsigned)(us >> 32); x.w = p.token;
                y.x = (unsigned)uq; y.y = p.token; y.z = (unsigned)(uq >> 32); y.w = p.token;
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %2, %3, off sc1" ::"v"(slot), "v"(x), "v"(slot + 1), "v"(y) : "memory");
            }
        }
        const int t = t0 + tid;
        const double n = (double)p.N * (double)(t + 1);
        const double m = S / n;
        const double var = (Q - 2.0 * m * S) / n + m * m;
        const float mf = (float)m, rf = (float)(1.0 / sqrt(var + 1e-8));
        smean[tid] = mf; srstd[tid] = rf;
        if (t < p.T) {
            if (p.mean) p.mean[(size_t)b * p.T + t] = mf;
            if (p.rstd) p.rstd[(size_t)b * p.T + t] = rf;
        }
    }
    __syncthreads();
    {
        float gv[4 * TNW], ev[4 * TNW];
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const float4 g4 = *reinterpret_cast<const float4*>(p.gain + n0 + 4 * j), e4 = *reinterpret_cast<const float4*>(p.beta + n0 + 4 * j);
            gv[4 * j] = g4.x; gv[4 * j + 1] = g4.y; gv[4 * j + 2] = g4.z; gv[4 * j + 3] = g4.w;
            ev[4 * j] = e4.x; ev[4 * j + 1] = e4.y; ev[4 * j + 2] = e4.z; ev[4 * j + 3] = e4.w;
        }
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int pos = pbase + 16 * k, t = t0 + pos;
            if (t >= p.T) continue;
            const float m = smean[pos], r = srstd[pos];
            const size_t orow = ((size_t)b * (p.T + p.padn) + p.padn + t) * p.N + n0;
            float o[4 * TNW];
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    float v = (acc[k][j][rr] - m) * r * gv[4 * j + rr] + ev[4 * j + rr];
                    o[4 * j + rr] = v > 0.f ? v : p.slope * v;
                }
            if (p.out16) {
#pragma unroll
                for (int h = 0; h < TNW / 2; ++h) {
                    bf16x8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (__bf16)o[8 * h + e];
                    *reinterpret_cast<bf16x8*>(p.out16 + orow + 8 * h) = hv;
                }
            }
            if (p.out32) {
#pragma unroll
                for (int j = 0; j < TNW; ++j) *reinterpret_cast<float4*>(p.out32 + orow + 4 * j) = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ weight fragments
// Wg [N][K * Cin] float32, k order (tap, c) (the GEMM layouts nele_weight_prep writes: forward, or flipped for the data gradient) -> bf16
// fragment stream [nsteps + GL_SB][WN][TNW][64 lanes][8]: step s = (slice cg, tap, half) = ((s / 2) / K, (s / 2) % K, s & 1); lane
// (m = lane & 15, gq = lane >> 4) of tile (wn, j) holds W[n][tap * Cin + 64 cg + 32 half + 8 gq ..] for the output channel
// n = 16 TNW wn + 4 TNW (m >> 2) + 4 j + (m & 3) - the permutation that leaves an accumulator lane with 4 TNW consecutive channels.
struct GLFragJobs { const float* Wg[16]; __bf16* Wfrag[16]; int N[16], Cin[16], K[16]; };
__global__ void glayer_frag_kernel(GLFragJobs J) {
    const int job = blockIdx.y;
    const float* __restrict__ Wg = J.Wg[job];
    __bf16* __restrict__ Wf = J.Wfrag[job];
    const int N = J.N[job], Cin = J.Cin[job], K = J.K[job];
    const int NT = N / 16, TNW = N >= 256 ? 8 : 4, nsteps = (Cin / GL_CS) * K * 2;
    const long long total = (long long)(nsteps + GL_SB) * NT * 512;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long long f = idx >> 9;
        const int tile = (int)(f % NT), s = (int)(f / NT);
        const int wn = tile / TNW, j = tile - wn * TNW, m = lane & 15, gq = lane >> 4;
        const int n = 16 * TNW * wn + 4 * TNW * (m >> 2) + 4 * j + (m & 3);
        float v = 0.f;
        if (s < nsteps) {
            const int half = s & 1, tp = (s >> 1) % K, cg = (s >> 1) / K;
            v = Wg[(size_t)n * K * Cin + (size_t)tp * Cin + cg * GL_CS + 32 * half + 8 * gq + e];
        }
        Wf[idx] = (__bf16)v;
    }
}

// cat(x, y) [B][T][64] x 2 float32 -> left-padded bf16 conv input [B][T + pad][128] (model.py:85-86); rows < pad stay zero
__global__ __launch_bounds__(256) void g_pack16_kernel(const float* __restrict__ x, const float* __restrict__ y, __bf16* __restrict__ out, int T, int pad, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c8 = (int)(i & 15);                  // group of 8 channels out of 128
        const long long bt = i >> 4;
        const int t = (int)(bt % T);
        const long long b = bt / T;
        const float* src = (c8 < 8 ? x : y) + (size_t)bt * 64 + 8 * (c8 & 7);
        const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
        bf16x8 v;
        v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
        v[4] = (__bf16)c.x; v[5] = (__bf16)c.y; v[6] = (__bf16)c.z; v[7] = (__bf16)c.w;
        *reinterpret_cast<bf16x8*>(out + ((size_t)b * (T + pad) + pad + t) * 128 + 8 * c8) = v;
    }
}

// ------------------------------------------------------------------------------------------ host side
static bool gl_shape_ok(int Cin, int N, int K) {
    return Cin >= GL_CS && Cin % GL_CS == 0 && Cin <= 1024 && (N == 64 || N == 256) && K >= 1 && K <= GL_NPOS - GL_TP + 1;
}
extern "C" int nele_glayer16_supported(int Cin, int N, int K) { return gl_shape_ok(Cin, N, K) ? 1 : 0; }
extern "C" long long nele_glayer16_wfrag_elems(int Cin, int N, int K) {
    if (!gl_shape_ok(Cin, N, K)) return 0;
    return (long long)((Cin / GL_CS) * K * 2 + GL_SB) * (N / 16) * 512;
}
/* bytes of the carry workspace for utterances of T frames (two tagged 16-byte slots per 256-frame strip); zero it once */
extern "C" long long nele_glayer16_carry_bytes(int B, int T) { return (long long)B * ((T + GL_TP - 1) / GL_TP) * 32; }

// ptrs_host: per job {Wg float32 [N][K * Cin], Wfrag bf16}; dims_host: per job {N, Cin, K}
extern "C" int nele_glayer16_weight_prep_batch(const void* const* ptrs_host, const int* dims_host, int jobs, void* stream) {
    NELE_CHECK_ARG(ptrs_host && dims_host && jobs >= 1 && jobs <= 16, "nele_glayer16_weight_prep_batch: 1..16 jobs");
    GLFragJobs J;
    long long mx = 0;
    for (int i = 0; i < jobs; ++i) {
        J.Wg[i] = (const float*)ptrs_host[2 * i]; J.Wfrag[i] = (__bf16*)ptrs_host[2 * i + 1];
        J.N[i] = dims_host[3 * i]; J.Cin[i] = dims_host[3 * i + 1]; J.K[i] = dims_host[3 * i + 2];
        NELE_CHECK_ARG(J.Wg[i] && J.Wfrag[i] && gl_shape_ok(J.Cin[i], J.N[i], J.K[i]), "nele_glayer16_weight_prep_batch: bad job %d", i);
        const long long t = nele_glayer16_wfrag_elems(J.Cin[i], J.N[i], J.K[i]);
        if (t > mx) mx = t;
    }
    const int blocks = (int)((mx + 255) / 256 < 512 ? (mx + 255) / 256 : 512);
    hipLaunchKernelGGL(glayer_frag_kernel, dim3(blocks, jobs), dim3(256), 0, as_stream(stream), J);
    NELE_CHECK_LAUNCH("glayer_frag_kernel");
    return NELE_OK;
}

extern "C" int nele_g_pack16(const float* x, const float* y, void* out16, int B, int T, int pad, void* stream) {
    NELE_CHECK_ARG(x && y && out16 && B > 0 && T > 0 && pad >= 0, "nele_g_pack16: bad arguments");
    const long long total = (long long)B * T * 16;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(g_pack16_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, as_stream(stream), x, y, (__bf16*)out16, T, pad, total);
    NELE_CHECK_LAUNCH("g_pack16_kernel");
    return NELE_OK;
}

template <int WN, int TNW, int NPW, int MODE>
static void gl_launch(const GLayerArgs& a, hipStream_t s) {
    constexpr size_t lds = (2 * (size_t)GL_SLICE + 2 * (size_t)GL_SB * WN * TNW * 512) * 2;
    static_assert(lds >= 2 * GL_SLICE * 2 + (WN * GL_TP * 2 + 10) * 8 + 2 * GL_TP * 4, "the statistics must fit into the weight ring");
    static unsigned long long attr = 0;
    if (nele_first_use_on_device(&attr))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(glayer16_kernel<WN, TNW, NPW, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((glayer16_kernel<WN, TNW, NPW, MODE>), dim3((unsigned)(a.B * a.nstrips)), dim3(512), lds, s, a);
}

static int gl_fill(GLayerArgs& a, const void* A16, const void* Wfrag, int B, int T, int Cin, int N, int K) {
    if (!gl_shape_ok(Cin, N, K)) return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_glayer16: unsupported layer (Cin %d N %d K %d)", Cin, N, K);
    a.A = (const __bf16*)A16; a.Wfrag = (const __bf16*)Wfrag; a.B = B; a.T = T; a.Cin = Cin; a.N = N; a.K = K;
    a.nstrips = (T + GL_TP - 1) / GL_TP; a.nsteps = (Cin / GL_CS) * K * 2;
    return NELE_OK;
}

/* Conv1d + Chomp1d + cLN + LeakyReLU of one generator layer (model.py:83-91).  A16 [B][T + K - 1][Cin] bf16 (K - 1 zero rows in front),
 * out16 [B][T + padn][N] bf16 (rows padn ..: the next layer's input; its first padn rows stay as they are).  Optional outputs (NULL = not
 * written): Y [B][T][N] float32 = convolution + bias, mean / rstd [B][T] (the backward pass reads the three), out32 = the activation in
 * float32, same layout as out16.  carry: nele_glayer16_carry_bytes(B, T) zero-initialised bytes, needed when T > 256; token: any value
 * that differs from call to call on the same carry buffer (e.g. a counter), never 0. */
extern "C" int nele_glayer16_fwd(const void* A16, const void* Wfrag, const float* bias, const float* gain, const float* beta, float* Y, float* mean,
                                 float* rstd, void* out16, float* out32, void* carry, unsigned token, int B, int T, int Cin, int N, int K,
                                 int padn, float slope, void* stream) {
    NELE_CHECK_ARG(A16 && Wfrag && bias && gain && beta && (out16 || out32) && B > 0 && T > 0 && padn >= 0, "nele_glayer16_fwd: bad arguments");
    GLayerArgs a;
    memset(&a, 0, sizeof(a));
    const int st = gl_fill(a, A16, Wfrag, B, T, Cin, N, K);
    if (st) return st;
    NELE_CHECK_ARG(a.nstrips == 1 || (carry && token != 0), "nele_glayer16_fwd: T > 256 needs the carry workspace and a non-zero token");
    a.bias = bias; a.gain = gain; a.beta = beta; a.Y = Y; a.mean = mean; a.rstd = rstd; a.out16 = (__bf16*)out16; a.out32 = out32;
    a.carry = (gl_u32x4*)carry; a.token = token; a.padn = padn; a.slope = slope;
    hipStream_t s = as_stream(stream);
    auto go = [&]() { if (N == 256) gl_launch<2, 8, 4, 0>(a, s); else gl_launch<1, 4, 2, 0>(a, s); };
    NELE_PROF("glayer16_kernel", s, go());
    NELE_CHECK_LAUNCH("glayer16_kernel");
    return NELE_OK;
}

/* The convolution alone, float32 result [B][T][N]: the data gradient of a generator layer = this over the END-padded bf16 output gradient
 * [B][T + K - 1][Cin = layer's output channels] with the flipped weights (N = layer's input channels). */
extern "C" int nele_glayer16_conv(const void* A16, const void* Wfrag, float* out32, int B, int T, int Cin, int N, int K, void* stream) {
    NELE_CHECK_ARG(A16 && Wfrag && out32 && B > 0 && T > 0, "nele_glayer16_conv: bad arguments");
    GLayerArgs a;
    memset(&a, 0, sizeof(a));
    const int st = gl_fill(a, A16, Wfrag, B, T, Cin, N, K);
    if (st) return st;
    a.out32 = out32;
    hipStream_t s = as_stream(stream);
    auto go = [&]() { if (N == 256) gl_launch<2, 8, 4, 1>(a, s); else gl_launch<1, 4, 2, 1>(a, s); };
    NELE_PROF("glayer16_conv_kernel", s, go());
    NELE_CHECK_LAUNCH("glayer16_kernel(conv)");
    return NELE_OK;
}
