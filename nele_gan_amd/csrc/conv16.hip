// Discriminator convolutions on bf16 activations (bf16 mode of model.Discriminator): forward and data-gradient of the
// spectral-norm Conv2d layers (model.py:105-109, 118-122) as implicit GEMM on v_mfma_f32_16x16x32_bf16, float32 accumulate.
//
// Round 3 redesign of conv_tile16_kernel (dense.hip), built around what its profile showed: one workgroup per CU
// with the whole (TH + KH - 1)-row halo resident, float32 activations converted while staged, two barriers per weight chunk, an
// LDS transpose in the epilogue - prologue and epilogue (24 % of a tile) overlapped with nothing, MFMA-busy 0.31.  Here:
//   * activations and output gradients are bfloat16 IN MEMORY (every consumer rounded them to bf16 anyway, so the operands are the
//     same numbers): half the staging traffic, no conversions, half the epilogue bytes;
//   * the halo is a RING of NR = TH + 1 (or TH + 2) rows instead of TH + KH - 1: kernel row kh reads input rows kh .. kh + TH - 1,
//     the row that comes next is loaded while the current kernel row is multiplied and lands in the slot of a dead row.  LDS per
//     workgroup drops from 139 KB to 67 KB for D.conv5, so TWO workgroups share a CU (2 waves per SIMD): one's prologue / epilogue /
//     barrier waits run under the other's MFMAs;
//   * weight fragments arrive in chunks of SB = 4 k-steps through a two-slot LDS ring, global loads issued one chunk ahead into
//     registers, ONE barrier per chunk (the store into the other slot happens before it);
//   * the k loop is flat over (kernel row, step): chunks may straddle kernel rows, so SB need not divide the steps of a row;
//   * operands are swapped (weights = MFMA A, positions = MFMA B) and the output channels are permuted inside the weight fragments so
//     that a lane ends up with 4 * TN CONSECUTIVE channels of one position: the epilogue is bias / LeakyReLU / mask in registers and
//     16-byte stores straight from the accumulators - no LDS round trip;
//   * positions sit in LDS at stride C with their 16-byte units XOR-swizzled by the position index where C is 32 or 64 (a power-of-two
//     stride would put the 16 lanes of a fragment read on 2 or 4 bank groups): conflict-free 16-byte fragment reads without padding.
// Layout contract: A [B][H][W][C] bf16, out [B][OH][OW][OC] bf16 or float32, aux (forward activation for the LeakyReLU mask of a
// data gradient) [B][Hout][Wout][N] bf16; C, N multiples of 8 / 4.  The data gradient is the same kernel over the zero-bordered
// gradient buffer with flipped weights, exactly as before.
#include "conv_common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

#define C16_TW 64        // output columns per tile
#ifndef C16_SB
#define C16_SB 4         // k-steps (of 32) per weight chunk
#endif

// -DC16_PROF: per-phase clock sums of thread 0 of every workgroup (s_memtime), read back by nele_conv16_prof_read - a diagnostic build only
// (tools/variants.sh conv16 prof:"-DC16_PROF"; tools/conv16_check.py prints the table when the library exports the reader).
#ifdef C16_PROF
__device__ unsigned long long c16_prof[8];       // 0 total, 1 prologue (to the first barrier), 2 DMA wait at chunk ends, 3 barrier at chunk ends, 4 epilogue, 5 workgroups
#define C16_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define C16_ACC(slot, t0_, t1_) do { c16_acc[slot] += (t1_) - (t0_); } while (0)
#else
#define C16_T(var) do {} while (0)
#define C16_ACC(slot, t0_, t1_) do {} while (0)
#endif

struct Conv16Args {
    const __bf16* A;
    const __bf16* Wfrag;    // [KH * sps][TN][64 lanes][8]: lane (m = lane & 15, g = lane >> 4) holds W[channel 4 TN (m >> 2) + 4 j + (m & 3)][32 u + 8 g ..]
    const float* bias;
    const __bf16* aux;
    void* out;
    int N, epi;
    float slope;
    int KH, KW, sps, nsteps;   // k-steps per kernel row (seglen rounded up to 32), KH * sps
    int RSP, NR;               // row stride in LDS (elements), ring rows
    int lgC, swz_sh, swz_mask; // unit u of position q sits at unit u ^ ((q >> swz_sh) & swz_mask) (C = 1 << lgC when swz_mask != 0)
    int ntiles, TH;
    double* gap;               // pooled partial sums [B][parts][N] (EPI_BIAS_LRELU only; NULL = none), parts = tiles per image * 4 waves
    const int* wvalid;         // valid output width per image for the pooling (NULL = Wout)
    ConvGeom g;
};

// one 1 KB piece global -> LDS without registers: lane l's 16 bytes land at lds_piece + 16 l (M0 carries the wave-uniform LDS address)
__device__ __forceinline__ void c16_dma(const __bf16* gsrc_lane, __bf16* lds_piece) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane, (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
}

template <int TN, int TH, bool OUT16>
__global__ __launch_bounds__(256, 2) void conv16_kernel(Conv16Args p) {
    C16_T(t_start);
#ifdef C16_PROF
    unsigned long long c16_acc[5] = {0, 0, 0, 0, 0};
#endif
    extern __shared__ __attribute__((aligned(16))) __bf16 lds16[];    // ring [NR][RSP] + 64 slack, then the weight ring [2][SB][TN][64][8]
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform: everything per position tile below is scalar
    // 1-D grid, XCD-aware: workgroup w runs on XCD w % 8; each XCD takes a contiguous run of tiles (tile columns fastest, then tile rows,
    // then utterances): tiles that share halo rows / columns are neighbours in one L2
    int wo0, ho0, b;
    {
        const int ntw = (g.Wout + C16_TW - 1) / C16_TW, nth = (g.Hout + TH - 1) / TH;
        const int per = (int)((gridDim.x + 7) >> 3);
        const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
        if (tile >= p.ntiles) return;
        const int tw_i = tile % ntw, rem = tile / ntw;
        wo0 = tw_i * C16_TW; ho0 = (rem % nth) * TH; b = rem / nth;
    }
    const int C = g.C, RSP = p.RSP, NR = p.NR;
    const int wcols = C16_TW + p.KW - 1;
    const int nrows = TH + p.KH - 1;                   // input rows this tile touches
    const int hi0 = ho0 + g.ih0, wi0 = wo0 + g.iw0;
    const int vcols = max(1, min(wcols, g.W - wi0));   // columns that exist in the input (the rest re-reads the last one: see below)
    const int vrows = max(1, min(nrows, g.H - hi0));
    const __bf16* abase = p.A + (((size_t)b * g.H + hi0) * g.W + wi0) * C;
    __bf16* wring = lds16 + NR * RSP + 64;
    constexpr int CH = C16_SB * TN * 512;              // elements per weight chunk

    // ---- input rows: global -> LDS by DMA, 1 KB pieces; piece k of a row image covers LDS elements [512 k, 512 k + 512) of the slot,
    // lane l supplies the 8 elements at o = 512 k + 8 l: position o / C, unit (o % C) / 8 - which holds the position's unit
    // u ^ swizzle(position); o >= wcols * C is the slot's tail (re-reads a valid element: it only ever meets zero weights).  Columns / rows outside the
    // input re-read the last valid column / row: they only feed outputs outside the output (never stored) and k-padding.
    // Wave w moves pieces w, w + 4, w + 8 (RSP <= 12 pieces).
    const int npieces = RSP >> 9;
    int rsrc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int o = 512 * (wave + 4 * t) + 8 * lane;
        const int q = o / C, un = ((o - q * C) >> 3) ^ ((q >> p.swz_sh) & p.swz_mask);
        rsrc[t] = min(q, vcols - 1) * C + 8 * un;
    }
    auto row_dma = [&](int r) {
        const __bf16* src = abase + (size_t)min(r, vrows - 1) * g.W * C;
        __bf16* dst = lds16 + (r % NR) * RSP;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (wave + 4 * t < npieces) c16_dma(src + rsrc[t], dst + 512 * (wave + 4 * t));
    };
    // ---- weight chunks: SB * TN pieces of 1 KB per chunk, wave w moves pieces w, w + 4, ..
    const int nchunk = (p.nsteps + C16_SB - 1) / C16_SB;
    const __bf16* wsrc = p.Wfrag + 512 * wave + 8 * lane;
    auto w_dma = [&](int c) {
        const __bf16* src = wsrc + (size_t)c * CH;
        __bf16* dst = wring + (c & 1) * CH + 512 * wave;
#pragma unroll
        for (int q = 0; q < (C16_SB * TN + 3) / 4; ++q)
            if ((C16_SB * TN) % 4 == 0 || wave + 4 * q < C16_SB * TN) c16_dma(src + 2048 * q, dst + 2048 * q);
    };
    int resident = min(NR, nrows) - 1;                 // rows 0 .. resident are (being) staged: every ring slot is written here once
    for (int r = 0; r <= resident; ++r) row_dma(r);
    w_dma(0);
    if (tid < 8) *reinterpret_cast<bf16x8*>(lds16 + NR * RSP + 8 * tid) = (bf16x8){(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};

    // ---- position tiles: the tile's nr x ncol valid 16-column groups are dealt round-robin to the 4 waves: wave w takes t = w, w + 4, ..
    // (NP = ceil(nr * ncol / 4) each; a surplus slot recomputes tile t mod total and stores nothing).  Full tiles: NP = TH.
    const int nr = min(TH, g.Hout - ho0), ncol = min(4, (g.Wout - wo0 + 15) >> 4), total = nr * ncol;
    const int NPr = (total + 3) >> 2;
    const int n0 = 4 * TN * lg;
    const int vb = li * C;                             // lane part of a fragment address (elements)

    auto run = [&](auto np_tag) {
        constexpr int NP = decltype(np_tag)::value;
        int prow[NP], pcol[NP];
        bool pok[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            int t = wave + 4 * k;
            pok[k] = t < total;
            if (!pok[k]) t -= total;                   // total >= 1; wave + 4 k < total + 4: one subtraction is enough when total >= 4, else modulo
            if (t >= total) t %= total;
            prow[k] = t / ncol; pcol[k] = t - prow[k] * ncol;
        }
        f32x4 acc[NP][TN];
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[k][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0): this wave's DMA pieces have landed
        __syncthreads();
        C16_T(t_pro);
        C16_ACC(1, t_start, t_pro);

        // Of the NEXT step to load: lu = step within its kernel row; per position tile the ring slot rr[k] of its input row and the
        // (wave-uniform) element offset so[k] of that slot + the tile's columns - advanced once per kernel row behind a real scalar
        // branch (recomputed per step, or as per-step selects, this scalar arithmetic was 2.3 instructions per MFMA)
        int lu = 0, rr[NP], so[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) { rr[k] = prow[k]; so[k] = prow[k] * RSP + 16 * pcol[k] * C; }
        const __bf16* wl = wring + lane * 8;
        auto ldfrag = [&](int uu, bf16x8 (&af)[NP], bf16x8 (&bfr)[TN]) {
            const int kk8 = lu * 32 + 8 * lg;
            const int vk = (vb + kk8) ^ ((((li + (kk8 >> p.lgC)) >> p.swz_sh) & p.swz_mask) << 3);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(wl + (uu * TN + j) * 512);
#pragma unroll
            for (int k = 0; k < NP; ++k) af[k] = *reinterpret_cast<const bf16x8*>(lds16 + so[k] + vk);
            if (__builtin_expect(++lu == p.sps, 0)) {
                asm volatile("" ::: "memory");             // (keeps the compiler from turning the branch into selects)
                lu = 0;
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    so[k] += RSP;
                    if (++rr[k] == NR) { rr[k] = 0; so[k] -= NR * RSP; }
                }
            }
        };
        auto mm = [&](const bf16x8 (&af)[NP], const bf16x8 (&bfr)[TN]) {
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[k][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[k], acc[k][j], 0, 0, 0);
        };
        // One ds_read (and its address arithmetic) in the shadow of every MPR MFMAs; the fences keep the compiler from sinking the next
        // step's reads back behind this step's MFMAs, and put the wait for them AFTER the MFMAs were issued.
        constexpr int MPR = (NP * TN) / (NP + TN) > 0 ? (NP * TN) / (NP + TN) : 1;
#define C16_INTERLEAVE()                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < NP + TN; ++q_) {                              \
        __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                \
    }                                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, NP * TN, 0);                              \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    __builtin_amdgcn_s_waitcnt(0xc07f);                                                   \
    __builtin_amdgcn_sched_barrier(0)
        const int nfull = p.nsteps / C16_SB, nrem = p.nsteps - nfull * C16_SB;
        int nl_row = (2 * C16_SB - 1) / p.sps, nl_rem = (2 * C16_SB - 1) - nl_row * p.sps;   // kernel row of step 2 SB - 1 (last step of chunk 1), kept by addition
        for (int c = 0; c < nfull; ++c) {
            // DMA behind this chunk's MFMAs: the weights of chunk c + 1 into the slot chunk c - 1 left, and the input row chunk c + 1 needs first
            if (c + 1 < nchunk) {
                w_dma(c + 1);
                const int need = min(nl_row + TH - 1, nrows - 1);                        // rows <= need must be resident when chunk c + 1 starts
                if (need > resident) { ++resident; row_dma(resident); }
                nl_rem += C16_SB;
                while (nl_rem >= p.sps) { nl_rem -= p.sps; ++nl_row; }
            }
            // ---- SB k-steps from LDS only; fragments double buffered in registers
            bf16x8 a0[NP], b0[TN], a1[NP], b1[TN];
            ldfrag(0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int uu = 0; uu + 1 < C16_SB; ++uu) {
                if (uu & 1) { ldfrag(uu + 1, a0, b0); mm(a1, b1); }
                else { ldfrag(uu + 1, a1, b1); mm(a0, b0); }
                C16_INTERLEAVE();
            }
            if (C16_SB & 1) mm(a0, b0); else mm(a1, b1);
            wl += (c & 1) ? -CH : CH;
            if (c + 1 < nchunk) {
                C16_T(t_w0);
                __builtin_amdgcn_s_waitcnt(0x0f70);    // vmcnt(0): the pieces this wave issued at the top of the chunk have landed
                C16_T(t_w1);
                __syncthreads();                       // ONE barrier: chunk c + 1 and the new row are visible, chunk c's slot is free
                C16_T(t_w2);
                C16_ACC(2, t_w0, t_w1);
                C16_ACC(3, t_w1, t_w2);
            }
        }
#undef C16_INTERLEAVE
        for (int uu = 0; uu < nrem; ++uu) {            // the last, short chunk (nothing left to prefetch)
            bf16x8 a0[NP], b0[TN];
            ldfrag(uu, a0, b0);
            mm(a0, b0);
        }

        C16_T(t_epi);
        // ---- epilogue, in registers: lane (li, lg) holds channels n0 .. n0 + 4 TN - 1 of position li of each of its position tiles
        float bv[4 * TN];
#pragma unroll
        for (int q = 0; q < 4 * TN; ++q)
            bv[q] = ((p.epi == EPI_BIAS || p.epi == EPI_BIAS_LRELU) && n0 + q < p.N) ? p.bias[n0 + q] : 0.f;
        // global average pooling fused into the last layer's forward pass (model.py:123 AdaptiveAvgPool2d(1) on the LeakyReLU output): every
        // lane sums its positions' float32 results in float64; the 16 lanes of a channel group are added below and each wave stores one
        // partial per channel - the consumer (gap_mlp_fwd_parts) adds the partials of an image in a fixed order
        const bool pool = p.gap != nullptr;
        const int wpool = pool ? (p.wvalid ? min(p.wvalid[b], g.Wout) : g.Wout) : 0;
        double gs[4 * TN];
#pragma unroll
        for (int q = 0; q < 4 * TN; ++q) gs[q] = 0.0;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int ho = ho0 + prow[k], wo = wo0 + 16 * pcol[k] + li;
            if (!pok[k] || wo >= g.Wout || n0 >= p.N) continue;
            float v[4 * TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) v[4 * j + reg] = acc[k][j][reg] + bv[4 * j + reg];
            if (p.epi == EPI_BIAS_LRELU) {
#pragma unroll
                for (int q = 0; q < 4 * TN; ++q) v[q] = v[q] > 0.f ? v[q] : p.slope * v[q];
                if (pool && wo < wpool) {
#pragma unroll
                    for (int q = 0; q < 4 * TN; ++q) gs[q] += (double)v[q];
                }
            } else if (p.epi == EPI_MASK_LRELU_GRAD) {
                const __bf16* xp = p.aux + (((size_t)b * g.Hout + ho) * g.Wout + wo) * p.N + n0;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (n0 + 4 * j < p.N) {
                        const bf16x4 x = *reinterpret_cast<const bf16x4*>(xp + 4 * j);
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) v[4 * j + reg] = (float)x[reg] > 0.f ? v[4 * j + reg] : p.slope * v[4 * j + reg];
                    }
                }
            }
            const size_t o_off = (((size_t)b * g.OH + ho + g.oh0) * g.OW + wo + g.ow0) * g.OC + n0;
            if (OUT16) {
                __bf16* op = reinterpret_cast<__bf16*>(p.out) + o_off;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (n0 + 4 * j < p.N) {
                        bf16x4 h;
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) h[reg] = (__bf16)v[4 * j + reg];
                        *reinterpret_cast<bf16x4*>(op + 4 * j) = h;
                    }
                }
            } else {
                float* op = reinterpret_cast<float*>(p.out) + o_off;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (n0 + 4 * j < p.N) *reinterpret_cast<float4*>(op + 4 * j) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
            }
        }
        if (pool) {
            const int ntw_ = (g.Wout + C16_TW - 1) / C16_TW, nth_ = (g.Hout + TH - 1) / TH;
            const int part = ((ho0 / TH) * ntw_ + wo0 / C16_TW) * 4 + wave;
            double* gp = p.gap + ((size_t)b * (ntw_ * nth_ * 4) + part) * p.N + n0;
#pragma unroll
            for (int q = 0; q < 4 * TN; ++q) {
                const double t = row16_sum_dpp(gs[q]);      // (DPP row operations: no traffic through the LDS crossbar this kernel is bound by)
                if (li == 0 && n0 + q < p.N) gp[q] = t;
            }
        }
        C16_T(t_end);
        C16_ACC(4, t_epi, t_end);
        C16_ACC(0, t_start, t_end);
#ifdef C16_PROF
        if (threadIdx.x == 0) { for (int q_ = 0; q_ < 5; ++q_) atomicAdd(&c16_prof[q_], c16_acc[q_]); atomicAdd(&c16_prof[5], 1ull); }
#endif
    };
    // NPr is workgroup-uniform: one scalar branch to the fully unrolled body for that many position tiles per wave
    if (TH == 8) {
        switch (NPr) {
            case 8: run(std::integral_constant<int, (TH == 8 ? 8 : 1)>{}); break;
            case 7: run(std::integral_constant<int, (TH == 8 ? 7 : 1)>{}); break;
            case 6: run(std::integral_constant<int, (TH == 8 ? 6 : 1)>{}); break;
            case 5: run(std::integral_constant<int, (TH == 8 ? 5 : 1)>{}); break;
            case 4: run(std::integral_constant<int, 4>{}); break;
            case 3: run(std::integral_constant<int, 3>{}); break;
            case 2: run(std::integral_constant<int, 2>{}); break;
            default: run(std::integral_constant<int, 1>{}); break;
        }
    } else {
        switch (NPr) {
            case 4: run(std::integral_constant<int, 4>{}); break;
            case 3: run(std::integral_constant<int, 3>{}); break;
            case 2: run(std::integral_constant<int, 2>{}); break;
            default: run(std::integral_constant<int, 1>{}); break;
        }
    }
}

// ------------------------------------------------------------------------------------------ D's first layer (1 x 1, 4 -> 8 channels)
// out[m][n] = LeakyReLU(bias[n] + sum_c in[m][c] * W[n][c]), exact float32 arithmetic, result stored as bf16: the layer is a stream of
// 16 bytes in, 16 bytes out per position (model.py:105, 118 first Conv2d); one thread per position.  The fma chain runs over the
// channels in the order 0, 2, 1, 3 - the order in which conv_gemm_kernel's two v_mfma_f32_16x16x4_f32 per k-step of 8 (even k, then
// odd k) accumulate them - so the float32 values are the ones the float32-buffer path produces.
__global__ __launch_bounds__(256) void conv16_pointwise_kernel(const float4* __restrict__ in, const float* __restrict__ W, const float* __restrict__ bias,
                                                               __bf16* __restrict__ out, long long M, float slope) {
    __shared__ float w[8][4], bs[8];
    if (threadIdx.x < 32) w[threadIdx.x >> 2][threadIdx.x & 3] = W[threadIdx.x];
    if (threadIdx.x < 8) bs[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256) {
        const float4 a = in[m];
        bf16x8 h;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            float v = fmaf(a.w, w[n][3], fmaf(a.y, w[n][1], fmaf(a.z, w[n][2], a.x * w[n][0]))) + bs[n];
            v = v > 0.f ? v : slope * v;
            h[n] = (__bf16)v;
        }
        *reinterpret_cast<bf16x8*>(out + 8 * m) = h;
    }
}

// in [M][4] float32 (channels-last D input), Wf [8][4] float32 (nele_weight_prep's forward layout, sigma-normalised), out [M][8] bf16
extern "C" int nele_conv16_pointwise_fwd(const float* in, const float* Wf, const float* bias, void* out16, long long M, int N, float slope, void* stream) {
    NELE_CHECK_ARG(in && Wf && bias && out16 && M > 0 && N == 8, "nele_conv16_pointwise_fwd: needs 4 -> 8 channels");
    const long long blocks = (M + 255) / 256;
    hipLaunchKernelGGL(conv16_pointwise_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(in), Wf, bias, reinterpret_cast<__bf16*>(out16), M, slope);
    NELE_CHECK_LAUNCH("conv16_pointwise_kernel");
    return NELE_OK;
}

// ------------------------------------------------------------------------------------------ weight fragments
// Wg [N][Ktot] float32, k order (kh, kw, c) (the layouts nele_weight_prep writes: forward, or flipped for the data gradient) ->
// bf16 fragment stream [KH * sps][TN][64][8] with the channel permutation of conv16_kernel's epilogue.
struct C16FragJobs { const float* Wg[16]; __bf16* Wfrag[16]; int N[16], Ktot[16], seglen[16], KH[16]; };
__global__ void conv16_frag_kernel(C16FragJobs J) {
    const int job = blockIdx.y;
    const float* __restrict__ Wg = J.Wg[job];
    __bf16* __restrict__ Wf = J.Wfrag[job];
    const int N = J.N[job], Ktot = J.Ktot[job], seglen = J.seglen[job], KH = J.KH[job];
    const int TN = (N + 15) >> 4, sps = (seglen + 31) >> 5;
    const long long total = (long long)(KH * sps + C16_SB) * TN * 512;       // + one chunk of zeros: the last chunk's loads run past the stream
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long long f = idx >> 9;
        const int j = (int)(f % TN), s = (int)(f / TN);
        const int m = lane & 15, gq = lane >> 4;
        const int n = 4 * TN * (m >> 2) + 4 * j + (m & 3);
        float v = 0.f;
        if (s < KH * sps && n < N) {
            const int kh = s / sps, us = s - kh * sps;
            const int ko = us * 32 + 8 * gq + e;
            if (ko < seglen) v = Wg[(size_t)n * Ktot + kh * seglen + ko];
        }
        Wf[idx] = (__bf16)v;
    }
}

// ------------------------------------------------------------------------------------------ host side
struct C16Plan { int TN, TH, NR, RSP, lgC, swz_sh, swz_mask, sps; size_t lds; };
static bool c16_plan(int N, const ConvGeom& g, int KH, int KW, C16Plan* pl, int epi = EPI_NONE) {
    if (N < 4 || N > 64 || (N & 3) || g.C < 8 || g.C > 64 || (g.C & 7) || KH < 2 || KW < 2 || KW > 9 || KH > 9) return false;
    if (g.seglen != KW * g.C || g.Ktot != KH * KW * g.C) return false;
    C16Plan q;
    q.TN = (N + 15) >> 4;
    q.sps = (g.seglen + 31) >> 5;
    const int nsteps = KH * q.sps;
    // tile rows: 8 where the layer is bound by memory (little work per staged byte) and the registers / LDS allow, else 4
    const int th_env = NELE_SWITCH_INT("NELE_CONV16_TH", 0);
    int th = (nsteps * q.TN <= 64 && q.TN <= 2) ? 8 : 4;      // (TN >= 3 with 8 rows: 256 registers, fragments no longer double-buffered - D.conv5 forward 1.54 against 1.30 ms)
    // ... except for data gradients: their epilogue loads the forward activation for the LeakyReLU mask, and twice the workgroups hide that
    // latency better than 8-row tiles save prologues (conv3's data gradient 0.285 -> 0.222 ms, conv2's 0.137 -> 0.118 at B = 256; the
    // forward passes of the same layers lose 4 - 8 % with 4 rows)
    if (epi == EPI_MASK_LRELU_GRAD) th = 4;
    if (th_env == 4 || (th_env == 8 && q.TN <= 2)) th = th_env;
    q.TH = th;
    // 16-byte fragment reads of 16 consecutive positions: conflict-free at stride 16, 32 and 96 bytes (C = 8, 16, 48); at 64 / 128 bytes
    // the units of a position are XOR-swizzled by the position index (checked against the ds_read_b128 lane groups of MI355X_MICROARCH.md)
    q.lgC = 0; q.swz_sh = 0; q.swz_mask = 0;
    if (g.C == 64) { q.lgC = 6; q.swz_mask = 7; }
    else if (g.C == 32) { q.lgC = 5; q.swz_sh = 1; q.swz_mask = 3; }
    q.RSP = ((C16_TW + KW - 1) * g.C + 511) & ~511;          // whole 1 KB DMA pieces per ring slot
    if (q.RSP > 12 * 512) return false;                       // a wave moves at most 3 pieces of a row
    const int full = q.TH + KH - 1;
    const int ring = q.sps >= 2 * C16_SB - 1 ? q.TH + 1 : q.sps >= C16_SB ? q.TH + 2 : full;
    q.NR = ring < full ? ring : full;
    q.lds = ((size_t)q.NR * q.RSP + 64 + 2 * C16_SB * q.TN * 512) * 2;
    if (q.lds > 160 * 1024) return false;
    *pl = q;
    return true;
}

#ifdef C16_PROF
extern "C" int nele_conv16_prof_read(unsigned long long* out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(c16_prof), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(c16_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" int nele_conv16_supported(int M, int N, const int* geom, int KH, int KW) {
    ConvGeom g;
    memcpy(&g, geom, sizeof(ConvGeom));
    C16Plan pl;
    return c16_plan(N, g, KH, KW, &pl) ? 1 : 0;
}

extern "C" long long nele_conv16_wfrag_elems(int N, int seglen, int KH) {
    const int TN = (N + 15) >> 4, sps = (seglen + 31) >> 5;
    return (long long)(KH * sps + C16_SB) * TN * 512;
}

// ptrs_host: per job {Wg float32 [N][Ktot], Wfrag bf16}; dims_host: per job {N, Ktot, seglen, KH}
extern "C" int nele_conv16_weight_prep_batch(const void* const* ptrs_host, const int* dims_host, int jobs, void* stream) {
    NELE_CHECK_ARG(ptrs_host && dims_host && jobs >= 1 && jobs <= 16, "nele_conv16_weight_prep_batch: 1..16 jobs");
    C16FragJobs J;
    long long mx = 0;
    for (int i = 0; i < jobs; ++i) {
        J.Wg[i] = (const float*)ptrs_host[2 * i]; J.Wfrag[i] = (__bf16*)ptrs_host[2 * i + 1];
        J.N[i] = dims_host[4 * i]; J.Ktot[i] = dims_host[4 * i + 1]; J.seglen[i] = dims_host[4 * i + 2]; J.KH[i] = dims_host[4 * i + 3];
        NELE_CHECK_ARG(J.Wg[i] && J.Wfrag[i] && J.Ktot[i] == J.seglen[i] * J.KH[i], "nele_conv16_weight_prep_batch: bad job %d", i);
        const long long t = nele_conv16_wfrag_elems(J.N[i], J.seglen[i], J.KH[i]);
        if (t > mx) mx = t;
    }
    const int blocks = (int)((mx + 255) / 256 < 512 ? (mx + 255) / 256 : 512);
    hipLaunchKernelGGL(conv16_frag_kernel, dim3(blocks, jobs), dim3(256), 0, as_stream(stream), J);
    NELE_CHECK_LAUNCH("conv16_frag_kernel");
    return NELE_OK;
}

template <int TN, int TH, bool OUT16>
static void c16_launch(const Conv16Args& a, dim3 grid, size_t lds, hipStream_t s) {
    static unsigned long long attr = 0;            // (per device: a process that switches devices sets the attribute on each)
    if (nele_first_use_on_device(&attr)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv16_kernel<TN, TH, OUT16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL((conv16_kernel<TN, TH, OUT16>), grid, dim3(256), lds, s, a);
}

// A16 [B][H][W][C] bf16, Wfrag from nele_conv16_weight_prep_batch, out bf16 (out_bf16 != 0) or float32; aux16: forward activation
// (bf16, [B][Hout][Wout][N]) for EPI_MASK_LRELU_GRAD.  Reference op: F.conv2d + LeakyReLU (model.py:118-122) / its autograd.
static int conv16_impl(const void* A16, const void* Wfrag, const float* bias, const void* aux16, void* out, int out_bf16, int M, int N, int epi,
                       float slope, const int* geom, int KH, int KW, const int* wvalid, double* gap_part, void* stream) {
    NELE_CHECK_ARG(A16 && Wfrag && out && geom, "nele_conv16: null pointer");
    ConvGeom g;
    memcpy(&g, geom, sizeof(ConvGeom));
    C16Plan pl;
    if (!c16_plan(N, g, KH, KW, &pl, epi)) return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_conv16: unsupported geometry (C %d N %d %dx%d)", g.C, N, KH, KW);
    NELE_CHECK_ARG(epi == EPI_NONE || epi == EPI_BIAS || epi == EPI_BIAS_LRELU || epi == EPI_MASK_LRELU_GRAD, "nele_conv16: epilogue %d", epi);
    NELE_CHECK_ARG(epi != EPI_MASK_LRELU_GRAD || aux16, "nele_conv16: the mask epilogue needs the forward activation");
    NELE_CHECK_ARG((epi != EPI_BIAS && epi != EPI_BIAS_LRELU) || bias, "nele_conv16: bias missing");
    NELE_CHECK_ARG(g.Hout > 0 && g.Wout > 0 && M % (g.Hout * g.Wout) == 0, "nele_conv16: M is not a multiple of Hout * Wout");
    NELE_CHECK_ARG((g.OC & 3) == 0, "nele_conv16: output channel stride must be a multiple of 4");
    const int B = M / (g.Hout * g.Wout);
    Conv16Args a;
    a.A = (const __bf16*)A16; a.Wfrag = (const __bf16*)Wfrag; a.bias = bias; a.aux = (const __bf16*)aux16; a.out = out;
    a.N = N; a.epi = epi; a.slope = slope; a.KH = KH; a.KW = KW; a.sps = pl.sps; a.nsteps = KH * pl.sps;
    a.RSP = pl.RSP; a.NR = pl.NR; a.lgC = pl.lgC; a.swz_sh = pl.swz_sh; a.swz_mask = pl.swz_mask; a.TH = pl.TH; a.g = g;
    a.gap = gap_part; a.wvalid = wvalid;
    a.ntiles = ((g.Wout + C16_TW - 1) / C16_TW) * ((g.Hout + pl.TH - 1) / pl.TH) * B;
    const dim3 grid((unsigned)((a.ntiles + 7) / 8 * 8));
    hipStream_t s = as_stream(stream);
#define C16_GO(TN_, TH_) do { if (out_bf16) c16_launch<TN_, TH_, true>(a, grid, pl.lds, s); else c16_launch<TN_, TH_, false>(a, grid, pl.lds, s); } while (0)
#define C16_PICK(TH_) switch (pl.TN) { case 1: C16_GO(1, TH_); break; case 2: C16_GO(2, TH_); break; case 3: C16_GO(3, TH_); break; default: C16_GO(4, TH_); break; }
    // measurement hook: "conv16_kernel" = every launch; "conv16_N<N>_K<Ktot>_e<epi>" = one layer's (D.conv5 forward: conv16_N64_K3888_e2)
    char ptag[48] = "conv16_kernel";
    if (nele_prof_armed() && !nele_prof_match("conv16_kernel")) snprintf(ptag, sizeof(ptag), "conv16_N%d_K%d_e%d", N, g.Ktot, epi);
    NELE_PROF(ptag, s, if (pl.TH == 8) { C16_PICK(8); } else { C16_PICK(4); });
#undef C16_PICK
#undef C16_GO
    NELE_CHECK_LAUNCH("conv16_kernel");
    return NELE_OK;
}

extern "C" int nele_conv16(const void* A16, const void* Wfrag, const float* bias, const void* aux16, void* out, int out_bf16, int M, int N, int epi,
                           float slope, const int* geom, int KH, int KW, void* stream) {
    return conv16_impl(A16, Wfrag, bias, aux16, out, out_bf16, M, N, epi, slope, geom, KH, KW, nullptr, nullptr, stream);
}

// Partial sums per image that nele_conv16_gap writes for this geometry (tiles per image x 4 waves); 0 = unsupported geometry
extern "C" int nele_conv16_gap_parts(int N, const int* geom, int KH, int KW) {
    if (!geom) return 0;
    ConvGeom g;
    memcpy(&g, geom, sizeof(ConvGeom));
    C16Plan pl;
    if (!c16_plan(N, g, KH, KW, &pl, EPI_BIAS_LRELU)) return 0;
    return ((g.Wout + C16_TW - 1) / C16_TW) * ((g.Hout + pl.TH - 1) / pl.TH) * 4;
}

// The discriminator's last conv layer with the pooling fused (model.py:109,121-123: Conv2d -> LeakyReLU -> AdaptiveAvgPool2d(1)): out16 =
// bf16(LeakyReLU(conv + bias)) [B][Hout][Wout][N] (kept for the backward pass's LeakyReLU mask), gap_part [B][parts][N] float64 = sums of the
// float32 LeakyReLU outputs over each wave's positions with column < wvalid[b] (wvalid NULL: all); parts = nele_conv16_gap_parts().
extern "C" int nele_conv16_gap(const void* A16, const void* Wfrag, const float* bias, void* out16, int M, int N, float slope, const int* geom, int KH,
                               int KW, const int* wvalid, double* gap_part, void* stream) {
    NELE_CHECK_ARG(gap_part, "nele_conv16_gap: null pointer");
    return conv16_impl(A16, Wfrag, bias, nullptr, out16, 1, M, N, EPI_BIAS_LRELU, slope, geom, KH, KW, wvalid, gap_part, stream);
}
