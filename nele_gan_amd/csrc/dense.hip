// Dense kernels of the generator / discriminator: convolution as implicit GEMM on the f32-input
// matrix cores (v_mfma_f32_16x16x4_f32: exact float32 products and accumulation, so results agree
// with the reference's float32 torch modules to rounding-order level).
//
//   nele_conv_gemm  : out[m][n] = epi( sum_kk A_view[m][kk] * Wg[n][kk] )      forward and data-gradient
//   nele_conv_wgrad : part[s][n][kk] = sum_{m in split s} dOut[m][n] * A_view[m][kk]   weight gradient
//
// A_view is never materialised: activations are channels-last [B][H][W][C], so the im2col row of
// output position (b,ho,wo) is KH contiguous runs of KW*C floats (run stride W*C).  Conv1d of the
// generator (model.py:49-77) is the H=1 case on a time-padded buffer (Chomp1d == left padding only),
// Conv2d of the discriminator (model.py:105-109) the general case, Linear the KH=KW=1 case.
// Data gradients are the same kernel run over a zero-bordered gradient buffer with flipped weights.
//
// Tiling: 256 threads = 4 waves stacked along M; wave tile 64 x (16*TN); block tile 256 x (16*TN);
// K step 8 (two 16x16x4 MFMAs per tile), LDS double-buffered, one barrier per step.
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include <cstring>

#include "conv_common.h"

struct GemmArgs {
    const float* A;
    const float* Wg;     // [N][Ktot]
    const float* bias;   // [N] or null
    const float* aux;    // EPI_MASK_LRELU_GRAD: forward activation, unpadded [B][Hout][Wout][OC]
    float* out;
    int M, N;
    int epi;
    float slope;
    ConvGeom g;
};

__device__ __forceinline__ void decode_m(int m, const ConvGeom& g, int& b, int& ho, int& wo) {
    const int hw = g.Hout * g.Wout;
    b = m / hw;
    const int r = m - b * hw;
    ho = r / g.Wout;
    wo = r - ho * g.Wout;
}

#define GEMM_BM 256
#define GEMM_BK 8
#define LDS_STRIDE 8

// TM = 16-row tiles per wave (block rows BM = 64*TM): TM = 4 for big M, smaller TM when M alone cannot fill the chip.
template <int TN, int TM>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(GemmArgs p) {
    constexpr int BN = 16 * TN;
    constexpr int BM = 64 * TM;
    constexpr int ALOADS = (BM * 2 + 255) / 256;   // float4 loads of the A tile per thread
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Bs[2][64 * LDS_STRIDE];
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    const int kq = tid & 1;
    size_t a_off[ALOADS];
    bool a_ok[ALOADS];
#pragma unroll
    for (int i = 0; i < ALOADS; ++i) {
        const int row = (tid >> 1) + 128 * i;
        const int m = m0 + row;
        a_ok[i] = (row < BM) && (m < p.M);
        int b = 0, ho = 0, wo = 0;
        if (a_ok[i]) decode_m(m, g, b, ho, wo);
        a_off[i] = (((size_t)b * g.H + ho + g.ih0) * g.W + wo + g.iw0) * g.C;
    }
    const int bn = tid >> 1;
    const bool b_ok = (tid < 2 * BN) && (n0 + bn < p.N);
    const float* wrow = p.Wg + (size_t)(n0 + bn) * g.Ktot;

    int kk = 4 * kq, kh = 0, r = 4 * kq;
    while (r >= g.seglen) { r -= g.seglen; ++kh; }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsteps = (g.Ktot + GEMM_BK - 1) / GEMM_BK;
    float4 ra[ALOADS], rb;
    auto gload = [&]() {
        const bool kin = kk < g.Ktot;
        const size_t koff = (size_t)kh * g.segstride + r;
#pragma unroll
        for (int i = 0; i < ALOADS; ++i)
            ra[i] = (a_ok[i] && kin) ? *reinterpret_cast<const float4*>(p.A + a_off[i] + koff) : make_float4(0, 0, 0, 0);
        rb = (b_ok && kin) ? *reinterpret_cast<const float4*>(wrow + kk) : make_float4(0, 0, 0, 0);
        kk += GEMM_BK;
        r += GEMM_BK;
        while (r >= g.seglen) { r -= g.seglen; ++kh; }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ALOADS; ++i) {
            const int row = (tid >> 1) + 128 * i;
            if (row < BM) *reinterpret_cast<float4*>(&As[buf][row * LDS_STRIDE + 4 * kq]) = ra[i];
        }
        if (tid < 2 * BN) *reinterpret_cast<float4*>(&Bs[buf][bn * LDS_STRIDE + 4 * kq]) = rb;
    };

    gload();
    lstore(0);
    __syncthreads();
    const int li = lane & 15, lg = lane >> 4;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload();
        float2 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            af[i] = *reinterpret_cast<const float2*>(&As[buf][(wave * 16 * TM + i * 16 + li) * LDS_STRIDE + 2 * lg]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float2*>(&Bs[buf][(j * 16 + li) * LDS_STRIDE + 2 * lg]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
        if (s + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = m0 + wave * 16 * TM + i * 16 + 4 * lg + reg;
            if (m >= p.M) continue;
            int b, ho, wo;
            decode_m(m, g, b, ho, wo);
            const size_t o_off = (((size_t)b * g.OH + ho + g.oh0) * g.OW + wo + g.ow0) * g.OC;
            const size_t x_off = (size_t)m * g.OC;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + j * 16 + li;
                if (n >= p.N) continue;
                float v = acc[i][j][reg];
                switch (p.epi) {
                    case EPI_BIAS: v += p.bias[n]; break;
                    case EPI_BIAS_LRELU: v += p.bias[n]; v = v > 0.f ? v : p.slope * v; break;
                    case EPI_MASK_LRELU_GRAD: v = p.aux[x_off + n] > 0.f ? v : p.slope * v; break;
                    case EPI_BIAS_EXPTANH: v += p.bias[n]; v = expf(3.2f * tanhf(v)); break;
                    default: break;
                }
                p.out[o_off + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ span-staged conv
// Same product as conv_gemm_kernel, for 2-D convolutions whose output rows are long (Wout >= 64):
// the 256 consecutive output positions of a block form <= SPAN_MAXRUN runs inside output rows; for the
// current kernel row kh each run needs ONE contiguous input span of (len + KW - 1) * C floats, which is
// staged in LDS once and then read KW times with shifted addresses (im2col traffic / KW).  Weights
// arrive in "fragment-major" order (one coalesced 512 B load per MFMA operand: nele_weight_prep_frag),
// so the K loop has no barrier; barriers only bracket the span staging (KH times per block).
#define SPAN_MAXRUN 8

struct SpanArgs {
    const float* A;
    const float* Wfrag;  // [Ktot/8][NT][64] float2, NT = number of 16-wide n tiles
    const float* bias;
    const float* aux;
    float* out;
    int M, N, NT;
    int epi;
    float slope;
    int KH, KW;
    ConvGeom g;
};

template <int TN>
__global__ __launch_bounds__(256, 2) void conv_span_kernel(SpanArgs p) {
    extern __shared__ __attribute__((aligned(16))) float span[];
    __shared__ int run_gbase[SPAN_MAXRUN];   // element offset of the run's span origin in A, for kh = 0
    __shared__ int run_len[SPAN_MAXRUN];     // output positions in the run
    __shared__ int run_off[SPAN_MAXRUN + 1]; // float offset of the run's span in LDS
    __shared__ int posbase[GEMM_BM];
    __shared__ int nrun_s;
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * GEMM_BM;
    const int mcount = min(GEMM_BM, p.M - m0);
    const int halo = (p.KW - 1) * g.C;

    if (tid == 0) {
        int b, ho, wo;
        decode_m(m0, g, b, ho, wo);
        int left = mcount, r = 0, off = 0, done = 0;
        while (left > 0 && r < SPAN_MAXRUN) {
            const int len = min(left, g.Wout - wo);
            run_gbase[r] = (int)((((size_t)b * g.H + ho + g.ih0) * g.W + wo + g.iw0) * g.C);  // fits int: checked on the host
            run_len[r] = len;
            run_off[r] = off;
            for (int i = 0; i < len; ++i) posbase[done + i] = off + i * g.C;
            off += len * g.C + halo;
            done += len;
            left -= len;
            wo = 0;
            if (++ho == g.Hout) { ho = 0; ++b; }
            ++r;
        }
        run_off[r] = off;
        nrun_s = r;
        for (int i = done; i < GEMM_BM; ++i) posbase[i] = 0;
    }
    __syncthreads();
    const int nrun = nrun_s;
    int pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pb[i] = posbase[wave * 64 + i * 16 + (lane & 15)];

    f32x4 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int lg = lane >> 4;
    const int steps_per_seg = g.seglen / GEMM_BK;
    const float2* wf = reinterpret_cast<const float2*>(p.Wfrag) + lane;
    int ks = 0;
    for (int kh = 0; kh < p.KH; ++kh) {
        __syncthreads();  // previous segment fully consumed
        for (int r = 0; r < nrun; ++r) {
            const int nfl = run_len[r] * g.C + halo;
            const float* src = p.A + (size_t)run_gbase[r] + (size_t)kh * g.segstride;
            float* dst = span + run_off[r];
            for (int e = tid * 4; e < nfl; e += 1024) *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(src + e);
        }
        __syncthreads();
        float2 bf[TN], bn[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = wf[((size_t)ks * p.NT + j) * 64];
        for (int s = 0; s < steps_per_seg; ++s, ++ks) {
            const bool more = (s + 1 < steps_per_seg) || (kh + 1 < p.KH);
            if (more) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bn[j] = wf[((size_t)(ks + 1) * p.NT + j) * 64];
            }
            float2 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const float2*>(span + pb[i] + s * GEMM_BK + 2 * lg);
            // two passes over the tiles so that the two MFMAs of one accumulator are 4*TN issues apart
            // (16x16x4 f32: 32-cycle issue, 40-cycle dependent latency)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
            if (more) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = bn[j];
            }
        }
    }

    const int li = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = m0 + wave * 64 + i * 16 + 4 * lg + reg;
            if (m >= p.M) continue;
            int b, ho, wo;
            decode_m(m, g, b, ho, wo);
            const size_t o_off = (((size_t)b * g.OH + ho + g.oh0) * g.OW + wo + g.ow0) * g.OC;
            const size_t x_off = (size_t)m * g.OC;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = j * 16 + li;
                if (n >= p.N) continue;
                float v = acc[i][j][reg];
                switch (p.epi) {
                    case EPI_BIAS: v += p.bias[n]; break;
                    case EPI_BIAS_LRELU: v += p.bias[n]; v = v > 0.f ? v : p.slope * v; break;
                    case EPI_MASK_LRELU_GRAD: v = p.aux[x_off + n] > 0.f ? v : p.slope * v; break;
                    case EPI_BIAS_EXPTANH: v += p.bias[n]; v = expf(3.2f * tanhf(v)); break;
                    default: break;
                }
                p.out[o_off + n] = v;
            }
        }
    }
}

// W in GEMM layout [N][Ktot] -> fragment-major [Ktot/8][NT][64][2]: element (ks, j, lane, e) =
// W[j*16 + (lane&15)][ks*8 + 2*(lane>>4) + e] (zero for n >= N).
__global__ void weight_frag_kernel(const float* __restrict__ Wg, int N, int Ktot, int NT, float* __restrict__ Wfrag) {
    const int total = (Ktot / 8) * NT * 64 * 2;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 1, lane = (idx >> 1) & 63, t = idx >> 7, j = t % NT, ks = t / NT;
        const int n = j * 16 + (lane & 15), kk = ks * 8 + 2 * (lane >> 4) + e;
        Wfrag[idx] = (n < N) ? Wg[(size_t)n * Ktot + kk] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------ span-staged conv, bf16 operands
// Same structure as conv_span_kernel with bf16 operands and float32 accumulation (v_mfma_f32_16x16x32_bf16): the input
// span is converted to bf16 while it is staged in LDS; a lane's A fragment is 8 consecutive channels (one ds_read_b128),
// the weights arrive fragment-major in bf16.  Wave tile 128 x (16*TN), block = 4 waves = 512 output positions.
// Each kernel row is padded to a multiple of 32 k-values with zero weights (the A side then reads finite neighbouring data).
#define SPAN16_BM 512
#define SPAN16_MAXRUN 10
// Position stride of the staged span in LDS.  With C = 64 (D.conv5's data gradient: 64 gradient channels) the plain stride is 128 bytes and
// the 16 positions of an A-fragment read fall on two alternating bank groups - 69 % of the kernel's LDS cycles were bank conflicts (PMC,
// rounds 1 and 2).  8 elements of padding per position make consecutive positions 144 bytes apart: 16 lanes x 16 bytes cover all banks
// once.  A k-step never straddles positions when 32 | C; its offset becomes tap * CP + channel offset.
#define SPAN16_PAD(C_) ((((C_) & ((C_) - 1)) == 0 && (C_) >= 64) ? 8 : 0)

struct Span16Args {
    const float* A;
    const __bf16* Wfrag;   // [KH*steps_per_seg][NT][64][8]
    const float* bias;
    const float* aux;
    float* out;
    int M, N, NT;
    int epi;
    float slope;
    int KH, KW, steps_per_seg;
    ConvGeom g;
    int a16;               // A is bf16 in memory (same layout): staged as is - half the bytes of the re-staging, no conversion
};

template <int TN>
__global__ __launch_bounds__(256, 2) void conv_span16_kernel(Span16Args p) {
    extern __shared__ __attribute__((aligned(16))) __bf16 span16[];
    __shared__ int run_gbase[SPAN16_MAXRUN];
    __shared__ int run_len[SPAN16_MAXRUN];
    __shared__ int run_off[SPAN16_MAXRUN + 1];
    __shared__ int posbase[SPAN16_BM];
    __shared__ int nrun_s;
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware ids (workgroup w runs on XCD w % 8): each XCD takes a contiguous run of position tiles, so the KH input rows that
    // consecutive tiles share are re-read from one L2 (scattered over the eight XCDs every tile fetched its rows from HBM: 11 x the input)
    const int mt = (int)(blockIdx.x & 7) * (int)((gridDim.x + 7) >> 3) + (int)(blockIdx.x >> 3);
    const int m0 = mt * SPAN16_BM;
    if (m0 >= p.M) return;
    const int mcount = min(SPAN16_BM, p.M - m0);
    const int CP = g.C + SPAN16_PAD(g.C), lgC = (CP == g.C) ? 30 : __ffs(g.C) - 1;
    const int halo = (p.KW - 1) * g.C, halo_p = (p.KW - 1) * CP;

    if (tid == 0) {
        int b, ho, wo;
        decode_m(m0, g, b, ho, wo);
        int left = mcount, r = 0, off = 0, done = 0;
        while (left > 0 && r < SPAN16_MAXRUN) {
            const int len = min(left, g.Wout - wo);
            run_gbase[r] = (int)((((size_t)b * g.H + ho + g.ih0) * g.W + wo + g.iw0) * g.C);
            run_len[r] = len;
            run_off[r] = off;
            for (int i = 0; i < len; ++i) posbase[done + i] = off + i * CP;
            off += len * CP + halo_p;
            done += len;
            left -= len;
            wo = 0;
            if (++ho == g.Hout) { ho = 0; ++b; }
            ++r;
        }
        run_off[r] = off;
        nrun_s = r;
        for (int i = done; i < SPAN16_BM; ++i) posbase[i] = 0;
    }
    __syncthreads();
    const int nrun = nrun_s;
    // k-padding slack after the last span is read with zero weights: it must hold finite values
    if (tid < 64) span16[run_off[nrun] + tid] = (__bf16)0.f;
    const int li = lane & 15, lg = lane >> 4;
    int pb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pb[i] = posbase[wave * 128 + i * 16 + li] + 8 * lg;

    f32x4 acc[8][TN];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The weight fragments come straight from L2 (every wave of every workgroup reads the same stream, 16 B per lane and
    // fragment): a 4-deep register ring keeps the loads three k-steps (~1500 MFMA cycles per wave) ahead of their use, which is
    // what it takes to cover the L2 round trip with only two waves per SIMD.  The ring runs across kernel-row boundaries.
    const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.Wfrag) + lane;
    const int KS = p.KH * p.steps_per_seg;
    bf16x8 bq[4][TN];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int j = 0; j < TN; ++j) bq[u][j] = wf[((size_t)min(u, KS - 1) * p.NT + j) * 64];
    int s = 0, kh = 0;
    for (int ks0 = 0; ks0 < KS; ks0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ks = ks0 + u;
            if (ks < KS) {
                if (s == 0) {                                  // new kernel row: stage its input span (bf16) in LDS
                    __syncthreads();
                    for (int r = 0; r < nrun; ++r) {
                        const int nfl = run_len[r] * g.C + halo;          // multiple of 8
                        const float* src = p.A + (size_t)run_gbase[r] + (size_t)kh * g.segstride;
                        __bf16* dst = span16 + run_off[r];
                        if (p.a16) {
                            const __bf16* src16 = reinterpret_cast<const __bf16*>(p.A) + (size_t)run_gbase[r] + (size_t)kh * g.segstride;
                            for (int e = tid * 8; e < nfl; e += 2048)
                                *reinterpret_cast<bf16x8*>(dst + e + (e >> lgC) * (CP - g.C)) = *reinterpret_cast<const bf16x8*>(src16 + e);
                        } else
                        for (int e = tid * 8; e < nfl; e += 2048) {
                            const float4 a = *reinterpret_cast<const float4*>(src + e);
                            const float4 c = *reinterpret_cast<const float4*>(src + e + 4);
                            bf16x8 v;
                            v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
                            v[4] = (__bf16)c.x; v[5] = (__bf16)c.y; v[6] = (__bf16)c.z; v[7] = (__bf16)c.w;
                            *reinterpret_cast<bf16x8*>(dst + e + (e >> lgC) * (CP - g.C)) = v;
                        }
                    }
                    __syncthreads();
                }
                if (ks + 3 < KS) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) bq[(u + 3) & 3][j] = wf[((size_t)(ks + 3) * p.NT + j) * 64];
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x8 af[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(span16 + pb[4 * h + i] + s * 32 + ((s * 32) >> lgC) * (CP - g.C));
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[4 * h + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bq[u][j], acc[4 * h + i][j], 0, 0, 0);
                }
                if (++s == p.steps_per_seg) { s = 0; ++kh; }
            }
        }
    }

#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = m0 + wave * 128 + i * 16 + 4 * lg + reg;
            if (m >= p.M) continue;
            int b, ho, wo;
            decode_m(m, g, b, ho, wo);
            const size_t o_off = (((size_t)b * g.OH + ho + g.oh0) * g.OW + wo + g.ow0) * g.OC;
            const size_t x_off = (size_t)m * g.OC;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = j * 16 + li;
                if (n >= p.N) continue;
                float v = acc[i][j][reg];
                switch (p.epi) {
                    case EPI_BIAS: v += p.bias[n]; break;
                    case EPI_BIAS_LRELU: v += p.bias[n]; v = v > 0.f ? v : p.slope * v; break;
                    case EPI_MASK_LRELU_GRAD: v = p.aux[x_off + n] > 0.f ? v : p.slope * v; break;
                    case EPI_BIAS_EXPTANH: v += p.bias[n]; v = expf(3.2f * tanhf(v)); break;
                    default: break;
                }
                p.out[o_off + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ 2-D tile conv, bf16 operands
// conv_span16_kernel re-stages every input row once per kernel row and lets each wave stream the weights from L2: at bf16 MFMA
// rates that L2->CU traffic (7 MB per CU and launch for D.conv5), not the matrix pipe, bounds it.  Here a workgroup owns a 2-D
// tile of TH x 64 output positions and keeps the tile's whole input halo ((TH+KH-1) x (64+KW-1) x C, bf16) resident in LDS:
// every input element is fetched once per tile (2.25x redundancy instead of 9x) - the first TH rows before the k loop, row
// TH+kh while kernel row kh is being multiplied.  The weight fragments are shared through LDS in chunks of SB k-steps
// (SB divides the steps of a kernel row): the next chunk's global loads are issued before the chunk's MFMA loop and only
// consumed after it, so the loop itself contains nothing but ds_read_b128 and MFMA - no global-load wait can stall it.
// Position tile (row i, column tile c) belongs to wave (i + c) & 3, so edge tiles (rows >= Hout, columns >= Wout) drop out
// evenly; their MFMAs are skipped.
struct Tile16Args {
    const float* A;
    const __bf16* Wfrag;   // [KH*steps_per_seg][NT][64][8]
    const float* bias;
    const float* aux;
    float* out;
    int N, NT;
    int epi;
    float slope;
    int KH, KW, steps_per_seg, SB;
    ConvGeom g;
    long long* dbg;        // optional per-phase clock totals of workgroup 0 (benchmark harness only)
    int ntiles;            // tile columns x tile rows x utterances (conv1d strip kernel: N chunks per strip that have their own workgroup)
    int SBH;               // conv1d strip kernel: utterances x rows
    int ncl;               // conv1d strip kernel: N chunks a workgroup walks itself over its staged strip (1 or N / 64)
};
__device__ __forceinline__ int g_wout(const Tile16Args& p) { return p.g.Wout; }
__device__ __forceinline__ int g_hout(const Tile16Args& p) { return p.g.Hout; }
#define TILE16_TW 64
#define TILE16_SBMAX 9

template <int TN, int TH>
__global__ __launch_bounds__(256) void conv_tile16_kernel(Tile16Args p) {
    extern __shared__ __attribute__((aligned(16))) __bf16 halo[];    // [TH+KH-1][RS] + 64 slack, then the weight chunk [SB][TN][64][8]
#ifdef T16_PROF
    const long long t16_wstart = wall_clock64();
#endif
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    // 1-D grid, XCD-aware: workgroup w runs on XCD w % 8; each XCD takes a contiguous run of tiles (tile columns fastest, then tile
    // rows, then utterances), so the tiles that share halo rows / columns are neighbours in one L2 instead of strangers in eight
    int wo0, ho0, b;
    {
        const int ntw = (g_wout(p) + TILE16_TW - 1) / TILE16_TW, nth = (g_hout(p) + TH - 1) / TH;
        const int per = (int)((gridDim.x + 7) >> 3);
        const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
        if (tile >= p.ntiles) return;
        const int tw_i = tile % ntw, rem = tile / ntw;
        wo0 = tw_i * TILE16_TW; ho0 = (rem % nth) * TH; b = rem / nth;
    }
    const int wcols = TILE16_TW + p.KW - 1;            // halo columns
    const int RS = wcols * g.C;                        // halo row stride (elements)
    const int nrows = TH + p.KH - 1;
    const int hi0 = ho0 + g.ih0, wi0 = wo0 + g.iw0;    // window origin in the input buffer
    const int vcols = max(0, min(wcols, g.W - wi0));   // columns that exist in the input
    const float* abase = p.A + (((size_t)b * g.H + hi0) * g.W + wi0) * g.C;
    __bf16* wbuf = halo + nrows * RS + 64;
    const int SB = p.SB, cps = p.steps_per_seg / SB, nchunk = p.KH * cps;
    const int cfrag = SB * TN * 64;                    // 16-byte fragments per chunk

    // stage halo row r (float32 -> bf16); elements beyond the input are zero
    auto stage_row = [&](int r) {
        const bool rin = hi0 + r < g.H;
        const float* src = abase + (size_t)r * g.W * g.C;
        __bf16* dst = halo + r * RS;
        const int nval = rin ? vcols * g.C : 0;        // multiple of 8
        for (int e = tid * 8; e < RS; e += 2048) {
            bf16x8 v;
            if (e < nval) {
                const float4 a = *reinterpret_cast<const float4*>(src + e);
                const float4 c = *reinterpret_cast<const float4*>(src + e + 4);
                v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
                v[4] = (__bf16)c.x; v[5] = (__bf16)c.y; v[6] = (__bf16)c.z; v[7] = (__bf16)c.w;
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (__bf16)0.f;
            }
            *reinterpret_cast<bf16x8*>(dst + e) = v;
        }
    };
    for (int r = 0; r < TH; ++r) stage_row(r);
    // k-padding reads run up to 31 elements past a position's window: into the slack behind the last row, and into the head of
    // the next halo row, which may not have arrived yet (zero weights there, but the data must be finite)
    if (tid < 64)
        for (int r = TH; r <= nrows; ++r) halo[r * RS + tid] = (__bf16)0.f;

    constexpr int NBR = (TILE16_SBMAX * TN + 3) / 4;   // 16-byte weight fragments per thread and chunk
    bf16x8 breg[NBR];
    const int nq = (cfrag + 255) >> 8;                 // groups of 256 fragments per chunk (the last one may run into the next chunk / slack)
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(p.Wfrag) + tid;
    auto wload = [&]() {                               // loads the chunk at wp, then advances wp
#pragma unroll
        for (int q = 0; q < NBR; ++q)
            if (q < nq) breg[q] = wp[256 * q];
        wp += cfrag;
    };
    auto wstore = [&]() {
#pragma unroll
        for (int q = 0; q < NBR; ++q)
            if (q < nq) reinterpret_cast<bf16x8*>(wbuf)[tid + 256 * q] = breg[q];
    };
    wload();
    wstore();

    // this wave's position tiles: row i, column tile (wave - i) & 3
    int pb[TH];
    unsigned valid = 0;
#pragma unroll
    for (int i = 0; i < TH; ++i) {
        const int c = (wave - i) & 3;
        pb[i] = i * RS + (16 * c + li) * g.C + 8 * lg;
        if (ho0 + i < g.Hout && wo0 + 16 * c < g.Wout) valid |= 1u << i;
    }
    f32x4 acc[TH][TN];
#pragma unroll
    for (int i = 0; i < TH; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // the next halo row travels through 4 float4 registers per thread (2 groups of 8 elements: covers RS <= 4096); plain scalars,
    // unconditional loads from clamped addresses + select: an indexed array here ends up in scratch with a vmcnt(0) behind it
    float4 r0a, r0c, r1a, r1c;
    r0a = r0c = r1a = r1c = make_float4(0.f, 0.f, 0.f, 0.f);
    const int e0 = tid * 8, e1 = tid * 8 + 2048;
    int rnval = 0;
    const bf16x8* wl = reinterpret_cast<const bf16x8*>(wbuf) + lane;
#ifdef T16_PROF
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tq0 = clock64(), tq1;
    const long long tw0 = wall_clock64(), tc0 = tq0;
    if (p.dbg && tid == 0 && blockIdx.x == 8) p.dbg[8] = tw0 - t16_wstart;
#define T16_T(j) do { tq1 = clock64(); tacc[j] += tq1 - tq0; tq0 = tq1; } while (0)
#else
#define T16_T(j)
#endif
    T16_T(0);
    auto mainloop = [&](auto all_valid) {
        for (int c = 0; c < nchunk; ++c) {
            const int kh = c / cps, s0 = (c - kh * cps) * SB;
            const bool seg_first = (s0 == 0), seg_last = (s0 + SB == p.steps_per_seg);
            if (c + 1 < nchunk) wload();
            if (seg_first && kh + 1 < p.KH) {          // issue the loads of halo row TH + kh
                const int r = TH + kh;
                const bool rin = hi0 + r < g.H;
                const float* src = rin ? abase + (size_t)r * g.W * g.C : abase;
                const int nval = rin ? vcols * g.C : 0;
                const int emax = max(vcols * g.C - 8, 0);
                const float* s0p = src + min(e0, emax);
                const float* s1p = src + min(e1, emax);
                r0a = *reinterpret_cast<const float4*>(s0p); r0c = *reinterpret_cast<const float4*>(s0p + 4);   // raw: the zero
                r1a = *reinterpret_cast<const float4*>(s1p); r1c = *reinterpret_cast<const float4*>(s1p + 4);   // select happens at publish time
                rnval = nval;
            }
            T16_T(1);
            // ---- SB k-steps from LDS only.  Fragments are double buffered in registers: the ds_reads of step u+1 are issued
            // before the 8*TN MFMAs of step u (sched_barrier keeps the compiler from sinking them back next to their use).
            const __bf16* hk = halo + kh * RS + s0 * 32;
            auto ldfrag = [&](int u, bf16x8 (&af)[TH], bf16x8 (&bfr)[TN]) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = wl[(u * TN + j) * 64];
#pragma unroll
                for (int i = 0; i < TH; ++i) af[i] = *reinterpret_cast<const bf16x8*>(hk + pb[i] + u * 32);
            };
            auto mm = [&](const bf16x8 (&af)[TH], const bf16x8 (&bfr)[TN]) {
#pragma unroll
                for (int i = 0; i < TH; ++i) {
                    if (decltype(all_valid)::value || (valid & (1u << i))) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    }
                }
            };
            bf16x8 a0[TH], b0[TN], a1[TH], b1[TN];
            ldfrag(0, a0, b0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            int u = 0;
            // One wave per SIMD: while a wave sits in its 8*TN back-to-back MFMA issues nothing else of it can issue, and while it
            // issues the next step's ds_reads the matrix pipe drains.  The sched_group_barrier pattern interleaves one ds_read (and
            // its address VALU op) after every two MFMAs, so the reads issue in the MFMAs' 16-cycle shadows.
#define T16_INTERLEAVE()                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < TH + TN; ++q_) {                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                \
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                \
    }                                                                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, TH * TN - 2 * (TH + TN) > 0 ? TH * TN - 2 * (TH + TN) : 0, 0)
            for (; u + 1 < SB; u += 2) {               // branch-free body: a conditional load would force lgkmcnt(0) at the join
                ldfrag(u + 1, a1, b1);
                mm(a0, b0);
                T16_INTERLEAVE();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0) AFTER the MFMAs were issued: the loads above had their shadow;
                __builtin_amdgcn_sched_barrier(0);     // without it the compiler puts the wait in front of the MFMAs (loop-header join)
                ldfrag(min(u + 2, SB - 1), a0, b0);
                mm(a1, b1);
                T16_INTERLEAVE();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef T16_INTERLEAVE
            if (u < SB) mm(a0, b0);
            T16_T(2);
            if (c + 1 < nchunk) {
                __syncthreads();                       // every wave is done with this weight chunk
                T16_T(3);
                wstore();
                if (seg_last && kh + 1 < p.KH) {       // publish halo row TH + kh for the next kernel row
                    __bf16* dst = halo + (TH + kh) * RS;
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!(e0 < rnval)) { r0a = z; r0c = z; }
                    if (!(e1 < rnval)) { r1a = z; r1c = z; }
                    if (e0 < RS) {
                        bf16x8 v;
                        v[0] = (__bf16)r0a.x; v[1] = (__bf16)r0a.y; v[2] = (__bf16)r0a.z; v[3] = (__bf16)r0a.w;
                        v[4] = (__bf16)r0c.x; v[5] = (__bf16)r0c.y; v[6] = (__bf16)r0c.z; v[7] = (__bf16)r0c.w;
                        *reinterpret_cast<bf16x8*>(dst + e0) = v;
                    }
                    if (e1 < RS) {
                        bf16x8 v;
                        v[0] = (__bf16)r1a.x; v[1] = (__bf16)r1a.y; v[2] = (__bf16)r1a.z; v[3] = (__bf16)r1a.w;
                        v[4] = (__bf16)r1c.x; v[5] = (__bf16)r1c.y; v[6] = (__bf16)r1c.z; v[7] = (__bf16)r1c.w;
                        *reinterpret_cast<bf16x8*>(dst + e1) = v;
                    }
                }
                T16_T(4);
                __syncthreads();
                T16_T(5);
            }
        }
    };
    if (valid == (1u << TH) - 1u) mainloop(std::true_type{});
    else mainloop(std::false_type{});
#ifdef T16_PROF
    if (p.dbg && tid == 0 && blockIdx.x == 8)
    { for (int j = 0; j < 6; ++j) p.dbg[j] = tacc[j]; p.dbg[6] = wall_clock64() - tw0; p.dbg[7] = clock64() - tc0; }
#endif

    // ---- epilogue: each position tile (16 positions x 16*TN channels) is transposed through this wave's private 4 KB of LDS
    // (the halo is dead now) so that the stores are float4 runs along the channels: 16 positions x OC floats are contiguous.
    __syncthreads();
    float* ep = reinterpret_cast<float*>(halo) + wave * (16 * 68);          // [16 positions][64 + 4] floats
    constexpr int NQ = TN * 4;                                            // float4 groups per position
#pragma unroll
    for (int i = 0; i < TH; ++i) {
        if (!(valid & (1u << i))) continue;
        const int c = (wave - i) & 3;
        const int ho = ho0 + i;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) ep[(4 * lg + reg) * 68 + 16 * j + li] = acc[i][j][reg];
        // lane -> (position row, float4 column); TN*4 float4 per position, 16 positions: TN passes of 64 lanes
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            const int idx = q * 64 + lane, pr = idx / NQ, cq = idx - pr * NQ;
            const int wo = wo0 + 16 * c + pr, n = 4 * cq;
            if (wo < g.Wout && n < p.N) {
                float4 v = *reinterpret_cast<const float4*>(&ep[pr * 68 + n]);
                const size_t o_off = (((size_t)b * g.OH + ho + g.oh0) * g.OW + wo + g.ow0) * g.OC + n;
                if (p.epi == EPI_BIAS || p.epi == EPI_BIAS_LRELU || p.epi == EPI_BIAS_EXPTANH) {
                    v.x += p.bias[n]; v.y += p.bias[n + 1]; v.z += p.bias[n + 2]; v.w += p.bias[n + 3];   // bias may be only 4-byte aligned
                }
                if (p.epi == EPI_BIAS_LRELU) {
                    v.x = v.x > 0.f ? v.x : p.slope * v.x; v.y = v.y > 0.f ? v.y : p.slope * v.y;
                    v.z = v.z > 0.f ? v.z : p.slope * v.z; v.w = v.w > 0.f ? v.w : p.slope * v.w;
                } else if (p.epi == EPI_MASK_LRELU_GRAD) {
                    const float4 x = *reinterpret_cast<const float4*>(p.aux + (((size_t)b * g.Hout + ho) * g.Wout + wo) * g.OC + n);
                    v.x = x.x > 0.f ? v.x : p.slope * v.x; v.y = x.y > 0.f ? v.y : p.slope * v.y;
                    v.z = x.z > 0.f ? v.z : p.slope * v.z; v.w = x.w > 0.f ? v.w : p.slope * v.w;
                } else if (p.epi == EPI_BIAS_EXPTANH) {
                    v.x = expf(3.2f * tanhf(v.x)); v.y = expf(3.2f * tanhf(v.y)); v.z = expf(3.2f * tanhf(v.z)); v.w = expf(3.2f * tanhf(v.w));
                }
                *reinterpret_cast<float4*>(p.out + o_off) = v;
            }
        }
    }
#ifdef T16_PROF
    if (p.dbg && tid == 0 && blockIdx.x == 8) p.dbg[9] = wall_clock64() - t16_wstart;
#endif
}

// ------------------------------------------------------------------------------------------ Conv1d / Linear tile kernel, bf16
// The generator's causal Conv1d layers (and any KH = 1 conv): M = B*T is small (8032 rows at B = 32), K = k*C is long (up to
// 1792) and N up to 256, so the generic GEMM ran 504 tiny workgroups with a barrier every 32 k-values (46 us per layer).  Here a
// workgroup owns 128 consecutive output positions x 64 output channels: the im2col rows of consecutive positions overlap
// (row t is the contiguous run inp[t*C .. t*C + k*C)), so the tile's WHOLE A operand is the contiguous strip of 128 + k - 1 input
// positions, staged once in LDS (bf16) and addressed Toeplitz-style; the weight fragments of the workgroup's 4 n-tiles stream
// through LDS in chunks of SB k-steps as in conv_tile16_kernel.  grid (ceil(Wout/128), N/64 chunks, B*Hout).
#define C1D_TW 128
// Position stride of the strip in LDS.  With C a multiple of 128 the plain stride (2 C bytes) is a multiple of the 256-byte bank cycle: the
// 16 lanes of an A-fragment read (16 consecutive positions, the same 16 bytes of each) all hit the same banks.  PAD elements per position
// shift consecutive positions; a k-step (32 consecutive k) never straddles positions when 32 | C, so its offset is tap * CP + channel offset.
#ifndef C1D_PADE
#define C1D_PADE 8
#endif
#define C1D_PAD(C_) (((C_) % 128 == 0) ? C1D_PADE : 0)
// -DC1_PROF: phase clocks of thread 0 of every workgroup (diagnostic build; tools/g_check.py prints them)
#ifdef C1_PROF
__device__ unsigned long long c1_prof[8];        // 0 strip staging, 1 first weights + barrier, 2 MFMA chunks, 3 chunk-end store + barrier, 4 epilogue, 5 workgroups
#define C1_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define C1_ACC(slot, a_, b_) do { c1_acc[slot] += (b_) - (a_); } while (0)
#else
#define C1_T(var) do {} while (0)
#define C1_ACC(slot, a_, b_) do {} while (0)
#endif
template <int TN>
__global__ __launch_bounds__(256) void conv1d_tile16_kernel(Tile16Args p) {
    C1_T(t_k0);
#ifdef C1_PROF
    unsigned long long c1_acc[5] = {0, 0, 0, 0, 0};
#endif
    extern __shared__ __attribute__((aligned(16))) __bf16 halo[];    // [RS] + 64 slack, then the weight chunk [SB][TN][64][8]
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    // 1-D grid, XCD-aware: the N-chunk workgroups of one input strip (they stage the same strip) get ids of one XCD
    int wo0, nb, bz;
    {
        const int nchN = p.ntiles;                        // N chunks per strip (host)
        const int gx = (g.Wout + C1D_TW - 1) / C1D_TW;
        const int w = (int)blockIdx.x, slot = w >> 3;
        const int strip = (slot / nchN) * 8 + (w & 7);
        nb = slot % nchN;
        if (strip >= gx * p.SBH) return;
        wo0 = (strip % gx) * C1D_TW; bz = strip / gx;
    }
    const int b = bz / g.Hout, ho = bz - b * g.Hout;
    const int wcols = C1D_TW + p.KW - 1;
    const int CP = g.C + (((g.C & (g.C - 1)) == 0) ? C1D_PAD(g.C) : 0);   // position stride in LDS (see C1D_PAD; padded only for powers of two)
    const int lgC = (CP == g.C) ? 30 : __ffs(g.C) - 1;
    const int RS = wcols * CP;
    const int wi0 = wo0 + g.iw0;
    const int vcols = max(0, min(wcols, g.W - wi0));
    const float* src = p.A + (((size_t)b * g.H + ho + g.ih0) * g.W + wi0) * g.C;
    __bf16* wbuf = halo + RS + 64;
    const int SB = p.SB, nchunk = p.steps_per_seg / SB;
    const int cfrag = SB * TN * 64;
    {   // stage the input strip (float32 -> bf16), zero beyond the input
        // groups of six 32-byte pieces per thread with all their loads in flight together: a load - convert - store loop paid a memory
        // latency per piece (17 pieces for a 256-channel strip: 16 k of the workgroup's 92 k clocks)
        const int nval = vcols * g.C, ntot = wcols * g.C;
        constexpr int SG = 6;
        for (int e0 = tid * 8; e0 < ntot; e0 += 2048 * SG) {
            float4 ra[SG], rc[SG];
            const int emax = max(nval - 8, 0);                  // (unconditional loads from clamped addresses: a test around a load makes the compiler wait for it at once)
#pragma unroll
            for (int q = 0; q < SG; ++q) {
                const int e = min(e0 + 2048 * q, emax);
                ra[q] = *reinterpret_cast<const float4*>(src + e); rc[q] = *reinterpret_cast<const float4*>(src + e + 4);
            }
#pragma unroll
            for (int q = 0; q < SG; ++q) {
                const int e = e0 + 2048 * q;
                if (e >= ntot) continue;
                bf16x8 v;
                if (e < nval) {
                    v[0] = (__bf16)ra[q].x; v[1] = (__bf16)ra[q].y; v[2] = (__bf16)ra[q].z; v[3] = (__bf16)ra[q].w;
                    v[4] = (__bf16)rc[q].x; v[5] = (__bf16)rc[q].y; v[6] = (__bf16)rc[q].z; v[7] = (__bf16)rc[q].w;
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = (__bf16)0.f;
                }
                const int px = (CP == g.C) ? 0 : e / g.C;
                *reinterpret_cast<bf16x8*>(halo + e + px * (CP - g.C)) = v;
            }
        }
        if (tid < 64) halo[RS + tid] = (__bf16)0.f;
    }
    // weight chunk c: steps [c SB, (c+1) SB) x this workgroup's TN n-tiles (stride NT tiles per step in the fragment stream)
    C1_T(t_k1);
    C1_ACC(0, t_k0, t_k1);
    // Weight chunks travel global -> registers -> LDS TWO chunks ahead (two register sets, two LDS slots): with one wave per SIMD and one
    // workgroup per CU nothing else covers a memory latency, and a chunk's MFMAs (SB x 2 x TN x 16 cycles) are shorter than one - fetched
    // one chunk ahead through one slot, every chunk ended in a wait for its successor and two barriers (122 us per layer at B = 256 for
    // 16 us of MFMA issue).  Chunk c + 2 is requested at the top of chunk c, chunk c + 1 is written to the other slot after chunk c's MFMAs,
    // ONE barrier per chunk.
    constexpr int NBR = (TILE16_SBMAX * TN + 3) / 4;
    bf16x8 bregA[NBR], bregB[NBR];
    const int nq = (cfrag + 255) >> 8;
    const int wsl = (cfrag * 8 + 2047) & ~2047;           // elements per LDS weight slot
    const bf16x8* wsrc = nullptr;
    int woff[NBR];                                       // fragment offsets inside a chunk: the same for every chunk
#pragma unroll
    for (int q = 0; q < NBR; ++q) {
        const int f = min(tid + 256 * q, cfrag - 1), u = f / (TN * 64), r = f - u * (TN * 64);
        woff[q] = u * p.NT * 64 + r;
    }
    auto wload = [&](int c, bf16x8 (&breg)[NBR]) {
        const bf16x8* wc = wsrc + (size_t)c * SB * p.NT * 64;
#pragma unroll
        for (int q = 0; q < NBR; ++q)
            if (q < nq) breg[q] = wc[woff[q]];
    };
    auto wstore = [&](int slot, const bf16x8 (&breg)[NBR]) {
#pragma unroll
        for (int q = 0; q < NBR; ++q)
            if (q < nq) reinterpret_cast<bf16x8*>(wbuf + slot * wsl)[tid + 256 * q] = breg[q];
    };
    // this wave's two position tiles: columns 16 (wave + 4 t)
    int pb[2];
    unsigned valid = 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = wave + 4 * t;
        pb[t] = (16 * c + li) * CP + 8 * lg;
        if (wo0 + 16 * c < g.Wout) valid |= 1u << t;
    }
    // Large batches (every CU has strips of its own): ONE workgroup per strip walks all N / 64 output-channel chunks over the strip it
    // staged instead of four workgroups staging the same strip (128 + k - 1 input positions x C channels, converted to bf16).  The
    // epilogue then transposes through its own LDS scratch (behind the weight chunk), not through the strip.
    const int nb_end = nb + p.ncl;
    float* epbase = (p.ncl > 1) ? reinterpret_cast<float*>(wbuf + 2 * wsl) : reinterpret_cast<float*>(halo);
    wsrc = reinterpret_cast<const bf16x8*>(p.Wfrag) + (size_t)nb * 4 * 64;
    wload(0, bregA);                                      // (the first two chunks of every later N chunk are requested under the epilogue before it)
    if (nchunk > 1) wload(1, bregB);
    for (; nb < nb_end; ++nb) {
    C1_T(t_n0);
    wstore(0, bregA);
    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    C1_T(t_n1);
    C1_ACC(1, t_n0, t_n1);
    auto chunk = [&](int c, bf16x8 (&bcur)[NBR], bf16x8 (&bnext)[NBR]) {   // bcur held chunk c (already in LDS), bnext holds chunk c + 1
        C1_T(t_c0);
        const bf16x8* wl = reinterpret_cast<const bf16x8*>(wbuf + (c & 1) * wsl) + lane;
        if (c + 2 < nchunk) wload(c + 2, bcur);
        const __bf16* hk = halo;
        const int ug0 = c * SB;                            // k-step ug: elements [32 ug, 32 ug + 32) of the im2col row
        // fragments double buffered in registers: the ds_reads of step u + 1 are issued before the MFMAs of step u (one wave per SIMD and
        // one workgroup per CU here: nothing else hides an LDS round trip per k-step)
        auto ldfrag = [&](int u, bf16x8 (&af)[2], bf16x8 (&bfr)[TN]) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = wl[(u * TN + j) * 64];
            const int k0 = (ug0 + u) * 32;
            const int koff = k0 + (k0 >> lgC) * (CP - g.C);     // = tap * CP + channel offset (C is a power of two whenever CP != C)
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(hk + pb[i] + koff);
        };
        auto mm = [&](const bf16x8 (&af)[2], const bf16x8 (&bfr)[TN]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (valid & (1u << i)) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            }
        };
        bf16x8 a0[2], b0[TN], a1[2], b1[TN];
        ldfrag(0, a0, b0);
        int u = 0;
        for (; u + 1 < SB; u += 2) {
            ldfrag(u + 1, a1, b1);
            mm(a0, b0);
            ldfrag(min(u + 2, SB - 1), a0, b0);
            mm(a1, b1);
        }
        if (u < SB) mm(a0, b0);
        C1_T(t_c1);
        C1_ACC(2, t_c0, t_c1);
        if (c + 1 < nchunk) {
            wstore((c + 1) & 1, bnext);                    // its slot was last read in chunk c - 1, behind that chunk's barrier
            __syncthreads();
        }
        C1_T(t_c2);
        C1_ACC(3, t_c1, t_c2);
    };
    for (int c = 0; c < nchunk; c += 2) {
        chunk(c, bregA, bregB);
        if (c + 1 < nchunk) chunk(c + 1, bregB, bregA);
    }
    // ---- epilogue through LDS (float4 stores along the channels)
    C1_T(t_e0);
    if (nb + 1 < nb_end) {                                 // every chunk of this N chunk has left the registers: the next one's first two come in now
        wsrc = reinterpret_cast<const bf16x8*>(p.Wfrag) + (size_t)(nb + 1) * 4 * 64;
        wload(0, bregA);
        if (nchunk > 1) wload(1, bregB);
    }
    __syncthreads();
    float* ep = epbase + wave * (16 * 68);
    constexpr int NQ = TN * 4;
    const int n0 = nb * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (!(valid & (1u << i))) continue;
        const int c = wave + 4 * i;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) ep[(4 * lg + reg) * 68 + 16 * j + li] = acc[i][j][reg];
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            const int idx = q * 64 + lane, pr = idx / NQ, cq = idx - pr * NQ;
            const int wo = wo0 + 16 * c + pr, n = n0 + 4 * cq;
            if (wo < g.Wout && n < p.N) {
                float4 v = *reinterpret_cast<const float4*>(&ep[pr * 68 + 4 * cq]);
                const size_t o_off = (((size_t)b * g.OH + ho + g.oh0) * g.OW + wo + g.ow0) * g.OC + n;
                if (p.epi == EPI_BIAS || p.epi == EPI_BIAS_LRELU || p.epi == EPI_BIAS_EXPTANH) {
                    v.x += p.bias[n]; v.y += p.bias[n + 1]; v.z += p.bias[n + 2]; v.w += p.bias[n + 3];
                }
                if (p.epi == EPI_BIAS_LRELU) {
                    v.x = v.x > 0.f ? v.x : p.slope * v.x; v.y = v.y > 0.f ? v.y : p.slope * v.y;
                    v.z = v.z > 0.f ? v.z : p.slope * v.z; v.w = v.w > 0.f ? v.w : p.slope * v.w;
                } else if (p.epi == EPI_MASK_LRELU_GRAD) {
                    const float4 x = *reinterpret_cast<const float4*>(p.aux + (((size_t)b * g.Hout + ho) * g.Wout + wo) * g.OC + n);
                    v.x = x.x > 0.f ? v.x : p.slope * v.x; v.y = x.y > 0.f ? v.y : p.slope * v.y;
                    v.z = x.z > 0.f ? v.z : p.slope * v.z; v.w = x.w > 0.f ? v.w : p.slope * v.w;
                } else if (p.epi == EPI_BIAS_EXPTANH) {
                    v.x = expf(3.2f * tanhf(v.x)); v.y = expf(3.2f * tanhf(v.y)); v.z = expf(3.2f * tanhf(v.z)); v.w = expf(3.2f * tanhf(v.w));
                }
                *reinterpret_cast<float4*>(p.out + o_off) = v;
            }
        }
    }
    if (nb + 1 < nb_end) __syncthreads();                  // the next chunk's first weights replace this chunk's last ones
    C1_T(t_e1);
    C1_ACC(4, t_e0, t_e1);
    }
#ifdef C1_PROF
    if (tid == 0) { for (int q_ = 0; q_ < 5; ++q_) atomicAdd(&c1_prof[q_], c1_acc[q_]); atomicAdd(&c1_prof[5], 1ull); }
#endif
}

// Wg [N][Ktot] f32 -> bf16 fragment-major [KH*sps][NT][64][8], sps = ceil(seglen/32); element (ks = kh*sps + s, j, lane, e) =
// Wg[j*16 + (lane&15)][kh*seglen + s*32 + 8*(lane>>4) + e], zero where n >= N or the in-segment index >= seglen.
__global__ void weight_frag16_kernel(const float* __restrict__ Wg, int N, int Ktot, int seglen, int KH, int NT, int sps,
                                     __bf16* __restrict__ Wfrag) {
    const int total = KH * sps * NT * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, t = idx >> 9, j = t % NT, ks = t / NT;
        const int kh = ks / sps, s_ = ks - kh * sps;
        const int n = j * 16 + (lane & 15), kin = s_ * 32 + 8 * (lane >> 4) + e;
        float v = 0.f;
        if (n < N && kin < seglen) v = Wg[(size_t)n * Ktot + (size_t)kh * seglen + kin];
        Wfrag[idx] = (__bf16)v;
    }
}

// ------------------------------------------------------------------------------------------ generic implicit GEMM, bf16 operands
// conv_gemm_kernel with bf16 MFMA operands (inputs rounded to bf16 while staged, float32 accumulation): K step 32 = one
// v_mfma_f32_16x16x32_bf16 per tile.  Used for the generator's Conv1d / Linear layers in bf16 mode.
#define G16_BK 32
#define G16_LD 40     // LDS row stride (bf16 elements): 80 B keeps the 16-byte fragment reads aligned and staggers banks
template <int TN, int TM>
__global__ __launch_bounds__(256, 2) void conv_gemm16_kernel(GemmArgs p) {
    constexpr int BN = 16 * TN;
    constexpr int BM = 64 * TM;
    constexpr int ALOADS = (BM * 8 + 255) / 256;    // float4 loads of the A tile per thread (BM rows x 8 quads)
    constexpr int BLOADS = (BN * 8 + 255) / 256;
    __shared__ __attribute__((aligned(16))) __bf16 As[2][BM * G16_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][64 * G16_LD];
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    const int kq = tid & 7;                          // k quad (4 floats) inside the 32-wide step
    size_t a_off[ALOADS];
    bool a_ok[ALOADS];
#pragma unroll
    for (int i = 0; i < ALOADS; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const int m = m0 + row;
        a_ok[i] = (row < BM) && (m < p.M);
        int b = 0, ho = 0, wo = 0;
        if (a_ok[i]) decode_m(m, g, b, ho, wo);
        a_off[i] = (((size_t)b * g.H + ho + g.ih0) * g.W + wo + g.iw0) * g.C;
    }
    bool b_ok[BLOADS];
    const float* wrow[BLOADS];
#pragma unroll
    for (int i = 0; i < BLOADS; ++i) {
        const int bn = (tid >> 3) + 32 * i;
        b_ok[i] = (bn < BN) && (n0 + bn < p.N);
        wrow[i] = p.Wg + (size_t)(n0 + bn) * g.Ktot;
    }
    int kk = 4 * kq, kh = 0, r = 4 * kq;
    while (r >= g.seglen) { r -= g.seglen; ++kh; }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsteps = (g.Ktot + G16_BK - 1) / G16_BK;
    float4 ra[ALOADS], rb[BLOADS];
    auto gload = [&]() {
        const bool kin = kk < g.Ktot;
        const size_t koff = (size_t)kh * g.segstride + r;
#pragma unroll
        for (int i = 0; i < ALOADS; ++i)
            ra[i] = (a_ok[i] && kin) ? *reinterpret_cast<const float4*>(p.A + a_off[i] + koff) : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < BLOADS; ++i)
            rb[i] = (b_ok[i] && kin) ? *reinterpret_cast<const float4*>(wrow[i] + kk) : make_float4(0, 0, 0, 0);
        kk += G16_BK;
        r += G16_BK;
        while (r >= g.seglen) { r -= g.seglen; ++kh; }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ALOADS; ++i) {
            const int row = (tid >> 3) + 32 * i;
            if (row < BM) {
                bf16x4 v;
                v[0] = (__bf16)ra[i].x; v[1] = (__bf16)ra[i].y; v[2] = (__bf16)ra[i].z; v[3] = (__bf16)ra[i].w;
                *reinterpret_cast<bf16x4*>(&As[buf][row * G16_LD + 4 * kq]) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < BLOADS; ++i) {
            const int bn = (tid >> 3) + 32 * i;
            if (bn < BN) {
                bf16x4 v;
                v[0] = (__bf16)rb[i].x; v[1] = (__bf16)rb[i].y; v[2] = (__bf16)rb[i].z; v[3] = (__bf16)rb[i].w;
                *reinterpret_cast<bf16x4*>(&Bs[buf][bn * G16_LD + 4 * kq]) = v;
            }
        }
    };

    gload();
    lstore(0);
    __syncthreads();
    const int li = lane & 15, lg = lane >> 4;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload();
        bf16x8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&As[buf][(wave * 16 * TM + i * 16 + li) * G16_LD + 8 * lg]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][(j * 16 + li) * G16_LD + 8 * lg]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        if (s + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = m0 + wave * 16 * TM + i * 16 + 4 * lg + reg;
            if (m >= p.M) continue;
            int b, ho, wo;
            decode_m(m, g, b, ho, wo);
            const size_t o_off = (((size_t)b * g.OH + ho + g.oh0) * g.OW + wo + g.ow0) * g.OC;
            const size_t x_off = (size_t)m * g.OC;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + j * 16 + li;
                if (n >= p.N) continue;
                float v = acc[i][j][reg];
                switch (p.epi) {
                    case EPI_BIAS: v += p.bias[n]; break;
                    case EPI_BIAS_LRELU: v += p.bias[n]; v = v > 0.f ? v : p.slope * v; break;
                    case EPI_MASK_LRELU_GRAD: v = p.aux[x_off + n] > 0.f ? v : p.slope * v; break;
                    case EPI_BIAS_EXPTANH: v += p.bias[n]; v = expf(3.2f * tanhf(v)); break;
                    default: break;
                }
                p.out[o_off + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ wgrad
struct WgradArgs {
    const float* A;      // forward input activations (geometry g: H,W,C,ih0,iw0,seglen,segstride,Ktot,Hout,Wout)
    const float* dOut;   // gradient wrt the layer output, buffer [B][OH][OW][OC] at offset (oh0,ow0)
    float* part;         // [splits][N][Ktot]
    float* bpart;        // [splits][N] (bias gradient partials) or null
    int M, N;
    int rows_per_split;  // multiple of 8
    ConvGeom g;
};

#define WG_BKK 128
#define WG_BN 64

#define WG_MS 16   // reduction rows per step (4 MFMA k-steps of 4)
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs p) {
    // row strides padded by 16 floats: the 4 k-rows read by one MFMA fragment load fall on disjoint banks
    constexpr int DSS = WG_BN + 16, AVS = WG_BKK + 16;
    __shared__ __attribute__((aligned(16))) float Ds[2][WG_MS * DSS];
    __shared__ __attribute__((aligned(16))) float Av[2][WG_MS * AVS];
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kk0 = blockIdx.x * WG_BKK, split = blockIdx.y, n0 = blockIdx.z * WG_BN;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    // Av loader: rows (tid>>5) and 8 + (tid>>5), quad tid&31 -> kk fixed for the whole loop
    const int arow = tid >> 5, akk = kk0 + 4 * (tid & 31);
    const bool a_kin = akk < g.Ktot;
    int kh = 0, r = akk;
    if (a_kin) { kh = akk / g.seglen; r = akk - kh * g.seglen; }
    const size_t a_koff = (size_t)kh * g.segstride + r;
    // Ds loader: row tid>>4 (0..15), quad tid&15
    const int drow = tid >> 4, dn = n0 + 4 * (tid & 15);
    const bool d_nin = dn < p.N;  // N is a multiple of 4 for every layer

    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    // positions of this thread's three loader rows, advanced by WG_MS per step (no divisions in the loop)
    int pb_[3], ph_[3], pw_[3];
    {
        const int rows[3] = {arow, arow + 8, drow};
#pragma unroll
        for (int q = 0; q < 3; ++q) decode_m(min(m_begin + rows[q], p.M - 1), g, pb_[q], ph_[q], pw_[q]);
    }
    float4 ra[2], rd;
    auto gload = [&](int mbase) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = mbase + arow + 8 * h;
            ra[h] = make_float4(0, 0, 0, 0);
            if (a_kin && m < m_end) {
                const size_t off = (((size_t)pb_[h] * g.H + ph_[h] + g.ih0) * g.W + pw_[h] + g.iw0) * g.C;
                ra[h] = *reinterpret_cast<const float4*>(p.A + off + a_koff);
            }
        }
        {
            const int m = mbase + drow;
            rd = make_float4(0, 0, 0, 0);
            if (d_nin && m < m_end) {
                const size_t off = (((size_t)pb_[2] * g.OH + ph_[2] + g.oh0) * g.OW + pw_[2] + g.ow0) * g.OC;
                rd = *reinterpret_cast<const float4*>(p.dOut + off + dn);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            pw_[q] += WG_MS;
            while (pw_[q] >= g.Wout) { pw_[q] -= g.Wout; if (++ph_[q] == g.Hout) { ph_[q] = 0; ++pb_[q]; } }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<float4*>(&Av[buf][(arow + 8 * h) * AVS + 4 * (tid & 31)]) = ra[h];
        *reinterpret_cast<float4*>(&Ds[buf][drow * DSS + 4 * (tid & 15)]) = rd;
    };

    const int li = lane & 15, lg = lane >> 4;
    if (m_begin < m_end) {
        gload(m_begin);
        lstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (int mb = m_begin; mb < m_end; mb += WG_MS, buf ^= 1) {
        const bool more = mb + WG_MS < m_end;
        if (more) gload(mb + WG_MS);
#pragma unroll
        for (int s = 0; s < WG_MS / 4; ++s) {
            float af[4], bf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = Ds[buf][(4 * s + lg) * DSS + i * 16 + li];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Av[buf][(4 * s + lg) * AVS + wave * 32 + j * 16 + li];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (p.bpart && blockIdx.x == 0 && tid < WG_BN) {
#pragma unroll
            for (int rr = 0; rr < WG_MS; ++rr) bsum += Ds[buf][rr * DSS + tid];
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
    }
    // acc[i][j][reg]: n = n0 + 16 i + 4 lg + reg, kk = kk0 + 32 wave + 16 j + li
    float* part = p.part + (size_t)split * p.N * g.Ktot;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int n = n0 + i * 16 + 4 * lg + reg;
            if (n >= p.N) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kk = kk0 + wave * 32 + j * 16 + li;
                if (kk < g.Ktot) part[(size_t)n * g.Ktot + kk] = acc[i][j][reg];
            }
        }
    if (p.bpart && blockIdx.x == 0 && tid < WG_BN && n0 + tid < p.N) p.bpart[(size_t)split * p.N + n0 + tid] = bsum;
}

// bf16-operand weight gradient (v_mfma_f32_16x16x32_bf16, f32 accumulate).  Both MFMA operands want the reduction
// index m contiguous per lane while memory has n / kk contiguous; the tiles are therefore staged row-major as loaded
// ([m][n], one 8-byte LDS store per float4) and read back with the hardware transpose read ds_read_b64_tr_b16
// (per 16-lane group: a 4-row x 16-column block, lane i receives column i): two reads = the 8 consecutive m of a lane's
// fragment.  32 reduction rows per step.
#define WG16_MS 32
#define WG16_LDD (WG_BN + 16)    // bf16 elements per LDS row of the dOut tile (80: see tr_frag)
#define WG16_LDA (WG_BKK + 16)   // ... of the A_view tile (144)

// MFMA operand fragment (16 columns x 32 rows of a row-major [row][column] tile, rows = the reduction index) by the hardware transpose
// read: two ds_read_b64_tr_b16, each lane addressing 4 contiguous elements of one row.  Which rows a lane group takes only permutes the
// reduction index - the same permutation for both operands - so the rows are dealt for the LDS banks: a 32-lane half of one read
// covers EIGHT CONSECUTIVE rows (lanes 16 g .. 16 g + 15: rows 4 (g & 1) + 16 (g >> 1) + 0 .. 3, then + 8 for the second read), which
// tile the 64 banks exactly when the row stride is 16, 48, 80 or 112 elements modulo 128.  (With rows 8 g + 0 .. 3 per group - 0 .. 3 and
// 8 .. 11 in one half - no stride is conflict-free: PMC showed 43 % of the LDS cycles of D.conv5's weight gradient as bank conflicts.)
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int ld, int col0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    const __bf16* a0 = tile + (4 * (g & 1) + 16 * (g >> 1) + q) * ld + col0 + 4 * pq;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 8 * ld));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

__global__ __launch_bounds__(256, 2) void conv_wgrad16_kernel(WgradArgs p) {
    __shared__ __attribute__((aligned(16))) __bf16 Ds16[2][WG16_MS * WG16_LDD];
    __shared__ __attribute__((aligned(16))) __bf16 Av16[2][WG16_MS * WG16_LDA];
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kk0 = blockIdx.x * WG_BKK, split = blockIdx.y, n0 = blockIdx.z * WG_BN;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    const int arow = tid >> 5, akq = tid & 31, akk = kk0 + 4 * akq;      // Av loader: rows arow + 8h (h < 4), kk quad akq
    const bool a_kin = akk < g.Ktot;
    int kh = 0, r = akk;
    if (a_kin) { kh = akk / g.seglen; r = akk - kh * g.seglen; }
    const size_t a_koff = (size_t)kh * g.segstride + r;
    const int drow = tid >> 4, dnq = tid & 15, dn = n0 + 4 * dnq;          // Ds loader: rows drow + 16h (h < 2), n quad dnq
    const bool d_nin = dn < p.N;

    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int pb_[6], ph_[6], pw_[6];
    {
        const int rows[6] = {arow, arow + 8, arow + 16, arow + 24, drow, drow + 16};
#pragma unroll
        for (int q = 0; q < 6; ++q) decode_m(min(m_begin + rows[q], p.M - 1), g, pb_[q], ph_[q], pw_[q]);
    }
    float4 ra[4], rd[2];
    auto gload = [&](int mbase) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int m = mbase + arow + 8 * h;
            ra[h] = make_float4(0, 0, 0, 0);
            if (a_kin && m < m_end) {
                const size_t off = (((size_t)pb_[h] * g.H + ph_[h] + g.ih0) * g.W + pw_[h] + g.iw0) * g.C;
                ra[h] = *reinterpret_cast<const float4*>(p.A + off + a_koff);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = mbase + drow + 16 * h;
            rd[h] = make_float4(0, 0, 0, 0);
            if (d_nin && m < m_end) {
                const size_t off = (((size_t)pb_[4 + h] * g.OH + ph_[4 + h] + g.oh0) * g.OW + pw_[4 + h] + g.ow0) * g.OC;
                rd[h] = *reinterpret_cast<const float4*>(p.dOut + off + dn);
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            pw_[q] += WG16_MS;
            while (pw_[q] >= g.Wout) { pw_[q] -= g.Wout; if (++ph_[q] == g.Hout) { ph_[q] = 0; ++pb_[q]; } }
        }
    };
    float bcol = 0.f;   // bias gradient: this thread's partial of column dn..dn+3 is folded at the end (threads with equal dnq)
    float bq[4] = {0.f, 0.f, 0.f, 0.f};
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            bf16x4 v;
            v[0] = (__bf16)ra[h].x; v[1] = (__bf16)ra[h].y; v[2] = (__bf16)ra[h].z; v[3] = (__bf16)ra[h].w;
            *reinterpret_cast<bf16x4*>(&Av16[buf][(arow + 8 * h) * WG16_LDA + 4 * akq]) = v;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x4 v;
            v[0] = (__bf16)rd[h].x; v[1] = (__bf16)rd[h].y; v[2] = (__bf16)rd[h].z; v[3] = (__bf16)rd[h].w;
            *reinterpret_cast<bf16x4*>(&Ds16[buf][(drow + 16 * h) * WG16_LDD + 4 * dnq]) = v;
            bq[0] += rd[h].x; bq[1] += rd[h].y; bq[2] += rd[h].z; bq[3] += rd[h].w;     // float32 column sums for the bias gradient
        }
    };
    (void)bcol;

    if (m_begin < m_end) {
        gload(m_begin);
        lstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (int mb = m_begin; mb < m_end; mb += WG16_MS, buf ^= 1) {
        const bool more = mb + WG16_MS < m_end;
        if (more) gload(mb + WG16_MS);
        bf16x8 af[4], bf[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = tr_frag(Ds16[buf], WG16_LDD, i * 16, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = tr_frag(Av16[buf], WG16_LDA, wave * 32 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        if (more) lstore(buf ^ 1);
        __syncthreads();
    }
    const int li = lane & 15, lg = lane >> 4;
    float* part = p.part + (size_t)split * p.N * g.Ktot;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int n = n0 + i * 16 + 4 * lg + reg;
            if (n >= p.N) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kk = kk0 + wave * 32 + j * 16 + li;
                if (kk < g.Ktot) part[(size_t)n * g.Ktot + kk] = acc[i][j][reg];
            }
        }
    // bias gradient: fold the 16 row-lanes (tid>>4) of each n quad through LDS (reuse Ds16 as float scratch)
    if (p.bpart && blockIdx.x == 0) {
        float* scr = reinterpret_cast<float*>(&Ds16[0][0]);     // 256 threads x 4 floats = 4 KB <= sizeof(Ds16[0])
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) scr[(tid >> 4) * 64 + 4 * dnq + q] = bq[q];
        __syncthreads();
        if (tid < WG_BN && n0 + tid < p.N) {
            float sacc = 0.f;
            for (int rr = 0; rr < 16; ++rr) sacc += scr[rr * 64 + tid];
            p.bpart[(size_t)split * p.N + n0 + tid] = sacc;
        }
    }
}

// ------------------------------------------------------------------------------------------ weight gradient, 2-D tiles, bf16
// conv_wgrad16_kernel gathers the im2col view: every input element is read KH*KW times per launch (5 GB for D.conv5 at B = 32)
// and the kernel runs at the memory system's speed.  Here a workgroup owns ONE kernel row kh and accumulates
// dW[n][kh][kw][c] = sum_pos dOut[pos][n] * X[pos + (kh, kw)][c] for all (kw, c) of that row in registers while it walks over
// position tiles of 4 x 64 outputs: per tile it stages the 4 input rows it needs (shifted by kh, 64 + KW - 1 columns, bf16) and
// the dOut tile ([256 positions][N], bf16) in LDS.  The (kw, c) index is contiguous in the staged rows (Toeplitz view, row
// stride C), so both MFMA operands come from the hardware transpose read exactly as in conv_wgrad16_kernel, and the input is
// read KH * 1.1 times instead of KH*KW times.  Partials per workgroup group, reduced by wgrad_reduce_kernel.
#define WT_TH 4
#define WT_TW 64
// LDS row stride (bf16) of the dOut tile for NT n-tiles: 16, 48 or 80 - conflict-free for tr_frag, and no wider than the channels need
// (a 72-element row for every layer kept D.conv3's weight gradient at 2 workgroups per CU)
#define WT_NP_OF(NT_) ((NT_) <= 1 ? 16 : (NT_) <= 3 ? 48 : 80)
#define WT_SLACK 512          // elements between the halo image and the dOut tile: Toeplitz over-read of the last row, overrun of the last 1 KB DMA piece

// -DWT_PROF: phase clocks of thread 0 of every workgroup (diagnostic build: tools/variants.sh dense prof:"-DWT_PROF"; tools/d_check.py prints them)
#ifdef WT_PROF
__device__ unsigned long long wt_prof[8];        // 0 loads issued, 1 load wait + LDS writes, 2 barrier, 3 MFMA loop, 4 top barrier, 5 tiles
#define WT_T(var) const unsigned long long var = __builtin_readcyclecounter()
#ifndef WT_PROF_NT
#define WT_PROF_NT 4       // which kernel (n tiles) reports
#endif
#else
#define WT_T(var) do {} while (0)
#endif

struct WgradTileArgs {
    const float* A;
    const float* dOut;
    float* part;         // [G][N][Ktot]
    float* bpart;        // [G][N] or null
    int N, B, KH, KW;
    int nth, ntw, ntiles, G;
    ConvGeom g;
    int d16;             // dOut is bf16 in memory
    // Sliced problems (layers too wide for one workgroup's accumulators: the generator's 256 -> 256 Conv1d).  A workgroup then owns a
    // chunk of p.N output channels and a slice of g.C input channels (g describes the SLICE: C, seglen = KW * C); ics is the position
    // stride of the input buffer (the layer's full channel count), subs_n x subs_c sub-problems share every group's tiles, Kfull = KH * KW * ics
    // is the row length of the partials.  Unsliced: ics = g.C, subs_n = subs_c = 1, Kfull = g.Ktot.
    int ics, subs_n, subs_c, Kfull;
};

template <int NT, int KTW, bool D16 = false, bool A16 = false>      // n tiles (16 each), kk tiles per wave (16 each; tile index = wave + 4 * jj); D16 / A16: dOut / the input activation is bf16 in memory
__global__ __launch_bounds__(256, 2) void conv_wgrad_tile16_kernel(WgradTileArgs p) {
    static_assert(!A16 || D16, "a bf16 activation comes with a bf16 output gradient");
    constexpr int WT_NP = WT_NP_OF(NT);
    extern __shared__ __attribute__((aligned(16))) __bf16 wt_lds[];     // halo rows [WT_TH][RS] + WT_SLACK, then dOut tile [256][WT_NP]
    __shared__ float bred[16][64];
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform: the tile tests below are scalar branches
    // grid: 1-D, KH * G workgroups.  Workgroup w runs on XCD w % 8 (round-robin dispatch), and the KH kernel-row workgroups of a group
    // read the same input / dOut tiles at about the same time: they are given ids of one XCD so that its L2 serves the repeats
    // (with the natural (kh, group) order the KH readers of a tile sat on different XCDs and every read went to HBM: 1.55 GB per launch
    // for 160 MB of operands)
    int kh, grp, sub;
    {
        const int inner = p.KH * p.subs_n * p.subs_c;      // the workgroups that read one group's tiles: ids of one XCD
        int rest;
        if ((p.G & 7) == 0) {
            const int w = blockIdx.x, xcd = w & 7, slot = w >> 3;
            rest = slot % inner;
            grp = (slot / inner) * 8 + xcd;
        } else {
            rest = blockIdx.x % inner;
            grp = blockIdx.x / inner;
        }
        kh = rest % p.KH;
        sub = rest / p.KH;
    }
    const int sn = sub / p.subs_c, sc = sub - sn * p.subs_c;
    const int n0 = sn * p.N, ic0 = sc * g.C, Nfull = p.N * p.subs_n;
    const int wcols = WT_TW + p.KW - 1, RS = wcols * g.C;
    __bf16* halo = wt_lds;
    __bf16* dt = wt_lds + WT_TH * RS + WT_SLACK;
    const int nkt = (g.seglen + 15) >> 4;               // kk tiles of this kernel row
    f32x4 acc[NT][KTW];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int jj = 0; jj < KTW; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < WT_SLACK; e += 256) halo[WT_TH * RS + e] = (__bf16)0.f;    // (the last DMA piece may run into the slack)
    const int dnq = tid & 15, dp0 = tid >> 4;            // dOut staging: n quad, first position
    float bq[4] = {0.f, 0.f, 0.f, 0.f};
    const bool want_bias = (p.bpart != nullptr) && (kh == 0) && (sc == 0);
#ifdef WT_PROF
    unsigned long long pacc[6] = {0, 0, 0, 0, 0, 0};
#endif

    for (int t = grp; t < p.ntiles; t += p.G) {
        const int tw_i = t % p.ntw, rem = t / p.ntw, th_i = rem % p.nth, b = rem / p.nth;
        const int ho0 = th_i * WT_TH, wo0 = tw_i * WT_TW;
        const int wi0 = wo0 + g.iw0;
        const int vcols = max(0, min(wcols, g.W - wi0));
        WT_T(t_a);
        __syncthreads();                                 // previous tile fully consumed
        WT_T(t_b);
        // ---- input rows ho0 + r + kh (float32 -> bf16, or bf16 as stored), zero outside the input
        // bf16 input: the four halo rows go global -> LDS by DMA (1 KB pieces of the contiguous [4][RS] image, wave w moves pieces w, w + 4,
        // ..; lane l of a piece supplies the 16 bytes at image element 512 k + 8 l), issued ahead of the dOut loads below: ONE memory
        // latency per tile and no staging registers (a row-by-row load / store loop paid a latency per load: 9.5 k of the 18 k clocks of
        // a conv5 tile visit).  Rows / columns outside the input re-read the last valid ones: they only meet output positions outside
        // the output, whose dOut is staged as zero, and reduction columns >= seglen, which are never stored.
        if (A16) {
            // (sliced problems - the generator's layers in bf16 mode, round 5: a position's 64-channel slice sits at position * ics + ic0)
            const int emax = max(vcols * g.C - 8, 0), himax = g.H - 1;
            const __bf16* src0 = reinterpret_cast<const __bf16*>(p.A) + (((size_t)b * g.H) * g.W + wi0) * p.ics + ic0;
            const int npc = (WT_TH * RS + 511) >> 9;
            for (int k = wave; k < npc; k += 4) {
                const int ef = min(512 * k + 8 * lane, WT_TH * RS - 8);
                const int r = (ef >= RS) + (ef >= 2 * RS) + (ef >= 3 * RS), e = min(ef - r * RS, emax);
                const int hi = min(ho0 + r + kh + g.ih0, himax);
                const int pos = e / g.C, within = e - pos * g.C;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + ((size_t)hi * g.W + pos) * p.ics + within),
                                                 (__attribute__((address_space(3))) void*)(halo + 512 * k), 16, 0, 0);
            }
        }
        if (!A16) {
            // float32 input (the generator's layers, D's first layers in float32-buffer mode): pieces of 8 elements of the [4][RS] image,
            // loads unconditional from clamped addresses and all in flight together, then converted and stored; a piece sits inside one
            // position's channel slice (C is a multiple of 8), whose address in the buffer is position * ics + ic0
            int tl = tid;                                  // laundered per tile: the piece addresses are tile-invariant and the compiler would
            asm volatile("" : "+v"(tl));                   // otherwise keep all of them in registers across the MFMA loop (187 VGPRs for the 32-channel layer)
            constexpr int HNF = 9, HB = 3;                 // pieces per thread, in batches of HB (registers: 8 floats per piece in flight)
            const int pmax = max(vcols - 1, 0), himax = g.H - 1;
            const float* src0 = p.A + (((size_t)b * g.H) * g.W + wi0) * p.ics + ic0;
#pragma unroll
            for (int it0 = 0; it0 < HNF; it0 += HB) {
                float4 fa[HB], fc[HB];
#pragma unroll
                for (int q = 0; q < HB; ++q) {
                    const int ef = min(tl * 8 + 2048 * (it0 + q), WT_TH * RS - 8);
                    const int r = (ef >= RS) + (ef >= 2 * RS) + (ef >= 3 * RS), e = ef - r * RS;
                    const int pos = e / g.C, within = e - pos * g.C;
                    const float* sp = src0 + ((size_t)min(ho0 + r + kh + g.ih0, himax) * g.W + min(pos, pmax)) * p.ics + within;
                    fa[q] = *reinterpret_cast<const float4*>(sp); fc[q] = *reinterpret_cast<const float4*>(sp + 4);
                }
#pragma unroll
                for (int q = 0; q < HB; ++q) {
                    const int ef = tl * 8 + 2048 * (it0 + q);
                    if (ef < WT_TH * RS) {
                        const int r = (ef >= RS) + (ef >= 2 * RS) + (ef >= 3 * RS), e = ef - r * RS;
                        const bool in = (ho0 + r + kh + g.ih0 < g.H) && (e < vcols * g.C);
                        bf16x8 v;
                        v[0] = (__bf16)(in ? fa[q].x : 0.f); v[1] = (__bf16)(in ? fa[q].y : 0.f); v[2] = (__bf16)(in ? fa[q].z : 0.f); v[3] = (__bf16)(in ? fa[q].w : 0.f);
                        v[4] = (__bf16)(in ? fc[q].x : 0.f); v[5] = (__bf16)(in ? fc[q].y : 0.f); v[6] = (__bf16)(in ? fc[q].z : 0.f); v[7] = (__bf16)(in ? fc[q].w : 0.f);
                        *reinterpret_cast<bf16x8*>(halo + ef) = v;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);         // (keeps the next batch's loads behind this batch's stores: registers)
            }
        }
        // ---- dOut tile [256 positions][N] (zero for positions outside the output and n >= N)
        if (D16) {                                       // bf16 in memory: staged as is
            bf16x4 rh[16];
            const __bf16* d16 = reinterpret_cast<const __bf16*>(p.dOut);
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int pos = dp0 + 16 * it, r = pos >> 6, c = pos & 63;
                const int ho = min(ho0 + r, g.Hout - 1), wo = min(wo0 + c, g.Wout - 1);
                const size_t off = (((size_t)b * g.OH + ho + g.oh0) * g.OW + wo + g.ow0) * g.OC + n0 + min(4 * dnq, p.N - 4);
                rh[it] = *reinterpret_cast<const bf16x4*>(d16 + off);
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int pos = dp0 + 16 * it, r = pos >> 6, c = pos & 63;
                const bool ok = (ho0 + r < g.Hout) && (wo0 + c < g.Wout) && (4 * dnq < p.N);
                bf16x4 v = rh[it];
                if (!ok) { v[0] = (__bf16)0.f; v[1] = (__bf16)0.f; v[2] = (__bf16)0.f; v[3] = (__bf16)0.f; }
                if (4 * dnq < WT_NP) *reinterpret_cast<bf16x4*>(dt + pos * WT_NP + 4 * dnq) = v;
                if (want_bias) { bq[0] += (float)v[0]; bq[1] += (float)v[1]; bq[2] += (float)v[2]; bq[3] += (float)v[3]; }
            }
        } else {
            int tq = tid;
            asm volatile("" : "+v"(tq));                   // (see the input rows above)
            const int dnq = tq & 15, dp0 = tq >> 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {                    // two batches of 8 positions: 32 instead of 64 staging registers
                float4 rd[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int pos = dp0 + 16 * (8 * h + q), r = pos >> 6, c = pos & 63;
                    const int ho = min(ho0 + r, g.Hout - 1), wo = min(wo0 + c, g.Wout - 1);
                    const size_t off = (((size_t)b * g.OH + ho + g.oh0) * g.OW + wo + g.ow0) * g.OC + n0 + min(4 * dnq, p.N - 4);
                    rd[q] = *reinterpret_cast<const float4*>(p.dOut + off);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int pos = dp0 + 16 * (8 * h + q), r = pos >> 6, c = pos & 63;
                    const bool ok = (ho0 + r < g.Hout) && (wo0 + c < g.Wout) && (4 * dnq < p.N);
                    const float4 v4 = ok ? rd[q] : make_float4(0.f, 0.f, 0.f, 0.f);
                    bf16x4 v;
                    v[0] = (__bf16)v4.x; v[1] = (__bf16)v4.y; v[2] = (__bf16)v4.z; v[3] = (__bf16)v4.w;
                    if (4 * dnq < WT_NP) *reinterpret_cast<bf16x4*>(dt + pos * WT_NP + 4 * dnq) = v;
                    if (want_bias) { bq[0] += v4.x; bq[1] += v4.y; bq[2] += v4.z; bq[3] += v4.w; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        WT_T(t_c);
        if (A16) __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): this wave's DMA pieces have landed
        __syncthreads();
        WT_T(t_d);
        // ---- 8 reduction steps of 32 positions (row r, columns 32 * half ..).  A read-then-multiply loop left the LDS latency of every
        // step exposed (9 k clocks per tile for 3.6 k of MFMA issue), two full register sets of fragments do not fit beside the 112
        // accumulators of D.conv5: the dOut fragments and the first PF input fragments of step ks + 1 are read during step ks, the
        // remaining input fragments at the top of their own step - they arrive behind the first PF x NT MFMAs.
#ifndef WT_PFB
#define WT_PFB 2
#endif
#ifndef WT_PFA
#define WT_PFA 1
#endif
        constexpr int PF = KTW < WT_PFB ? KTW : WT_PFB;
        constexpr int PFS = PF > 0 ? PF : 1;
        auto lda = [&](int ks, bf16x8 (&af)[NT]) {
#pragma unroll
            for (int i = 0; i < NT; ++i) af[i] = tr_frag(dt + (ks * 32) * WT_NP, WT_NP, 16 * i, lane);
        };
        auto ldb = [&](int ks, int jj) {
            const __bf16* hb = halo + (ks >> 1) * RS + (ks & 1) * 32 * g.C;
            return tr_frag(hb, g.C, 16 * min(wave + 4 * jj, nkt - 1), lane);
        };
        {
            bf16x8 ac[NT], an[NT], bc[KTW], bn[PFS];
            lda(0, ac);
#pragma unroll
            for (int jj = 0; jj < PF; ++jj) bc[jj] = ldb(0, jj);
#pragma unroll 1
            for (int ks = 0; ks < WT_TH * 2; ++ks) {
                if (!WT_PFA && ks > 0) lda(ks, ac);
#pragma unroll
                for (int jj = PF; jj < KTW; ++jj) bc[jj] = ldb(ks, jj);
                if (ks + 1 < WT_TH * 2) {
                    if (WT_PFA) lda(ks + 1, an);
#pragma unroll
                    for (int jj = 0; jj < PF; ++jj) bn[jj] = ldb(ks + 1, jj);
                }
                // (a wave's surplus tile slots, wave + 4 jj >= nkt, recompute the last tile and are not stored: tests here would cut the
                //  step's MFMAs into seven basic blocks with a wait in front of each)
#pragma unroll
                for (int jj = 0; jj < KTW; ++jj)
#pragma unroll
                    for (int i = 0; i < NT; ++i) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[i], bc[jj], acc[i][jj], 0, 0, 0);
                if (ks + 1 < WT_TH * 2) {
                    if (WT_PFA) {
#pragma unroll
                        for (int i = 0; i < NT; ++i) ac[i] = an[i];
                    }
#pragma unroll
                    for (int jj = 0; jj < PF; ++jj) bc[jj] = bn[jj];
                }
                __builtin_amdgcn_sched_barrier(0);           // the reads of later steps stay behind this step's MFMAs (registers)
            }
        }
#ifdef WT_PROF
        { WT_T(t_e); pacc[4] += t_b - t_a; pacc[1] += t_c - t_b; pacc[2] += t_d - t_c; pacc[3] += t_e - t_d; pacc[5] += 1; }
#endif
    }
#ifdef WT_PROF
    if (tid == 0 && NT == WT_PROF_NT) for (int q_ = 0; q_ < 6; ++q_) atomicAdd(&wt_prof[q_], pacc[q_]);
#endif
    // ---- partials: acc[i][jj][reg] = dW[n = 16 i + 4 lg + reg][kk = kh * seglen + 16 (wave + 4 jj) + li]
    const int li = lane & 15, lg = lane >> 4;
    // (sliced problems: local column kk = kw * C + c of the slice is column kw * ics + ic0 + c of the layer's kernel row, rows n0 + n)
    float* part = p.part + (size_t)grp * Nfull * p.Kfull + (size_t)n0 * p.Kfull + (size_t)kh * p.KW * p.ics + ic0;
    int kcol[KTW];
#pragma unroll
    for (int jj = 0; jj < KTW; ++jj) {
        const int kk = 16 * (wave + 4 * jj) + li, kw = kk / g.C;
        kcol[jj] = (wave + 4 * jj < nkt && kk < g.seglen) ? kw * p.ics + (kk - kw * g.C) : -1;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int n = 16 * i + 4 * lg + reg;
            if (n >= p.N) continue;
#pragma unroll
            for (int jj = 0; jj < KTW; ++jj)
                if (kcol[jj] >= 0) part[(size_t)n * p.Kfull + kcol[jj]] = acc[i][jj][reg];
        }
    if (want_bias) {                                      // fold the 16 threads that share an n quad
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) bred[dp0][4 * dnq + q] = bq[q];
        __syncthreads();
        if (tid < 64 && tid < p.N) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += bred[q][tid];
            p.bpart[(size_t)grp * Nfull + n0 + tid] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------ weight gradient, bf16 in memory, DMA double buffer
// Round 4.  conv_wgrad_tile16_kernel ran at MFMA-busy 0.35 (D.conv5) .. 0.07 (D.conv2): one kernel row per workgroup, i.e. every staged
// position tile was fetched KH times and used for 1 / KH of the arithmetic; the dOut tile travelled through registers; and a tile's
// staging (one memory latency, 5.6 k clocks) was only hidden by the OTHER workgroup of the CU.  This kernel is for the layers whose
// input activation AND output gradient are bfloat16 in memory (D.conv2 .. conv5 in bf16 mode):
//  * a workgroup owns KR kernel rows (all of them for D.conv2 .. conv4, three of nine for D.conv5): the halo holds TH + KR - 1 input
//    rows and the dOut tile is staged once for KR x the MFMAs.  The (kernel row, 16-wide reduction tile) work items are dealt
//    round-robin to the 8 waves; a wave keeps NT x NJ accumulator tiles for the whole kernel;
//  * BOTH operands go global -> LDS by DMA (global_load_lds_dwordx4, 1 KB pieces, per-lane source addresses): no staging registers,
//    no conversion.  The dOut image is [positions][NP] with the conflict-free row stride of tr_frag; lanes that land on pad columns
//    or on positions outside the output read from row 0 of the utterance's zero-bordered gradient buffer (zeros);
//  * two LDS buffers: the DMA of tile t + 1 is issued right after the barrier that opens tile t and lands under tile t's MFMAs -
//    one barrier per tile, no exposed memory latency; one 512-thread workgroup per CU (two waves per SIMD).
// Partials [group][N][Ktot] + bias partials, reduced by wgrad_reduce_kernel as before (fixed order: deterministic).
#define WD_TW 64
#define WD_MAXHP 6           // halo DMA pieces per wave and tile (host checks)
#define WD_MAXDP 4           // dOut DMA pieces per wave and tile
struct WgradDmaArgs {
    const __bf16* A;
    const __bf16* dOut;
    float* part;             // [G][N][Ktot]
    float* bpart;            // [G][N] or null
    int N, KH, KW, KR, nkt;
    int nth, ntw, ntiles, G;
    ConvGeom g;
};

template <int NT, int NJ, int TH>
__global__ __launch_bounds__(512, 1) void conv_wgrad_dma_kernel(WgradDmaArgs p) {
    constexpr int NP = WT_NP_OF(NT);
    constexpr int NPOS = TH * WD_TW, NS = NPOS / 32;                  // positions per tile, reduction steps of 32 positions
    constexpr int NDP = NPOS * NP / 512;                              // DMA pieces of the dOut image (exact: NP is 16, 48 or 80)
    extern __shared__ __attribute__((aligned(16))) __bf16 wd_lds[];   // 2 x { halo [TH + KR - 1][RS], WT_SLACK, dOut [NPOS][NP] }
    __shared__ float bred[32][64];
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsub = p.KH / p.KR;                                     // workgroups that share a group's tiles: ids of one XCD
    int grp, rest;
    if (nsub > 1 && (p.G & 7) == 0) {
        const int w = blockIdx.x, xcd = w & 7, slot = w >> 3;
        rest = slot % nsub;
        grp = (slot / nsub) * 8 + xcd;
    } else {
        rest = blockIdx.x % nsub;
        grp = blockIdx.x / nsub;
    }
    const int kh0 = rest * p.KR;
    const int HR = TH + p.KR - 1, wcols = WD_TW + p.KW - 1, RS = wcols * g.C;
    const int himg = HR * RS, dtoff = himg + WT_SLACK, bufsz = dtoff + NPOS * NP;
    const int nitems = p.KR * p.nkt;
    // ---- this wave's work items q = wave + 8 jj -> (kernel row kr, reduction tile): element offset of the B fragment inside the halo image
    int boff[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int q = min(wave + 8 * jj, nitems - 1), kr = q / p.nkt;
        boff[jj] = kr * RS + 16 * (q - kr * p.nkt);
    }
    f32x4 acc[NT][NJ];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < WT_SLACK; e += 512) { wd_lds[himg + e] = (__bf16)0.f; wd_lds[bufsz + himg + e] = (__bf16)0.f; }
    // ---- tile-invariant parts of the DMA source addresses
    const int npc = (himg + 511) >> 9;
    int hrow[WD_MAXHP], hel[WD_MAXHP];
#pragma unroll
    for (int u = 0; u < WD_MAXHP; ++u) {
        const int ef = min(512 * (wave + 8 * u) + 8 * lane, himg - 8);
        hrow[u] = ef / RS;
        hel[u] = ef - hrow[u] * RS;
    }
    int dofs[WD_MAXDP], drc[WD_MAXDP];                                // dOut pieces: offset inside the utterance's buffer, (row << 8 | column) or -1 for pad columns
#pragma unroll
    for (int u = 0; u < WD_MAXDP; ++u) {
        const int e = 512 * (wave + 8 * u) + 8 * lane;
        const int pos = e / NP, col = e - pos * NP, r = pos >> 6, c = pos & 63;
        dofs[u] = ((r + g.oh0) * g.OW + c + g.ow0) * g.OC + col;
        drc[u] = (col < p.N && wave + 8 * u < NDP) ? ((r << 8) | c) : -1;
    }
    auto issue = [&](int t, __bf16* buf) {
        const int tw_i = t % p.ntw, rem = t / p.ntw, th_i = rem % p.nth, b = rem / p.nth;
        const int ho0 = th_i * TH, wo0 = tw_i * WD_TW, wi0 = wo0 + g.iw0;
        const int vcols = max(0, min(wcols, g.W - wi0));
        const int emax = max(vcols * g.C - 8, 0), himax = g.H - 1;
        const __bf16* src0 = p.A + (((size_t)b * g.H) * g.W + wi0) * g.C;
#pragma unroll
        for (int u = 0; u < WD_MAXHP; ++u) {
            const int k = wave + 8 * u;
            if (k < npc) {
                const int hi = min(ho0 + hrow[u] + kh0 + g.ih0, himax);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + (size_t)hi * g.W * g.C + min(hel[u], emax)),
                                                 (__attribute__((address_space(3))) void*)(buf + 512 * k), 16, 0, 0);
            }
        }
        const __bf16* dimg = p.dOut + (size_t)b * g.OH * g.OW * g.OC;   // row 0 of the bordered buffer is zero: the source of everything outside
        const int tofs = (ho0 * g.OW + wo0) * g.OC;
#pragma unroll
        for (int u = 0; u < WD_MAXDP; ++u) {
            const int k = wave + 8 * u;
            if (k < NDP) {
                const bool ok = drc[u] >= 0 && (ho0 + (drc[u] >> 8) < g.Hout) && (wo0 + (drc[u] & 255) < g.Wout);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dimg + (ok ? tofs + dofs[u] : 0)),
                                                 (__attribute__((address_space(3))) void*)(buf + dtoff + 512 * k), 16, 0, 0);
            }
        }
    };
    const bool want_bias = (p.bpart != nullptr) && (kh0 == 0);
    float bq[4] = {0.f, 0.f, 0.f, 0.f};
    const int bnq = tid & 15, bp0 = tid >> 4;                            // bias: n quad, first position (32 position groups)
    // lane parts of the tr_frag addresses (see tr_frag): row (4 (g & 1) + 16 (g >> 1) + q), 4 elements at 4 pq
    const int lg_ = lane >> 4, lq_ = (lane & 15) >> 2, lpq = lane & 3;
    const int lrow = 4 * (lg_ & 1) + 16 * (lg_ >> 1) + lq_;
    const int la_off = lrow * NP + 4 * lpq, lb_off = lrow * g.C + 4 * lpq;
    const int ldb8 = 8 * g.C;

    int t = grp, it = 0;
    if (t < p.ntiles) issue(t, wd_lds);
    for (; t < p.ntiles; t += p.G, ++it) {
        __bf16* buf = wd_lds + (it & 1) * bufsz;
        __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0): this wave's pieces of tile t have landed
        __syncthreads();                                   // ... everyone's; and every wave is done with the other buffer (tile t - G)
        if (t + p.G < p.ntiles) issue(t + p.G, wd_lds + ((it + 1) & 1) * bufsz);
        const __bf16* dt = buf + dtoff;
        if (want_bias && 4 * bnq < p.N) {
#pragma unroll
            for (int u = 0; u < NPOS / 32; ++u) {
                const bf16x4 v = *reinterpret_cast<const bf16x4*>(dt + (bp0 + 32 * u) * NP + 4 * bnq);
                bq[0] += (float)v[0]; bq[1] += (float)v[1]; bq[2] += (float)v[2]; bq[3] += (float)v[3];
            }
        }
        // ---- F = NS x NJ work items (reduction step ks, item jj) as ONE unrolled software pipeline: the B fragment of item f + LA and the
        // A fragments of the step that starts LA items ahead are requested before the MFMAs of item f (a read-then-multiply item left
        // one LDS latency exposed per item: MFMA-busy 0.55); sched_barrier pins the order, the wait-count pass then waits only for
        // the fragment an item needs.
        {
            constexpr int F = NS * NJ;
            constexpr int LA = NT >= 3 ? 2 : (NT == 2 ? 3 : 6);             // look-ahead in items (an item = NT MFMAs of 16 cycles)
            constexpr int AS = LA / NJ + 2;                                  // A-fragment sets in flight
            bf16x8 bring[LA + 1], aring[AS][NT];
            auto rdA = [&](int ks, bf16x8 (&a)[NT]) {
                const __bf16* ha = dt + ks * 32 * NP + la_off;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(ha + 16 * i));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(ha + 16 * i + 8 * NP));
                    a[i][0] = lo[0]; a[i][1] = lo[1]; a[i][2] = lo[2]; a[i][3] = lo[3]; a[i][4] = hi[0]; a[i][5] = hi[1]; a[i][6] = hi[2]; a[i][7] = hi[3];
                }
            };
            auto rdB = [&](int ks, int jj) {
                const __bf16* hb = buf + (ks >> 1) * RS + (ks & 1) * 32 * g.C + lb_off + boff[jj];
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)hb);
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(hb + ldb8));
                bf16x8 r;
                r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
                return r;
            };
            // prologue: the A sets of the steps that start within the first LA items, the B fragments of items 0 .. LA - 1
#pragma unroll
            for (int ks = 0; ks < NS; ++ks)
                if (ks * NJ <= LA) rdA(ks, aring[ks % AS]);
#pragma unroll
            for (int f = 0; f < LA; ++f)
                if (f < F) bring[f % (LA + 1)] = rdB(f / NJ, f % NJ);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int ks = f / NJ, jj = f % NJ;
                if (f + LA < F) {
                    const int fn = f + LA, ksn = fn / NJ;
                    bring[fn % (LA + 1)] = rdB(ksn, fn % NJ);
                    if (fn % NJ == 0 && ksn * NJ > LA) rdA(ksn, aring[ksn % AS]);     // first item of a later step: its A set
                }
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aring[ks % AS][i], bring[f % (LA + 1)], acc[i][jj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- partials: acc[i][jj][reg] = dW[n = 16 i + 4 lg + reg][(kh0 + kr) * seglen + 16 tile + li]
    const int li = lane & 15, lgp = lane >> 4;
    float* part = p.part + (size_t)grp * p.N * g.Ktot;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int q = wave + 8 * jj;
        if (q >= nitems) continue;
        const int kr = q / p.nkt, kk = 16 * (q - kr * p.nkt) + li;
        if (kk >= g.seglen) continue;
        const int col = (kh0 + kr) * g.seglen + kk;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int n = 16 * i + 4 * lgp + reg;
                if (n < p.N) part[(size_t)n * g.Ktot + col] = acc[i][jj][reg];
            }
    }
    if (want_bias) {                                      // fold the 32 threads that share an n quad
#pragma unroll
        for (int q = 0; q < 4; ++q) bred[bp0][4 * bnq + q] = bq[q];
        __syncthreads();
        if (tid < 64 && tid < p.N) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 32; ++q) sum += bred[q][tid];
            p.bpart[(size_t)grp * p.N + tid] = sum;
        }
    }
}

// Pointwise (1 x 1) layer with 4 input channels and N <= 8 outputs (D's first layer: 3 -> 8): the weight gradient is a plain reduction
// over the M output positions, 16 B + 32 B per row - memory bound, no matrix cores.  grid = splits workgroups of 1024 threads,
// rows strided over all threads, 4N + N accumulators per thread, fixed-order wave / workgroup reduction.  part [splits][N][4].
#define WP_NMAX 8
__global__ __launch_bounds__(1024) void conv_wgrad_pointwise_kernel(WgradArgs p) {
    __shared__ float red[16][WP_NMAX * 5];
    const ConvGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, N = p.N;
    float acc[WP_NMAX][4], bacc[WP_NMAX];
#pragma unroll
    for (int n = 0; n < WP_NMAX; ++n) { bacc[n] = 0.f; acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.f; }
    for (int m = blockIdx.x * 1024 + tid; m < p.M; m += gridDim.x * 1024) {
        int b, ho, wo;
        decode_m(m, g, b, ho, wo);
        const float4 a = *reinterpret_cast<const float4*>(p.A + (((size_t)b * g.H + ho + g.ih0) * g.W + wo + g.iw0) * g.C);
        const float* dp = p.dOut + (((size_t)b * g.OH + ho + g.oh0) * g.OW + wo + g.ow0) * g.OC;
        float d[WP_NMAX];
        const float4 d0 = *reinterpret_cast<const float4*>(dp);
        d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w;
        if (N > 4) {
            const float4 d1 = *reinterpret_cast<const float4*>(dp + 4);
            d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
        } else { d[4] = d[5] = d[6] = d[7] = 0.f; }
#pragma unroll
        for (int n = 0; n < WP_NMAX; ++n) {
            acc[n][0] += d[n] * a.x; acc[n][1] += d[n] * a.y; acc[n][2] += d[n] * a.z; acc[n][3] += d[n] * a.w;
            bacc[n] += d[n];
        }
    }
#pragma unroll
    for (int n = 0; n < WP_NMAX; ++n) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float v = wave_sum(acc[n][c]); if (lane == 0) red[wave][n * 5 + c] = v; }
        const float v = wave_sum(bacc[n]);
        if (lane == 0) red[wave][n * 5 + 4] = v;
    }
    __syncthreads();
    if (tid < WP_NMAX * 5) {
        float s_ = 0.f;
        for (int w_ = 0; w_ < 16; ++w_) s_ += red[w_][tid];
        const int n = tid / 5, c = tid - n * 5;
        if (n < N) {
            if (c < 4) p.part[((size_t)blockIdx.x * N + n) * 4 + c] = s_;
            else if (p.bpart) p.bpart[(size_t)blockIdx.x * N + n] = s_;
        }
    }
}

// Sum the split partials in fixed order and scatter from GEMM layout [n][kh][kw][ci] to the
// PyTorch parameter layout [n][ci][kh][kw] (flip != 0: the partials are in the flipped data-gradient
// layout, never used for weights).  One thread per weight element.
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bpart, int splits, int N,
                                    int KH, int KW, int C, int Cvalid, float* __restrict__ dW, float* __restrict__ db,
                                    int accumulate) {
    const int Ktot = KH * KW * C;
    const int total = N * Ktot;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int n = idx / Ktot, kk = idx - n * Ktot;
        const int ci = kk % C, t = kk / C, kw = t % KW, kh = t / KW;
        if (ci >= Cvalid) continue;
        // eight partials in flight per round (a dependent load per partial made the small layers' reductions 100 us latency chains);
        // fixed summation order
        float s = 0.f;
        int sp = 0;
        for (; sp + 8 <= splits; sp += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(sp + u) * total + idx];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; sp < splits; ++sp) s += part[(size_t)sp * total + idx];
        const size_t o = (((size_t)n * Cvalid + ci) * KH + kh) * KW + kw;
        dW[o] = accumulate ? dW[o] + s : s;
    }
    if (db && bpart) {
        for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
            float s = 0.f;
            int sp = 0;
            for (; sp + 8 <= splits; sp += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = bpart[(size_t)(sp + u) * N + n];
                s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            for (; sp < splits; ++sp) s += bpart[(size_t)sp * N + n];
            db[n] = accumulate ? db[n] + s : s;
        }
    }
}

// PyTorch weight [N][Cvalid][KH][KW] (optionally scaled by 1/sigma[0]) -> forward GEMM layout
// Wf[n][kh][kw][c] (c padded to C with zeros) and, if Wb != null, data-gradient layout
// Wb[c][KH-1-kh][KW-1-kw][n] (rows c < Cvalid only; Cb = N is its channel count).
__global__ void weight_prep_kernel(const float* __restrict__ Wt, const float* __restrict__ sigma, int N, int Cvalid, int C,
                                   int KH, int KW, float* __restrict__ Wf, float* __restrict__ Wb) {
    const int total = N * Cvalid * KH * KW;
    const float inv = sigma ? 1.f / sigma[0] : 1.f;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int t = idx;
        const int kw = t % KW; t /= KW;
        const int kh = t % KH; t /= KH;
        const int ci = t % Cvalid;
        const int n = t / Cvalid;
        const float v = Wt[idx] * inv;
        Wf[(((size_t)n * KH + kh) * KW + kw) * C + ci] = v;
        if (Wb) Wb[(((size_t)ci * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * N + n] = v;
    }
}

// ------------------------------------------------------------------------------------------ C ABI
static int check_geom(const char* who, const ConvGeom& g, int M, int N) {
    if (M <= 0 || N <= 0) return nele_set_error(NELE_ERR_INVALID_ARG, "%s: M=%d N=%d", who, M, N);
    if (g.C % 4 || g.seglen % 4 || g.segstride % 4 || g.Ktot % 4)
        return nele_set_error(NELE_ERR_UNSUPPORTED, "%s: channel counts must be multiples of 4 (C=%d seglen=%d)", who, g.C, g.seglen);
    if (g.Ktot % g.seglen) return nele_set_error(NELE_ERR_INVALID_ARG, "%s: Ktot %% seglen != 0", who);
    return NELE_OK;
}

// geom: 15 ints in ConvGeom order
extern "C" int nele_conv_gemm(const float* A, const float* Wg, const float* bias, const float* aux, float* out, int M, int N,
                              int epi, float slope, const int* geom, void* stream) {
    NELE_CHECK_ARG(A && Wg && out && geom, "nele_conv_gemm: null pointer");
    GemmArgs p;
    p.A = A; p.Wg = Wg; p.bias = bias; p.aux = aux; p.out = out; p.M = M; p.N = N; p.epi = epi; p.slope = slope;
    memcpy(&p.g, geom, sizeof(ConvGeom));
    int st = check_geom("nele_conv_gemm", p.g, M, N);
    if (st) return st;
    NELE_CHECK_ARG(!(epi == EPI_BIAS || epi == EPI_BIAS_LRELU || epi == EPI_BIAS_EXPTANH) || bias, "nele_conv_gemm: epilogue needs bias");
    NELE_CHECK_ARG(epi != EPI_MASK_LRELU_GRAD || aux, "nele_conv_gemm: epilogue needs aux");
    hipStream_t s = as_stream(stream);
    // rows per block: 256 when M alone yields >= 2 blocks per CU, else 128 / 64 so that the grid still covers the chip
    const int ny = (N + 63) / 64;
    int TMsel = 4;
    if ((long long)((M + 255) / 256) * ny < 512) TMsel = 2;
    if ((long long)((M + 127) / 128) * ny < 512) TMsel = 1;
    const int BM = 64 * TMsel, gx = (M + BM - 1) / BM;
#define LAUNCH_GEMM(TN_) \
    do { \
        if (TMsel == 4) hipLaunchKernelGGL((conv_gemm_kernel<TN_, 4>), dim3(gx, (TN_ == 4) ? ny : 1), dim3(256), 0, s, p); \
        else if (TMsel == 2) hipLaunchKernelGGL((conv_gemm_kernel<TN_, 2>), dim3(gx, (TN_ == 4) ? ny : 1), dim3(256), 0, s, p); \
        else hipLaunchKernelGGL((conv_gemm_kernel<TN_, 1>), dim3(gx, (TN_ == 4) ? ny : 1), dim3(256), 0, s, p); \
    } while (0)
    if (N <= 16) LAUNCH_GEMM(1);
    else if (N <= 32) LAUNCH_GEMM(2);
    else if (N <= 48) LAUNCH_GEMM(3);
    else LAUNCH_GEMM(4);
#undef LAUNCH_GEMM
    NELE_CHECK_LAUNCH("nele_conv_gemm");
    return NELE_OK;
}

// nele_conv_gemm with bf16 MFMA operands (N > 48 only: the generator's layers)
extern "C" int nele_conv_gemm_bf16(const float* A, const float* Wg, const float* bias, const float* aux, float* out, int M, int N,
                                   int epi, float slope, const int* geom, void* stream) {
    NELE_CHECK_ARG(A && Wg && out && geom, "nele_conv_gemm_bf16: null pointer");
    GemmArgs p;
    p.A = A; p.Wg = Wg; p.bias = bias; p.aux = aux; p.out = out; p.M = M; p.N = N; p.epi = epi; p.slope = slope;
    memcpy(&p.g, geom, sizeof(ConvGeom));
    int st = check_geom("nele_conv_gemm_bf16", p.g, M, N);
    if (st) return st;
    NELE_CHECK_ARG(!(epi == EPI_BIAS || epi == EPI_BIAS_LRELU || epi == EPI_BIAS_EXPTANH) || bias, "nele_conv_gemm_bf16: epilogue needs bias");
    NELE_CHECK_ARG(epi != EPI_MASK_LRELU_GRAD || aux, "nele_conv_gemm_bf16: epilogue needs aux");
    hipStream_t s = as_stream(stream);
    const int ny = (N + 63) / 64;
    const int TMsel = ((long long)((M + 127) / 128) * ny >= 512) ? 2 : 1;
    const int BM = 64 * TMsel, gx = (M + BM - 1) / BM;
    if (TMsel == 2) hipLaunchKernelGGL((conv_gemm16_kernel<4, 2>), dim3(gx, ny), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_gemm16_kernel<4, 1>), dim3(gx, ny), dim3(256), 0, s, p);
    NELE_CHECK_LAUNCH("nele_conv_gemm_bf16");
    return NELE_OK;
}

extern "C" int nele_weight_prep_frag(const float* Wg, int N, int Ktot, float* Wfrag, void* stream) {
    NELE_CHECK_ARG(Wg && Wfrag && N > 0 && Ktot > 0 && Ktot % 8 == 0, "nele_weight_prep_frag: bad arguments (Ktot must be a multiple of 8)");
    const int NT = (N + 15) / 16;
    const int total = (Ktot / 8) * NT * 128;
    hipLaunchKernelGGL(weight_frag_kernel, dim3(min(1024, (total + 255) / 256)), dim3(256), 0, as_stream(stream), Wg, N, Ktot, NT, Wfrag);
    NELE_CHECK_LAUNCH("nele_weight_prep_frag");
    return NELE_OK;
}

// 1 if nele_conv_span supports this geometry (long output rows, span fits in LDS twice per CU), else 0
extern "C" int nele_conv_span_supported(int M, int N, const int* geom, int KH, int KW) {
    ConvGeom g;
    memcpy(&g, geom, sizeof(ConvGeom));
    if (N > 64 || g.seglen % 8 || g.C % 4 || g.Wout < 64 || KH * g.seglen != g.Ktot || KW * g.C != g.seglen) return 0;
    const int maxrun = (GEMM_BM + g.Wout - 1) / g.Wout + 1;
    if (maxrun > SPAN_MAXRUN) return 0;
    const long long fl = (long long)GEMM_BM * g.C + (long long)maxrun * (KW - 1) * g.C;
    if (fl * 4 > 76 * 1024) return 0;
    return 1;
}

extern "C" int nele_conv_span(const float* A, const float* Wfrag, const float* bias, const float* aux, float* out, int M, int N, int epi,
                              float slope, const int* geom, int KH, int KW, long long a_elems, void* stream) {
    NELE_CHECK_ARG(A && Wfrag && out && geom, "nele_conv_span: null pointer");
    if (!nele_conv_span_supported(M, N, geom, KH, KW)) return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_conv_span: geometry not supported");
    NELE_CHECK_ARG(a_elems < 2147483647LL, "nele_conv_span: input buffer too large for 32-bit offsets");
    SpanArgs p;
    p.A = A; p.Wfrag = Wfrag; p.bias = bias; p.aux = aux; p.out = out; p.M = M; p.N = N; p.NT = (N + 15) / 16; p.epi = epi; p.slope = slope;
    p.KH = KH; p.KW = KW;
    memcpy(&p.g, geom, sizeof(ConvGeom));
    NELE_CHECK_ARG(!(epi == EPI_BIAS || epi == EPI_BIAS_LRELU || epi == EPI_BIAS_EXPTANH) || bias, "nele_conv_span: epilogue needs bias");
    NELE_CHECK_ARG(epi != EPI_MASK_LRELU_GRAD || aux, "nele_conv_span: epilogue needs aux");
    const int maxrun = (GEMM_BM + p.g.Wout - 1) / p.g.Wout + 1;
    const size_t lds = ((size_t)GEMM_BM * p.g.C + (size_t)maxrun * (KW - 1) * p.g.C) * sizeof(float);
    const int gx = (M + GEMM_BM - 1) / GEMM_BM;
    hipStream_t s = as_stream(stream);
    static unsigned long long attr_done = 0;            // (per device: a process that switches devices sets the attribute on each)
    if (nele_first_use_on_device(&attr_done)) {  // allow > 64 KB of dynamic LDS
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_span_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_span_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_span_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_span_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    }
    switch (p.NT) {
        case 1: hipLaunchKernelGGL(conv_span_kernel<1>, dim3(gx), dim3(256), lds, s, p); break;
        case 2: hipLaunchKernelGGL(conv_span_kernel<2>, dim3(gx), dim3(256), lds, s, p); break;
        case 3: hipLaunchKernelGGL(conv_span_kernel<3>, dim3(gx), dim3(256), lds, s, p); break;
        default: hipLaunchKernelGGL(conv_span_kernel<4>, dim3(gx), dim3(256), lds, s, p); break;
    }
    NELE_CHECK_LAUNCH("nele_conv_span");
    return NELE_OK;
}

// elements (bf16) of the fragment-major weight buffer for nele_conv_span_bf16
extern "C" long long nele_weight_frag16_elems(int N, int seglen, int KH) {
    return (long long)KH * ((seglen + 31) / 32) * ((N + 15) / 16) * 512 + 2048;   // + one 256-fragment group of read slack
}

extern "C" int nele_weight_prep_frag16(const float* Wg, int N, int Ktot, int seglen, int KH, void* Wfrag, void* stream) {
    NELE_CHECK_ARG(Wg && Wfrag && N > 0 && KH > 0 && seglen > 0 && KH * seglen == Ktot, "nele_weight_prep_frag16: bad arguments");
    const int NT = (N + 15) / 16, sps = (seglen + 31) / 32;
    const long long total = nele_weight_frag16_elems(N, seglen, KH);
    hipLaunchKernelGGL(weight_frag16_kernel, dim3((unsigned)min(2048LL, (total + 255) / 256)), dim3(256), 0, as_stream(stream), Wg, N, Ktot, seglen,
                       KH, NT, sps, (__bf16*)Wfrag);
    NELE_CHECK_LAUNCH("nele_weight_prep_frag16");
    return NELE_OK;
}

// LDS bytes of the 2-D tile kernel, or 0 when the geometry does not fit it
static size_t tile16_lds(const ConvGeom& g, int N, int KH, int KW, int TH) {
    if (N > 64 || N % 4 || g.OC % 4 || g.C % 8 || KH * g.seglen != g.Ktot || KW * g.C != g.seglen) return 0;
    const long long RS = (long long)(TILE16_TW + KW - 1) * g.C;
    if (RS > 4096) return 0;                                            // one row travels as 2 x 8 floats per thread
    const int sps = (g.seglen + 31) / 32, NT = (N + 15) / 16;
    int SB = 1;
    for (int d = 1; d <= TILE16_SBMAX; ++d)
        if (sps % d == 0) SB = d;
    const long long bytes = ((long long)(TH + KH - 1) * RS + 64) * 2 + (((long long)SB * NT * 1024 + 4095) & ~4095LL);
    if (bytes > 158 * 1024) return 0;
    return (size_t)(bytes < 4 * 16 * 68 * 4 ? 4 * 16 * 68 * 4 : bytes);        // the epilogue transposes through 4 x 4352 B
}
static int tile16_sb(const ConvGeom& g) {
    const int sps = (g.seglen + 31) / 32;
    int SB = 1;
    for (int d = 1; d <= TILE16_SBMAX; ++d)
        if (sps % d == 0) SB = d;
    return SB;
}
// Conv1d / Linear tile kernel: steps per weight chunk and LDS bytes, or 0
static int conv1d16_sb(const ConvGeom& g, int N, int KH, int KW, size_t* lds_out) {
    if (KH != 1 || N % 4 || g.OC % 4 || g.C % 8 || KW * g.C != g.seglen || g.seglen != g.Ktot || g.seglen % 32) return 0;
    if (N > 64 && N % 64) return 0;
    const long long RS = (long long)(C1D_TW + KW - 1) * (g.C + (((g.C & (g.C - 1)) == 0) ? C1D_PAD(g.C) : 0));
    const int sps = g.seglen / 32, TNsel = (N >= 64) ? 4 : (N + 15) / 16;
    for (int d = TILE16_SBMAX; d >= 1; --d) {
        if (sps % d) continue;
        const long long bytes = (RS + 64) * 2 + 2 * (((long long)d * TNsel * 1024 + 4095) & ~4095LL);     // strip + two weight slots
        if (bytes <= 158 * 1024) {
            if (lds_out) *lds_out = (size_t)(bytes < 4 * 16 * 68 * 4 ? 4 * 16 * 68 * 4 : bytes);
            return d;
        }
    }
    return 0;
}
// Tile height.  Default: see below.  NELE_CONV_TH=4|8|10|11|13 forces one height where it fits; NELE_CONV_TALL=1 picks among {13, 11, 10, 8, 4} the candidate that
// wastes the fewest output rows (D.conv5, Hout = 44: 11; D.conv4, 52: 13; the 58-row gradients: 10) - a taller tile also needs fewer LDS
// reads per MFMA (TH + TN fragment reads per TH * TN MFMAs).  Measured at B = 256 (A/B inside one run, tools/ab.sh): D.conv5 forward alone
// 1.70 -> 1.62 ms (0.305 -> 0.32 of the bf16 peak), but the whole step 75.6 -> 76.9 ms: the taller tile's 160 KB of LDS leaves no room for
// the metric streams' workgroups beside it, and at B = 256 the step is bound by the sum of all kernels, not by the convolutions.  (The
// converse also holds: 4-row tiles make the kernel 17 % slower alone and the step 0.5 ms faster.)  Hence opt-in.
static int tile16_th(const ConvGeom& g, int N, int KH, int KW) {
    const int tall = NELE_SWITCH_INT("NELE_CONV_TALL", 0);
    const int force = NELE_SWITCH_INT("NELE_CONV_TH", 0);                                     // NELE_CONV_TH=4|8: force one tile height where it fits (A/B)
    if (force == 4 || force == 8 || force == 10 || force == 11 || force == 13) {
        if ((force == 4 || g.Hout >= 8) && tile16_lds(g, N, KH, KW, force)) return force;
    }
    if (!tall) {
        // 64 output channels (D.conv5 forward, the 64-wide gradients): 8-row tiles; fewer channels: 4-row tiles.  Measured inside the
        // B = 256 step (tools/ab.sh, three alternating repetitions): all 8-row 75.5 ms, all 4-row 73.9 ms (but D.conv5 forward alone
        // 1.96 instead of 1.70 ms), this mix 73.9 ms with D.conv5 at 1.74 ms: the narrow layers (TN <= 3) are bound by their LDS reads
        // either way, and the smaller halo leaves room for the workgroups of the other streams.
        if (N < 64 && tile16_lds(g, N, KH, KW, 4)) return 4;
        if (g.Hout >= 8 && tile16_lds(g, N, KH, KW, 8)) return 8;
        if (tile16_lds(g, N, KH, KW, 4)) return 4;
        return 0;
    }
    const int cand[5] = {13, 11, 10, 8, 4};
    int best = 0, bestrows = 1 << 30;
    for (int q = 0; q < 5; ++q) {
        const int th = cand[q];
        if (th > 4 && g.Hout < 8) continue;
        if (!tile16_lds(g, N, KH, KW, th)) continue;
        const int rows = (g.Hout + th - 1) / th * th;
        if (rows < bestrows) { bestrows = rows; best = th; }
    }
    return best;
}
static int span16_supported(int M, int N, const ConvGeom& g, int KH, int KW) {
    if (N > 64 || g.C % 8 || g.Wout < 64 || KH * g.seglen != g.Ktot || KW * g.C != g.seglen) return 0;
    const int maxrun = (SPAN16_BM + g.Wout - 1) / g.Wout + 1;
    if (maxrun > SPAN16_MAXRUN) return 0;
    const int cp = g.C + SPAN16_PAD(g.C);
    const long long el = (long long)SPAN16_BM * cp + (long long)maxrun * (KW - 1) * cp + 64;
    if (el * 2 > 79 * 1024) return 0;
    return 1;
}
extern "C" int nele_conv_span_bf16_supported(int M, int N, const int* geom, int KH, int KW) {
    ConvGeom g;
    memcpy(&g, geom, sizeof(ConvGeom));
    if (g.Wout >= 32 && tile16_th(g, N, KH, KW)) return 1;
    if (g.Wout >= 32 && conv1d16_sb(g, N, KH, KW, nullptr)) return 1;
    return span16_supported(M, N, g, KH, KW);
}
static int conv_span_bf16_impl(const float* A, const void* Wfrag, const float* bias, const float* aux, float* out, int M, int N, int epi,
                               float slope, const int* geom, int KH, int KW, long long a_elems, int a16, void* stream);
extern "C" int nele_conv_span_bf16(const float* A, const void* Wfrag, const float* bias, const float* aux, float* out, int M, int N, int epi,
                                   float slope, const int* geom, int KH, int KW, long long a_elems, void* stream) {
    return conv_span_bf16_impl(A, Wfrag, bias, aux, out, M, N, epi, slope, geom, KH, KW, a_elems, 0, stream);
}
// The same with the input tensor stored as bf16 (same [B][H][W][C] layout and zero border).  Only geometries that run on the span kernel
// (NELE_ERR_UNSUPPORTED otherwise): D.conv5's data gradient, whose input - the pooling gradient - is re-staged once per kernel row.
extern "C" int nele_conv_span_bf16_a16(const void* A16, const void* Wfrag, const float* bias, const float* aux, float* out, int M, int N, int epi,
                                       float slope, const int* geom, int KH, int KW, long long a_elems, void* stream) {
    return conv_span_bf16_impl(reinterpret_cast<const float*>(A16), Wfrag, bias, aux, out, M, N, epi, slope, geom, KH, KW, a_elems, 1, stream);
}
extern "C" int nele_conv_span_bf16_a16_supported(int M, int N, const int* geom, int KH, int KW) {
    ConvGeom g;
    memcpy(&g, geom, sizeof(ConvGeom));
    if (g.Wout >= 32 && tile16_th(g, N, KH, KW)) return 0;
    if (g.Wout >= 32 && conv1d16_sb(g, N, KH, KW, nullptr)) return 0;
    return span16_supported(M, N, g, KH, KW);
}
static int conv_span_bf16_impl(const float* A, const void* Wfrag, const float* bias, const float* aux, float* out, int M, int N, int epi,
                               float slope, const int* geom, int KH, int KW, long long a_elems, int a16, void* stream) {
    NELE_CHECK_ARG(A && Wfrag && out && geom, "nele_conv_span_bf16: null pointer");
    if (!nele_conv_span_bf16_supported(M, N, geom, KH, KW)) return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_conv_span_bf16: geometry not supported");
    NELE_CHECK_ARG(a_elems < 2147483647LL, "nele_conv_span_bf16: input buffer too large for 32-bit offsets");
    Span16Args p;
    p.A = A; p.Wfrag = (const __bf16*)Wfrag; p.bias = bias; p.aux = aux; p.out = out; p.M = M; p.N = N; p.NT = (N + 15) / 16; p.epi = epi;
    p.slope = slope; p.KH = KH; p.KW = KW;
    memcpy(&p.g, geom, sizeof(ConvGeom));
    p.steps_per_seg = (p.g.seglen + 31) / 32;
    p.a16 = a16;
    if (a16 && !nele_conv_span_bf16_a16_supported(M, N, geom, KH, KW))
        return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_conv_span_bf16_a16: this geometry does not run on the span kernel");
    NELE_CHECK_ARG(!(epi == EPI_BIAS || epi == EPI_BIAS_LRELU || epi == EPI_BIAS_EXPTANH) || bias, "nele_conv_span_bf16: epilogue needs bias");
    NELE_CHECK_ARG(epi != EPI_MASK_LRELU_GRAD || aux, "nele_conv_span_bf16: epilogue needs aux");
    hipStream_t s = as_stream(stream);
    const int tile_on = NELE_SWITCH_INT("NELE_CONV_TILE", 1);
    const int th = (tile_on && p.g.Wout >= 32) ? tile16_th(p.g, N, KH, KW) : 0;
    {   // Conv1d / Linear geometry (KH = 1): strip kernel
        size_t clds = 0;
        const int csb = (tile_on && KH == 1 && p.g.Wout >= 32) ? conv1d16_sb(p.g, N, KH, KW, &clds) : 0;
        if (csb) {
            Tile16Args t;
            t.A = A; t.Wfrag = p.Wfrag; t.bias = bias; t.aux = aux; t.out = out; t.N = N; t.NT = p.NT; t.epi = epi; t.slope = slope;
            t.KH = KH; t.KW = KW; t.steps_per_seg = p.g.seglen / 32; t.g = p.g; t.SB = csb; t.dbg = nullptr;
            const int BH = M / p.g.Wout;
            static unsigned long long cattr = 0;            // (per device: a process that switches devices sets the attribute on each)
            if (nele_first_use_on_device(&cattr)) {
#define C1D_ATTR(TN_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_tile16_kernel<TN_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                C1D_ATTR(1); C1D_ATTR(2); C1D_ATTR(3); C1D_ATTR(4);
#undef C1D_ATTR
            }
            const int nchunksN = (N > 64) ? N / 64 : 1, TNsel = (N >= 64) ? 4 : p.NT;
            const int nstrips = ((p.g.Wout + C1D_TW - 1) / C1D_TW) * BH;
            const int walk_on = NELE_SWITCH_INT("NELE_CONV1D_WALK", 1);                       // NELE_CONV1D_WALK=0: one workgroup per (strip, N chunk) at every batch (A/B)
            const size_t ep_bytes = 4 * 16 * 68 * 4;
            const bool walk = walk_on && nchunksN > 1 && nstrips >= 256 && clds + 4096 + ep_bytes <= 158 * 1024;
            t.ntiles = walk ? 1 : nchunksN; t.ncl = walk ? nchunksN : 1; t.SBH = BH;
            if (walk) clds = ((clds + 2047) & ~(size_t)2047) + 4096 + ep_bytes;
            const dim3 grid((unsigned)(t.ntiles * 8 * ((nstrips + 7) / 8)));
            switch (TNsel) {
                case 1: hipLaunchKernelGGL((conv1d_tile16_kernel<1>), grid, dim3(256), clds, s, t); break;
                case 2: hipLaunchKernelGGL((conv1d_tile16_kernel<2>), grid, dim3(256), clds, s, t); break;
                case 3: hipLaunchKernelGGL((conv1d_tile16_kernel<3>), grid, dim3(256), clds, s, t); break;
                default: hipLaunchKernelGGL((conv1d_tile16_kernel<4>), grid, dim3(256), clds, s, t); break;
            }
            NELE_CHECK_LAUNCH("nele_conv_span_bf16(conv1d)");
            return NELE_OK;
        }
    }
    if (th) {
        Tile16Args t;
        t.A = A; t.Wfrag = p.Wfrag; t.bias = bias; t.aux = aux; t.out = out; t.N = N; t.NT = p.NT; t.epi = epi; t.slope = slope;
        t.KH = KH; t.KW = KW; t.steps_per_seg = p.steps_per_seg; t.g = p.g; t.SB = tile16_sb(p.g); t.dbg = nullptr;
        const int B_ = M / (p.g.Hout * p.g.Wout);
        const size_t lds = tile16_lds(p.g, N, KH, KW, th);
        static unsigned long long tattr = 0;            // (per device: a process that switches devices sets the attribute on each)
        if (nele_first_use_on_device(&tattr)) {
#define TILE16_ATTR(TN_, TH_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tile16_kernel<TN_, TH_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
            TILE16_ATTR(1, 8); TILE16_ATTR(2, 8); TILE16_ATTR(3, 8); TILE16_ATTR(4, 8);
            TILE16_ATTR(1, 4); TILE16_ATTR(2, 4); TILE16_ATTR(3, 4); TILE16_ATTR(4, 4);
            TILE16_ATTR(1, 10); TILE16_ATTR(2, 10); TILE16_ATTR(3, 10); TILE16_ATTR(4, 10);
            TILE16_ATTR(1, 11); TILE16_ATTR(2, 11); TILE16_ATTR(3, 11); TILE16_ATTR(4, 11);
            TILE16_ATTR(1, 13); TILE16_ATTR(2, 13); TILE16_ATTR(3, 13); TILE16_ATTR(4, 13);
#undef TILE16_ATTR
        }
        t.ntiles = ((p.g.Wout + TILE16_TW - 1) / TILE16_TW) * ((p.g.Hout + th - 1) / th) * B_;
        const dim3 grid((unsigned)((t.ntiles + 7) / 8 * 8));      // a multiple of 8: every XCD gets the same number of ids
#define TILE16_LAUNCH(TN_, TH_) hipLaunchKernelGGL((conv_tile16_kernel<TN_, TH_>), grid, dim3(256), lds, s, t)
#define TILE16_PICK(TH_) switch (p.NT) { case 1: TILE16_LAUNCH(1, TH_); break; case 2: TILE16_LAUNCH(2, TH_); break; case 3: TILE16_LAUNCH(3, TH_); break; default: TILE16_LAUNCH(4, TH_); break; }
        switch (th) {
            case 13: TILE16_PICK(13); break;
            case 11: TILE16_PICK(11); break;
            case 10: TILE16_PICK(10); break;
            case 8: TILE16_PICK(8); break;
            default: TILE16_PICK(4); break;
        }
#undef TILE16_PICK
#undef TILE16_LAUNCH
        NELE_CHECK_LAUNCH("nele_conv_span_bf16(tile)");
        return NELE_OK;
    }
    const int maxrun = (SPAN16_BM + p.g.Wout - 1) / p.g.Wout + 1;
    const int cp16 = p.g.C + SPAN16_PAD(p.g.C);
    const size_t lds = ((size_t)SPAN16_BM * cp16 + (size_t)maxrun * (KW - 1) * cp16 + 64) * 2;
    const int gx = ((M + SPAN16_BM - 1) / SPAN16_BM + 7) / 8 * 8;     // a multiple of 8: the same number of ids per XCD
    static unsigned long long attr_done = 0;            // (per device: a process that switches devices sets the attribute on each)
    if (nele_first_use_on_device(&attr_done)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_span16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_span16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_span16_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_span16_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    }
    switch (p.NT) {
        case 1: hipLaunchKernelGGL(conv_span16_kernel<1>, dim3(gx), dim3(256), lds, s, p); break;
        case 2: hipLaunchKernelGGL(conv_span16_kernel<2>, dim3(gx), dim3(256), lds, s, p); break;
        case 3: hipLaunchKernelGGL(conv_span16_kernel<3>, dim3(gx), dim3(256), lds, s, p); break;
        default: hipLaunchKernelGGL(conv_span16_kernel<4>, dim3(gx), dim3(256), lds, s, p); break;
    }
    NELE_CHECK_LAUNCH("nele_conv_span_bf16");
    return NELE_OK;
}

extern "C" long long nele_conv_wgrad_workspace_floats(int M, int N, int Ktot, int* splits_out) {
    // enough splits to fill the chip (>= ~1024 blocks) while keeping >= 512 rows per split
    const int kt = (Ktot + WG_BKK - 1) / WG_BKK, nt = (N + WG_BN - 1) / WG_BN;
    int splits = (1024 + kt * nt - 1) / (kt * nt);
    const int max_splits = (M + 511) / 512;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 256) splits = 256;
    if (splits_out) *splits_out = splits;
    int slots = splits;                                    // the 2-D tile kernel uses up to 64 workgroup groups
    if (slots < 64) slots = (max_splits < 64) ? (max_splits > splits ? max_splits : splits) : 64;
    // the DMA kernel (bf16 activations and gradients in memory) runs one workgroup per CU and kernel-row block: up to 256 groups
    if (N <= 64 && slots < 256) slots = max_splits < 256 ? (max_splits > slots ? max_splits : slots) : 256;
    return (long long)slots * ((long long)N * Ktot + N);
}

static int conv_wgrad_impl(const float* A, const float* dOut, float* workspace, long long workspace_floats, int M, int N,
                           const int* geom, int KH, int KW, int Cvalid, float* dW, float* db, int accumulate, int bf16, void* stream);

extern "C" int nele_conv_wgrad(const float* A, const float* dOut, float* workspace, long long workspace_floats, int M, int N,
                               const int* geom, int KH, int KW, int Cvalid, float* dW, float* db, int accumulate, void* stream) {
    return conv_wgrad_impl(A, dOut, workspace, workspace_floats, M, N, geom, KH, KW, Cvalid, dW, db, accumulate, 0, stream);
}

// same with bf16 MFMA operands (float32 accumulation, float32 partials and result)
extern "C" int nele_conv_wgrad_bf16(const float* A, const float* dOut, float* workspace, long long workspace_floats, int M, int N,
                                    const int* geom, int KH, int KW, int Cvalid, float* dW, float* db, int accumulate, void* stream) {
    return conv_wgrad_impl(A, dOut, workspace, workspace_floats, M, N, geom, KH, KW, Cvalid, dW, db, accumulate, 1, stream);
}
static bool wgrad_dma_on() {                               // NELE_WGRAD_DMA=0: the one-kernel-row tile kernel for bf16 operands too (A/B diagnostic)
    const int on = NELE_SWITCH_INT("NELE_WGRAD_DMA", 1);
    return on != 0;
}
static bool wgrad_tile_eligible(int M, int N, const ConvGeom& g, int KH, int KW) {
    const int wt_on = NELE_SWITCH_INT("NELE_WGRAD_TILE", 1);
    const int nkt = (g.seglen + 15) / 16;
    const long long wt_lds = ((long long)WT_TH * (WT_TW + KW - 1) * g.C + WT_SLACK + 256 * WT_NP_OF((N + 15) / 16)) * 2;
    return wt_on && N <= 64 && g.C % 8 == 0 && KW * g.C == g.seglen && nkt <= 28 && g.Wout >= 32 && wt_lds <= 76 * 1024 &&
           M % (g.Hout * g.Wout) == 0;
}
extern "C" int nele_conv_wgrad_bf16_d16_supported(int M, int N, const int* geom, int KH, int KW) {
    ConvGeom g;
    memcpy(&g, geom, sizeof(ConvGeom));
    return wgrad_tile_eligible(M, N, g, KH, KW) ? 1 : 0;
}
// ... and the output gradient stored as bf16 (same layout); only for layers that run on the 2-D tile kernel (NELE_ERR_UNSUPPORTED otherwise)
extern "C" int nele_conv_wgrad_bf16_d16(const float* A, const void* dOut16, float* workspace, long long workspace_floats, int M, int N,
                                        const int* geom, int KH, int KW, int Cvalid, float* dW, float* db, int accumulate, void* stream) {
    return conv_wgrad_impl(A, reinterpret_cast<const float*>(dOut16), workspace, workspace_floats, M, N, geom, KH, KW, Cvalid, dW, db, accumulate, 3, stream);
}

// ... and the input activation stored as bf16 too (model.Discriminator's bf16 mode: every activation and output gradient of layers 2-5 is bf16)
#ifdef C1_PROF
extern "C" int nele_conv1d_prof_read(unsigned long long* out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(c1_prof), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(c1_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

#ifdef WT_PROF
extern "C" int nele_wgrad_tile_prof_read(unsigned long long* out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(wt_prof), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(wt_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" int nele_conv_wgrad_bf16_a16d16(const void* A16, const void* dOut16, float* workspace, long long workspace_floats, int M, int N,
                                           const int* geom, int KH, int KW, int Cvalid, float* dW, float* db, int accumulate, void* stream) {
    return conv_wgrad_impl(reinterpret_cast<const float*>(A16), reinterpret_cast<const float*>(dOut16), workspace, workspace_floats, M, N, geom, KH, KW, Cvalid, dW, db,
                           accumulate, 7, stream);
}

static int conv_wgrad_impl(const float* A, const float* dOut, float* workspace, long long workspace_floats, int M, int N,
                           const int* geom, int KH, int KW, int Cvalid, float* dW, float* db, int accumulate, int bf16, void* stream) {
    NELE_CHECK_ARG(A && dOut && workspace && geom && dW, "nele_conv_wgrad: null pointer");
    const int d16 = (bf16 >> 1) & 1;                       // dOut stored as bf16
    const int a16 = (bf16 >> 2) & 1;                       // the input activation stored as bf16 (with d16 only)
    bf16 &= 1;
    WgradArgs p;
    memcpy(&p.g, geom, sizeof(ConvGeom));
    int st = check_geom("nele_conv_wgrad", p.g, M, N);
    if (st) return st;
    NELE_CHECK_ARG(N % 4 == 0 && p.g.OC % 4 == 0, "nele_conv_wgrad: N and OC must be multiples of 4");
    NELE_CHECK_ARG(KH * KW * p.g.C == p.g.Ktot, "nele_conv_wgrad: KH*KW*C != Ktot");
    int splits = 1;
    const long long need = nele_conv_wgrad_workspace_floats(M, N, p.g.Ktot, &splits);
    if (workspace_floats < need) return nele_set_error(NELE_ERR_WORKSPACE, "nele_conv_wgrad: workspace %lld < %lld floats", workspace_floats, need);
    p.A = A; p.dOut = dOut; p.M = M; p.N = N;
    p.part = workspace;
    p.bpart = db ? workspace + (size_t)splits * N * p.g.Ktot : nullptr;
    hipStream_t s = as_stream(stream);
    // measurement hook: "wgrad_N<N>_K<Ktot>" times one layer's weight gradient including its partial reduction (D.conv5: wgrad_N64_K3888)
    bool prof_w = false;
    if (nele_prof_armed()) {
        char ptag[48];
        snprintf(ptag, sizeof(ptag), "wgrad_N%d_K%d", N, p.g.Ktot);
        prof_w = nele_prof_match(ptag);
        if (prof_w) nele_prof_mark(s);
    }
    // 2-D tile kernel (bf16): one kernel row per workgroup, accumulators in registers over all position tiles
    const int wt_on = NELE_SWITCH_INT("NELE_WGRAD_TILE", 1);
    // Layers too wide for one workgroup's accumulators (N > 64 or more than 28 reduction tiles per kernel row: the generator's Conv1d
    // layers, 256 x 7 x 256) run on the same tile kernel as subs_n x subs_c sub-problems of 64 output x 64 input channels that share
    // every group's position tiles; a Conv1d batch is viewed as ONE image whose rows are the utterances (KH = 1: rows are independent),
    // so that a tile is 4 utterances x 64 frames.  The im2col-gathering kernel read the input KW times per 128-column block of the
    // gradient: 1.44 GB per launch for 130 MB of operands, 240 us per layer at B = 256.
    ConvGeom gt = p.g;                                     // geometry the tile kernel sees
    int Nt = N, subs_n = 1, subs_c = 1, Bt = M / (p.g.Hout * p.g.Wout);
    bool sliced = false;
    if (bf16 && (!d16 || a16) && wt_on && !wgrad_tile_eligible(M, N, p.g, KH, KW) && N % 64 == 0 && p.g.C % 64 == 0 && KW * p.g.C == p.g.seglen &&
        KW * 4 <= 28 && p.g.Wout >= 32 && M % (p.g.Hout * p.g.Wout) == 0) {
        gt.C = 64; gt.seglen = KW * 64; gt.Ktot = KH * KW * 64;
        Nt = 64; subs_n = N / 64; subs_c = p.g.C / 64;
        if (KH == 1 && p.g.H == 1 && p.g.Hout == 1 && p.g.OH == 1 && p.g.ih0 == 0 && p.g.oh0 == 0) { gt.H = Bt; gt.Hout = Bt; gt.OH = Bt; Bt = 1; }
        sliced = (((long long)WT_TH * (WT_TW + KW - 1) * 64 + WT_SLACK + 256 * WT_NP_OF(4)) * 2 <= 76 * 1024);
        if (!sliced) { gt = p.g; Nt = N; subs_n = subs_c = 1; Bt = M / (p.g.Hout * p.g.Wout); }
    }
    const int nkt = (gt.seglen + 15) / 16, NT = (Nt + 15) / 16;
    const long long wt_lds = ((long long)WT_TH * (WT_TW + KW - 1) * gt.C + WT_SLACK + 256 * WT_NP_OF(NT)) * 2;
    const int max_splits = (M + 511) / 512;
    int G = (max_splits < 64) ? (max_splits > splits ? max_splits : splits) : 64;
    if (G < splits) G = splits;
    {   // experiment knob: fewer groups = fewer partials to reduce, but fewer workgroups to pull the memory system
        const int genv = NELE_SWITCH_INT("NELE_WGRAD_GROUPS", 0);
        if (genv > 0 && genv < G) G = genv;
    }
    bool tiled = false;
    // ---- DMA double-buffer kernel: both operands bf16 in memory (D.conv2 .. conv5 in bf16 mode)
    if (a16 && d16 && wgrad_dma_on()) {
        const ConvGeom& gg = p.g;
        const int nkt_ = (gg.seglen + 15) / 16, NT_ = (N + 15) / 16;
        // kernel rows per workgroup: the largest divisor of KH whose accumulators fit (NT x items-per-wave tiles of 4 registers: 176 at most)
        int KR = 0;
        for (int cand = KH; cand >= 1; --cand)
            if (KH % cand == 0 && NT_ * ((cand * nkt_ + 7) / 8) <= 44) { KR = cand; break; }
        const int NJ_ = KR ? (KR * nkt_ + 7) / 8 : 0;
        int TH_ = 0;
        const int th_env = NELE_SWITCH_INT("NELE_WGRAD_DMA_TH", 4);                              // NELE_WGRAD_DMA_TH=2|3|4: cap on the tile height (A/B diagnostic)
        for (int cand = th_env < 4 ? th_env : 4; cand >= 2 && KR; --cand) {
            const long long bufsz = (long long)(cand + KR - 1) * (WD_TW + KW - 1) * gg.C + WT_SLACK + (long long)cand * WD_TW * WT_NP_OF(NT_);
            const int npc = (int)(((long long)(cand + KR - 1) * (WD_TW + KW - 1) * gg.C + 511) / 512), ndp = cand * WD_TW * WT_NP_OF(NT_) / 512;
            if (bufsz * 4 <= 150 * 1024 && npc <= 8 * WD_MAXHP && ndp <= 8 * WD_MAXDP) { TH_ = cand; break; }
        }
        // compiled shapes (NT, NJ, TH): D.conv5 (4, 11, 3 | 2), D.conv4 (3, 13, 4), D.conv3 (2, 4, 4), D.conv2 (1, 1, 4)
        const bool shape_ok = (NT_ == 4 && NJ_ == 11 && (TH_ == 2 || TH_ == 3)) || (NT_ == 3 && NJ_ == 13 && TH_ == 4) || (NT_ == 2 && NJ_ == 4 && TH_ == 4) ||
                              (NT_ == 1 && NJ_ == 1 && TH_ == 4);
        if (KR && TH_ && shape_ok && N <= 64 && N == gg.OC && gg.C % 8 == 0 && KW * gg.C == gg.seglen && gg.oh0 >= 1 && gg.Wout >= 1 &&
            M % (gg.Hout * gg.Wout) == 0 && (long long)(M / (gg.Hout * gg.Wout)) * gg.OH * gg.OW * gg.OC < (1ll << 31) &&
            (long long)gg.OH * gg.OW * gg.OC < (1 << 30)) {
            WgradDmaArgs t;
            t.A = reinterpret_cast<const __bf16*>(A); t.dOut = reinterpret_cast<const __bf16*>(dOut); t.part = workspace;
            t.N = N; t.KH = KH; t.KW = KW; t.KR = KR; t.nkt = nkt_; t.g = gg;
            const int Bt_ = M / (gg.Hout * gg.Wout);
            t.nth = (gg.Hout + TH_ - 1) / TH_; t.ntw = (gg.Wout + WD_TW - 1) / WD_TW; t.ntiles = Bt_ * t.nth * t.ntw;
            static int ncu_d = 0;
            if (!ncu_d) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&ncu_d, hipDeviceAttributeMultiprocessorCount, dev); if (ncu_d <= 0) ncu_d = 256; }
            const int nsub = KH / KR;
            const long long slots_ws = workspace_floats / ((long long)N * gg.Ktot + N);
            int Gd = (ncu_d / nsub) & ~7;                   // one workgroup per CU
            if (Gd > slots_ws) Gd = (int)(slots_ws & ~7ll);
            if (Gd > t.ntiles) Gd = t.ntiles;
            if (Gd >= 1) {
                t.G = Gd;
                t.bpart = db ? workspace + (size_t)Gd * N * gg.Ktot : nullptr;
                const size_t lds = (size_t)((long long)(TH_ + KR - 1) * (WD_TW + KW - 1) * gg.C + WT_SLACK + (long long)TH_ * WD_TW * WT_NP_OF(NT_)) * 4;
                const dim3 grid(nsub * Gd);
#define WD_LAUNCH(NT__, NJ__, TH__) do { static unsigned long long once_ = 0; \
                    if (nele_first_use_on_device(&once_)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_dma_kernel<NT__, NJ__, TH__>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); \
                    hipLaunchKernelGGL((conv_wgrad_dma_kernel<NT__, NJ__, TH__>), grid, dim3(512), lds, s, t); } while (0)
                if (NT_ == 4 && TH_ == 3) WD_LAUNCH(4, 11, 3); else if (NT_ == 4) WD_LAUNCH(4, 11, 2); else if (NT_ == 3) WD_LAUNCH(3, 13, 4); else if (NT_ == 2) WD_LAUNCH(2, 4, 4); else WD_LAUNCH(1, 1, 4);
#undef WD_LAUNCH
                NELE_CHECK_LAUNCH("nele_conv_wgrad(dma)");
                tiled = true;
                splits = Gd;
                p.part = workspace;
                p.bpart = t.bpart;
            }
        }
    }
    if (!tiled && bf16 && (sliced || wgrad_tile_eligible(M, N, p.g, KH, KW))) {
        WgradTileArgs t;
        t.A = A; t.dOut = dOut; t.part = workspace; t.N = Nt; t.B = Bt; t.KH = KH; t.KW = KW; t.d16 = d16;
        t.ics = p.g.C; t.subs_n = subs_n; t.subs_c = subs_c; t.Kfull = p.g.Ktot;
        t.nth = (gt.Hout + WT_TH - 1) / WT_TH; t.ntw = (gt.Wout + WT_TW - 1) / WT_TW; t.ntiles = t.B * t.nth * t.ntw;
        if (G > t.ntiles) G = t.ntiles;
        const int wg_per_group = KH * subs_n * subs_c;
        const int ktw = (nkt + 3) / 4;
        {   // Workgroups that do not fit the chip at once run as a second, nearly empty round: KH * 64 = 576 workgroups on 512 slots (two
            // per CU for D.conv5) took 1.64 x as long as KH * 56 = 504 (3.20 -> 1.95 ms at B = 256).  Pick the group count (a multiple
            // of 8, at most the partial slots of the workspace) that minimises rounds x tiles per workgroup.
            static int ncu = 0;
            static int occ_tab[5][3] = {{0}};
            const int ki = ktw <= 2 ? 0 : (ktw <= 4 ? 1 : 2), ni = NT < 1 ? 1 : (NT > 4 ? 4 : NT);
            if (!ncu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); if (ncu <= 0) ncu = 256; }
            if (!occ_tab[ni][ki]) {
                int occ = 0;
                const void* fn = nullptr;
#define WT_FN(NT_, K_) reinterpret_cast<const void*>(conv_wgrad_tile16_kernel<NT_, K_>)
#define WT_FNK(K_) (ni == 1 ? WT_FN(1, K_) : ni == 2 ? WT_FN(2, K_) : ni == 3 ? WT_FN(3, K_) : WT_FN(4, K_))
                fn = ki == 0 ? WT_FNK(2) : (ki == 1 ? WT_FNK(4) : WT_FNK(7));
#undef WT_FNK
#undef WT_FN
                (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 256, (size_t)wt_lds) != hipSuccess || occ < 1) occ = 1;
                hipFuncAttributes fa;
                if (hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.numRegs > 0) {      // registers: 512 per SIMD lane, one wave of the workgroup per SIMD
                    const int by_regs = 512 / ((fa.numRegs + 7) & ~7);
                    if (by_regs >= 1 && by_regs < occ) occ = by_regs;
                }
                if (NELE_SWITCH_INT("NELE_DEBUG_WGRAD", 0)) fprintf(stderr, "wgrad tile <%d,%d>: occupancy %d (regs %d, lds %lld + %zu)\n", ni, ki, occ, fa.numRegs, wt_lds, fa.sharedSizeBytes);
                occ_tab[ni][ki] = occ;
            }
            const long long slots = (long long)ncu * occ_tab[ni][ki];
            const int gauto = NELE_SWITCH_INT("NELE_WGRAD_AUTOGROUPS", 1);                        // NELE_WGRAD_AUTOGROUPS=0: the fixed 64 groups (A/B diagnostic)
            if (gauto && G >= 16) {
                long long best = -1;
                int bestG = G;
                for (int cand = G & ~7; cand >= 8; cand -= 8) {    // (any count up to the workspace's partial slots works for this kernel)
                    const long long rounds = ((long long)wg_per_group * cand + slots - 1) / slots, per = (t.ntiles + cand - 1) / cand;
                    const long long cost = rounds * per;
                    if (best < 0 || cost < best) { best = cost; bestG = cand; }
                }
                G = bestG;
            }
        }
        t.G = G; t.g = gt;
        t.bpart = db ? workspace + (size_t)G * N * p.g.Ktot : nullptr;
        const dim3 grid(wg_per_group * G);
        static unsigned long long wattr = 0;            // (per device: a process that switches devices sets the attribute on each)
        if (nele_first_use_on_device(&wattr)) {
#define WT_ATTR(NT_, K_) do { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_tile16_kernel<NT_, K_>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
                              (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_tile16_kernel<NT_, K_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
                              (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_tile16_kernel<NT_, K_, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); } while (0)
            WT_ATTR(1, 2); WT_ATTR(2, 2); WT_ATTR(3, 2); WT_ATTR(4, 2); WT_ATTR(1, 4); WT_ATTR(2, 4); WT_ATTR(3, 4); WT_ATTR(4, 4);
            WT_ATTR(1, 7); WT_ATTR(2, 7); WT_ATTR(3, 7); WT_ATTR(4, 7);
#undef WT_ATTR
        }
#define WT_LAUNCH(NT_, K_) do { if (a16) hipLaunchKernelGGL((conv_wgrad_tile16_kernel<NT_, K_, true, true>), grid, dim3(256), (size_t)wt_lds, s, t); \
                                else if (d16) hipLaunchKernelGGL((conv_wgrad_tile16_kernel<NT_, K_, true>), grid, dim3(256), (size_t)wt_lds, s, t); \
                                else hipLaunchKernelGGL((conv_wgrad_tile16_kernel<NT_, K_>), grid, dim3(256), (size_t)wt_lds, s, t); } while (0)
#define WT_PICK(K_) switch (NT) { case 1: WT_LAUNCH(1, K_); break; case 2: WT_LAUNCH(2, K_); break; case 3: WT_LAUNCH(3, K_); break; default: WT_LAUNCH(4, K_); break; }
        if (ktw <= 2) { WT_PICK(2) } else if (ktw <= 4) { WT_PICK(4) } else { WT_PICK(7) }
#undef WT_PICK
#undef WT_LAUNCH
        NELE_CHECK_LAUNCH("nele_conv_wgrad(tile)");
        tiled = true;
        splits = G;
        p.part = workspace;
        p.bpart = t.bpart;
    }
    if (!tiled && d16) return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_conv_wgrad_bf16_d16: this layer does not run on the tile kernel");
    if (a16 && !d16) return nele_set_error(NELE_ERR_INVALID_ARG, "nele_conv_wgrad: a bf16 activation needs a bf16 output gradient");
    if (!tiled && KH == 1 && KW == 1 && p.g.C == 4 && p.g.Ktot == 4 && (N == 4 || N == 8) && p.g.OC % 4 == 0 && splits >= 1) {
        // pointwise layer (D conv1): memory-bound row reduction; as many workgroups as there are partial slots, at most 64
        int sp = splits < 64 ? splits : 64;
        p.bpart = db ? workspace + (size_t)sp * N * p.g.Ktot : nullptr;
        hipLaunchKernelGGL(conv_wgrad_pointwise_kernel, dim3(sp), dim3(1024), 0, s, p);
        NELE_CHECK_LAUNCH("nele_conv_wgrad(pointwise)");
        splits = sp;
        tiled = true;
    }
    if (!tiled) {
    int rps = (M + splits - 1) / splits;
    rps = (rps + WG16_MS - 1) / WG16_MS * WG16_MS;
    p.rows_per_split = rps;
    dim3 grid((p.g.Ktot + WG_BKK - 1) / WG_BKK, splits, (N + WG_BN - 1) / WG_BN);
    if (bf16) hipLaunchKernelGGL(conv_wgrad16_kernel, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), 0, s, p);
    NELE_CHECK_LAUNCH("nele_conv_wgrad");
    }
    const int total = N * p.g.Ktot;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(min(1024, (total + 255) / 256)), dim3(256), 0, s, p.part, p.bpart, splits, N, KH, KW,
                       p.g.C, Cvalid, dW, db, accumulate);
    NELE_CHECK_LAUNCH("nele_conv_wgrad(reduce)");
    if (prof_w) nele_prof_mark(s);
    return NELE_OK;
}

// ---- batched variants: all layers of a model in one launch each (blockIdx.y = job); host pointer / dims arrays as in nele_spectral_norm
#define WB_MAXJ 16
struct PrepJobs { const float* Wt[WB_MAXJ]; const float* sigma[WB_MAXJ]; float* Wf[WB_MAXJ]; float* Wb[WB_MAXJ]; int N[WB_MAXJ], Cv[WB_MAXJ], C[WB_MAXJ], KH[WB_MAXJ], KW[WB_MAXJ]; };
__global__ void weight_prep_batch_kernel(PrepJobs J) {
    const int q = blockIdx.y;
    const float* __restrict__ Wt = J.Wt[q];
    float* __restrict__ Wf = J.Wf[q];
    float* __restrict__ Wb = J.Wb[q];
    const int N = J.N[q], Cvalid = J.Cv[q], C = J.C[q], KH = J.KH[q], KW = J.KW[q];
    const int total = N * Cvalid * KH * KW;
    const float inv = J.sigma[q] ? 1.f / J.sigma[q][0] : 1.f;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int t = idx;
        const int kw = t % KW; t /= KW;
        const int kh = t % KH; t /= KH;
        const int ci = t % Cvalid;
        const int n = t / Cvalid;
        const float v = Wt[idx] * inv;
        Wf[(((size_t)n * KH + kh) * KW + kw) * C + ci] = v;
        if (Wb) Wb[(((size_t)ci * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * N + n] = v;
    }
}
struct Frag16Jobs { const float* Wg[WB_MAXJ]; __bf16* Wfrag[WB_MAXJ]; int N[WB_MAXJ], Ktot[WB_MAXJ], seglen[WB_MAXJ], KH[WB_MAXJ]; };
__global__ void weight_frag16_batch_kernel(Frag16Jobs J) {
    const int q = blockIdx.y;
    const float* __restrict__ Wg = J.Wg[q];
    __bf16* __restrict__ Wfrag = J.Wfrag[q];
    const int N = J.N[q], Ktot = J.Ktot[q], seglen = J.seglen[q], KH = J.KH[q];
    const int NT = (N + 15) / 16, sps = (seglen + 31) / 32;
    const int total = KH * sps * NT * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, t = idx >> 9, j = t % NT, ks = t / NT;
        const int kh = ks / sps, s_ = ks - kh * sps;
        const int n = j * 16 + (lane & 15), kin = s_ * 32 + 8 * (lane >> 4) + e;
        float v = 0.f;
        if (n < N && kin < seglen) v = Wg[(size_t)n * Ktot + (size_t)kh * seglen + kin];
        Wfrag[idx] = (__bf16)v;
    }
}

// ptrs_host [4 * jobs] = (Wt, sigma or null, Wf, Wb or null), dims_host [5 * jobs] = (N, Cvalid, C, KH, KW): nele_weight_prep per job
extern "C" int nele_weight_prep_batch(const void* const* ptrs_host, const int* dims_host, int jobs, void* stream) {
    NELE_CHECK_ARG(ptrs_host && dims_host && jobs >= 1 && jobs <= WB_MAXJ, "nele_weight_prep_batch: bad arguments (jobs <= 16)");
    PrepJobs J;
    int maxtotal = 1;
    for (int q = 0; q < jobs; ++q) {
        J.Wt[q] = (const float*)ptrs_host[4 * q]; J.sigma[q] = (const float*)ptrs_host[4 * q + 1];
        J.Wf[q] = (float*)ptrs_host[4 * q + 2]; J.Wb[q] = (float*)ptrs_host[4 * q + 3];
        J.N[q] = dims_host[5 * q]; J.Cv[q] = dims_host[5 * q + 1]; J.C[q] = dims_host[5 * q + 2]; J.KH[q] = dims_host[5 * q + 3]; J.KW[q] = dims_host[5 * q + 4];
        NELE_CHECK_ARG(J.Wt[q] && J.Wf[q] && J.N[q] > 0 && J.Cv[q] > 0 && J.C[q] >= J.Cv[q], "nele_weight_prep_batch: job %d invalid", q);
        const int total = J.N[q] * J.Cv[q] * J.KH[q] * J.KW[q];
        if (total > maxtotal) maxtotal = total;
    }
    hipLaunchKernelGGL(weight_prep_batch_kernel, dim3(min(256, (maxtotal + 255) / 256), jobs), dim3(256), 0, as_stream(stream), J);
    NELE_CHECK_LAUNCH("nele_weight_prep_batch");
    return NELE_OK;
}

// ptrs_host [2 * jobs] = (Wg [N][Ktot] f32, Wfrag), dims_host [4 * jobs] = (N, Ktot, seglen, KH): nele_weight_prep_frag16 per job
extern "C" int nele_weight_prep_frag16_batch(const void* const* ptrs_host, const int* dims_host, int jobs, void* stream) {
    NELE_CHECK_ARG(ptrs_host && dims_host && jobs >= 1 && jobs <= WB_MAXJ, "nele_weight_prep_frag16_batch: bad arguments (jobs <= 16)");
    Frag16Jobs J;
    long long maxtotal = 1;
    for (int q = 0; q < jobs; ++q) {
        J.Wg[q] = (const float*)ptrs_host[2 * q]; J.Wfrag[q] = (__bf16*)ptrs_host[2 * q + 1];
        J.N[q] = dims_host[4 * q]; J.Ktot[q] = dims_host[4 * q + 1]; J.seglen[q] = dims_host[4 * q + 2]; J.KH[q] = dims_host[4 * q + 3];
        NELE_CHECK_ARG(J.Wg[q] && J.Wfrag[q] && J.N[q] > 0 && J.KH[q] > 0 && J.seglen[q] > 0 && J.KH[q] * J.seglen[q] == J.Ktot[q],
                       "nele_weight_prep_frag16_batch: job %d invalid", q);
        const long long total = nele_weight_frag16_elems(J.N[q], J.seglen[q], J.KH[q]) - 2048;
        if (total > maxtotal) maxtotal = total;
    }
    hipLaunchKernelGGL(weight_frag16_batch_kernel, dim3((unsigned)min(512LL, (maxtotal + 255) / 256), jobs), dim3(256), 0, as_stream(stream), J);
    NELE_CHECK_LAUNCH("nele_weight_prep_frag16_batch");
    return NELE_OK;
}

extern "C" int nele_weight_prep(const float* Wt, const float* sigma, int N, int Cvalid, int C, int KH, int KW, float* Wf, float* Wb,
                                void* stream) {
    NELE_CHECK_ARG(Wt && Wf && N > 0 && Cvalid > 0 && C >= Cvalid, "nele_weight_prep: bad arguments");
    const int total = N * Cvalid * KH * KW;
    hipLaunchKernelGGL(weight_prep_kernel, dim3(min(1024, (total + 255) / 256)), dim3(256), 0, as_stream(stream), Wt, sigma, N, Cvalid, C,
                       KH, KW, Wf, Wb);
    NELE_CHECK_LAUNCH("nele_weight_prep");
    return NELE_OK;
}
