// Discriminator-side kernels other than the conv GEMMs (model.py:101-166) and the optimiser:
//   nele_spectral_norm     : one power iteration (train) or none (eval) + sigma = u^T W v   (torch.nn.utils.spectral_norm, model.py:105-116)
//   nele_sn_grad           : gradient through W/sigma with u,v held constant
//   nele_gap_mlp_fwd       : AdaptiveAvgPool2d(1) + 3 spectral-norm Linear + LeakyReLU + sigmoid (model.py:123-132)
//   nele_gap_mlp_bwd       : MSE-loss gradient back to the last conv activation (into the zero-bordered gradient buffer)
//   nele_mlp_wgrad         : Linear weight / bias gradients, fixed-order over the batch
//   nele_adam_step         : torch.optim.Adam update on a flat parameter buffer            (train_nele.py:89-91)
#include "common.h"

// ------------------------------------------------------------------------------------------ spectral norm
// W [N][K] row-major (weight_orig viewed as (out, -1)); u [N], v [K] updated in place when n_iter == 1.
// v = normalize(W^T u), u = normalize(W v), sigma = u . (W v); normalize(x) = x / max(||x||, 1e-12).
// One 1024-thread block per layer; all layers of a discriminator go in ONE launch (grid = layers).
#define SN_MAXL 8
struct SnLayers {
    const float* W[SN_MAXL];
    float* u[SN_MAXL];
    float* v[SN_MAXL];
    int N[SN_MAXL], K[SN_MAXL];
    float* sigma;   // [layers]
};

__global__ __launch_bounds__(1024) void spectral_norm_kernel(SnLayers L, int n_iter) {
    __shared__ double red[16];
    __shared__ float su[64], swv[64];
    const int l = blockIdx.x;
    const float* __restrict__ W = L.W[l];
    float* u = L.u[l];
    float* v = L.v[l];
    const int N = L.N[l], K = L.K[l];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < N) su[tid] = u[tid];
    __syncthreads();
    if (n_iter > 0) {
        double nrm = 0.0;
        for (int k = tid; k < K; k += 1024) {
            float a = 0.f;
            for (int n = 0; n < N; ++n) a += W[(size_t)n * K + k] * su[n];
            v[k] = a;
            nrm += (double)a * (double)a;
        }
        nrm = block_sum(nrm, red);
        const float inv = 1.f / fmaxf((float)sqrt(nrm), 1e-12f);
        for (int k = tid; k < K; k += 1024) v[k] *= inv;
        __syncthreads();
    }
    for (int n = wave; n < N; n += 16) {
        float a = 0.f;
        for (int k = lane; k < K; k += 64) a += W[(size_t)n * K + k] * v[k];
        a = wave_sum(a);
        if (lane == 0) swv[n] = a;
    }
    __syncthreads();
    if (n_iter > 0) {
        double nrm = 0.0;
        if (tid < N) nrm = (double)swv[tid] * (double)swv[tid];
        nrm = block_sum(nrm, red);
        const float inv = 1.f / fmaxf((float)sqrt(nrm), 1e-12f);
        if (tid < N) {
            su[tid] = swv[tid] * inv;
            u[tid] = su[tid];
        }
        __syncthreads();
    }
    double s = 0.0;
    if (tid < N) s = (double)su[tid] * (double)swv[tid];
    s = block_sum(s, red);
    if (tid == 0) L.sigma[l] = (float)s;
}

// dst (+)= (dWsn - <dWsn, W/sigma> u v^T) / sigma      ([N][K] parameter layout)
// two passes: per-block partial dot products (fixed order), then the element-wise update.
#define SNG_CHUNK 4096
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ dW, const float* __restrict__ W,
                                                     const float* __restrict__ sigma, int total, double* __restrict__ partial) {
    __shared__ double red[8];
    const float inv = 1.f / sigma[0];
    const int i0 = blockIdx.x * SNG_CHUNK;
    double d = 0.0;
    for (int i = i0 + threadIdx.x; i < min(total, i0 + SNG_CHUNK); i += 256) d += (double)dW[i] * (double)(W[i] * inv);
    d = block_sum(d, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = d;
}

__global__ __launch_bounds__(256) void sn_grad_kernel(const float* __restrict__ dW, const float* __restrict__ u, const float* __restrict__ v,
                                                      const float* __restrict__ sigma, const double* __restrict__ partial, int nblk, int N,
                                                      int K, float* __restrict__ dst, int accumulate) {
    const float inv = 1.f / sigma[0];
    double d = 0.0;
    for (int i = 0; i < nblk; ++i) d += partial[i];
    const float dot = (float)d;
    const int total = N * K, i0 = blockIdx.x * SNG_CHUNK;
    for (int i = i0 + threadIdx.x; i < min(total, i0 + SNG_CHUNK); i += 256) {
        const int n = i / K, k = i - n * K;
        const float val = (dW[i] - dot * u[n] * v[k]) * inv;
        dst[i] = accumulate ? dst[i] + val : val;
    }
}

// ------------------------------------------------------------------------------------------ GAP + MLP
struct MlpW {
    const float *w1, *b1, *s1;  // [64][64], [64], sigma
    const float *w2, *b2, *s2;  // [16][64]
    const float *w3, *b3, *s3;  // [nout][16]
};

// Pooling partial sums: grid (GAP_CHUNKS, B), block 256: act [B][P][64] -> part [B][GAP_CHUNKS][64] (float64)
#define GAP_CHUNKS 32
// wvalid (may be NULL): valid output width of each utterance inside a padded batch (T_b - 20 for the last conv layer); columns at or
// behind it come from zero padding and take no part in the pooling (the reference pools each utterance over its own extent).
__global__ __launch_bounds__(256) void gap_partial_kernel(const float* __restrict__ act, int P, double* __restrict__ part, int Wout,
                                                          const int* __restrict__ wvalid) {
    // thread = (position group g of 16, channel quad q): 16-byte loads, four float64 sums per thread (4-byte loads ran at 3.1 TB/s)
    __shared__ double sp[16][64];
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, q = tid & 15, g = tid >> 4;
    const int per = (P + GAP_CHUNKS - 1) / GAP_CHUNKS, p0 = ch * per, p1 = min(P, p0 + per);
    const float* a = act + (size_t)b * P * 64 + 4 * q;
    const int wv = wvalid ? min(wvalid[b], Wout) : Wout;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int pos = p0 + g; pos < p1; pos += 16)
        if (pos % Wout < wv) {
            const float4 v = *reinterpret_cast<const float4*>(a + (size_t)pos * 64);
            s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
        }
    sp[g][4 * q] = s0; sp[g][4 * q + 1] = s1; sp[g][4 * q + 2] = s2; sp[g][4 * q + 3] = s3;
    __syncthreads();
    if (tid < 64) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += sp[r][tid];
        part[((size_t)b * GAP_CHUNKS + ch) * 64 + tid] = t;
    }
}

// One block per utterance: finish the pooling from the partial sums, then the 3-layer head.
__global__ __launch_bounds__(256) void gap_mlp_fwd_kernel(const double* __restrict__ part, int P, MlpW w, int nout, float slope,
                                                          float* __restrict__ pooled, float* __restrict__ h1, float* __restrict__ h2,
                                                          float* __restrict__ score, int Wout, const int* __restrict__ wvalid, int nparts) {
    __shared__ float sp[64], sh1[64], sh2[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (wvalid) P = (P / Wout) * min(wvalid[b], Wout);
    {   // partial sums: four runs of consecutive partials per channel (one per wave), added in run order - the same association for every batch
        __shared__ double sq[4][64];
        const int c = tid & 63, q = tid >> 6, per = (nparts + 3) >> 2;
        const int c0 = q * per, c1 = min(nparts, c0 + per);
        double s = 0.0;
        for (int ch = c0; ch < c1; ++ch) s += part[((size_t)b * nparts + ch) * 64 + c];
        sq[q][c] = s;
        __syncthreads();
        if (tid < 64) {
            const double t = (sq[0][tid] + sq[1][tid]) + (sq[2][tid] + sq[3][tid]);
            const float m = (float)(t / (double)P);
            sp[tid] = m;
            pooled[(size_t)b * 64 + tid] = m;
        }
    }
    __syncthreads();
    if (tid < 64) {
        const float inv = 1.f / w.s1[0];
        float z = 0.f;
        for (int k = 0; k < 64; ++k) z += w.w1[tid * 64 + k] * inv * sp[k];
        z += w.b1[tid];
        z = z > 0.f ? z : slope * z;
        sh1[tid] = z;
        h1[(size_t)b * 64 + tid] = z;
    }
    __syncthreads();
    if (tid < 16) {
        const float inv = 1.f / w.s2[0];
        float z = 0.f;
        for (int k = 0; k < 64; ++k) z += w.w2[tid * 64 + k] * inv * sh1[k];
        z += w.b2[tid];
        z = z > 0.f ? z : slope * z;
        sh2[tid] = z;
        h2[(size_t)b * 16 + tid] = z;
    }
    __syncthreads();
    if (tid < nout) {
        const float inv = 1.f / w.s3[0];
        float z = 0.f;
        for (int k = 0; k < 16; ++k) z += w.w3[tid * 16 + k] * inv * sh2[k];
        z += w.b3[tid];
        score[(size_t)b * nout + tid] = 1.f / (1.f + expf(-z));
    }
}

// dscore [B][nout] (gradient wrt the sigmoid outputs) -> dz3, dz2, dz1 (pre-activation gradients) and
// dpooled; one block (64 threads) per utterance.
__global__ __launch_bounds__(64) void mlp_bwd_kernel(const float* __restrict__ dscore, const float* __restrict__ score,
                                                     const float* __restrict__ h1, const float* __restrict__ h2, MlpW w, int nout,
                                                     float slope, float* __restrict__ dz3, float* __restrict__ dz2,
                                                     float* __restrict__ dz1, float* __restrict__ dpooled) {
    __shared__ float s3[4], s2[16], s1[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < nout) {
        const float sc = score[(size_t)b * nout + tid];
        const float d = dscore[(size_t)b * nout + tid] * sc * (1.f - sc);
        s3[tid] = d;
        dz3[(size_t)b * nout + tid] = d;
    }
    __syncthreads();
    if (tid < 16) {
        const float inv = 1.f / w.s3[0];
        float d = 0.f;
        for (int n = 0; n < nout; ++n) d += w.w3[n * 16 + tid] * inv * s3[n];
        d *= (h2[(size_t)b * 16 + tid] > 0.f ? 1.f : slope);
        s2[tid] = d;
        dz2[(size_t)b * 16 + tid] = d;
    }
    __syncthreads();
    {
        const float inv = 1.f / w.s2[0];
        float d = 0.f;
        for (int n = 0; n < 16; ++n) d += w.w2[n * 64 + tid] * inv * s2[n];
        d *= (h1[(size_t)b * 64 + tid] > 0.f ? 1.f : slope);
        s1[tid] = d;
        dz1[(size_t)b * 64 + tid] = d;
    }
    __syncthreads();
    {
        const float inv = 1.f / w.s1[0];
        float d = 0.f;
        for (int n = 0; n < 64; ++n) d += w.w1[n * 64 + tid] * inv * s1[n];
        dpooled[(size_t)b * 64 + tid] = d;
    }
}

// d act[b][pos][c] = dpooled[b][c] / P * lrelu'(act), written into the zero-bordered gradient buffer
// [B][OH][OW][64] at offset (oh0, ow0).
// T = float, or __bf16: the consumers (the layer's data- and weight-gradient kernels) round their operands to bf16 anyway, and the data
// gradient re-stages this buffer once per kernel row - half the bytes and no conversion there (see conv_span16_kernel).
typedef __bf16 disc_bf16x4 __attribute__((ext_vector_type(4)));
template <typename T, typename TA = float>
__global__ void gap_bwd_kernel(const float* __restrict__ dpooled, const TA* __restrict__ act, int Hout, int Wout, int OH, int OW,
                               int oh0, int ow0, float slope, T* __restrict__ gbuf, const int* __restrict__ wvalid) {
    const int b = blockIdx.y, P = Hout * Wout;
    const int wv = wvalid ? min(wvalid[b], Wout) : Wout;
    const float invP = 1.f / (float)(Hout * wv);
    // four channels of a position per thread: one 16-byte load, one 8- or 16-byte store
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P * 16; i += gridDim.x * blockDim.x) {
        const int pos = i >> 4, c = (i & 15) * 4;
        const int ho = pos / Wout, wo = pos - ho * Wout;
        float4 a;
        if constexpr (sizeof(TA) == 2) {                       // bf16 activation (the LeakyReLU mask only needs its sign): one 8-byte load
            const disc_bf16x4 h = *reinterpret_cast<const disc_bf16x4*>(act + ((size_t)b * P + pos) * 64 + c);
            a = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        } else {
            a = *reinterpret_cast<const float4*>(act + ((size_t)b * P + pos) * 64 + c);
        }
        const float4 dp = *reinterpret_cast<const float4*>(dpooled + (size_t)b * 64 + c);
        const bool in = wo < wv;
        const float d0 = in ? dp.x * invP * (a.x > 0.f ? 1.f : slope) : 0.f, d1 = in ? dp.y * invP * (a.y > 0.f ? 1.f : slope) : 0.f;
        const float d2 = in ? dp.z * invP * (a.z > 0.f ? 1.f : slope) : 0.f, d3 = in ? dp.w * invP * (a.w > 0.f ? 1.f : slope) : 0.f;
        T* o = gbuf + (((size_t)b * OH + ho + oh0) * OW + wo + ow0) * 64 + c;
        o[0] = (T)d0; o[1] = (T)d1; o[2] = (T)d2; o[3] = (T)d3;
    }
}

// dW[n][k] = sum_b dz[b][n] * x[b][k] (sigma-normalised weight gradient; nele_sn_grad maps it to
// weight_orig), db[n] = sum_b dz[b][n].  One thread per (n,k); batch summed in order.
__global__ void mlp_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x, int B, int N, int K, float* __restrict__ dW,
                                 float* __restrict__ db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // (the batch loop in groups of 8 with the loads issued together: one memory latency per group instead of one per item - this
    //  16-wave kernel took 110 us at B = 256; the sums stay in batch order)
    if (i < N * K) {
        const int n = i / K, k = i - n * K;
        float s = 0.f;
        int b = 0;
        for (; b + 8 <= B; b += 8) {
            float dv[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { dv[u] = dz[(size_t)(b + u) * N + n]; xv[u] = x[(size_t)(b + u) * K + k]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += dv[u] * xv[u];
        }
        for (; b < B; ++b) s += dz[(size_t)b * N + n] * x[(size_t)b * K + k];
        dW[i] = s;
    }
    if (i < N) {
        float s = 0.f;
        int b = 0;
        for (; b + 8 <= B; b += 8) {
            float dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) dv[u] = dz[(size_t)(b + u) * N + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += dv[u];
        }
        for (; b < B; ++b) s += dz[(size_t)b * N + i];
        db[i] = s;
    }
}

// ------------------------------------------------------------------------------------------ Adam
// torch.optim.Adam (no amsgrad, no weight decay): one fused pass over the flat buffers.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt) {
    const float step_size = lr / bc1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// Guarded update: a non-finite gradient (NaN / inf metric target, poisoned eigen-decomposition, overflow) must not reach the
// parameters or the Adam moments - one such step would make every later step NaN.  grad_nonfinite_kernel marks guard[0] = step when
// any element of g is not finite (atomicMax: the stored value is the same whichever thread wins); adam_guarded_kernel does nothing
// when guard[0] == step and counts the skipped step in guard[1].  No host synchronisation: the host reads guard[1] when it wants to.
__global__ void grad_nonfinite_kernel(const float* __restrict__ g, size_t n, int* guard, int step) {
    bool bad = false;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned u = __float_as_uint(g[i]);
        bad |= ((u & 0x7f800000u) == 0x7f800000u);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicMax(&guard[0], step);
}

__global__ void adam_guarded_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                                    float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt, int* guard, int step) {
    if (guard[0] == step) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&guard[1], 1);
        return;
    }
    // torch.optim.Adam's bias correction counts the updates that HAPPENED: a masked step never existed for it.  bc1 / bc2_sqrt come from
    // the host for update number `step`; when steps were masked before this one (guard[1] > 0, a rare event) the corrections are recomputed
    // here for update number step - guard[1] (until round 5 the masked steps were counted: a documented deviation, now gone).
    const int skipped = guard[1];
    if (skipped > 0) {
        const double eff = (double)(step - skipped);
        bc1 = (float)(1.0 - pow((double)beta1, eff));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, eff));
    }
    const float step_size = lr / bc1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// ------------------------------------------------------------------------------------------ C ABI
// ptrs_host: HOST array of 3*layers device pointers {W, u, v} per layer; dims_host: HOST array of 2*layers ints {N, K};
// sigma: device [layers].
extern "C" int nele_spectral_norm(const void* const* ptrs_host, const int* dims_host, int layers, float* sigma, int n_iter, void* stream) {
    NELE_CHECK_ARG(ptrs_host && dims_host && sigma && layers >= 1 && layers <= SN_MAXL, "nele_spectral_norm: bad arguments (layers <= 8)");
    SnLayers L;
    for (int l = 0; l < layers; ++l) {
        L.W[l] = (const float*)ptrs_host[3 * l];
        L.u[l] = (float*)ptrs_host[3 * l + 1];
        L.v[l] = (float*)ptrs_host[3 * l + 2];
        L.N[l] = dims_host[2 * l];
        L.K[l] = dims_host[2 * l + 1];
        NELE_CHECK_ARG(L.W[l] && L.u[l] && L.v[l] && L.N[l] > 0 && L.N[l] <= 64 && L.K[l] > 0, "nele_spectral_norm: layer %d invalid (N must be <= 64)", l);
    }
    L.sigma = sigma;
    hipLaunchKernelGGL(spectral_norm_kernel, dim3(layers), dim3(1024), 0, as_stream(stream), L, n_iter);
    NELE_CHECK_LAUNCH("nele_spectral_norm");
    return NELE_OK;
}

// scratch: float64 [nele_sn_grad_scratch_doubles(N*K)]
extern "C" int nele_sn_grad_scratch_doubles(int total) { return (total + SNG_CHUNK - 1) / SNG_CHUNK; }

extern "C" int nele_sn_grad(const float* dW, const float* W, const float* u, const float* v, const float* sigma, int N, int K,
                            float* dst, int accumulate, double* scratch, void* stream) {
    NELE_CHECK_ARG(dW && W && u && v && sigma && dst && scratch && N > 0 && K > 0, "nele_sn_grad: bad arguments");
    const int total = N * K, nblk = (total + SNG_CHUNK - 1) / SNG_CHUNK;
    hipLaunchKernelGGL(sn_dot_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), dW, W, sigma, total, scratch);
    hipLaunchKernelGGL(sn_grad_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), dW, u, v, sigma, scratch, nblk, N, K, dst, accumulate);
    NELE_CHECK_LAUNCH("nele_sn_grad");
    return NELE_OK;
}

// mlp: 9 device pointers {w1,b1,sigma1,w2,b2,sigma2,w3,b3,sigma3}
// scratch: float64 [B][32][64] pooling partial sums
extern "C" int nele_gap_mlp_fwd_var(const float* act, int B, int P, int Wout, const int* wvalid, const float* const* mlp_host, int nout, float slope,
                                    float* pooled, float* h1, float* h2, float* score, double* scratch, void* stream) {
    NELE_CHECK_ARG(act && mlp_host && pooled && h1 && h2 && score && scratch && B > 0 && P > 0 && nout >= 1 && nout <= 4,
                   "nele_gap_mlp_fwd: bad arguments");
    NELE_CHECK_ARG(Wout > 0 && P % Wout == 0, "nele_gap_mlp_fwd: P=%d is not a multiple of the output width %d", P, Wout);
    const float* const* mlp = mlp_host;
    MlpW w = {mlp[0], mlp[1], mlp[2], mlp[3], mlp[4], mlp[5], mlp[6], mlp[7], mlp[8]};
    hipLaunchKernelGGL(gap_partial_kernel, dim3(GAP_CHUNKS, B), dim3(256), 0, as_stream(stream), act, P, scratch, Wout, wvalid);
    hipLaunchKernelGGL(gap_mlp_fwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), scratch, P, w, nout, slope, pooled, h1, h2, score, Wout, wvalid, GAP_CHUNKS);
    NELE_CHECK_LAUNCH("nele_gap_mlp_fwd");
    return NELE_OK;
}
extern "C" int nele_gap_mlp_fwd(const float* act, int B, int P, const float* const* mlp_host, int nout, float slope, float* pooled, float* h1,
                                float* h2, float* score, double* scratch, void* stream) {
    return nele_gap_mlp_fwd_var(act, B, P, P, nullptr, mlp_host, nout, slope, pooled, h1, h2, score, scratch, stream);
}

// The head alone, on pooled partial sums the producing conv kernel already wrote (nele_conv16_gap): part [B][nparts][64] float64.
extern "C" int nele_gap_mlp_fwd_parts(const double* part, int nparts, int B, int P, int Wout, const int* wvalid, const float* const* mlp_host, int nout,
                                      float slope, float* pooled, float* h1, float* h2, float* score, void* stream) {
    NELE_CHECK_ARG(part && nparts > 0 && mlp_host && pooled && h1 && h2 && score && B > 0 && P > 0, "nele_gap_mlp_fwd_parts: bad arguments");
    NELE_CHECK_ARG(nout >= 1 && nout <= 4, "nele_gap_mlp_fwd_parts: nout");
    NELE_CHECK_ARG(Wout > 0 && P % Wout == 0, "nele_gap_mlp_fwd_parts: P=%d is not a multiple of the output width %d", P, Wout);
    const float* const* mlp = mlp_host;
    MlpW w = {mlp[0], mlp[1], mlp[2], mlp[3], mlp[4], mlp[5], mlp[6], mlp[7], mlp[8]};
    hipLaunchKernelGGL(gap_mlp_fwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), part, P, w, nout, slope, pooled, h1, h2, score, Wout, wvalid, nparts);
    NELE_CHECK_LAUNCH("nele_gap_mlp_fwd_parts");
    return NELE_OK;
}

extern "C" int nele_gap_mlp_bwd_var(const float* dscore, const float* score, const float* h1, const float* h2, const float* act,
                                    const float* const* mlp_host, int nout, float slope, int B, int Hout, int Wout, const int* wvalid, int OH, int OW,
                                    int oh0, int ow0, float* dz3, float* dz2, float* dz1, float* dpooled, float* gbuf, void* stream);
extern "C" int nele_gap_mlp_bwd(const float* dscore, const float* score, const float* h1, const float* h2, const float* act,
                                const float* const* mlp_host, int nout, float slope, int B, int Hout, int Wout, int OH, int OW, int oh0,
                                int ow0, float* dz3, float* dz2, float* dz1, float* dpooled, float* gbuf, void* stream) {
    return nele_gap_mlp_bwd_var(dscore, score, h1, h2, act, mlp_host, nout, slope, B, Hout, Wout, nullptr, OH, OW, oh0, ow0, dz3, dz2, dz1, dpooled,
                                gbuf, stream);
}
static int gap_mlp_bwd_impl(const float* dscore, const float* score, const float* h1, const float* h2, const void* act,
                            const float* const* mlp_host, int nout, float slope, int B, int Hout, int Wout, const int* wvalid, int OH, int OW,
                            int oh0, int ow0, float* dz3, float* dz2, float* dz1, float* dpooled, void* gbuf, int gbuf_bf16, void* stream,
                            int act_bf16 = 0) {
    NELE_CHECK_ARG(dscore && score && h1 && h2 && mlp_host && dz3 && dz2 && dz1 && dpooled && B > 0, "nele_gap_mlp_bwd: bad arguments");
    const float* const* mlp = mlp_host;
    MlpW w = {mlp[0], mlp[1], mlp[2], mlp[3], mlp[4], mlp[5], mlp[6], mlp[7], mlp[8]};
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(mlp_bwd_kernel, dim3(B), dim3(64), 0, s, dscore, score, h1, h2, w, nout, slope, dz3, dz2, dz1, dpooled);
    NELE_CHECK_LAUNCH("nele_gap_mlp_bwd(mlp)");
    if (gbuf) {
        NELE_CHECK_ARG(act, "nele_gap_mlp_bwd: act required for the pooling gradient");
        const int P = Hout * Wout;
        if (gbuf_bf16 && act_bf16) hipLaunchKernelGGL((gap_bwd_kernel<__bf16, __bf16>), dim3(min(512, (P * 16 + 255) / 256), B), dim3(256), 0, s, dpooled,
                                                      (const __bf16*)act, Hout, Wout, OH, OW, oh0, ow0, slope, (__bf16*)gbuf, wvalid);
        else if (gbuf_bf16) hipLaunchKernelGGL((gap_bwd_kernel<__bf16, float>), dim3(min(512, (P * 16 + 255) / 256), B), dim3(256), 0, s, dpooled,
                                               (const float*)act, Hout, Wout, OH, OW, oh0, ow0, slope, (__bf16*)gbuf, wvalid);
        else hipLaunchKernelGGL((gap_bwd_kernel<float, float>), dim3(min(512, (P * 16 + 255) / 256), B), dim3(256), 0, s, dpooled, (const float*)act, Hout,
                                Wout, OH, OW, oh0, ow0, slope, (float*)gbuf, wvalid);
        NELE_CHECK_LAUNCH("nele_gap_mlp_bwd(gap)");
    }
    return NELE_OK;
}
extern "C" int nele_gap_mlp_bwd_var(const float* dscore, const float* score, const float* h1, const float* h2, const float* act,
                                    const float* const* mlp_host, int nout, float slope, int B, int Hout, int Wout, const int* wvalid, int OH, int OW,
                                    int oh0, int ow0, float* dz3, float* dz2, float* dz1, float* dpooled, float* gbuf, void* stream) {
    return gap_mlp_bwd_impl(dscore, score, h1, h2, act, mlp_host, nout, slope, B, Hout, Wout, wvalid, OH, OW, oh0, ow0, dz3, dz2, dz1, dpooled, gbuf, 0,
                            stream);
}
extern "C" int nele_gap_mlp_bwd_var16(const float* dscore, const float* score, const float* h1, const float* h2, const float* act,
                                      const float* const* mlp_host, int nout, float slope, int B, int Hout, int Wout, const int* wvalid, int OH, int OW,
                                      int oh0, int ow0, float* dz3, float* dz2, float* dz1, float* dpooled, void* gbuf16, void* stream) {
    return gap_mlp_bwd_impl(dscore, score, h1, h2, act, mlp_host, nout, slope, B, Hout, Wout, wvalid, OH, OW, oh0, ow0, dz3, dz2, dz1, dpooled, gbuf16,
                            1, stream);
}

// ... with the last conv layer's activation as bf16 too (nele_conv16_gap's out16: the mask only needs the sign)
extern "C" int nele_gap_mlp_bwd_var16a(const float* dscore, const float* score, const float* h1, const float* h2, const void* act16,
                                       const float* const* mlp_host, int nout, float slope, int B, int Hout, int Wout, const int* wvalid, int OH, int OW,
                                       int oh0, int ow0, float* dz3, float* dz2, float* dz1, float* dpooled, void* gbuf16, void* stream) {
    return gap_mlp_bwd_impl(dscore, score, h1, h2, act16, mlp_host, nout, slope, B, Hout, Wout, wvalid, OH, OW, oh0, ow0, dz3, dz2, dz1, dpooled, gbuf16,
                            1, stream, 1);
}

extern "C" int nele_mlp_wgrad(const float* dz, const float* x, int B, int N, int K, float* dW, float* db, void* stream) {
    NELE_CHECK_ARG(dz && x && dW && db && B > 0 && N > 0 && K > 0, "nele_mlp_wgrad: bad arguments");
    hipLaunchKernelGGL(mlp_wgrad_kernel, dim3((N * K + 255) / 256), dim3(256), 0, as_stream(stream), dz, x, B, N, K, dW, db);
    NELE_CHECK_LAUNCH("nele_mlp_wgrad");
    return NELE_OK;
}

extern "C" int nele_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                              int step, void* stream) {
    NELE_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "nele_adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)min((long long)2048, (n + 255) / 256)), dim3(256), 0, as_stream(stream), p, g, m, v,
                       (size_t)n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
    NELE_CHECK_LAUNCH("nele_adam_step");
    return NELE_OK;
}

// guard: device int[2], zero-initialised by the caller once: {last step with a non-finite gradient, number of skipped steps}.
extern "C" int nele_adam_step_guarded(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                                      int step, int* guard, void* stream) {
    NELE_CHECK_ARG(p && g && m && v && guard && n > 0 && step >= 1, "nele_adam_step_guarded: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const dim3 grid((unsigned)min((long long)2048, (n + 255) / 256));
    hipLaunchKernelGGL(grad_nonfinite_kernel, grid, dim3(256), 0, as_stream(stream), g, (size_t)n, guard, step);
    hipLaunchKernelGGL(adam_guarded_kernel, grid, dim3(256), 0, as_stream(stream), p, g, m, v, (size_t)n, lr, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2), guard, step);
    NELE_CHECK_LAUNCH("nele_adam_step_guarded");
    return NELE_OK;
}
