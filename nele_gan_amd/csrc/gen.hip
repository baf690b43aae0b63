// Generator-side kernels other than the GEMMs (model.py:43-98, 168-205; train_nele.py:130-146):
//   nele_g_pack          : cat(x,y) -> time-padded channels-last conv input        (model.py:85-86)
//   nele_cln_fwd / _bwd  : cumulative layer norm + LeakyReLU(0.3)                   (model.py:180-205, 88-91)
//   nele_exptanh_bwd     : gradient of exp(3.2*tanh(o))                             (model.py:98)
//   nele_energy_norm_fwd / _bwd : utterance-level energy normalisation + D input    (train_nele.py:133-146)
//   nele_colsum          : fixed-order reduction of per-utterance partials
#include "common.h"

// ------------------------------------------------------------------------------------------ pack
__global__ void g_pack_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int T, int pad) {
    // out [B][T+pad][128]; rows < pad stay zero (buffer is zero-initialised once by the host)
    const int b = blockIdx.y, t = blockIdx.x, c = threadIdx.x;  // 128 threads
    const float v = (c < 64) ? x[((size_t)b * T + t) * 64 + c] : y[((size_t)b * T + t) * 64 + c - 64];
    out[((size_t)b * (T + pad) + pad + t) * 128 + c] = v;
}

// ------------------------------------------------------------------------------------------ cLN
// cum_mean = S/n, cum_var = (Q - 2 mean S)/n + mean^2, n = C (t+1), eps = 1e-8 (model.py:188-199).
//   stats kernel : wave per frame, s_t = sum_c y, q_t = sum_c y^2 (float64)              grid (ceil(T/4), B)
//   apply kernel : block-wide float64 prefix sum over the T frame sums (every block redoes it: T <= 1024 adds),
//                  then normalise + affine + LeakyReLU a chunk of CLN_FR frames               grid (ceil(T/CLN_FR), B)
#define CLN_MAX_T 1024
#define CLN_FR 16

__global__ __launch_bounds__(256) void cln_stats_kernel(const float* __restrict__ Y, double* __restrict__ stats, int T, int C) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const float* y = Y + ((size_t)b * T + t) * C;
    double s = 0.0, q = 0.0;
    for (int c = lane; c < C; c += 64) {
        const float v = y[c];
        s += (double)v;
        q += (double)v * (double)v;
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) { stats[((size_t)b * T + t) * 2] = s; stats[((size_t)b * T + t) * 2 + 1] = q; }
}

// inclusive prefix sums of two float64 sequences of length T (<= 1024) held in LDS; 256 threads, 4 items each
__device__ __forceinline__ void block_scan2(double* a, double* b2, int T, bool reverse) {
    __shared__ double ta[256], tb[256];
    const int tid = threadIdx.x;
    const int per = (T + 255) / 256;
    const int lo = tid * per, hi = min(T, lo + per);
    double sa = 0.0, sb = 0.0;
    for (int i = lo; i < hi; ++i) {
        const int j = reverse ? T - 1 - i : i;
        sa += a[j]; sb += b2[j];
        a[j] = sa; b2[j] = sb;
    }
    ta[tid] = sa; tb[tid] = sb;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const double va = (tid >= o) ? ta[tid - o] : 0.0, vb = (tid >= o) ? tb[tid - o] : 0.0;
        __syncthreads();
        ta[tid] += va; tb[tid] += vb;
        __syncthreads();
    }
    const double oa = (tid > 0) ? ta[tid - 1] : 0.0, ob = (tid > 0) ? tb[tid - 1] : 0.0;
    for (int i = lo; i < hi; ++i) {
        const int j = reverse ? T - 1 - i : i;
        a[j] += oa; b2[j] += ob;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void cln_fwd_kernel(const float* __restrict__ Y, const double* __restrict__ stats,
                                                      const float* __restrict__ gain, const float* __restrict__ bias,
                                                      float* __restrict__ out, float* __restrict__ mean, float* __restrict__ rstd, int T,
                                                      int C, int pad, float slope) {
    __shared__ double ss[CLN_MAX_T], qq[CLN_MAX_T];
    __shared__ float smean[CLN_FR], srstd[CLN_FR];
    const int b = blockIdx.y, tid = threadIdx.x, t0 = blockIdx.x * CLN_FR;
    for (int t = tid; t < T; t += 256) { ss[t] = stats[((size_t)b * T + t) * 2]; qq[t] = stats[((size_t)b * T + t) * 2 + 1]; }
    __syncthreads();
    block_scan2(ss, qq, T, false);
    if (tid < CLN_FR && t0 + tid < T) {
        const int t = t0 + tid;
        const double n = (double)C * (double)(t + 1);
        const double m = ss[t] / n;
        const double var = (qq[t] - 2.0 * m * ss[t]) / n + m * m;
        smean[tid] = (float)m;
        srstd[tid] = (float)(1.0 / sqrt(var + 1e-8));
        mean[(size_t)b * T + t] = smean[tid];
        rstd[(size_t)b * T + t] = srstd[tid];
    }
    __syncthreads();
    const int nfr = min(CLN_FR, T - t0);
    const float* y = Y + ((size_t)b * T + t0) * C;
    float* o = out + ((size_t)b * (T + pad) + pad + t0) * C;
    const int total = nfr * C;
    for (int i = tid; i < total; i += 256) {
        const int f = i / C, c = i - f * C;
        float v = (y[i] - smean[f]) * srstd[f] * gain[c] + bias[c];
        v = v > 0.f ? v : slope * v;
        o[i] = v;
    }
}

// Backward of out = lrelu(cLN(y)):  with xh = (y-mu_t) r_t, dxh = dAct * lrelu'(.) * gain_c,
//   A_t = sum_c dxh, B_t = sum_c dxh*xh, n_t = C(t+1)
//   dL/dS_t = (-A_t r_t + B_t mu_t r_t^2) / n_t,   dL/dQ_t = -B_t r_t^2 / (2 n_t)
//   RS_t = sum_{t'>=t} dL/dS_t', RQ_t likewise;   dy[t,c] = dxh r_t + RS_t + 2 y RQ_t.
// dY is written into an END-padded buffer [B][T+pade][C] (rows >= T stay zero) for the data-gradient GEMM.
__global__ __launch_bounds__(256) void cln_bwd_stats_kernel(const float* __restrict__ dAct, const float* __restrict__ Y,
                                                            const float* __restrict__ gain, const float* __restrict__ bias,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            double* __restrict__ ab, int T, int C, float slope) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const float* y = Y + ((size_t)b * T + t) * C;
    const float* da = dAct + ((size_t)b * T + t) * C;
    const float mu = mean[(size_t)b * T + t], r = rstd[(size_t)b * T + t];
    double a = 0.0, bb = 0.0;
    for (int c = lane; c < C; c += 64) {
        const float xh = (y[c] - mu) * r;
        const float x = xh * gain[c] + bias[c];
        const float d = da[c] * (x > 0.f ? 1.f : slope) * gain[c];
        a += (double)d;
        bb += (double)d * (double)xh;
    }
    a = wave_sum(a);
    bb = wave_sum(bb);
    if (lane == 0) { ab[((size_t)b * T + t) * 2] = a; ab[((size_t)b * T + t) * 2 + 1] = bb; }
}

// grid (ceil(T/CLN_FR), B); dgain/dbias partials per (utterance, chunk): [B*nchunks][C]
__global__ __launch_bounds__(256) void cln_bwd_kernel(const float* __restrict__ dAct, const float* __restrict__ Y,
                                                      const double* __restrict__ ab, const float* __restrict__ gain,
                                                      const float* __restrict__ bias, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, float* __restrict__ dY,
                                                      float* __restrict__ dgain_part, float* __restrict__ dbias_part, int T, int C, int pade,
                                                      float slope, __bf16* __restrict__ dY16) {
    __shared__ double ds[CLN_MAX_T], dq[CLN_MAX_T];
    const int b = blockIdx.y, tid = threadIdx.x, t0 = blockIdx.x * CLN_FR;
    for (int t = tid; t < T; t += 256) {
        const double n = (double)C * (double)(t + 1), r = (double)rstd[(size_t)b * T + t], mu = (double)mean[(size_t)b * T + t];
        const double A = ab[((size_t)b * T + t) * 2], Bv = ab[((size_t)b * T + t) * 2 + 1];
        ds[t] = (-A * r + Bv * mu * r * r) / n;
        dq[t] = -Bv * r * r / (2.0 * n);
    }
    __syncthreads();
    block_scan2(ds, dq, T, true);   // suffix sums
    const int nfr = min(CLN_FR, T - t0);
    const float* y = Y + ((size_t)b * T + t0) * C;
    const float* da = dAct + ((size_t)b * T + t0) * C;
    float* o = dY ? dY + ((size_t)b * (T + pade) + t0) * C : nullptr;     // (either output may be absent, not both)
    __bf16* o16 = dY16 ? dY16 + ((size_t)b * (T + pade) + t0) * C : nullptr;       // the same gradient as bf16: the data-gradient convolution's operand
    const size_t prow = (size_t)b * gridDim.x + blockIdx.x;
    for (int c = tid; c < C; c += 256) {
        const float g = gain[c], bs = bias[c];
        double dg = 0.0, db = 0.0;
        for (int f = 0; f < nfr; ++f) {
            const int t = t0 + f;
            const float mu = mean[(size_t)b * T + t], r = rstd[(size_t)b * T + t];
            const float yy = y[(size_t)f * C + c];
            const float xh = (yy - mu) * r;
            const float x = xh * g + bs;
            const float dx = da[(size_t)f * C + c] * (x > 0.f ? 1.f : slope);
            dg += (double)dx * (double)xh;
            db += (double)dx;
            const float dyv = dx * g * r + (float)ds[t] + 2.f * yy * (float)dq[t];
            if (o) o[(size_t)f * C + c] = dyv;
            if (o16) o16[(size_t)f * C + c] = (__bf16)dyv;
        }
        dgain_part[prow * C + c] = (float)dg;
        dbias_part[prow * C + c] = (float)db;
    }
}

// out[c] (+)= sum_r part[r][c]: block = 64 columns x 16 row groups, fixed summation order; blockIdx.y selects one of two
// (part, out) pairs so that the gain and bias partials of a cLN layer reduce in one launch
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ part0, float* __restrict__ out0, const float* __restrict__ part1,
                                                      float* __restrict__ out1, int rows, int cols, int accumulate) {
    __shared__ float sp[16][64];
    const float* part = blockIdx.y ? part1 : part0;
    float* out = blockIdx.y ? out1 : out0;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float s = 0.f;
    if (c < cols) {
        // groups of 8 loads in flight (one memory latency per group: the serial load-add loop of these 8 workgroups took 75 us for 2 MB);
        // the sum stays in row order
        int r = g;
        for (; r + 7 * 16 < rows; r += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(r + 16 * u) * cols + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; r < rows; r += 16) s += part[(size_t)r * cols + c];
    }
    sp[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q += 4) t += (sp[q][threadIdx.x] + sp[q + 1][threadIdx.x]) + (sp[q + 2][threadIdx.x] + sp[q + 3][threadIdx.x]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

// ------------------------------------------------------------------------------------------ tail
__global__ void exptanh_bwd_kernel(const float* __restrict__ dmask, const float* __restrict__ mask, float* __restrict__ dout, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float m = mask[i];
        const float th = logf(m) * (1.f / 3.2f);
        dout[i] = dmask[i] * m * 3.2f * (1.f - th * th);
    }
}

// ------------------------------------------------------------------------------------------ energy norm
// train_nele.py:133-146 per utterance (the reference is batch 1, so torch.sum is per utterance):
//   cp = clean^inv_p; beta2 = sum(cp) / sum(mask*cp); enh = clean * mask^p * beta2^p
//   D input channels-last [B][64][T][4] = (enh, noise, clean, 0) transposed to (band, frame).
// Also emits alpha2 = mask*beta2 (train_nele.py:307) for the resynthesis path.
// One workgroup of 1024 threads per utterance (the sums are per utterance).  Frames are processed in tiles of 64: the band features
// are [t][c] (c contiguous) and the D input is [c][t][4], so each tile goes through LDS and both sides are accessed in their
// contiguous direction (the scattered 16-byte stores of a thread-per-element version made this kernel 70 us, its backward 136 us).
#define EN_TT 64
__global__ __launch_bounds__(1024) void energy_norm_fwd_kernel(const float* __restrict__ clean, const float* __restrict__ mask,
                                                               const float* __restrict__ noise, float p, float inv_p,
                                                               float* __restrict__ beta2_out, float* __restrict__ s2_out,
                                                               float* __restrict__ din, float* __restrict__ alpha2, int T) {
    __shared__ double red[16];
    __shared__ float te[EN_TT][65], tn[EN_TT][65], tc[EN_TT][65];
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)b * T * 64;
    const int n = T * 64;
    double s1 = 0.0, s2 = 0.0;
    // a thread's elements tid, tid + 1024, .. are ADDED in that order (the sums are part of what the tests pin), but loaded eight at a time:
    // one element per iteration paid a memory round trip each - 31 of them at T = 501, two thirds of this kernel's 60 us
    for (int i0 = tid; i0 < n; i0 += 8 * 1024) {
        float cv[8], mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 1024 * u;
            cv[u] = i < n ? clean[base + i] : 0.f;
            mv[u] = i < n ? mask[base + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + 1024 * u < n) {
                const float cp = nele_powi_f32(cv[u], inv_p);
                s1 += (double)cp;
                s2 += (double)(mv[u] * cp);
            }
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float beta2 = (float)(s1 / s2);
    if (tid == 0) {
        beta2_out[b] = beta2;
        if (s2_out) s2_out[b] = (float)s2;
    }
    if (!din) {                                       // the enhancement path: alpha2 only, no tiles, no barriers
        if (alpha2) {
            for (int i0 = tid; i0 < n; i0 += 8 * 1024) {
                float mv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) mv[u] = (i0 + 1024 * u < n) ? mask[base + i0 + 1024 * u] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + 1024 * u < n) alpha2[base + i0 + 1024 * u] = mv[u] * beta2;
            }
        }
        return;
    }
    const float beta_p = nele_pow_f32(beta2, p);
    for (int t0 = 0; t0 < T; t0 += EN_TT) {
        const int nt = min(EN_TT, T - t0);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 1024 * e, tl = idx >> 6, c = idx & 63;
            if (tl < nt) {
                const size_t g = base + (size_t)(t0 + tl) * 64 + c;
                const float cb = clean[g], m = mask[g];
                if (alpha2) alpha2[g] = m * beta2;
                if (din) {
                    te[tl][c] = cb * nele_pow_f32(m, p) * beta_p;
                    tn[tl][c] = noise[g];
                    tc[tl][c] = cb;
                }
            }
        }
        if (!din) continue;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 1024 * e, c = idx >> 6, tl = idx & 63;
            if (tl < nt)
                *reinterpret_cast<float4*>(din + (((size_t)b * 64 + c) * T + t0 + tl) * 4) = make_float4(te[tl][c], tn[tl][c], tc[tl][c], 0.f);
        }
    }
}

// dmask from d(din): only channel 0 (enh) depends on the mask.
//   enh = cb m^p bp,  bp = (S1/S2)^p,  S2 = sum m cp
//   dm = dE cb p m^(p-1) bp  -  (sum dE cb m^p) p bp / S2 * cp
__global__ __launch_bounds__(1024) void energy_norm_bwd_kernel(const float* __restrict__ clean, const float* __restrict__ mask,
                                                               const float* __restrict__ beta2_in, const float* __restrict__ s2_in,
                                                               const float* __restrict__ ddin, const float* __restrict__ din, float p,
                                                               float inv_p, float* __restrict__ dmask, int T) {
    // With the forward pass's D input at hand, enh = cb m^p bp is read back instead of recomputed: cb m^p = enh / bp and
    // cb m^(p-1) bp = enh / m, which leaves one powf per element (cb^inv_p) instead of four (this kernel was powf-bound).
    __shared__ double red[16];
    __shared__ float td[EN_TT][65], te[EN_TT][65];
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)b * T * 64;
    const float beta_p = nele_pow_f32(beta2_in[b], p);
    double r = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
        float k2 = 0.f;
        if (pass == 1) {
            r = block_sum(r, red);
            k2 = (float)(r * (double)p * (double)beta_p / (double)s2_in[b]);
        }
        for (int t0 = 0; t0 < T; t0 += EN_TT) {
            const int nt = min(EN_TT, T - t0);
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) {                       // dE (and enh) of the tile, read along t (contiguous in [c][t][4])
                const int idx = tid + 1024 * e, c = idx >> 6, tl = idx & 63;
                if (tl < nt) {
                    const size_t g = (((size_t)b * 64 + c) * T + t0 + tl) * 4;
                    td[tl][c] = ddin[g];
                    if (din) te[tl][c] = din[g];
                }
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = tid + 1024 * e, tl = idx >> 6, c = idx & 63;
                if (tl < nt) {
                    const size_t g = base + (size_t)(t0 + tl) * 64 + c;
                    const float dE = td[tl][c];
                    if (din) {
                        const float enh = te[tl][c];
                        if (pass == 0) r += (double)(dE * (enh / beta_p));
                        else dmask[g] = dE * p * (enh / mask[g]) - k2 * nele_powi_f32(clean[g], inv_p);
                    } else {
                        const float cb = clean[g], m = mask[g];
                        if (pass == 0) r += (double)(dE * cb * nele_pow_f32(m, p));
                        else dmask[g] = dE * cb * p * powf(m, p - 1.f) * beta_p - k2 * nele_powi_f32(cb, inv_p);
                    }
                }
            }
        }
    }
}

// D input from three band-feature tensors [B][T][64] (dataloader.py:76-84: (enhanced, noise, clean)):
// channels-last [B][64][T][4], 4th channel zero.  c2 may be null (D_Qua: (enhanced, clean)).
__global__ void d_pack_kernel(const float* __restrict__ c0, const float* __restrict__ c1, const float* __restrict__ c2,
                              float* __restrict__ din, int T) {
    const int b = blockIdx.y;
    const size_t base = (size_t)b * T * 64;
    const int n = T * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int t = i >> 6, c = i & 63;
        float4 v = make_float4(c0[base + i], c1[base + i], c2 ? c2[base + i] : 0.f, 0.f);
        *reinterpret_cast<float4*>(din + (((size_t)b * 64 + c) * T + t) * 4) = v;
    }
}

// Reference layout [B][Cin][64][T] (model.py:118) <-> channels-last [B][64][T][4]
__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ din, int Cin, int T) {
    const int b = blockIdx.y;
    const int n = 64 * T;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < Cin; ++ch) v[ch] = x[((size_t)b * Cin + ch) * n + i];
        *reinterpret_cast<float4*>(din + ((size_t)b * n + i) * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}
__global__ void nhwc4_to_nchw_kernel(const float* __restrict__ ddin, float* __restrict__ dx, int Cin, int T) {
    const int b = blockIdx.y;
    const int n = 64 * T;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 v = *reinterpret_cast<const float4*>(ddin + ((size_t)b * n + i) * 4);
        const float a[4] = {v.x, v.y, v.z, v.w};
        for (int ch = 0; ch < Cin; ++ch) dx[((size_t)b * Cin + ch) * n + i] = a[ch];
    }
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int nele_g_pack(const float* x, const float* y, float* out, int B, int T, int pad, void* stream) {
    NELE_CHECK_ARG(x && y && out && B > 0 && T > 0 && pad >= 0, "nele_g_pack: bad arguments");
    hipLaunchKernelGGL(g_pack_kernel, dim3(T, B), dim3(128), 0, as_stream(stream), x, y, out, T, pad);
    NELE_CHECK_LAUNCH("nele_g_pack");
    return NELE_OK;
}

// scratch: float64 [B][T][2]; part buffers: [B * nele_cln_chunks(T)][C]
extern "C" int nele_cln_chunks(int T) { return (T + CLN_FR - 1) / CLN_FR; }

extern "C" int nele_cln_fwd(const float* Y, const float* gain, const float* bias, float* out, float* mean, float* rstd, double* scratch,
                            int B, int T, int C, int pad, float slope, void* stream) {
    NELE_CHECK_ARG(Y && gain && bias && out && mean && rstd && scratch && B > 0, "nele_cln_fwd: bad arguments");
    if (T > CLN_MAX_T) return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_cln_fwd: T=%d > %d", T, CLN_MAX_T);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(cln_stats_kernel, dim3((T + 3) / 4, B), dim3(256), 0, s, Y, scratch, T, C);
    hipLaunchKernelGGL(cln_fwd_kernel, dim3((T + CLN_FR - 1) / CLN_FR, B), dim3(256), 0, s, Y, scratch, gain, bias, out, mean, rstd, T, C, pad,
                       slope);
    NELE_CHECK_LAUNCH("nele_cln_fwd");
    return NELE_OK;
}

extern "C" int nele_cln_bwd(const float* dAct, const float* Y, const float* gain, const float* bias, const float* mean,
                            const float* rstd, float* dY, void* dY16, float* dgain_part, float* dbias_part, double* scratch, int B, int T, int C,
                            int pade, float slope, void* stream) {
    NELE_CHECK_ARG(dAct && Y && gain && bias && mean && rstd && (dY || dY16) && dgain_part && dbias_part && scratch && B > 0,
                   "nele_cln_bwd: bad arguments");
    if (T > CLN_MAX_T) return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_cln_bwd: T=%d > %d", T, CLN_MAX_T);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(cln_bwd_stats_kernel, dim3((T + 3) / 4, B), dim3(256), 0, s, dAct, Y, gain, bias, mean, rstd, scratch, T, C, slope);
    hipLaunchKernelGGL(cln_bwd_kernel, dim3((T + CLN_FR - 1) / CLN_FR, B), dim3(256), 0, s, dAct, Y, scratch, gain, bias, mean, rstd, dY,
                       dgain_part, dbias_part, T, C, pade, slope, (__bf16*)dY16);
    NELE_CHECK_LAUNCH("nele_cln_bwd");
    return NELE_OK;
}

extern "C" int nele_colsum(const float* part, int rows, int cols, float* out, int accumulate, void* stream) {
    NELE_CHECK_ARG(part && out && rows > 0 && cols > 0, "nele_colsum: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, 1), dim3(1024), 0, as_stream(stream), part, out, part, out, rows, cols, accumulate);
    NELE_CHECK_LAUNCH("nele_colsum");
    return NELE_OK;
}

extern "C" int nele_colsum2(const float* part0, float* out0, const float* part1, float* out1, int rows, int cols, int accumulate, void* stream) {
    NELE_CHECK_ARG(part0 && out0 && part1 && out1 && rows > 0 && cols > 0, "nele_colsum2: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, 2), dim3(1024), 0, as_stream(stream), part0, out0, part1, out1, rows, cols, accumulate);
    NELE_CHECK_LAUNCH("nele_colsum2");
    return NELE_OK;
}

extern "C" int nele_exptanh_bwd(const float* dmask, const float* mask, float* dout, long long n, void* stream) {
    NELE_CHECK_ARG(dmask && mask && dout && n > 0, "nele_exptanh_bwd: bad arguments");
    hipLaunchKernelGGL(exptanh_bwd_kernel, dim3((unsigned)min((long long)2048, (n + 255) / 256)), dim3(256), 0, as_stream(stream), dmask,
                       mask, dout, (size_t)n);
    NELE_CHECK_LAUNCH("nele_exptanh_bwd");
    return NELE_OK;
}

extern "C" int nele_energy_norm_fwd(const float* clean, const float* mask, const float* noise, float p, float inv_p, float* beta2,
                                    float* s2, float* din, float* alpha2, int B, int T, void* stream) {
    NELE_CHECK_ARG(clean && mask && beta2 && B > 0 && T > 0, "nele_energy_norm_fwd: bad arguments");
    NELE_CHECK_ARG(!din || noise, "nele_energy_norm_fwd: din requested without noise features");
    hipLaunchKernelGGL(energy_norm_fwd_kernel, dim3(B), dim3(1024), 0, as_stream(stream), clean, mask, noise, p, inv_p, beta2, s2, din,
                       alpha2, T);
    NELE_CHECK_LAUNCH("nele_energy_norm_fwd");
    return NELE_OK;
}

extern "C" int nele_energy_norm_bwd(const float* clean, const float* mask, const float* beta2, const float* s2, const float* ddin, const float* din,
                                    float p, float inv_p, float* dmask, int B, int T, void* stream) {
    NELE_CHECK_ARG(clean && mask && beta2 && s2 && ddin && dmask && B > 0, "nele_energy_norm_bwd: bad arguments");
    hipLaunchKernelGGL(energy_norm_bwd_kernel, dim3(B), dim3(1024), 0, as_stream(stream), clean, mask, beta2, s2, ddin, din, p, inv_p, dmask, T);
    NELE_CHECK_LAUNCH("nele_energy_norm_bwd");
    return NELE_OK;
}

extern "C" int nele_d_pack(const float* c0, const float* c1, const float* c2, float* din, int B, int T, void* stream) {
    NELE_CHECK_ARG(c0 && c1 && din && B > 0 && T > 0, "nele_d_pack: bad arguments");
    hipLaunchKernelGGL(d_pack_kernel, dim3((T * 64 + 255) / 256, B), dim3(256), 0, as_stream(stream), c0, c1, c2, din, T);
    NELE_CHECK_LAUNCH("nele_d_pack");
    return NELE_OK;
}

// A D training batch gathered from per-utterance items (dataloader.py:54-84 + train_nele.py:349-367: the reference's D loader hands out
// one [3][64][T_k] item at a time; here a shuffled list of items becomes padded batches).  items: n device pointers to channels-last
// [64][T_k][4] float32 (band rows `stride` floats apart: a row of a larger padded batch is a valid item), out [rows][64][Tm][4]: item r in
// row r, columns >= T_k and rows >= n zero; frames_out [rows] = T_k (Tm for the fill rows).  One launch per 64 items instead of one copy
// per item (round 6: 830 small copies per 256-utterance epoch were a fifth of run_epoch's host time).
struct GatherJobs { const float* src[64]; int T[64]; long long stride[64]; int n; };
__global__ __launch_bounds__(256) void d_gather_kernel(GatherJobs j, int r0, int Tm, float* __restrict__ out, int* __restrict__ frames_out) {
    const int r = blockIdx.y, band = blockIdx.x;
    const int Tk = r < j.n ? j.T[r] : 0;
    const float4* src = r < j.n ? reinterpret_cast<const float4*>(j.src[r] + (size_t)band * j.stride[r]) : nullptr;
    float4* dst = reinterpret_cast<float4*>(out + (((size_t)(r0 + r) * 64 + band) * Tm) * 4);
    for (int t = threadIdx.x; t < Tm; t += 256) dst[t] = t < Tk ? src[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (frames_out && band == 0 && threadIdx.x == 0) frames_out[r0 + r] = r < j.n ? Tk : Tm;
}

extern "C" int nele_d_gather_items(const void* const* items_host, const int* frames_host, const long long* strides_host, int n, int rows, int Tm,
                                   float* din_out, int* frames_out, void* stream) {
    NELE_CHECK_ARG(items_host && frames_host && din_out && n > 0 && rows >= n && Tm > 0, "nele_d_gather_items: bad arguments");
    for (int i = 0; i < n; ++i)
        NELE_CHECK_ARG(items_host[i] && frames_host[i] > 0 && frames_host[i] <= Tm && ((size_t)items_host[i] % 16) == 0 &&
                       (!strides_host || (strides_host[i] >= 4LL * frames_host[i] && strides_host[i] % 4 == 0)),
                       "nele_d_gather_items: item %d (T = %d, Tm = %d)", i, frames_host[i], Tm);
    for (int r0 = 0; r0 < rows; r0 += 64) {
        GatherJobs j;
        const int m = rows - r0 < 64 ? rows - r0 : 64;
        j.n = n - r0 < 0 ? 0 : (n - r0 < 64 ? n - r0 : 64);
        for (int i = 0; i < 64; ++i) {
            const bool live = i < j.n;
            j.src[i] = live ? reinterpret_cast<const float*>(items_host[r0 + i]) : nullptr;
            j.T[i] = live ? frames_host[r0 + i] : 0;
            j.stride[i] = live ? (strides_host ? strides_host[r0 + i] : 4LL * frames_host[r0 + i]) : 0;
        }
        hipLaunchKernelGGL(d_gather_kernel, dim3(64, m), dim3(256), 0, as_stream(stream), j, r0, Tm, din_out, frames_out);
    }
    NELE_CHECK_LAUNCH("nele_d_gather_items");
    return NELE_OK;
}

extern "C" int nele_d_layout(const float* src, float* dst, int B, int Cin, int T, int to_nhwc, void* stream) {
    NELE_CHECK_ARG(src && dst && B > 0 && T > 0 && Cin >= 1 && Cin <= 4, "nele_d_layout: bad arguments");
    dim3 grid((T * 64 + 255) / 256, B);
    if (to_nhwc) hipLaunchKernelGGL(nchw_to_nhwc4_kernel, grid, dim3(256), 0, as_stream(stream), src, dst, Cin, T);
    else hipLaunchKernelGGL(nhwc4_to_nchw_kernel, grid, dim3(256), 0, as_stream(stream), src, dst, Cin, T);
    NELE_CHECK_LAUNCH("nele_d_layout");
    return NELE_OK;
}
