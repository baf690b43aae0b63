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
// One block (256 threads) per utterance.
//   phase 1: wave per frame: s_t = sum_c y, q_t = sum_c y^2 (float32 lanes, float64 wave reduce)
//   phase 2: cumulative sums over t (float64, serial: T <= a few hundred)
//   phase 3: normalise, affine, LeakyReLU, store into the next layer's padded buffer
// cum_mean = S/n, cum_var = (Q - 2 mean S)/n + mean^2, n = C (t+1), eps = 1e-8 (model.py:188-199).
#define CLN_MAX_T 1024

__global__ __launch_bounds__(256) void cln_fwd_kernel(const float* __restrict__ Y, const float* __restrict__ gain,
                                                      const float* __restrict__ bias, float* __restrict__ out, float* __restrict__ mean,
                                                      float* __restrict__ rstd, int T, int C, int pad, float slope) {
    __shared__ double ss[CLN_MAX_T], qq[CLN_MAX_T];
    __shared__ float smean[CLN_MAX_T], srstd[CLN_MAX_T];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* y = Y + (size_t)b * T * C;
    for (int t = wave; t < T; t += 4) {
        double s = 0.0, q = 0.0;
        for (int c = lane; c < C; c += 64) {
            const float v = y[(size_t)t * C + c];
            s += (double)v;
            q += (double)v * (double)v;
        }
        s = wave_sum(s);
        q = wave_sum(q);
        if (lane == 0) { ss[t] = s; qq[t] = q; }
    }
    __syncthreads();
    if (tid == 0) {
        double S = 0.0, Q = 0.0;
        for (int t = 0; t < T; ++t) {
            S += ss[t];
            Q += qq[t];
            const double n = (double)C * (double)(t + 1);
            const double m = S / n;
            const double var = (Q - 2.0 * m * S) / n + m * m;
            smean[t] = (float)m;
            srstd[t] = (float)(1.0 / sqrt(var + 1e-8));
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
        mean[(size_t)b * T + t] = smean[t];
        rstd[(size_t)b * T + t] = srstd[t];
    }
    float* o = out + ((size_t)b * (T + pad) + pad) * C;
    const int total = T * C;
    for (int i = tid; i < total; i += 256) {
        const int t = i / C, c = i - t * C;
        float v = (y[i] - smean[t]) * srstd[t] * gain[c] + bias[c];
        v = v > 0.f ? v : slope * v;
        o[i] = v;
    }
}

// Backward of out = lrelu(cLN(y)):  with xh = (y-mu_t) r_t, dxh = dAct * lrelu'(.) * gain_c,
//   A_t = sum_c dxh, B_t = sum_c dxh*xh, n_t = C(t+1)
//   dL/dS_t = (-A_t r_t + B_t mu_t r_t^2) / n_t,   dL/dQ_t = -B_t r_t^2 / (2 n_t)
//   RS_t = sum_{t'>=t} dL/dS_t', RQ_t likewise;   dy[t,c] = dxh r_t + RS_t + 2 y RQ_t.
// dY is written into an END-padded buffer [B][T+pade][C] (rows >= T stay zero) for the data-gradient GEMM.
__global__ __launch_bounds__(256) void cln_bwd_kernel(const float* __restrict__ dAct, const float* __restrict__ Y,
                                                      const float* __restrict__ gain, const float* __restrict__ bias,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      float* __restrict__ dY, float* __restrict__ dgain_part,
                                                      float* __restrict__ dbias_part, int T, int C, int pade, float slope) {
    __shared__ double sa[CLN_MAX_T], sb[CLN_MAX_T];
    __shared__ float rs[CLN_MAX_T], rq[CLN_MAX_T], smean[CLN_MAX_T], srstd[CLN_MAX_T];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* y = Y + (size_t)b * T * C;
    const float* da = dAct + (size_t)b * T * C;
    for (int t = tid; t < T; t += 256) {
        smean[t] = mean[(size_t)b * T + t];
        srstd[t] = rstd[(size_t)b * T + t];
    }
    __syncthreads();
    for (int t = wave; t < T; t += 4) {
        double a = 0.0, bb = 0.0;
        const float mu = smean[t], r = srstd[t];
        for (int c = lane; c < C; c += 64) {
            const float xh = (y[(size_t)t * C + c] - mu) * r;
            const float x = xh * gain[c] + bias[c];
            const float d = da[(size_t)t * C + c] * (x > 0.f ? 1.f : slope) * gain[c];
            a += (double)d;
            bb += (double)d * (double)xh;
        }
        a = wave_sum(a);
        bb = wave_sum(bb);
        if (lane == 0) { sa[t] = a; sb[t] = bb; }
    }
    __syncthreads();
    if (tid == 0) {
        double RS = 0.0, RQ = 0.0;
        for (int t = T - 1; t >= 0; --t) {
            const double n = (double)C * (double)(t + 1), r = (double)srstd[t], mu = (double)smean[t];
            RS += (-sa[t] * r + sb[t] * mu * r * r) / n;
            RQ += -sb[t] * r * r / (2.0 * n);
            rs[t] = (float)RS;
            rq[t] = (float)RQ;
        }
    }
    __syncthreads();
    float* o = dY + (size_t)b * (T + pade) * C;
    // thread per channel (C <= 256): walks t, accumulates dgain / dbias, writes dy
    for (int c = tid; c < C; c += 256) {
        const float g = gain[c], bs = bias[c];
        double dg = 0.0, db = 0.0;
        for (int t = 0; t < T; ++t) {
            const float yy = y[(size_t)t * C + c];
            const float xh = (yy - smean[t]) * srstd[t];
            const float x = xh * g + bs;
            const float dx = da[(size_t)t * C + c] * (x > 0.f ? 1.f : slope);
            dg += (double)dx * (double)xh;
            db += (double)dx;
            o[(size_t)t * C + c] = dx * g * srstd[t] + rs[t] + 2.f * yy * rq[t];
        }
        dgain_part[(size_t)b * C + c] = (float)dg;
        dbias_part[(size_t)b * C + c] = (float)db;
    }
}

__global__ void colsum_kernel(const float* __restrict__ part, int rows, int cols, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += part[(size_t)r * cols + c];
    out[c] = accumulate ? out[c] + s : s;
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
__global__ __launch_bounds__(256) void energy_norm_fwd_kernel(const float* __restrict__ clean, const float* __restrict__ mask,
                                                              const float* __restrict__ noise, float p, float inv_p,
                                                              float* __restrict__ beta2_out, float* __restrict__ s2_out,
                                                              float* __restrict__ din, float* __restrict__ alpha2, int T) {
    __shared__ double red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)b * T * 64;
    const int n = T * 64;
    double s1 = 0.0, s2 = 0.0;
    for (int i = tid; i < n; i += 256) {
        const float cp = powf(clean[base + i], inv_p);
        s1 += (double)cp;
        s2 += (double)(mask[base + i] * cp);
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float beta2 = (float)(s1 / s2);
    if (tid == 0) {
        beta2_out[b] = beta2;
        if (s2_out) s2_out[b] = (float)s2;
    }
    const float beta_p = powf(beta2, p);
    for (int i = tid; i < n; i += 256) {
        const int t = i >> 6, c = i & 63;
        const float cb = clean[base + i], m = mask[base + i];
        if (alpha2) alpha2[base + i] = m * beta2;
        if (din) {
            float4 v;
            v.x = cb * powf(m, p) * beta_p;
            v.y = noise[base + i];
            v.z = cb;
            v.w = 0.f;
            *reinterpret_cast<float4*>(din + (((size_t)b * 64 + c) * T + t) * 4) = v;
        }
    }
}

// dmask from d(din): only channel 0 (enh) depends on the mask.
//   enh = cb m^p bp,  bp = (S1/S2)^p,  S2 = sum m cp
//   dm = dE cb p m^(p-1) bp  -  (sum dE cb m^p) p bp / S2 * cp
__global__ __launch_bounds__(256) void energy_norm_bwd_kernel(const float* __restrict__ clean, const float* __restrict__ mask,
                                                              const float* __restrict__ beta2_in, const float* __restrict__ s2_in,
                                                              const float* __restrict__ ddin, float p, float inv_p,
                                                              float* __restrict__ dmask, int T) {
    __shared__ double red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)b * T * 64;
    const int n = T * 64;
    const float beta_p = powf(beta2_in[b], p);
    double r = 0.0;
    for (int i = tid; i < n; i += 256) {
        const int t = i >> 6, c = i & 63;
        const float dE = ddin[(((size_t)b * 64 + c) * T + t) * 4];
        r += (double)(dE * clean[base + i] * powf(mask[base + i], p));
    }
    r = block_sum(r, red);
    const float k2 = (float)(r * (double)p * (double)beta_p / (double)s2_in[b]);
    for (int i = tid; i < n; i += 256) {
        const int t = i >> 6, c = i & 63;
        const float dE = ddin[(((size_t)b * 64 + c) * T + t) * 4];
        const float cb = clean[base + i], m = mask[base + i];
        dmask[base + i] = dE * cb * p * powf(m, p - 1.f) * beta_p - k2 * powf(cb, inv_p);
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

extern "C" int nele_cln_fwd(const float* Y, const float* gain, const float* bias, float* out, float* mean, float* rstd, int B, int T,
                            int C, int pad, float slope, void* stream) {
    NELE_CHECK_ARG(Y && gain && bias && out && mean && rstd && B > 0, "nele_cln_fwd: bad arguments");
    if (T > CLN_MAX_T) return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_cln_fwd: T=%d > %d", T, CLN_MAX_T);
    hipLaunchKernelGGL(cln_fwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), Y, gain, bias, out, mean, rstd, T, C, pad, slope);
    NELE_CHECK_LAUNCH("nele_cln_fwd");
    return NELE_OK;
}

extern "C" int nele_cln_bwd(const float* dAct, const float* Y, const float* gain, const float* bias, const float* mean,
                            const float* rstd, float* dY, float* dgain_part, float* dbias_part, int B, int T, int C, int pade,
                            float slope, void* stream) {
    NELE_CHECK_ARG(dAct && Y && gain && bias && mean && rstd && dY && dgain_part && dbias_part && B > 0, "nele_cln_bwd: bad arguments");
    if (T > CLN_MAX_T) return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_cln_bwd: T=%d > %d", T, CLN_MAX_T);
    hipLaunchKernelGGL(cln_bwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), dAct, Y, gain, bias, mean, rstd, dY, dgain_part,
                       dbias_part, T, C, pade, slope);
    NELE_CHECK_LAUNCH("nele_cln_bwd");
    return NELE_OK;
}

extern "C" int nele_colsum(const float* part, int rows, int cols, float* out, int accumulate, void* stream) {
    NELE_CHECK_ARG(part && out && rows > 0 && cols > 0, "nele_colsum: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 255) / 256), dim3(256), 0, as_stream(stream), part, rows, cols, out, accumulate);
    NELE_CHECK_LAUNCH("nele_colsum");
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
    hipLaunchKernelGGL(energy_norm_fwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), clean, mask, noise, p, inv_p, beta2, s2, din,
                       alpha2, T);
    NELE_CHECK_LAUNCH("nele_energy_norm_fwd");
    return NELE_OK;
}

extern "C" int nele_energy_norm_bwd(const float* clean, const float* mask, const float* beta2, const float* s2, const float* ddin,
                                    float p, float inv_p, float* dmask, int B, int T, void* stream) {
    NELE_CHECK_ARG(clean && mask && beta2 && s2 && ddin && dmask && B > 0, "nele_energy_norm_bwd: bad arguments");
    hipLaunchKernelGGL(energy_norm_bwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), clean, mask, beta2, s2, ddin, p, inv_p, dmask, T);
    NELE_CHECK_LAUNCH("nele_energy_norm_bwd");
    return NELE_OK;
}

extern "C" int nele_d_pack(const float* c0, const float* c1, const float* c2, float* din, int B, int T, void* stream) {
    NELE_CHECK_ARG(c0 && c1 && din && B > 0 && T > 0, "nele_d_pack: bad arguments");
    hipLaunchKernelGGL(d_pack_kernel, dim3((T * 64 + 255) / 256, B), dim3(256), 0, as_stream(stream), c0, c1, c2, din, T);
    NELE_CHECK_LAUNCH("nele_d_pack");
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
