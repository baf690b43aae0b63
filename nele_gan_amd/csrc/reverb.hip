// Evaluation-path signal conditioning (reference eval_metrics.py:124-167, audio_util.py:67-74):
//   nele_fir_filter : scipy.signal.lfilter(h, [1], x) - a room impulse response applied to a batch of utterances
//   nele_norm_clip  : x / rms(x) * target, optional second summand, then clip() (divide by 1.05, 1.10, ... until inside [-1, 1))
// float64 arithmetic like the reference (lfilter promotes float32 inputs because a = [1] is an integer array).
#include "common.h"

#define FIR_TB 256          // threads per workgroup
#define FIR_PER 4           // outputs per thread (n0 + t + 256 j)
#define FIR_NOUT (FIR_TB * FIR_PER)
#define FIR_KC 512          // taps per LDS chunk

// y[b][n] = sum_{k <= n, k < Lh} h[k] x[b][n - k], accumulated from the oldest tap to the newest (lfilter's transposed direct form II
// adds the terms in that order).  grid (ceil(L / 1024), B), block 256.  Per chunk of 512 taps the workgroup keeps the taps and the
// 1024 + 511 input samples they touch in LDS; thread t owns outputs n0 + t + 256 j, so the input reads of a wave are consecutive
// (conflict-free) and the tap is a broadcast.
__global__ __launch_bounds__(FIR_TB) void fir_filter_kernel(const float* __restrict__ x, int L, const double* __restrict__ h, int Lh,
                                                            double* __restrict__ y) {
    __shared__ double hs[FIR_KC];
    __shared__ double xs[FIR_NOUT + FIR_KC];
    const int b = blockIdx.y, n0 = blockIdx.x * FIR_NOUT, t = threadIdx.x;
    const float* xb = x + (size_t)b * L;
    double acc[FIR_PER];
#pragma unroll
    for (int j = 0; j < FIR_PER; ++j) acc[j] = 0.0;
    const int kmax = min(Lh, n0 + FIR_NOUT);                 // taps beyond the block's last output index never meet a sample
    const int nchunk = (kmax + FIR_KC - 1) / FIR_KC;
    for (int c = nchunk - 1; c >= 0; --c) {                  // oldest taps first
        const int k0 = c * FIR_KC;
        __syncthreads();
        for (int i = t; i < FIR_KC; i += FIR_TB) hs[i] = (k0 + i < Lh) ? h[k0 + i] : 0.0;
        // xs[i] = x[n0 - k0 - (FIR_KC - 1) + i]
        const int base = n0 - k0 - (FIR_KC - 1);
        for (int i = t; i < FIR_NOUT + FIR_KC; i += FIR_TB) {
            const int n = base + i;
            xs[i] = (n >= 0 && n < L) ? (double)xb[n] : 0.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = FIR_KC - 1; kk >= 0; --kk) {
            const double hk = hs[kk];
#pragma unroll
            for (int j = 0; j < FIR_PER; ++j) acc[j] = fma(hk, xs[t + FIR_TB * j + (FIR_KC - 1) - kk], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < FIR_PER; ++j) {
        const int n = n0 + t + FIR_TB * j;
        if (n < L) y[(size_t)b * L + n] = acc[j];
    }
}

// One workgroup per utterance: v = a (+ add); v = v / rms(v) * target (target > 0); clip(v) of audio_util.py:67-74; out = (float)v.
// The divisors of clip() depend only on max / min, and x / c is monotonic in x, so the loop runs on the two extremes and every element
// then goes through the same sequence of divisions.
__global__ __launch_bounds__(1024) void norm_clip_kernel(const double* __restrict__ a64, const float* __restrict__ a32, const float* __restrict__ add,
                                                         int N, double target, float* __restrict__ out, double* __restrict__ out64,
                                                         int* __restrict__ nclip) {
    __shared__ double red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    auto val = [&](int i) {
        double v = a64 ? a64[(size_t)b * N + i] : (double)a32[(size_t)b * N + i];
        if (add) v += (double)add[(size_t)b * N + i];
        return v;
    };
    double scale_num = 1.0, scale_den = 1.0;
    if (target > 0.0) {
        double s = 0.0;
        for (int i = tid; i < N; i += 1024) { const double v = val(i); s += v * v; }
        s = block_sum(s, red);
        scale_den = sqrt(s / (double)N);
        scale_num = target;
    }
    double mx = -1e300, mn = 1e300;
    for (int i = tid; i < N; i += 1024) {
        double v = val(i);
        if (target > 0.0) v = v / scale_den * scale_num;
        mx = fmax(mx, v); mn = fmin(mn, v);
    }
    mx = block_max(mx, red);
    mn = -block_max(-mn, red);
    int steps = 0;
    {
        double small = 0.05, hi = mx, lo = mn;
        while ((hi >= 1.0 || lo < -1.0) && steps < 4096) { hi = hi / (1.0 + small); lo = lo / (1.0 + small); small = small + 0.05; ++steps; }
    }
    for (int i = tid; i < N; i += 1024) {
        double v = val(i);
        if (target > 0.0) v = v / scale_den * scale_num;
        double small = 0.05;
        for (int s_ = 0; s_ < steps; ++s_) { v = v / (1.0 + small); small = small + 0.05; }
        if (out) out[(size_t)b * N + i] = (float)v;
        if (out64) out64[(size_t)b * N + i] = v;
    }
    if (nclip && tid == 0) nclip[b] = steps;
}

extern "C" int nele_fir_filter(const float* x, int B, int L, const double* h, int Lh, double* y, void* stream) {
    NELE_CHECK_ARG(x && h && y && B > 0 && L > 0 && Lh > 0, "nele_fir_filter: bad arguments");
    hipLaunchKernelGGL(fir_filter_kernel, dim3((L + FIR_NOUT - 1) / FIR_NOUT, B), dim3(FIR_TB), 0, as_stream(stream), x, L, h, Lh, y);
    NELE_CHECK_LAUNCH("nele_fir_filter");
    return NELE_OK;
}

extern "C" int nele_norm_clip(const double* a64, const float* a32, const float* add, int B, int N, double target_rms, float* out,
                              double* out64, int* nclip, void* stream) {
    NELE_CHECK_ARG((a64 != nullptr) != (a32 != nullptr), "nele_norm_clip: exactly one of a64 / a32 must be given");
    NELE_CHECK_ARG((out || out64) && B > 0 && N > 0, "nele_norm_clip: bad arguments");
    hipLaunchKernelGGL(norm_clip_kernel, dim3(B), dim3(1024), 0, as_stream(stream), a64, a32, add, N, target_rms, out, out64, nclip);
    NELE_CHECK_LAUNCH("nele_norm_clip");
    return NELE_OK;
}
