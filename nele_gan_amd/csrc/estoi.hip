// Batched ESTOI (reference intel.py:122-140 -> pystoi.stoi(x, y, 16000, extended=True); the
// third-party algorithm is restated in oracle/estoi.py, PARITY UNPINNED vs pystoi itself).
//
// Per utterance (x = clean, y = degraded, both [L] float32 at 16 kHz), all arithmetic float64:
//   k1 resample 16k -> 10k : polyphase FIR (581-tap Kaiser sinc, up 5 / down 8), one thread per output
//   k2 frame energies (256 / hop 128, Hann(258)[1:-1]), 40 dB gate vs the loudest clean frame,
//      ordered compaction of the kept frames
//   k3 per kept frame: rebuild the silence-removed signal on the fly (overlap-add of the two
//      neighbouring kept frames), window, 512-point FFT (x and y share one complex transform),
//      15 third-octave band magnitudes
//   k4 per 30-frame segment: row then column mean/variance normalisation, mean product
//   k5 fixed-order sum over segments, logistic map (intel.py:136-140)
#include "common.h"
#include "fft512.h"

#define ES_NFRAME 256
#define ES_HOP 128
#define ES_NB 15
#define ES_NSEG 30
#define ES_HALF 290   // half length of the resampling filter
#define ES_EPS 2.220446049250313e-16

// pystoi thirdoct(10000, 512, 15, 150): bins [lo, hi) nearest to the band edges (checked against oracle/estoi.py)
__constant__ int c_tob_lo[ES_NB] = {7, 9, 11, 14, 17, 22, 27, 34, 43, 55, 69, 87, 109, 138, 174};
__constant__ int c_tob_hi[ES_NB] = {9, 11, 14, 17, 22, 27, 34, 43, 55, 69, 87, 109, 138, 174, 219};

struct EstoiWs {
    double* xr;      // [B][2][n10]
    double* en;      // [B][F]
    int* keep;       // [B][F]  original frame index of kept frame k
    int* nkept;      // [B]
    double* tob;     // [B][2][ES_NB][F]
    double* dseg;    // [B][F]
    int n10, F;      // of the longest row: buffer strides
    const int* lens; // [B] samples per utterance inside the padded [B][L] buffers, or NULL
    int L;
};
// per-utterance sizes (the reference scores files of any length one at a time: intel.py:122-134, audio_util.py:134-141)
__device__ __forceinline__ int es_len(const EstoiWs& ws, int b) { return ws.lens ? min(ws.lens[b], ws.L) : ws.L; }
__device__ __forceinline__ int es_n10(const EstoiWs& ws, int b) {
    const long long n = (long long)es_len(ws, b) * 5;
    return (int)(n / 8 + (n % 8 ? 1 : 0));
}
__device__ __forceinline__ int es_frames(const EstoiWs& ws, int b) {
    const int n10 = es_n10(ws, b);
    return (n10 >= ES_NFRAME) ? (n10 - ES_NFRAME) / ES_HOP + 1 : 0;
}

__device__ __forceinline__ double hann_sym256(int j) { return 0.5 - 0.5 * cospi(2.0 * (double)(j + 1) / 257.0); }

__device__ __forceinline__ double bessel_i0(double x) {
    // power series, converges quickly for the beta used here (5.65)
    double s = 1.0, t = 1.0;
    const double q = 0.25 * x * x;
    for (int k = 1; k < 60; ++k) {
        t *= q / ((double)k * (double)k);
        s += t;
        if (t < 1e-20 * s) break;
    }
    return s;
}

// filter taps: 5 * kaiser(581, beta) * sinc-lowpass, normalised to unit DC gain before the *5
__global__ void estoi_filter_kernel(double* __restrict__ h) {
    __shared__ double red[8];
    const int t = threadIdx.x;  // 1024 threads >= 581
    const int Lh = ES_HALF, n = 2 * Lh + 1;
    const double beta = 0.1102 * (60.0 - 8.7);
    const double fc = 1.0 / 16.0;
    double v = 0.0;
    if (t < n) {
        const double tt = (double)(t - Lh);
        const double arg = 2.0 * fc * tt;
        const double sinc = (t == Lh) ? 1.0 : sinpi(arg) / (M_PI * arg);
        const double ideal = 2.0 * 5.0 * fc * sinc;
        const double r = 2.0 * (double)t / (double)(n - 1) - 1.0;
        const double kais = bessel_i0(beta * sqrt(fmax(0.0, 1.0 - r * r))) / bessel_i0(beta);
        v = kais * ideal;
    }
    const double tot = block_sum(v, red);
    if (t < n) h[t] = 5.0 * v / tot;
}

// grid (ceil(n10/256), B, 2).  out[i] = sum_n src[n] * h[8 i + 290 - 5 n], n ascending (the order is part of the result).
// The taps an output uses are h[p + 5 m], p = (8 i + 290) mod 5: the filter sits in LDS by PHASE, hp[p][m] = h[p + 5 m], and every lane
// walks m downwards in step - 5 distinct LDS addresses per read instead of 64 lanes striding 64 bytes through h (16-way bank
// conflicts: 1.2 ms per call at B = 256).  The input samples a block needs are staged in LDS too.
#define ES_NPH ((2 * ES_HALF + 1 + 4) / 5)            // taps per phase (117)
#define ES_XIN ((256 * 8 + 2 * ES_HALF) / 5 + 4)     // input samples under 256 consecutive outputs
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(256) void estoi_resample_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const double* __restrict__ h, int L, EstoiWs ws) {
    __shared__ double hp[5][ES_NPH + 1];
    __shared__ float xs[ES_XIN];
    const int b = blockIdx.y, sig = blockIdx.z;
    const int i0 = blockIdx.x * 256, i = i0 + threadIdx.x;
    if (i0 >= es_n10(ws, b)) return;
    for (int k = threadIdx.x; k < 5 * (ES_NPH + 1); k += 256) {
        const int p = k / (ES_NPH + 1), m = k - p * (ES_NPH + 1), idx = p + 5 * m;
        hp[p][m] = (idx <= 2 * ES_HALF) ? h[idx] : 0.0;
    }
    const float* src = (sig == 0 ? x : y) + (size_t)b * L;
    L = es_len(ws, b);
    // inputs of the block: n from ceil((8 i0 + 290 - 580) / 5) (>= 0) to (8 (i0 + 255) + 290) / 5
    const int c0 = 8 * i0 + ES_HALF;
    const int nb0 = (c0 - 2 * ES_HALF < 0) ? 0 : (c0 - 2 * ES_HALF + 4) / 5;
    for (int k = threadIdx.x; k < ES_XIN; k += 256) xs[k] = (nb0 + k < L) ? src[nb0 + k] : 0.f;
    __syncthreads();
    if (i >= es_n10(ws, b)) return;
    const int c = 8 * i + ES_HALF;
    int n_lo = (c - 2 * ES_HALF + 4) / 5;  // ceil((c-580)/5) for c-580 >= 0
    if (c - 2 * ES_HALF < 0) n_lo = 0;
    int n_hi = c / 5;
    if (n_hi > L - 1) n_hi = L - 1;
    const int p = c % 5;                   // tap index c - 5 n = p + 5 m with m = (c - p) / 5 - n
    const int mtop = (c - p) / 5;
    double acc = 0.0;
    for (int n = n_lo; n <= n_hi; ++n) acc += (double)xs[n - nb0] * hp[p][mtop - n];
    ws.xr[((size_t)b * 2 + sig) * ws.n10 + i] = acc;
}
#endif  // NELE_AB

// ---- the same resampler, a thread per group of FIVE consecutive outputs (second session of round 3).  The kernel above pays two LDS reads
// and a conversion per tap and is bound by the LDS pipe.  Outputs i = 5 g + r (r = 0..4) have the phases p = (8 i + 290) mod 5 = {0, 3, 1, 4, 2}
// whatever g is, and their newest inputs sit at n = 8 g + 58 + {0, 1, 3, 4, 6}: walking n upwards, step s of output r meets input
// x[8 g - 58 + s + off_r] and tap h[p_r + 5 (116 - s)].  So (i) the five tap weights of a step are the same for every lane - they come from a
// [117][8] table in memory by scalar loads, no LDS read - and (ii) the thread's 123 inputs are read once (31 16-byte LDS reads) and
// converted once, then serve all five outputs from registers: 6 vector instructions per 5 taps instead of 10 + 10 LDS reads.  Per output
// the sum runs over n ascending with the same multiply-add as above; taps the kernel above does not execute (n < 0, n >= L, tap index
// > 580) meet a staged zero input or a zero weight, which leaves the sum unchanged: bit-identical outputs.
// grid (ceil(n10 / 1280), B, 2), block 256.
#define ES_M 117                          // steps per output: tap indices p + 5 m, m = 116 .. 0
#define ES_HPAD 584                       // filter length rounded up to 8 doubles: the step table follows it in the workspace
#define ES_HTAB (ES_M * 8)                // [step][5 outputs of a group + 3 pad]
__global__ void estoi_taps_kernel(double* __restrict__ h) {
    double* tab = h + ES_HPAD;
    for (int e = threadIdx.x; e < ES_HTAB; e += blockDim.x) {
        const int s_ = e >> 3, r = e & 7;
        double w = 0.0;
        if (r < 5) {
            const int p = (8 * r) % 5, idx = p + 5 * (ES_M - 1 - s_);
            if (idx <= 2 * ES_HALF) w = h[idx];
        }
        tab[e] = w;
    }
}

__global__ __launch_bounds__(256) void estoi_resample5_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const double* __restrict__ htab, int L, EstoiWs ws) {
    constexpr int NX = 8 * 255 + 124;
    __shared__ __attribute__((aligned(16))) float xs[NX + 4];
    __shared__ double outs[1280];
    const int b = blockIdx.y, sig = blockIdx.z, tid = threadIdx.x;
    const int g0 = blockIdx.x * 256, i0 = 5 * g0, n10b = es_n10(ws, b);
    if (i0 >= n10b) return;
    const float* src = (sig == 0 ? x : y) + (size_t)b * L;
    L = es_len(ws, b);
    const int nbase = 8 * g0 - (ES_M - 1) / 2;             // input index of xs[0]: x[8 g0 - 58] (may lie before the signal: zeros)
    for (int k = tid; k < NX; k += 256) {
        const int idx = nbase + k;
        xs[k] = (idx >= 0 && idx < L) ? src[idx] : 0.f;
    }
    __syncthreads();
    const float4* xq = reinterpret_cast<const float4*>(xs + 8 * tid);
    float X[124];
#pragma unroll
    for (int k = 0; k < 31; ++k) {
        const float4 v = xq[k];
        X[4 * k] = v.x; X[4 * k + 1] = v.y; X[4 * k + 2] = v.z; X[4 * k + 3] = v.w;
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
#pragma unroll
    for (int s_ = 0; s_ < ES_M; ++s_) {
        a0 += (double)X[s_] * htab[8 * s_];
        a1 += (double)X[s_ + 1] * htab[8 * s_ + 1];
        a2 += (double)X[s_ + 3] * htab[8 * s_ + 2];
        a3 += (double)X[s_ + 4] * htab[8 * s_ + 3];
        a4 += (double)X[s_ + 6] * htab[8 * s_ + 4];
    }
    outs[5 * tid] = a0; outs[5 * tid + 1] = a1; outs[5 * tid + 2] = a2; outs[5 * tid + 3] = a3; outs[5 * tid + 4] = a4;
    __syncthreads();
    double* dst = ws.xr + ((size_t)b * 2 + sig) * ws.n10 + i0;
    for (int k = tid; k < 1280; k += 256)
        if (i0 + k < n10b) dst[k] = outs[k];
}

// one block per utterance
__global__ __launch_bounds__(256) void estoi_vad_kernel(EstoiWs ws) {
    __shared__ double red[8];
    __shared__ int scan[256];
    __shared__ int base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* x = ws.xr + (size_t)b * 2 * ws.n10;
    double* en = ws.en + (size_t)b * ws.F;
    const int Fb = es_frames(ws, b);
    double mx = -1e300;
    double hwj[ES_NFRAME / 64];                              // the lane's four window values, once (not a float64 cospi per frame and sample)
#pragma unroll
    for (int i = 0; i < ES_NFRAME / 64; ++i) hwj[i] = hann_sym256(lane + 64 * i);
    for (int f = wave; f < Fb; f += 4) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < ES_NFRAME / 64; ++i) {
            const double v = hwj[i] * x[(size_t)f * ES_HOP + lane + 64 * i];
            s += v * v;
        }
        s = wave_sum(s);
        const double e = 20.0 * log10(sqrt(s) + ES_EPS);
        if (lane == 0) en[f] = e;
        mx = fmax(mx, e);
    }
    mx = block_max(mx, red);
    // ordered compaction of frames with (max - 40 - en) < 0
    if (tid == 0) base = 0;
    __syncthreads();
    for (int f0 = 0; f0 < Fb; f0 += 256) {
        const int f = f0 + tid;
        const int k = (f < Fb && (mx - 40.0 - en[f]) < 0.0) ? 1 : 0;
        scan[tid] = k;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int v = (tid >= o) ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        if (k) ws.keep[(size_t)b * ws.F + base + scan[tid] - 1] = f;
        __syncthreads();
        if (tid == 255) base += scan[255];
        __syncthreads();
    }
    if (tid == 0) ws.nkept[b] = base;
}

// grid (F, B), block 256: kept frame k of the silence-removed signals
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(256) void estoi_tob_kernel(EstoiWs ws) {
    __shared__ Fft512Lds s;
    __shared__ double p2[2][NELE_NBINS];
    const int b = blockIdx.y, k = blockIdx.x, tid = threadIdx.x;
    const int nk = ws.nkept[b];
    if (k >= nk) return;
    const int* keep = ws.keep + (size_t)b * ws.F;
    const double* x = ws.xr + (size_t)b * 2 * ws.n10;
    const double* y = x + ws.n10;
    fft512_init_twiddles(s);
    {
        const int j = tid;  // sample j of frame k of the silence-removed signal
        // x_sil[128 k + j] = sum over kept frames kk covering it of w[o] * x[128 f_kk + o], lower kk first
        double xs = 0.0, ys = 0.0;
        int kk0, o0, kk1, o1;
        if (j < ES_HOP) { kk0 = k - 1; o0 = j + ES_HOP; kk1 = k; o1 = j; }
        else { kk0 = k; o0 = j; kk1 = k + 1; o1 = j - ES_HOP; }
        if (kk0 >= 0 && kk0 < nk) {
            const double w = hann_sym256(o0);
            const size_t p = (size_t)keep[kk0] * ES_HOP + o0;
            xs += w * x[p];
            ys += w * y[p];
        }
        if (kk1 >= 0 && kk1 < nk) {
            const double w = hann_sym256(o1);
            const size_t p = (size_t)keep[kk1] * ES_HOP + o1;
            xs += w * x[p];
            ys += w * y[p];
        }
        const double w = hann_sym256(j);
        s.x[fft512_brev(j)] = make_double2(w * xs, w * ys);
        s.x[fft512_brev(j + 256)] = make_double2(0.0, 0.0);
    }
    __syncthreads();
    fft512_run<false>(s);
    for (int q = tid; q < NELE_NBINS; q += 256) {
        const double2 zk = s.x[q], zn = s.x[(512 - q) & 511];
        const double ar = 0.5 * (zk.x + zn.x), ai = 0.5 * (zk.y - zn.y);
        const double br = 0.5 * (zk.y + zn.y), bi = 0.5 * (zn.x - zk.x);
        p2[0][q] = ar * ar + ai * ai;
        p2[1][q] = br * br + bi * bi;
    }
    __syncthreads();
    if (tid < 2 * ES_NB) {
        const int sig = tid / ES_NB, band = tid - sig * ES_NB;
        double a = 0.0;
        for (int q = c_tob_lo[band]; q < c_tob_hi[band]; ++q) a += p2[sig][q];
        ws.tob[(((size_t)b * 2 + sig) * ES_NB + band) * ws.F + k] = sqrt(a);
    }
}
#endif  // NELE_AB

// The same, one WAVE per kept frame with the transform in registers (fft512_wave, third session of round 3; NELE_ESTOI_TOBW=0 = the
// kernel above): twiddles and the window once per workgroup of 4 x ETW_NP frames, no workgroup barrier per frame; only the bins of the
// fifteen bands (7 .. 218) are unpacked.  grid (ceil(F / (4 ETW_NP)), B), block 256.
#define ETW_NP 2
__global__ __launch_bounds__(256) void estoi_tob_wave_kernel(EstoiWs ws) {
    __shared__ double2 tw[256];
    __shared__ double hw[ES_NFRAME];
    __shared__ __attribute__((aligned(16))) double2 xs[4][FFTW_SLOTS];
    const int b = blockIdx.y, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int nk = ws.nkept[b];
    {
        double sn, cs;
        sincospi((double)tid / 256.0, &sn, &cs);
        tw[tid] = make_double2(cs, -sn);
        hw[tid] = hann_sym256(tid);
    }
    __syncthreads();
    if ((int)blockIdx.x * ETW_NP * 4 >= nk) return;
    const int* keep = ws.keep + (size_t)b * ws.F;
    const double* x = ws.xr + (size_t)b * 2 * ws.n10;
    const double* y = x + ws.n10;
    double2* xw = xs[wv];
    double* p2x = reinterpret_cast<double*>(xw);            // |X|^2 / |Y|^2 of the band bins overlay the exchange buffer
    double* p2y = p2x + 256;
    for (int it = 0; it < ETW_NP; ++it) {
        const int k = ((int)blockIdx.x * ETW_NP + it) * 4 + wv;
        if (k >= nk) break;
        double2 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int j = fftw_n(lane, r);                  // sample j of frame k of the silence-removed signal (zero padding behind 256)
            double xs_ = 0.0, ys_ = 0.0;
            if (j < ES_NFRAME) {
                // x_sil[128 k + j] = sum over kept frames kk covering it of w[o] * x[128 f_kk + o], lower kk first
                int kk0, o0, kk1, o1;
                if (j < ES_HOP) { kk0 = k - 1; o0 = j + ES_HOP; kk1 = k; o1 = j; }
                else { kk0 = k; o0 = j; kk1 = k + 1; o1 = j - ES_HOP; }
                if (kk0 >= 0 && kk0 < nk) {
                    const double w = hw[o0];
                    const size_t p = (size_t)keep[kk0] * ES_HOP + o0;
                    xs_ += w * x[p];
                    ys_ += w * y[p];
                }
                if (kk1 >= 0 && kk1 < nk) {
                    const double w = hw[o1];
                    const size_t p = (size_t)keep[kk1] * ES_HOP + o1;
                    xs_ += w * x[p];
                    ys_ += w * y[p];
                }
                const double w = hw[j];
                xs_ *= w; ys_ *= w;
            }
            v[r] = make_double2(xs_, ys_);
        }
        fft512_wave<false>(v, xw, tw, lane);
        fft512_wave_store(v, xw, lane);
        double px[4], py[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = lane + 64 * i;                    // bins 0 .. 255; the bands end at 218
            const double2 zk = xw[fftw_slot(q)], zn = xw[fftw_slot((512 - q) & 511)];
            const double ar = 0.5 * (zk.x + zn.x), ai = 0.5 * (zk.y - zn.y);
            const double br = 0.5 * (zk.y + zn.y), bi = 0.5 * (zn.x - zk.x);
            px[i] = ar * ar + ai * ai;
            py[i] = br * br + bi * bi;
        }
        fftw_wave_sync();                                   // every lane has read its bins
#pragma unroll
        for (int i = 0; i < 4; ++i) { p2x[lane + 64 * i] = px[i]; p2y[lane + 64 * i] = py[i]; }
        fftw_wave_sync();
        if (lane < 2 * ES_NB) {
            const int sig = lane / ES_NB, band = lane - sig * ES_NB;
            const double* pp = sig ? p2y : p2x;
            double a = 0.0;
            for (int q = c_tob_lo[band]; q < c_tob_hi[band]; ++q) a += pp[q];
            ws.tob[(((size_t)b * 2 + sig) * ES_NB + band) * ws.F + k] = sqrt(a);
        }
        fftw_wave_sync();                                   // the buffer is restaged by the next frame
    }
}

// grid (F, B), block 64: segment m = frames m .. m+29
__global__ __launch_bounds__(64) void estoi_seg_kernel(EstoiWs ws) {
    __shared__ double sx[ES_NB][ES_NSEG + 1], sy[ES_NB][ES_NSEG + 1];
    __shared__ double part[ES_NSEG];
    const int b = blockIdx.y, m = blockIdx.x, tid = threadIdx.x;
    const int nk = ws.nkept[b];
    if (m + ES_NSEG > nk) return;
    const double* tx = ws.tob + (size_t)b * 2 * ES_NB * ws.F;
    const double* ty = tx + (size_t)ES_NB * ws.F;
    for (int i = tid; i < ES_NB * ES_NSEG; i += 64) {
        const int r = i / ES_NSEG, c = i - r * ES_NSEG;
        sx[r][c] = tx[(size_t)r * ws.F + m + c];
        sy[r][c] = ty[(size_t)r * ws.F + m + c];
    }
    __syncthreads();
    if (tid < 2 * ES_NB) {  // rows: subtract mean over time, divide by norm
        double(*a)[ES_NSEG + 1] = (tid < ES_NB) ? sx : sy;
        const int r = tid % ES_NB;
        double mu = 0.0;
        for (int c = 0; c < ES_NSEG; ++c) mu += a[r][c];
        mu /= (double)ES_NSEG;
        double nn = 0.0;
        for (int c = 0; c < ES_NSEG; ++c) { const double v = a[r][c] - mu; a[r][c] = v; nn += v * v; }
        const double inv = 1.0 / sqrt(nn);
        for (int c = 0; c < ES_NSEG; ++c) a[r][c] *= inv;
    }
    __syncthreads();
    if (tid < 2 * ES_NSEG) {  // columns: subtract mean over bands, divide by norm
        double(*a)[ES_NSEG + 1] = (tid < ES_NSEG) ? sx : sy;
        const int c = tid % ES_NSEG;
        double mu = 0.0;
        for (int r = 0; r < ES_NB; ++r) mu += a[r][c];
        mu /= (double)ES_NB;
        double nn = 0.0;
        for (int r = 0; r < ES_NB; ++r) { const double v = a[r][c] - mu; a[r][c] = v; nn += v * v; }
        const double inv = 1.0 / sqrt(nn);
        for (int r = 0; r < ES_NB; ++r) a[r][c] *= inv;
    }
    __syncthreads();
    if (tid < ES_NSEG) {
        double d = 0.0;
        for (int r = 0; r < ES_NB; ++r) d += sx[r][tid] * sy[r][tid];
        part[tid] = d;
    }
    __syncthreads();
    if (tid == 0) {
        double d = 0.0;
        for (int c = 0; c < ES_NSEG; ++c) d += part[c];
        ws.dseg[(size_t)b * ws.F + m] = d / (double)ES_NSEG;
    }
}

__global__ void estoi_final_kernel(EstoiWs ws, float* __restrict__ raw, float* __restrict__ mapped, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int nk = ws.nkept[b];
    double d;
    if (nk < ES_NSEG) {
        d = 1e-5;  // pystoi: not enough frames
    } else {
        const int J = nk - ES_NSEG + 1;
        double s = 0.0;
        for (int m = 0; m < J; ++m) s += ws.dseg[(size_t)b * ws.F + m];
        d = s / (double)J;
    }
    if (raw) raw[b] = (float)d;
    if (mapped) mapped[b] = (float)(1.0 / (1.0 + exp(-8.0 * (d - 0.25))));
}

// ------------------------------------------------------------------------------------------ C ABI
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static void estoi_dims(int L, int* n10, int* F) {
    const long long n = (long long)L * 5;
    *n10 = (int)(n / 8 + (n % 8 ? 1 : 0));
    *F = (*n10 >= ES_NFRAME) ? (*n10 - ES_NFRAME) / ES_HOP + 1 : 0;
}

extern "C" long long nele_metric_estoi_workspace_bytes(int B, int L) {
    int n10, F;
    estoi_dims(L, &n10, &F);
    size_t t = align256(sizeof(double) * (ES_HPAD + ES_HTAB));
    t += align256(sizeof(double) * (size_t)B * 2 * n10);
    t += align256(sizeof(double) * (size_t)B * F);
    t += align256(sizeof(int) * (size_t)B * F);
    t += align256(sizeof(int) * (size_t)B);
    t += align256(sizeof(double) * (size_t)B * 2 * ES_NB * F);
    t += align256(sizeof(double) * (size_t)B * F);
    return (long long)t;
}

extern "C" int nele_metric_estoi_var(const float* x, const float* y, const int* lengths, int B, int L, void* workspace, long long workspace_bytes,
                                     float* raw, float* mapped, void* stream) {
    NELE_CHECK_ARG(x && y && workspace && (raw || mapped) && B > 0, "nele_metric_estoi: bad arguments");
    int n10, F;
    estoi_dims(L, &n10, &F);
    if (F < 1) return nele_set_error(NELE_ERR_SIGNAL, "nele_metric_estoi: L=%d too short", L);
    if (workspace_bytes < nele_metric_estoi_workspace_bytes(B, L))
        return nele_set_error(NELE_ERR_WORKSPACE, "nele_metric_estoi: workspace too small");
    char* p = (char*)workspace;
    double* h = (double*)p; p += align256(sizeof(double) * (ES_HPAD + ES_HTAB));     // filter + the five-output step table
    EstoiWs ws;
    ws.n10 = n10; ws.F = F; ws.lens = lengths; ws.L = L;
    ws.xr = (double*)p; p += align256(sizeof(double) * (size_t)B * 2 * n10);
    ws.en = (double*)p; p += align256(sizeof(double) * (size_t)B * F);
    ws.keep = (int*)p; p += align256(sizeof(int) * (size_t)B * F);
    ws.nkept = (int*)p; p += align256(sizeof(int) * (size_t)B);
    ws.tob = (double*)p; p += align256(sizeof(double) * (size_t)B * 2 * ES_NB * F);
    ws.dseg = (double*)p;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(estoi_filter_kernel, dim3(1), dim3(1024), 0, s, h);
    const int rs5 = NELE_SWITCH_INT("NELE_ESTOI_RS5", 1);                                   // NELE_ESTOI_RS5=0: the output-per-thread resampler (A/B diagnostic)
    if (rs5) {
        hipLaunchKernelGGL(estoi_taps_kernel, dim3(1), dim3(256), 0, s, h);
        hipLaunchKernelGGL(estoi_resample5_kernel, dim3((n10 + 1279) / 1280, B, 2), dim3(256), 0, s, x, y, h + ES_HPAD, L, ws);
    } else {
        NELE_AB_ONLY(hipLaunchKernelGGL(estoi_resample_kernel, dim3((n10 + 255) / 256, B, 2), dim3(256), 0, s, x, y, h, L, ws);)
    }
    hipLaunchKernelGGL(estoi_vad_kernel, dim3(B), dim3(256), 0, s, ws);
    const int tobw = NELE_SWITCH_INT("NELE_ESTOI_TOBW", 1);                                  // NELE_ESTOI_TOBW=0: the workgroup-per-frame kernel (A/B diagnostic)
    if (tobw) hipLaunchKernelGGL(estoi_tob_wave_kernel, dim3((F + 4 * ETW_NP - 1) / (4 * ETW_NP), B), dim3(256), 0, s, ws);
    else { NELE_AB_ONLY(hipLaunchKernelGGL(estoi_tob_kernel, dim3(F, B), dim3(256), 0, s, ws);) }
    hipLaunchKernelGGL(estoi_seg_kernel, dim3(F, B), dim3(64), 0, s, ws);
    hipLaunchKernelGGL(estoi_final_kernel, dim3((B + 63) / 64), dim3(64), 0, s, ws, raw, mapped, B);
    NELE_CHECK_LAUNCH("nele_metric_estoi");
    return NELE_OK;
}

extern "C" int nele_metric_estoi(const float* x, const float* y, int B, int L, void* workspace, long long workspace_bytes, float* raw,
                                 float* mapped, void* stream) {
    return nele_metric_estoi_var(x, y, nullptr, B, L, workspace, workspace_bytes, raw, mapped, stream);
}
