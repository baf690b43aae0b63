// Signal features / resynthesis kernels.
//   nele_stft_band  : wav -> complex64 spectrum + 64-band energies ** p      (audio_util.py:53-58, 30-50, 422-437)
//   nele_imcra_band : spectrum -> IMCRA noise PSD + band energies ** p        (noise_est/imcra.py:363-484, 521-577; audio_util.py:113-117, 439-456)
//   nele_gain_istft : alpha^2 band gains + spectrum -> waveform               (audio_util.py:76-110, 458-461, 60-65)
//   nele_wav_post   : optional RMS normalisation + PCM_16 round trip          (inference.py:109, 115; train_nele.py:313)
//
// Layouts in HBM (row-major): wav [B][L] f32; spec [B][T][257] float2 (frame-major so a frame's bins
// are one coalesced 2 KB run); band [B][T][64] f32; psd [B][T][257] f32.
// This file is compiled with -ffp-contract=off: the float32/float64 op sequence below mirrors numpy's.
#include "common.h"
#include "fft512.h"

__constant__ int c_gmt[NELE_NBANDS] = {0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23,
                                        24, 25, 26, 28, 30, 32, 34, 36, 38, 41, 43, 46, 49, 52, 55, 58, 62, 66, 70, 74,
                                        79, 83, 88, 93, 99, 105, 111, 117, 124, 131, 139, 147, 156, 165, 174, 184, 195,
                                        206, 218, 230, 243, 257};

// Band energy of band i from tmp[257] (= |X|^2, float32), in the accumulation order of
// compute_band_E (audio_util.py:40-48): first the "frac" shares of band i-1's bins, then the
// "1-frac" shares of band i's own bins; float32 products, float64 running sum.
__device__ __forceinline__ float band_energy(const float* tmp, int i) {
    double acc = 0.0;
    if (i >= 1) {
        const int g0 = c_gmt[i - 1], size = c_gmt[i] - g0;
        for (int j = 0; j < size; ++j) {
            const float fr = (float)((double)j / (double)size);
            acc += (double)(fr * tmp[g0 + j]);
        }
    }
    if (i <= NELE_NBANDS - 2) {
        const int g0 = c_gmt[i], size = c_gmt[i + 1] - g0;
        for (int j = 0; j < size; ++j) {
            const float fr = (float)(1.0 - (double)j / (double)size);
            acc += (double)(fr * tmp[g0 + j]);
        }
    }
    return (float)acc;
}

// The same sums with the per-bin weights from LDS tables: band_energy() pays a float64 division per term on the one wave that runs it
// (PMC: stft_band_wave_kernel and band_from_psd_kernel are bound by instruction issue, and the divisions are most of it).
// frw[k] = (float)(j / size), omw[k] = (float)(1 - j / size) for bin k = g0 + j of its band - the very expressions above, evaluated once
// per workgroup by one thread per BAND (<= 14 bins each; a per-bin fill that searches its band costs more than it saves: measured), so
// every product and the order of the float64 sum are unchanged: bit-identical.
__device__ __forceinline__ void band_weights_init(float* __restrict__ frw, float* __restrict__ omw) {
    const int i = threadIdx.x;
    if (i <= NELE_NBANDS - 2) {
        const int g0 = c_gmt[i], size = c_gmt[i + 1] - g0;
        for (int j = 0; j < size; ++j) {
            frw[g0 + j] = (float)((double)j / (double)size);
            omw[g0 + j] = (float)(1.0 - (double)j / (double)size);
        }
    }
}
__device__ __forceinline__ float band_energy_w(const float* tmp, const float* __restrict__ frw, const float* __restrict__ omw, int i) {
    double acc = 0.0;
    if (i >= 1) {
        const int g0 = c_gmt[i - 1], g1 = c_gmt[i];
        for (int k = g0; k < g1; ++k) acc += (double)(frw[k] * tmp[k]);
    }
    if (i <= NELE_NBANDS - 2) {
        const int g0 = c_gmt[i], g1 = c_gmt[i + 1];
        for (int k = g0; k < g1; ++k) acc += (double)(omw[k] * tmp[k]);
    }
    return (float)acc;
}

// np.abs(complex64) as numpy >= 1.25 computes it on FMA-capable x86 hosts (SIMD loop
// loops_unary_complex: larger * sqrt(fma(r, r, 1)), r = smaller / larger), so that |X|^2 is
// bit-identical to the oracle's and threshold decisions downstream (IMCRA, VAD) see the same values.
__device__ __forceinline__ float np_cabsf(float re, float im) {
    re = fabsf(re);
    im = fabsf(im);
    const float larger = fmaxf(re, im), smaller = fminf(im, re);
    if (larger == 0.f) return 0.f;
    const float ratio = smaller / larger;
    return sqrtf(__builtin_fmaf(ratio, ratio, 1.0f)) * larger;
}

// the power law of the band features: nele_pow_f32 (common.h)
__device__ __forceinline__ float pow_f32(float x, float p) { return nele_pow_f32(x, p); }

__device__ __forceinline__ double hann512(int n) { return 0.5 - 0.5 * cospi((double)n / 256.0); }

// ------------------------------------------------------------------------------------------ STFT
// grid (ceil(T/2), B), block 256.  Frames 2p and 2p+1 share one complex FFT.
// lens (may be NULL): samples of each utterance inside the padded [B][L] buffer (the reference handles files of any length one at a
// time, dataloader.py:30-42); frames at or behind a short row's own count T_b = 1 + L_b / 256 are written as zeros.
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(256) void stft_band_kernel(const float* __restrict__ wav, int L, int T, float power,
                                                        float2* __restrict__ spec, float* __restrict__ band, const int* __restrict__ lens) {
    __shared__ Fft512Lds s;
    __shared__ float tmp[2][NELE_NBINS + 3];
    const int b = blockIdx.y, t0 = 2 * blockIdx.x, t1 = t0 + 1;
    const float* x = wav + (size_t)b * L;
    int Tb = T;
    if (lens) { L = min(lens[b], L); Tb = 1 + L / NELE_HOP; }
    for (int t = max(t0, Tb); t <= t1 && t < T; ++t) {     // frames behind the end of a short row
        if (spec) for (int k = threadIdx.x; k < NELE_NBINS; k += 256) spec[((size_t)b * T + t) * NELE_NBINS + k] = make_float2(0.f, 0.f);
        if (band && threadIdx.x < NELE_NBANDS) band[((size_t)b * T + t) * NELE_NBANDS + threadIdx.x] = 0.f;
    }
    if (t0 >= Tb) return;
    const bool has1 = t1 < Tb;
    fft512_init_twiddles(s);
    for (int n = threadIdx.x; n < NELE_NFFT; n += 256) {
        int o0 = NELE_HOP * t0 + n - NELE_HOP;
        if (o0 < 0) o0 = -o0;
        if (o0 >= L) o0 = 2 * (L - 1) - o0;
        int o1 = o0;
        if (has1) {
            o1 = NELE_HOP * t1 + n - NELE_HOP;
            if (o1 >= L) o1 = 2 * (L - 1) - o1;
        }
        const double w = hann512(n);
        const double zr = w * (double)x[o0];
        const double zi = has1 ? w * (double)x[o1] : 0.0;
        s.x[fft512_brev(n)] = make_double2(zr, zi);
    }
    __syncthreads();
    fft512_run<false>(s);
    for (int k = threadIdx.x; k < NELE_NBINS; k += 256) {
        const double2 zk = s.x[k], zn = s.x[(NELE_NFFT - k) & (NELE_NFFT - 1)];
        const float2 A = make_float2((float)(0.5 * (zk.x + zn.x)), (float)(0.5 * (zk.y - zn.y)));
        const float2 Bv = make_float2((float)(0.5 * (zk.y + zn.y)), (float)(0.5 * (zn.x - zk.x)));
        if (spec) {
            spec[((size_t)b * T + t0) * NELE_NBINS + k] = A;
            if (has1) spec[((size_t)b * T + t1) * NELE_NBINS + k] = Bv;
        }
        // np.abs(complex64) -> float32 magnitude, squared in float32 (audio_util.py:428, 44)
        const float m0 = np_cabsf(A.x, A.y);
        const float m1 = np_cabsf(Bv.x, Bv.y);
        tmp[0][k] = m0 * m0;
        tmp[1][k] = m1 * m1;
    }
    __syncthreads();
    if (band && threadIdx.x < 2 * NELE_NBANDS) {
        const int f = threadIdx.x >> 6, i = threadIdx.x & 63, t = t0 + f;
        if (t < Tb) {
            const float e = band_energy(tmp[f], i);
            band[((size_t)b * T + t) * NELE_NBANDS + i] = pow_f32(e, power);
        }
    }
}
#endif  // NELE_AB

// ---- the same STFT, one WAVE per frame pair with the transform in registers (fft512_wave: two LDS exchanges instead of nine workgroup
// barriers; same butterflies, same twiddles, same window values: bit-identical spectra and band features, tests/test_features_gpu.py).
// A workgroup = 4 waves x STW_NP pairs; twiddles and the Hann window are computed once per workgroup.  A lane loads the 8 samples its
// registers start from straight from memory (n = fftw_n(lane, r): the lanes of a load cover 64 consecutive samples).
// grid (ceil(ceil(T / 2) / (4 STW_NP)), B), block 256.
#define STW_NP 2
__global__ __launch_bounds__(256) void stft_band_wave_kernel(const float* __restrict__ wav, int L, int T, float power,
                                                             float2* __restrict__ spec, float* __restrict__ band, const int* __restrict__ lens, float* __restrict__ pw) {
    __shared__ double2 tw[256];
    __shared__ double hw[NELE_NFFT];
    __shared__ __attribute__((aligned(16))) double2 xs[4][FFTW_SLOTS];
    __shared__ float frw[NELE_NBINS + 3], omw[NELE_NBINS + 3];
    const int b = blockIdx.y, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const float* x = wav + (size_t)b * L;
    if (band) band_weights_init(frw, omw);
    int Tb = T;
    if (lens) { L = min(lens[b], L); Tb = 1 + L / NELE_HOP; }
    {
        double sn, cs;
        sincospi((double)tid / 256.0, &sn, &cs);
        tw[tid] = make_double2(cs, -sn);
        hw[tid] = hann512(tid);
        hw[tid + 256] = hann512(tid + 256);
    }
    __syncthreads();
    double2* xw = xs[wv];
    for (int it = 0; it < STW_NP; ++it) {
        const int t0 = 2 * ((blockIdx.x * STW_NP + it) * 4 + wv), t1 = t0 + 1;
        if (t0 >= T) break;
        for (int t = max(t0, Tb); t <= t1 && t < T; ++t) {     // frames behind the end of a short row
            if (spec) for (int k = lane; k < NELE_NBINS; k += 64) spec[((size_t)b * T + t) * NELE_NBINS + k] = make_float2(0.f, 0.f);
            if (band) band[((size_t)b * T + t) * NELE_NBANDS + lane] = 0.f;
        }
        if (t0 >= Tb) continue;
        const bool has1 = t1 < Tb;
        double2 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n = fftw_n(lane, r);
            int o0 = NELE_HOP * t0 + n - NELE_HOP;
            if (o0 < 0) o0 = -o0;
            if (o0 >= L) o0 = 2 * (L - 1) - o0;
            int o1 = o0;
            if (has1) {
                o1 = NELE_HOP * t1 + n - NELE_HOP;
                if (o1 >= L) o1 = 2 * (L - 1) - o1;
            }
            const double w = hw[n];
            const double zr = w * (double)x[o0];
            const double zi = has1 ? w * (double)x[o1] : 0.0;
            v[r] = make_double2(zr, zi);
        }
        fft512_wave<false>(v, xw, tw, lane);
        fft512_wave_store(v, xw, lane);
        float p0[5], p1[5];                                 // |X|^2 of the lane's bins, both frames
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int k = lane + 64 * q;
            if (k < NELE_NBINS) {
                const double2 zk = xw[fftw_slot(k)], zn = xw[fftw_slot((NELE_NFFT - k) & (NELE_NFFT - 1))];
                const float2 A = make_float2((float)(0.5 * (zk.x + zn.x)), (float)(0.5 * (zk.y - zn.y)));
                const float2 Bv = make_float2((float)(0.5 * (zk.y + zn.y)), (float)(0.5 * (zn.x - zk.x)));
                if (spec) {
                    spec[((size_t)b * T + t0) * NELE_NBINS + k] = A;
                    if (has1) spec[((size_t)b * T + t1) * NELE_NBINS + k] = Bv;
                }
                const float m0 = np_cabsf(A.x, A.y);
                const float m1 = np_cabsf(Bv.x, Bv.y);
                p0[q] = m0 * m0;
                p1[q] = m1 * m1;
                if (pw) {                                   // |X|^2 as numpy squares np.abs(complex64): what IMCRA starts from (nele_imcra_band_pw)
                    pw[((size_t)b * T + t0) * NELE_NBINS + k] = p0[q];
                    if (has1) pw[((size_t)b * T + t1) * NELE_NBINS + k] = p1[q];
                }
            }
        }
        if (band) {
            fftw_wave_sync();                               // every lane has read its bins: the exchange buffer is free for |X|^2 (float32)
            float* tmp0 = reinterpret_cast<float*>(xw);
            float* tmp1 = tmp0 + NELE_NBINS + 3;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int k = lane + 64 * q;
                if (k < NELE_NBINS) { tmp0[k] = p0[q]; tmp1[k] = p1[q]; }
            }
            fftw_wave_sync();
            band[((size_t)b * T + t0) * NELE_NBANDS + lane] = pow_f32(band_energy_w(tmp0, frw, omw, lane), power);
            if (has1) band[((size_t)b * T + t1) * NELE_NBANDS + lane] = pow_f32(band_energy_w(tmp1, frw, omw, lane), power);
        }
        fftw_wave_sync();                                   // xw is rewritten by the next pair
    }
}

// ------------------------------------------------------------------------------------------ IMCRA
// One block (320 threads, bins 0..256 active) per utterance; serial over frames (true recurrence),
// neighbour bins exchanged through LDS.  float32 / float64 staging follows numpy (see oracle/features.py).
#define IMCRA_THREADS 320
struct ImcraLds {
    double y2[2][NELE_NBINS + 2];   // |Y|^2 of the frame, by frame parity
    double a[NELE_NBINS + 2];
    double b[NELE_NBINS + 2];
    double st[8][NELE_NBINS];
    double tst[8][NELE_NBINS];
    float tmp[NELE_NBINS + 3];
};

__device__ __forceinline__ double fsmooth3(const double* v, int k, double w0, double w1, double w2) {
    return (w0 * v[k] + w1 * v[k + 1]) + w2 * v[k + 2];  // v is offset by one: v[k+1] is bin k
}

// BAND_IN_LOOP = false: the band feature is left to band_from_psd_kernel (parallel over frames); the serial loop then has two
// barriers per frame instead of four and no 64-band reduction on the wave that paces it.
template <bool BAND_IN_LOOP>
__global__ __launch_bounds__(IMCRA_THREADS) void imcra_band_kernel(const float2* __restrict__ spec, int T, float power,
                                                                   float* __restrict__ psd, float* __restrict__ band,
                                                                   const int* __restrict__ tlens) {
    __shared__ ImcraLds s;
    const int b = blockIdx.x, k = threadIdx.x;
    const bool act = k < NELE_NBINS;
    const bool edge_lo = (k == 0), edge_hi = (k == NELE_NBINS - 1);
    const float2* Y = spec + (size_t)b * T * NELE_NBINS;
    const double alpha_s = 0.9, alpha_d = 0.85, Bmin = 3.2, Gamma0 = 4.6, Gamma1 = 3.0, zeta0 = 1.67, beta = 1.47;
    const double alpha_dd = 0.92, xi_min = pow(10.0, -25.0 / 20.0), p_up = 0.9;
    const double one_m_as = 1.0 - alpha_s, one_m_ad = 1.0 - alpha_d, one_m_add = 1.0 - alpha_dd;
    double w0 = 0.25, w1 = 0.5, w2 = 0.25;  // imcra.py:270-280 (sym_hanning(3), row-normalised)
    if (edge_lo) { w0 = 0.0; w1 = 1.0 / 1.5; w2 = 0.5 / 1.5; }
    if (edge_hi) { w0 = 0.5 / 1.5; w1 = 1.0 / 1.5; w2 = 0.0; }

    double S = 0, tS = 0, Smin = 0, tSmin = 0, Smin_sw = 0, tSmin_sw = 0, ovL = 0, Lam64 = 1e-6, G = 1.0, Gamma = 1.0;
    float Lam32 = 0.f;
    int j = 0, u = 0;
    // the spectrum of frame l+1 is loaded while frame l is processed: the recursion is serial over frames and would otherwise pay
    // a full global-load latency per frame
    float2 ynext = act ? Y[k] : make_float2(0.f, 0.f);
    const int Tb = tlens ? min(tlens[blockIdx.x], T) : T;   // frames of this utterance inside the padded batch
    for (int l = Tb; l < T; ++l) {                          // behind the end of a short row: zeros
        if (act && psd) psd[((size_t)b * T + l) * NELE_NBINS + k] = 0.f;
        if (BAND_IN_LOOP && band && k < NELE_NBANDS) band[((size_t)b * T + l) * NELE_NBANDS + k] = 0.f;
    }
    for (int l = 0; l < Tb; ++l) {
        float Y2f = 0.f, outv = 0.f;
        double xi = 0.0, I = 0.0;
        const float2 y = ynext;
        if (act && l + 1 < Tb) ynext = Y[(size_t)(l + 1) * NELE_NBINS + k];
        if (act) {
            const float h = np_cabsf(y.x, y.y);                                   // np.abs(complex64)
            Y2f = h * h;                                                          // **2 on a float32 array
            double* y2 = s.y2[l & 1];
            y2[k + 1] = (double)Y2f;
            if (edge_lo) y2[0] = (double)Y2f;
            if (edge_hi) y2[NELE_NBINS + 1] = (double)Y2f;
        }
        __syncthreads();
        if (act) {
            const double Y2 = (double)Y2f;
            const double Sf = fsmooth3(s.y2[l & 1], k, w0, w1, w2);
            // ---- decision-directed a-priori SNR (imcra.py:543-557)
            const double xi_G = (l == 0) ? 1.0 : (G * G) * Gamma;
            double term;
            if (l == 0 || l >= 16) {
                Gamma = Y2 / Lam64;
                double xi_ML = Gamma - 1.0;
                if (xi_ML < 1e-6) xi_ML = 1e-6;
                term = one_m_add * xi_ML;
            } else {  // Lambda_D (hence Gamma, xi_ML) is a float32 array for frames 1..15
                const float Gf = Y2f / Lam32;
                float xf = Gf - 1.0f;
                if (xf < (float)1e-6) xf = (float)1e-6;
                term = (double)((float)one_m_add * xf);
                Gamma = (double)Gf;
            }
            xi = alpha_dd * xi_G + term;
            if (xi < xi_min) xi = xi_min;
            G = xi / (1.0 + xi);
            // ---- imcra.update (imcra.py:363-484)
            if (l == 0) {  // init_params (imcra.py:338-361)
                S = Sf; tS = Sf; Smin = Sf; tSmin = Sf; Smin_sw = Sf; tSmin_sw = Sf;
                ovL = Y2;
                Lam32 = Y2f;
            }
            S = alpha_s * S + one_m_as * Sf;
            Smin = fmin(Smin, S);
            Smin_sw = fmin(Smin_sw, S);
            if (l < 15) {
                Lam32 = (float)alpha_d * Lam32 + (float)one_m_ad * Y2f;
                outv = Lam32;
            } else {
                const double Gamma_min = Y2 / (Bmin * Smin);
                const double zeta = S / (Bmin * Smin);
                I = (Gamma_min < Gamma0 && zeta < zeta0) ? 1.0 : 0.0;
            }
        }
        if (l >= 15) {  // block-uniform branch
            if (act) {
                const double IY = I * (double)Y2f;
                s.a[k + 1] = IY;
                s.b[k + 1] = I;
                if (edge_lo) { s.a[0] = IY; s.b[0] = I; }
                if (edge_hi) { s.a[NELE_NBINS + 1] = IY; s.b[NELE_NBINS + 1] = I; }
            }
            __syncthreads();
            if (act) {
                const double Y2 = (double)Y2f;
                const double norm = fsmooth3(s.b, k, w0, w1, w2);
                double tSf = fsmooth3(s.a, k, w0, w1, w2);
                if (norm > 0.0) tSf = tSf / norm;
                tS = alpha_s * tS + one_m_as * tSf;
                tSmin = fmin(tSmin, tS);
                tSmin_sw = fmin(tSmin_sw, tS);
                const double tG = Y2 / (Bmin * tSmin);
                const double tz = S / (Bmin * tSmin);
                double q = 0.0;
                if (tG <= 1.0 && tz < zeta0) q = 1.0;
                else if (1.0 < tG && tG < Gamma1 && tz < zeta0) q = (Gamma1 - tG) / (Gamma1 - 1.0);
                // post_speech_prob (imcra.py:22-36)
                const double nu = Gamma * xi / (1.0 + xi);
                double p = 0.0;
                if (q < 1.0) p = 1.0 / (1.0 + (q / (1.0 - q)) * (1.0 + xi) * exp(-nu));
                if (p > p_up) p = p_up;
                const double tad = alpha_d + one_m_ad * p;
                ovL = tad * ovL + (1.0 - tad) * Y2;
                Lam64 = beta * ovL;
                outv = (float)Lam64;
                // minimum tracking (imcra.py:452-481); np.roll of the U=8 store == ring buffer
                if (j + 1 == 15) {
                    const int slot = u & 7;
                    s.st[slot][k] = Smin_sw;
                    s.tst[slot][k] = tSmin_sw;
                    const int n = (u < 8) ? (u + 1) : 8;
                    double m = s.st[0][k], tm = s.tst[0][k];
                    for (int i = 1; i < n; ++i) { m = fmin(m, s.st[i][k]); tm = fmin(tm, s.tst[i][k]); }
                    Smin = m; tSmin = tm;
                    Smin_sw = S; tSmin_sw = tS;
                }
            }
            if (++j == 15) { j = 0; ++u; }
        }
        if (act && psd) psd[((size_t)b * T + l) * NELE_NBINS + k] = outv;
        if (BAND_IN_LOOP) {
            // ---- band feature of sqrt(PSD) (audio_util.py:446-451)
            if (act) {
                const float r = sqrtf(outv);
                s.tmp[k] = r * r;
            }
            __syncthreads();
            // the band energy goes out raw: the float64 pow (a few hundred instructions on the wave that paces the serial loop) is applied by
            // band_pow_kernel over all frames at once
            if (band && k < NELE_NBANDS) band[((size_t)b * T + l) * NELE_NBANDS + k] = band_energy(s.tmp, k);
        }
        // LDS hazards without those barriers: |Y|^2 is double-buffered by frame parity; s.a / s.b of frame l are read after frame l's
        // second barrier and rewritten after frame l+1's first one, which every reader has to reach first
    }
}

// ---- IMCRA without a workgroup-wide recursion (round 6).  The one-kernel form above paces every frame through two barriers and four
// dependent levels of float64 divisions (3 000 of its 3 800 cycles per frame - measured by splitting it: docs/REJECTED.md), because the
// neighbour exchange of the frequency smoothing sits inside the serial loop.  But the smoothing is the ONLY thing that couples bins, and
// what it smooths is available ahead of the recursion that consumes it:
//   imcra_pow_kernel    |Y|^2 (float32, as numpy computes it) for every frame and bin at once;
//   imcra_ind_kernel    one THREAD per utterance and bin, serial over frames, no LDS: S_f = smooth(|Y|^2) from the three neighbouring
//                       values, S, S_min and its minima store, the speech indicator I (imcra.py:363-412) -> one byte per frame and bin;
//   imcra_track_kernel  one thread per utterance and bin: S again (one multiply-add: cheaper than storing it), S~_f = smooth(I |Y|^2) /
//                       smooth(I) from the neighbours' indicators, S~, S~_min + store, the speech-absence prior q (:413-429, 452-484) and the
//                       tracker proper (Gamma, xi, G, p, lambda_D: :543-557, 22-36, 430-450) -> PSD.
// The loop-carried chains left are S (a multiply-add), the minima, and the tracker's lambda_D -> Gamma -> nu -> exp -> p -> lambda_D; every
// other division hangs off them and pipelines across the eight frames a thread keeps in flight.  257 bins pack into waves across
// utterances (the one-kernel form spends a fifth of its issue slots on a wave with one live lane).  Same operations on the same operands
// in the same order: bit-identical PSDs (tests/test_features_gpu.py).
__global__ __launch_bounds__(256) void imcra_pow_kernel(const float2* __restrict__ spec, size_t n, float* __restrict__ y2f) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float2 y = spec[i];
        const float h = np_cabsf(y.x, y.y);                                   // np.abs(complex64)
        y2f[i] = h * h;                                                       // **2 on a float32 array
    }
}

#define IMT_PF 8
struct ImcraBin {                                   // what a thread of the two serial kernels knows about its bin
    const float* y2;                                // |Y|^2 row pointer of (utterance, bin)
    int dl, dr;                                     // offsets of the left / right neighbour (0 at the edges: the edge bins stand in for themselves)
    double w0, w1, w2;                              // imcra.py:270-280 (sym_hanning(3), row-normalised)
};
__device__ __forceinline__ ImcraBin imcra_bin(const float* y2f, int b, int k, int T) {
    ImcraBin q;
    q.y2 = y2f + (size_t)b * T * NELE_NBINS + k;
    q.dl = (k == 0) ? 0 : -1;
    q.dr = (k == NELE_NBINS - 1) ? 0 : 1;
    q.w0 = 0.25; q.w1 = 0.5; q.w2 = 0.25;
    if (k == 0) { q.w0 = 0.0; q.w1 = 1.0 / 1.5; q.w2 = 0.5 / 1.5; }
    if (k == NELE_NBINS - 1) { q.w0 = 0.5 / 1.5; q.w1 = 1.0 / 1.5; q.w2 = 0.0; }
    return q;
}

__global__ __launch_bounds__(256) void imcra_ind_kernel(const float* __restrict__ y2f, int B, int T, unsigned char* __restrict__ ind,
                                                        const int* __restrict__ tlens) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= B * NELE_NBINS) return;
    const int b = gid / NELE_NBINS, k = gid - b * NELE_NBINS;
    const ImcraBin bn = imcra_bin(y2f, b, k, T);
    unsigned char* Iout = ind + (size_t)b * T * NELE_NBINS + k;
    const double alpha_s = 0.9, Bmin = 3.2, Gamma0 = 4.6, zeta0 = 1.67;
    const double one_m_as = 1.0 - alpha_s;
    const int Tb = tlens ? min(tlens[b], T) : T;
    double S = 0, Smin = 0, Smin_sw = 0, st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int j = 0, u = 0;
    float fa[IMT_PF][3], fb[IMT_PF][3];
    auto fetch = [&](int l0, float (&f)[IMT_PF][3]) {
#pragma unroll
        for (int d = 0; d < IMT_PF; ++d) {
            const int l = l0 + d;
            const float* r = bn.y2 + (size_t)l * NELE_NBINS;
            f[d][0] = (l < Tb) ? r[bn.dl] : 0.f;
            f[d][1] = (l < Tb) ? r[0] : 0.f;
            f[d][2] = (l < Tb) ? r[bn.dr] : 0.f;
        }
    };
    auto step = [&](int l, const float (&f)[3]) {
        const double Y2 = (double)f[1];
        const double Sf = (bn.w0 * (double)f[0] + bn.w1 * (double)f[1]) + bn.w2 * (double)f[2];
        if (l == 0) { S = Sf; Smin = Sf; Smin_sw = Sf; }
        S = alpha_s * S + one_m_as * Sf;
        Smin = fmin(Smin, S);
        Smin_sw = fmin(Smin_sw, S);
        if (l >= 15) {
            const double Gamma_min = Y2 / (Bmin * Smin);
            const double zeta = S / (Bmin * Smin);
            Iout[(size_t)l * NELE_NBINS] = (Gamma_min < Gamma0 && zeta < zeta0) ? 1 : 0;
            if (j + 1 == 15) {                                   // minimum tracking (imcra.py:452-481); np.roll of the U = 8 store == ring buffer
                const int slot = u & 7;
#pragma unroll
                for (int i = 0; i < 8; ++i) st[i] = (i == slot) ? Smin_sw : st[i];
                const int n = (u < 8) ? (u + 1) : 8;
                double m = st[0];
#pragma unroll
                for (int i = 1; i < 8; ++i) m = (i < n) ? fmin(m, st[i]) : m;
                Smin = m;
                Smin_sw = S;
            }
            if (++j == 15) { j = 0; ++u; }
        }
    };
    fetch(0, fa);
    for (int l0 = 0; l0 < Tb; l0 += 2 * IMT_PF) {
        fetch(l0 + IMT_PF, fb);
#pragma unroll
        for (int d = 0; d < IMT_PF; ++d)
            if (l0 + d < Tb) step(l0 + d, fa[d]);
        fetch(l0 + 2 * IMT_PF, fa);
#pragma unroll
        for (int d = 0; d < IMT_PF; ++d)
            if (l0 + IMT_PF + d < Tb) step(l0 + IMT_PF + d, fb[d]);
    }
}

// Two waves per 64 bins: wave 0 runs the speech-absence prior (S, S~ from the neighbours' indicators, S~_min + store, q -> r = q / (1 - q):
// five divisions that hang off short recurrences - issue-bound), wave 1 the tracker (lambda_D -> Gamma -> nu -> exp -> p -> lambda_D: three
// dependent divisions and an exponential per frame - latency-bound), one batch of IMT_PF frames behind, r through LDS.  In one instruction
// stream the compiler does not interleave the two across frames: a wave that did both took their SUM, 2 500 cycles per frame (0.52 ms at
// T = 501); side by side they take the longer one.
__global__ __launch_bounds__(128) void imcra_track_kernel(const float* __restrict__ y2f, const unsigned char* __restrict__ ind, int B, int T,
                                                          float* __restrict__ psd, const int* __restrict__ tlens) {
    __shared__ double rq[2][IMT_PF][64];
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;              // 0: prior, 1: tracker (wave-uniform)
    const int gid = blockIdx.x * 64 + lane;
    const bool live = gid < B * NELE_NBINS;
    const int b = live ? gid / NELE_NBINS : 0, k = live ? gid - b * NELE_NBINS : 0;
    const ImcraBin bn = imcra_bin(y2f, b, k, T);
    const unsigned char* Iin = ind + (size_t)b * T * NELE_NBINS + k;
    float* P = psd + (size_t)b * T * NELE_NBINS + k;
    const double alpha_s = 0.9, alpha_d = 0.85, Bmin = 3.2, Gamma1 = 3.0, zeta0 = 1.67, beta = 1.47;
    const double alpha_dd = 0.92, xi_min = pow(10.0, -25.0 / 20.0), p_up = 0.9;
    const double one_m_as = 1.0 - alpha_s, one_m_ad = 1.0 - alpha_d, one_m_add = 1.0 - alpha_dd;
    const int Tb = live ? (tlens ? min(tlens[b], T) : T) : 0;
    int Tmax = Tb;                                                            // the lanes of a wave may belong to two utterances
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o, 64));
    const int nbatch = (Tmax + IMT_PF - 1) / IMT_PF;
    if (role == 1 && live)
        for (int l = Tb; l < T; ++l) P[(size_t)l * NELE_NBINS] = 0.f;        // behind the end of a short row: zeros
    // ---- prior state (wave 0)
    double S = 0, tS = 0, tSmin = 0, tSmin_sw = 0, tst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int j = 0, u = 0;
    float fa[IMT_PF][3];
    unsigned char ia[IMT_PF][3];
    // ---- tracker state (wave 1)
    double ovL = 0, Lam64 = 1e-6, G = 1.0, Gamma = 1.0;
    float Lam32 = 0.f;
    float ya[IMT_PF];
    auto fetch_prior = [&](int l0) {
#pragma unroll
        for (int d = 0; d < IMT_PF; ++d) {
            const int l = l0 + d;
            const float* r = bn.y2 + (size_t)l * NELE_NBINS;
            const unsigned char* q = Iin + (size_t)l * NELE_NBINS;
            const bool in = l < Tb, in15 = in && l >= 15;
            fa[d][0] = in ? r[bn.dl] : 0.f;
            fa[d][1] = in ? r[0] : 0.f;
            fa[d][2] = in ? r[bn.dr] : 0.f;
            ia[d][0] = in15 ? q[bn.dl] : (unsigned char)0;
            ia[d][1] = in15 ? q[0] : (unsigned char)0;
            ia[d][2] = in15 ? q[bn.dr] : (unsigned char)0;
        }
    };
    auto fetch_track = [&](int l0) {
#pragma unroll
        for (int d = 0; d < IMT_PF; ++d) ya[d] = (l0 + d < Tb) ? bn.y2[(size_t)(l0 + d) * NELE_NBINS] : 0.f;
    };
    auto prior = [&](int l, const float (&f)[3], const unsigned char (&ii)[3]) -> double {
        const double Y2 = (double)f[1];
        const double Sf = (bn.w0 * (double)f[0] + bn.w1 * (double)f[1]) + bn.w2 * (double)f[2];
        if (l == 0) { S = Sf; tS = Sf; tSmin = Sf; tSmin_sw = Sf; }          // init_params (imcra.py:338-361)
        S = alpha_s * S + one_m_as * Sf;
        double r = -1.0;
        if (l >= 15) {
            const double I0 = (double)ii[0], I1 = (double)ii[1], I2 = (double)ii[2];
            const double a0 = I0 * (double)f[0], a1 = I1 * (double)f[1], a2 = I2 * (double)f[2];
            const double norm = (bn.w0 * I0 + bn.w1 * I1) + bn.w2 * I2;
            double tSf = (bn.w0 * a0 + bn.w1 * a1) + bn.w2 * a2;
            if (norm > 0.0) tSf = tSf / norm;
            tS = alpha_s * tS + one_m_as * tSf;
            tSmin = fmin(tSmin, tS);
            tSmin_sw = fmin(tSmin_sw, tS);
            const double tG = Y2 / (Bmin * tSmin);
            const double tz = S / (Bmin * tSmin);
            double q = 0.0;
            if (tG <= 1.0 && tz < zeta0) q = 1.0;
            else if (1.0 < tG && tG < Gamma1 && tz < zeta0) q = (Gamma1 - tG) / (Gamma1 - 1.0);
            if (q < 1.0) r = q / (1.0 - q);                                  // post_speech_prob's q / (1 - q) (imcra.py:22-36); -1: p = 0
            if (j + 1 == 15) {                                               // minimum tracking (imcra.py:452-481)
                const int slot = u & 7;
#pragma unroll
                for (int i = 0; i < 8; ++i) tst[i] = (i == slot) ? tSmin_sw : tst[i];
                const int n = (u < 8) ? (u + 1) : 8;
                double tm = tst[0];
#pragma unroll
                for (int i = 1; i < 8; ++i) tm = (i < n) ? fmin(tm, tst[i]) : tm;
                tSmin = tm;
                tSmin_sw = tS;
            }
            if (++j == 15) { j = 0; ++u; }
        }
        return r;
    };
    auto track = [&](int l, const float Y2f, const double r) {
        const double Y2 = (double)Y2f;
        float outv = 0.f;
        // ---- decision-directed a-priori SNR (imcra.py:543-557)
        const double xi_G = (l == 0) ? 1.0 : (G * G) * Gamma;
        double term;
        if (l == 0 || l >= 16) {
            Gamma = Y2 / Lam64;
            double xi_ML = Gamma - 1.0;
            if (xi_ML < 1e-6) xi_ML = 1e-6;
            term = one_m_add * xi_ML;
        } else {  // Lambda_D (hence Gamma, xi_ML) is a float32 array for frames 1..15
            const float Gf = Y2f / Lam32;
            float xf = Gf - 1.0f;
            if (xf < (float)1e-6) xf = (float)1e-6;
            term = (double)((float)one_m_add * xf);
            Gamma = (double)Gf;
        }
        double xi = alpha_dd * xi_G + term;
        if (xi < xi_min) xi = xi_min;
        G = xi / (1.0 + xi);
        if (l == 0) { ovL = Y2; Lam32 = Y2f; }
        if (l < 15) {
            Lam32 = (float)alpha_d * Lam32 + (float)one_m_ad * Y2f;
            outv = Lam32;
        } else {
            const double nu = Gamma * xi / (1.0 + xi);
            double p = 0.0;
            if (r >= 0.0) p = 1.0 / (1.0 + r * (1.0 + xi) * exp(-nu));
            if (p > p_up) p = p_up;
            const double tad = alpha_d + one_m_ad * p;
            ovL = tad * ovL + (1.0 - tad) * Y2;
            Lam64 = beta * ovL;
            outv = (float)Lam64;
        }
        P[(size_t)l * NELE_NBINS] = outv;
    };
    if (role == 0) fetch_prior(0); else fetch_track(0);
    for (int i = 0; i <= nbatch; ++i) {
        if (role == 0) {
            if (i < nbatch) {
                float f[IMT_PF][3];
                unsigned char ii[IMT_PF][3];
#pragma unroll
                for (int d = 0; d < IMT_PF; ++d)
#pragma unroll
                    for (int c = 0; c < 3; ++c) { f[d][c] = fa[d][c]; ii[d][c] = ia[d][c]; }
                fetch_prior((i + 1) * IMT_PF);                               // the next batch's loads fly under this batch's arithmetic
#pragma unroll
                for (int d = 0; d < IMT_PF; ++d) {
                    const int l = i * IMT_PF + d;
                    rq[i & 1][d][lane] = (l < Tb) ? prior(l, f[d], ii[d]) : -1.0;
                }
            }
        } else if (i >= 1) {
            float y[IMT_PF];
#pragma unroll
            for (int d = 0; d < IMT_PF; ++d) y[d] = ya[d];
            fetch_track(i * IMT_PF);
#pragma unroll
            for (int d = 0; d < IMT_PF; ++d) {
                const int l = (i - 1) * IMT_PF + d;
                const double r = rq[(i - 1) & 1][d][lane];
                if (l < Tb) track(l, y[d], r);
            }
        }
        __syncthreads();
    }
}

// band feature of sqrt(PSD) for every frame at once (audio_util.py:446-451): grid (ceil(T / 4), B), block 256 = 4 frames x 64 bands
__global__ __launch_bounds__(256) void band_from_psd_kernel(const float* __restrict__ psd, int T, float power, float* __restrict__ band) {
    __shared__ float tmp[4][NELE_NBINS + 3];
    __shared__ float frw[NELE_NBINS + 3], omw[NELE_NBINS + 3];
    const int b = blockIdx.y, l0 = blockIdx.x * 4;
    band_weights_init(frw, omw);
    for (int e = threadIdx.x; e < 4 * NELE_NBINS; e += 256) {
        const int f = e / NELE_NBINS, k = e - f * NELE_NBINS;
        if (l0 + f < T) {
            const float r = sqrtf(psd[((size_t)b * T + l0 + f) * NELE_NBINS + k]);
            tmp[f][k] = r * r;
        }
    }
    __syncthreads();
    const int f = threadIdx.x >> 6, i = threadIdx.x & 63;
    if (l0 + f < T) band[((size_t)b * T + l0 + f) * NELE_NBANDS + i] = pow_f32(band_energy_w(tmp[f], frw, omw, i), power);
}

// band[i] = band[i] ** power in place (the tail of imcra_band_kernel, parallel over frames)
__global__ void band_pow_kernel(float* __restrict__ band, size_t n, float power) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) band[i] = pow_f32(band[i], power);
}

// ------------------------------------------------------------------------------------------ iSTFT
// grid (T-1, B), block 256: output samples [256 j, 256 j + 256) of the trimmed signal = second half
// of frame j + first half of frame j+1; both real inverse transforms ride one complex FFT.
__device__ __forceinline__ double band_gain_sqrt(const float* __restrict__ a2, int k) {
    // interp_band_gain (audio_util.py:93-110) then np.sqrt (audio_util.py:85)
    if (k <= 1) return sqrt(1e-4);
    if (k == NELE_NBINS - 1) return sqrt(1e-2);
    int i = 0;
    while (c_gmt[i + 1] <= k) ++i;
    const int size = c_gmt[i + 1] - c_gmt[i], jj = k - c_gmt[i];
    const double frac = (double)jj / (double)size;
    const float g = (float)(1.0 - frac) * a2[i] + (float)frac * a2[i + 1];
    return sqrt((double)g);
}

// the same with the band of bin k looked up (gain_istft_wave_kernel fills the table once per workgroup instead of searching per bin and hop)
__device__ __forceinline__ double band_gain_sqrt_at(const float* __restrict__ a2, int k, int i) {
    if (k <= 1) return sqrt(1e-4);
    if (k == NELE_NBINS - 1) return sqrt(1e-2);
    const int size = c_gmt[i + 1] - c_gmt[i], jj = k - c_gmt[i];
    const double frac = (double)jj / (double)size;
    const float g = (float)(1.0 - frac) * a2[i] + (float)frac * a2[i + 1];
    return sqrt((double)g);
}

#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(256) void gain_istft_kernel(const float* __restrict__ alpha2, const float2* __restrict__ spec,
                                                         int T, float* __restrict__ wav, const int* __restrict__ tlens) {
    __shared__ Fft512Lds s;
    const int b = blockIdx.y, fa = blockIdx.x, fb = fa + 1;
    if (tlens && fb >= min(tlens[b], T)) {                  // behind the end of a short row (its signal has 256 (T_b - 1) samples): zeros
        wav[(size_t)b * (NELE_HOP * (T - 1)) + (size_t)NELE_HOP * fa + threadIdx.x] = 0.f;
        return;
    }
    const float* a2a = alpha2 ? alpha2 + ((size_t)b * T + fa) * NELE_NBANDS : nullptr;   // NULL: plain ISTFT (audio_util.py:60-65)
    const float* a2b = a2a + NELE_NBANDS;
    const float2* Xa = spec + ((size_t)b * T + fa) * NELE_NBINS;
    const float2* Xb = Xa + NELE_NBINS;
    fft512_init_twiddles(s);
    for (int k = threadIdx.x; k < NELE_NBINS; k += 256) {
        const double ga = a2a ? band_gain_sqrt(a2a, k) : 1.0, gb = a2a ? band_gain_sqrt(a2b, k) : 1.0;
        const float2 xa = Xa[k], xb = Xb[k];
        double ar = ga * (double)xa.x, ai = ga * (double)xa.y;
        double br = gb * (double)xb.x, bi = gb * (double)xb.y;
        if (k == 0 || k == NELE_NBINS - 1) { ai = 0.0; bi = 0.0; }  // c2r ignores these imaginary parts
        s.x[fft512_brev(k)] = make_double2(ar - bi, ai + br);
        if (k >= 1 && k <= 255) s.x[fft512_brev(NELE_NFFT - k)] = make_double2(ar + bi, br - ai);
    }
    __syncthreads();
    fft512_run<true>(s);
    {
        const int n = threadIdx.x;
        const double x1 = s.x[n + 256].x * (1.0 / 512.0);  // frame fa, second half
        const double x2 = s.x[n].y * (1.0 / 512.0);        // frame fb, first half
        const double wa = hann512(n + 256), wb = hann512(n);
        float y = (float)(wa * x1);                        // overlap-add into a float32 buffer
        y = (float)((double)y + wb * x2);
        float wss = (float)(wa * wa);                      // window_sumsquare, same float32 staging
        wss = (float)((double)wss + wb * wb);
        wav[(size_t)b * (NELE_HOP * (T - 1)) + (size_t)NELE_HOP * fa + n] = y / wss;
    }
}
#endif  // NELE_AB

// ---- the same resynthesis, one WAVE per output hop with the inverse transform in registers (fft512_wave<true>): the lane that holds
// X[64 m + lane] holds both halves it needs (frame fa's second half in v[m + 4].x, frame fb's first half in v[m].y), so nothing is
// exchanged after the transform.  Bit-identical to gain_istft_kernel.  grid (ceil((T - 1) / (4 STW_NP)), B), block 256.
// Round 5: a wave walks ISW_NH CONSECUTIVE hops, so the second frame of a hop (its spectrum bins and its 257 interpolated gains, each a
// float64 square root) is the first frame of the next one and stays in the lane's registers; the interpolation weights
// (float)(1 - j / size), (float)(j / size) of audio_util.py:100-106 come from a per-workgroup LDS table instead of a float64 division per
// bin and frame.  Same products, same order: bit-identical.  grid (ceil((T - 1) / (4 ISW_NH)), B), block 256.
#define ISW_NH 8
__global__ __launch_bounds__(256) void gain_istft_wave_kernel(const float* __restrict__ alpha2, const float2* __restrict__ spec,
                                                              int T, float* __restrict__ wav, const int* __restrict__ tlens) {
    __shared__ double2 tw[256];
    __shared__ double hw[NELE_NFFT];
    __shared__ __attribute__((aligned(16))) double2 xs[4][FFTW_SLOTS];
    __shared__ unsigned char bidx[NELE_NBINS + 3];        // band of every bin (the search of band_gain_sqrt, once per workgroup)
    __shared__ float wlo[NELE_NBINS + 3], whi[NELE_NBINS + 3];
    const int b = blockIdx.y, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    for (int k = tid; k < NELE_NBINS; k += 256) {
        int i = 0;
        while (i < NELE_NBANDS - 2 && c_gmt[i + 1] <= k) ++i;
        bidx[k] = (unsigned char)i;
        const int size = c_gmt[i + 1] - c_gmt[i], jj = k - c_gmt[i];
        const double frac = (double)jj / (double)size;
        wlo[k] = (float)(1.0 - frac);
        whi[k] = (float)frac;
    }
    {
        double sn, cs;
        sincospi((double)tid / 256.0, &sn, &cs);
        tw[tid] = make_double2(cs, -sn);
        hw[tid] = hann512(tid);
        hw[tid + 256] = hann512(tid + 256);
    }
    __syncthreads();
    double2* xw = xs[wv];
    float* out = wav + (size_t)b * (NELE_HOP * (T - 1));
    const int Tb = tlens ? min(tlens[b], T) : T;
    // gain of bin k of the frame whose band gains are a2 (interp_band_gain + np.sqrt, audio_util.py:93-110, 85)
    auto gain = [&](const float* __restrict__ a2, int k) -> double {
        if (!a2) return 1.0;                               // plain ISTFT (audio_util.py:60-65)
        if (k <= 1) return sqrt(1e-4);
        if (k == NELE_NBINS - 1) return sqrt(1e-2);
        const int i = bidx[k];
        const float g = wlo[k] * a2[i] + whi[k] * a2[i + 1];
        return sqrt((double)g);
    };
    double gprev[5];
    float2 xprev[5];
    bool have = false;                                    // gprev / xprev hold frame fa of the coming hop
    const int fa0 = (blockIdx.x * 4 + wv) * ISW_NH;
    for (int it = 0; it < ISW_NH; ++it) {
        const int fa = fa0 + it, fb = fa + 1;
        if (fa >= T - 1) break;
        if (fb >= Tb) {                                     // behind the end of a short row (its signal has 256 (T_b - 1) samples): zeros
#pragma unroll
            for (int m = 0; m < 4; ++m) out[(size_t)NELE_HOP * fa + 64 * m + lane] = 0.f;
            have = false;
            continue;
        }
        const float* a2a = alpha2 ? alpha2 + ((size_t)b * T + fa) * NELE_NBANDS : nullptr;
        const float* a2b = alpha2 ? a2a + NELE_NBANDS : nullptr;
        const float2* Xa = spec + ((size_t)b * T + fa) * NELE_NBINS;
        const float2* Xb = Xa + NELE_NBINS;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int k = lane + 64 * q;
            if (k < NELE_NBINS) {
                const double ga = have ? gprev[q] : gain(a2a, k), gb = gain(a2b, k);
                const float2 xa = have ? xprev[q] : Xa[k], xb = Xb[k];
                gprev[q] = gb; xprev[q] = xb;
                double ar = ga * (double)xa.x, ai = ga * (double)xa.y;
                double br = gb * (double)xb.x, bi = gb * (double)xb.y;
                if (k == 0 || k == NELE_NBINS - 1) { ai = 0.0; bi = 0.0; }  // c2r ignores these imaginary parts
                xw[fftw_slot(fft512_brev(k))] = make_double2(ar - bi, ai + br);
                if (k >= 1 && k <= 255) xw[fftw_slot(fft512_brev(NELE_NFFT - k))] = make_double2(ar + bi, br - ai);
            }
        }
        have = true;
        fftw_wave_sync();
        double2 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = xw[9 * lane + r];
        fft512_wave<true>(v, xw, tw, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int n = 64 * m + lane;
            const double x1 = v[m + 4].x * (1.0 / 512.0);      // frame fa, second half
            const double x2 = v[m].y * (1.0 / 512.0);          // frame fb, first half
            const double wa = hw[n + 256], wb = hw[n];
            float y = (float)(wa * x1);                        // overlap-add into a float32 buffer
            y = (float)((double)y + wb * x2);
            float wss = (float)(wa * wa);                      // window_sumsquare, same float32 staging
            wss = (float)((double)wss + wb * wb);
            out[(size_t)NELE_HOP * fa + n] = y / wss;
        }
        fftw_wave_sync();                                   // xw is restaged by the next hop
    }
}

// enh / rms(enh) * target (inference.py:109) and the optional PCM_16 round trip (libsndfile float->short with 0x7FFF scaling + lrintf,
// read back / 32768: PARITY UNPINNED), in two passes that both fill the chip: grid (chunks of WP_CHUNK samples, B).
//   pass 1  float64 sum of the float32 squares of one chunk -> part[b][chunk]                  (fixed order inside the block)
//   pass 2  every block adds its utterance's partials in chunk order (so all blocks of an utterance, and any batch the utterance is
//           part of, see the same rms), then scales / quantises its own chunk.
// (Until round 5: one 256-thread block per utterance - 357 us for 128 utterances of 8 s, 1/16 of the chip, in the inference path.)
#define WP_CHUNK 4096
__global__ __launch_bounds__(256) void wav_sumsq_kernel(const float* __restrict__ wav, int N, const int* __restrict__ tlens, double* __restrict__ part) {
    __shared__ double red[8];
    const int b = blockIdx.y, c = blockIdx.x;
    const float* x = wav + (size_t)b * N;
    const int n = tlens ? min(N, NELE_HOP * (tlens[b] - 1)) : N;
    const int i0 = c * WP_CHUNK, i1 = min(i0 + WP_CHUNK, n);
    double acc = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) acc += (double)(x[i] * x[i]);
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) part[(size_t)b * gridDim.x + c] = acc;
}
__global__ __launch_bounds__(256) void wav_scale_kernel(float* __restrict__ wav, int N, float target_rms, int pcm16, const int* __restrict__ tlens,
                                                        const double* __restrict__ part) {
    const int b = blockIdx.y, c = blockIdx.x;
    float* x = wav + (size_t)b * N;
    const int n = tlens ? min(N, NELE_HOP * (tlens[b] - 1)) : N;      // samples of this utterance; the zeros behind them stay zeros
    const int i0 = c * WP_CHUNK, i1 = min(i0 + WP_CHUNK, n);
    if (i0 >= i1) return;
    double acc = 0.0;
    for (int q = 0; q < (int)gridDim.x; ++q) acc += part[(size_t)b * gridDim.x + q];
    const float r = sqrtf((float)(acc / (double)n));
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        float v = x[i] / r * target_rms;
        if (pcm16) {
            float q = rintf(v * 32767.f);
            q = fminf(fmaxf(q, -32768.f), 32767.f);
            v = q / 32768.f;
        }
        x[i] = v;
    }
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int nele_stft_band_var(const float* wav, const int* lengths, int B, int L, float power, void* spec, float* band, void* stream) {
    NELE_CHECK_ARG(wav && B > 0, "nele_stft_band: null wav or B <= 0");
    NELE_CHECK_ARG(L > NELE_HOP, "nele_stft_band: L=%d must exceed 256 (reflect padding)", L);
    NELE_CHECK_ARG(spec || band, "nele_stft_band: no output requested");
    const int T = 1 + L / NELE_HOP;
    const int wave_on = NELE_SWITCH_INT("NELE_STFT_WAVE", 1);                                // NELE_STFT_WAVE=0: the workgroup-per-frame-pair kernels (A/B diagnostic)
    if (wave_on) {
        dim3 grid(((T + 1) / 2 + 4 * STW_NP - 1) / (4 * STW_NP), B);
        hipLaunchKernelGGL(stft_band_wave_kernel, grid, dim3(256), 0, as_stream(stream), wav, L, T, power, (float2*)spec, band, lengths, (float*)nullptr);
    } else {
        NELE_AB_ONLY(dim3 grid((T + 1) / 2, B);
                     hipLaunchKernelGGL(stft_band_kernel, grid, dim3(256), 0, as_stream(stream), wav, L, T, power, (float2*)spec, band, lengths);)
    }
    NELE_CHECK_LAUNCH("nele_stft_band");
    return NELE_OK;
}
// The noise file's side of the features (audio_util.py:439-456): IMCRA consumes |STFT|^2 only, so the spectrum need not exist in memory -
// pw [B][T][257] float32 = np.abs(STFT) ** 2 as numpy computes it in float32 (frames behind a short row's end are NOT written; nothing
// reads them), the input of nele_imcra_band_pw.  spec / band as in nele_stft_band_var (either may be NULL).
extern "C" int nele_stft_pow_var(const float* wav, const int* lengths, int B, int L, float power, void* spec, float* band, float* pw, void* stream) {
    NELE_CHECK_ARG(wav && B > 0 && pw, "nele_stft_pow: null wav / pw or B <= 0");
    NELE_CHECK_ARG(L > NELE_HOP, "nele_stft_pow: L=%d must exceed 256 (reflect padding)", L);
    const int T = 1 + L / NELE_HOP;
    dim3 grid(((T + 1) / 2 + 4 * STW_NP - 1) / (4 * STW_NP), B);
    hipLaunchKernelGGL(stft_band_wave_kernel, grid, dim3(256), 0, as_stream(stream), wav, L, T, power, (float2*)spec, band, lengths, pw);
    NELE_CHECK_LAUNCH("nele_stft_pow");
    return NELE_OK;
}
extern "C" int nele_stft_band(const float* wav, int B, int L, float power, void* spec, float* band, void* stream) {
    return nele_stft_band_var(wav, nullptr, B, L, power, spec, band, stream);
}

extern "C" int nele_imcra_band_var(const void* spec, const int* frames, int B, int T, float power, float* psd, float* band, void* stream) {
    NELE_CHECK_ARG(spec && B > 0 && T > 0, "nele_imcra_band: bad arguments");
    NELE_CHECK_ARG(psd || band, "nele_imcra_band: no output requested");
    if (psd && band) {      // noise PSD kept: the band feature is computed from it afterwards, off the serial loop
        hipLaunchKernelGGL(imcra_band_kernel<false>, dim3(B), dim3(IMCRA_THREADS), 0, as_stream(stream), (const float2*)spec, T, power, psd, band, frames);
        hipLaunchKernelGGL(band_from_psd_kernel, dim3((T + 3) / 4, B), dim3(256), 0, as_stream(stream), psd, T, power, band);
        NELE_CHECK_LAUNCH("nele_imcra_band");
        return NELE_OK;
    }
    hipLaunchKernelGGL(imcra_band_kernel<true>, dim3(B), dim3(IMCRA_THREADS), 0, as_stream(stream), (const float2*)spec, T, power,
                       psd, band, frames);
    if (band) {
        const size_t nb = (size_t)B * T * NELE_NBANDS;
        hipLaunchKernelGGL(band_pow_kernel, dim3((unsigned)((nb + 255) / 256 < 2048 ? (nb + 255) / 256 : 2048)), dim3(256), 0, as_stream(stream), band, nb, power);
    }
    NELE_CHECK_LAUNCH("nele_imcra_band");
    return NELE_OK;
}
extern "C" long long nele_imcra_workspace_bytes(int B, int T) {
    if (B <= 0 || T <= 0) return -1;
    return ((long long)B * T * NELE_NBINS * 5 + 255) / 256 * 256;             // |Y|^2 (float32) + the speech indicator (one byte)
}
// The same result as nele_imcra_band_var(spec, frames, .., psd, band) - bit for bit - without a workgroup-wide recursion: imcra_pow_kernel,
// imcra_ind_kernel, imcra_track_kernel (see there), |Y|^2 and the indicator in a caller-provided workspace of nele_imcra_workspace_bytes(B, T).
extern "C" int nele_imcra_band_ws(const void* spec, const int* frames, int B, int T, float power, float* psd, float* band, void* ws,
                                  long long ws_bytes, void* stream) {
    NELE_CHECK_ARG(spec && psd && B > 0 && T > 0, "nele_imcra_band_ws: bad arguments (the PSD buffer is required)");
    NELE_CHECK_ARG(ws && ws_bytes >= nele_imcra_workspace_bytes(B, T) && ((size_t)ws % 16) == 0, "nele_imcra_band_ws: workspace of %lld bytes needed",
                   nele_imcra_workspace_bytes(B, T));
    const size_t n = (size_t)B * T * NELE_NBINS;
    float* y2f = (float*)ws;
    unsigned char* ind = (unsigned char*)ws + 4 * n;
    const unsigned nb = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(imcra_pow_kernel, dim3(nb), dim3(256), 0, as_stream(stream), (const float2*)spec, n, y2f);
    const dim3 grid((B * NELE_NBINS + 255) / 256);
    hipLaunchKernelGGL(imcra_ind_kernel, grid, dim3(256), 0, as_stream(stream), (const float*)y2f, B, T, ind, frames);
    hipLaunchKernelGGL(imcra_track_kernel, dim3((B * NELE_NBINS + 63) / 64), dim3(128), 0, as_stream(stream), (const float*)y2f, (const unsigned char*)ind, B, T, psd, frames);
    if (band) hipLaunchKernelGGL(band_from_psd_kernel, dim3((T + 3) / 4, B), dim3(256), 0, as_stream(stream), psd, T, power, band);
    NELE_CHECK_LAUNCH("nele_imcra_band_ws");
    return NELE_OK;
}
// ... and from |Y|^2 itself (nele_stft_pow_var's pw): the first kernel and the spectrum's round trip through memory drop out.
// workspace: nele_imcra_workspace_bytes(B, T) as above (only its indicator part is used).
extern "C" int nele_imcra_band_pw(const float* pw, const int* frames, int B, int T, float power, float* psd, float* band, void* ws,
                                  long long ws_bytes, void* stream) {
    NELE_CHECK_ARG(pw && psd && B > 0 && T > 0, "nele_imcra_band_pw: bad arguments (the PSD buffer is required)");
    NELE_CHECK_ARG(ws && ws_bytes >= nele_imcra_workspace_bytes(B, T), "nele_imcra_band_pw: workspace of %lld bytes needed", nele_imcra_workspace_bytes(B, T));
    unsigned char* ind = (unsigned char*)ws;
    const dim3 grid((B * NELE_NBINS + 255) / 256);
    hipLaunchKernelGGL(imcra_ind_kernel, grid, dim3(256), 0, as_stream(stream), pw, B, T, ind, frames);
    hipLaunchKernelGGL(imcra_track_kernel, dim3((B * NELE_NBINS + 63) / 64), dim3(128), 0, as_stream(stream), pw, (const unsigned char*)ind, B, T, psd, frames);
    if (band) hipLaunchKernelGGL(band_from_psd_kernel, dim3((T + 3) / 4, B), dim3(256), 0, as_stream(stream), psd, T, power, band);
    NELE_CHECK_LAUNCH("nele_imcra_band_pw");
    return NELE_OK;
}
extern "C" int nele_imcra_band(const void* spec, int B, int T, float power, float* psd, float* band, void* stream) {
    return nele_imcra_band_var(spec, nullptr, B, T, power, psd, band, stream);
}

// compute_band_E (audio_util.py:30-50) on a magnitude spectrogram: X [N][257] f32 -> OUT [N][64] f32 (no power law).
// One wave per frame: |X|^2 in float32 staged in LDS, then band_energy() in the reference's accumulation order.
__global__ __launch_bounds__(256) void band_energy_kernel(const float* __restrict__ X, int N, float* __restrict__ out) {
    __shared__ float tmp[4][NELE_NBINS + 3];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, f = blockIdx.x * 4 + w;
    if (f < N)
        for (int k = lane; k < NELE_NBINS; k += 64) {
            const float m = X[(size_t)f * NELE_NBINS + k];
            tmp[w][k] = m * m;
        }
    __syncthreads();
    if (f < N) out[(size_t)f * NELE_NBANDS + lane] = band_energy(tmp[w], lane);
}

// interp_band_gain (audio_util.py:93-110): bandE [N][64] f32 -> g [N][257] f64 (the reference's np.ones(257) is float64; the
// interpolation itself runs on the float32 band values, as numpy does with float32 scalars and python-float weights under NEP 50).
__global__ void interp_gain_kernel(const float* __restrict__ bandE, int N, double* __restrict__ g) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)N * NELE_NBINS) return;
    const int f = (int)(i / NELE_NBINS), k = (int)(i % NELE_NBINS);
    double v;
    if (k <= 1) v = 1e-4;
    else if (k == NELE_NBINS - 1) v = 1e-2;
    else {
        int b = 0;
        while (c_gmt[b + 1] <= k) ++b;
        const int size = c_gmt[b + 1] - c_gmt[b], jj = k - c_gmt[b];
        const double frac = (double)jj / (double)size;
        v = (double)((float)(1.0 - frac) * bandE[(size_t)f * NELE_NBANDS + b] + (float)frac * bandE[(size_t)f * NELE_NBANDS + b + 1]);
    }
    g[i] = v;
}

extern "C" int nele_compute_band_E(const float* mag, int N, float* band, void* stream) {
    NELE_CHECK_ARG(mag && band && N > 0, "nele_compute_band_E: bad arguments");
    hipLaunchKernelGGL(band_energy_kernel, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), mag, N, band);
    NELE_CHECK_LAUNCH("nele_compute_band_E");
    return NELE_OK;
}

extern "C" int nele_interp_band_gain(const float* bandE, int N, double* g, void* stream) {
    NELE_CHECK_ARG(bandE && g && N > 0, "nele_interp_band_gain: bad arguments");
    const size_t n = (size_t)N * NELE_NBINS;
    hipLaunchKernelGGL(interp_gain_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), bandE, N, g);
    NELE_CHECK_LAUNCH("nele_interp_band_gain");
    return NELE_OK;
}

extern "C" int nele_gain_istft_var(const float* alpha2, const void* spec, const int* frames, int B, int T, float* wav, void* stream) {
    NELE_CHECK_ARG(spec && wav && B > 0, "nele_gain_istft: bad arguments");
    NELE_CHECK_ARG(T >= 2, "nele_gain_istft: T=%d < 2", T);
    const int wave_on = NELE_SWITCH_INT("NELE_STFT_WAVE", 1);                                // NELE_STFT_WAVE=0: the workgroup-per-hop kernel (A/B diagnostic)
    if (wave_on)
        hipLaunchKernelGGL(gain_istft_wave_kernel, dim3((T - 1 + 4 * ISW_NH - 1) / (4 * ISW_NH), B), dim3(256), 0, as_stream(stream), alpha2,
                           (const float2*)spec, T, wav, frames);
    else {
        NELE_AB_ONLY(hipLaunchKernelGGL(gain_istft_kernel, dim3(T - 1, B), dim3(256), 0, as_stream(stream), alpha2, (const float2*)spec, T, wav, frames);)
    }
    NELE_CHECK_LAUNCH("nele_gain_istft");
    return NELE_OK;
}
extern "C" int nele_gain_istft(const float* alpha2, const void* spec, int B, int T, float* wav, void* stream) {
    return nele_gain_istft_var(alpha2, spec, nullptr, B, T, wav, stream);
}

// PCM_16 round trip alone (no level normalisation: nothing per utterance to reduce): plain element-wise pass over the whole batch
__global__ void wav_quant_kernel(float* __restrict__ wav, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float q = rintf(wav[i] * 32767.f);
        q = fminf(fmaxf(q, -32768.f), 32767.f);
        wav[i] = q / 32768.f;
    }
}

extern "C" long long nele_wav_post_workspace_doubles(int B, int N) { return (long long)B * ((N + WP_CHUNK - 1) / WP_CHUNK); }
extern "C" int nele_wav_post_var(float* wav, const int* frames, int B, int N, float target_rms, int pcm16, double* workspace, void* stream);
extern "C" int nele_wav_post(float* wav, int B, int N, float target_rms, int pcm16, double* workspace, void* stream) {
    return nele_wav_post_var(wav, nullptr, B, N, target_rms, pcm16, workspace, stream);
}
extern "C" int nele_wav_post_var(float* wav, const int* frames, int B, int N, float target_rms, int pcm16, double* workspace, void* stream) {
    NELE_CHECK_ARG(wav && B > 0 && N > 0, "nele_wav_post: bad arguments");
    if (target_rms <= 0.f && !pcm16) return NELE_OK;
    if (target_rms <= 0.f) {
        const size_t n = (size_t)B * N;
        hipLaunchKernelGGL(wav_quant_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, as_stream(stream), wav, n);
        NELE_CHECK_LAUNCH("nele_wav_post");
        return NELE_OK;
    }
    NELE_CHECK_ARG(workspace, "nele_wav_post: the rms normalisation needs a workspace of nele_wav_post_workspace_doubles(B, N) doubles");
    const dim3 grid((N + WP_CHUNK - 1) / WP_CHUNK, B);
    hipLaunchKernelGGL(wav_sumsq_kernel, grid, dim3(256), 0, as_stream(stream), wav, N, frames, workspace);
    hipLaunchKernelGGL(wav_scale_kernel, grid, dim3(256), 0, as_stream(stream), wav, N, target_rms, pcm16, frames, workspace);
    NELE_CHECK_LAUNCH("nele_wav_post");
    return NELE_OK;
}
