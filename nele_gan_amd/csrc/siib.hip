// Batched SIIB^Gauss (reference intel.py:57-106: wrapper VAD + replication rule are the reference's own
// code; the SIIB core is pysiib's, restated in oracle/siib.py -- PARITY UNPINNED vs pysiib itself).
//
// Per utterance (x clean, y degraded, [L] float32 @16 kHz), float64 arithmetic:
//   s1  VAD on x (intel.py:37-50), active duration -> replication factor M (intel.py:93-97)
//   s2  VAD on the M-times tiled x (frames index the signal modulo L: nothing is materialised),
//       k-th order statistic with duplicates, ordered compaction of the active frames
//   s3  per active frame: Hann(400) window, 400-point DFT of x and y (201 bins), squared gammatone
//       weights (28 bands), log
//   s4  per band: minimum, forward temporal masking (serial over frames), mean removal
//   s5  stack K = 15 frames (420 rows), remove the row means
//   s6  Cxx = Xs Xs^T / (n-1)                      (tiled float64 GEMM)
//   s7  eigenvectors of Cxx                        (csrc/eigh.hip: tridiagonalisation + bisection + inverse iteration)
//   s8  Xp = U^T Xs, Yp = U^T Ys fused with the per-component sums of Xp^2, Yp^2, Xp Yp
//   s9  rho, I = -1/2 log2(1 - rho_p^2 rho^2), SIIB = R/K sum I, logistic map
#include "common.h"
#include <cstdlib>

#define SB_WLEN 400
#define SB_SHIFT 200
#define SB_NBIN 201
#define SB_J 28
#define SB_K 15
#define SB_D (SB_J * SB_K)   // 420
#define SB_TF 16
#define SB_EPS 2.220446049250313e-16
#define SB_MMAX 40

struct SiibWs {
    double* g2;      // [28][201] squared gammatone magnitude responses
    double* tab;     // [3][400] cos(2 pi j / 400), sin(2 pi j / 400), hann400(j)
    double* g2t;     // [201][28] g2 transposed (bands contiguous)
    double* rowstat; // [B][2][28][2] per (signal, band) row of XL: minimum before masking, mean after masking
    double* xdb;     // [B][NT]   frame power (dB) of the tiled clean signal
    int* list;       // [B][NA]   active frame indices (tiled frame numbering)
    int* info;       // [B][4]    {M, n_tiled_frames, n_active, status}
    int* nprim;      // [B]       active frames inside the first frame period of the tiled signal (their spectra are computed, the rest copied)
    double* XL;      // [B][2][28][NA] log band energies of the active frames (x then y)
    double* Xs;      // [B][2][420][NA] stacked, mean-removed (zero padded to NA columns)
    double* C;       // [B][420][420] covariance (destroyed by the eigensolver)
    double* U;       // [B][420][420] eigenvectors (row j = eigenvector j)
    char* eigws;     // eigensolver workspace
    double* lam;     // [B][420]
    double* part;    // [B][420][NTL][3]
    double* px;      // [B][7][NTL][16][256] projections of the clean signal in accumulator order (split mode: phase 3 -> phase 4)
    double* lagp;    // [B][3][SL_NSEG][29][784] lag-product partials: pair 0 = (x, x), 1 = (y, y), 2 = (x, y)                  (lag path)
    double* mu;      // [B][2][15][28] window means of the stacked rows                                                         (lag path)
    double* S2;      // [B][2][420][420] unscaled second-moment matrices of the stacked frames: Syy, sym(Sxy)                   (lag path)
    double* qpart;   // [B][420][14] partial quadratic forms u_k^T S u_k per column tile: 7 tiles of Syy, 7 of sym(Sxy)          (lag path)
    int nseg;        // workgroup groups the SL_NSEG frame segments of the lag products are dealt to (1, 2 or 4: more workgroups at small batches)
    int NT, NA, NTL;
    int Bn;          // utterances in this call
    const int* lens; // [B] samples per utterance inside the padded [B][L] buffers, or NULL (every row has L samples)
};
// per-utterance length (the reference scores files of any length one at a time: intel.py:58-60, audio_util.py:134-141)
__device__ __forceinline__ int sb_len(const SiibWs& ws, int b, int L) { return ws.lens ? min(ws.lens[b], L) : L; }

__device__ __forceinline__ double hann400(int n) { return 0.5 - 0.5 * cospi((double)n / 200.0); }

// ---------------------------------------------------------------- gammatone matrix (oracle/siib.py gammatone_matrix)
__global__ void siib_g2_kernel(double* __restrict__ g2, double* __restrict__ tab, double* __restrict__ g2t) {
    __shared__ double red[8];
    const int j = blockIdx.x, q = threadIdx.x;  // 256 threads >= 201
    if (j == 0)
        for (int i = q; i < SB_WLEN; i += 256) {
            double s_, c_;
            sincospi((double)i / 200.0, &s_, &c_);
            tab[i] = c_; tab[SB_WLEN + i] = s_; tab[2 * SB_WLEN + i] = hann400(i);
        }
    const double e0 = 21.4 * log10(4.37 * (100.0 / 1000.0) + 1.0), e1 = 21.4 * log10(4.37 * (6500.0 / 1000.0) + 1.0);
    const double cf_erb = e0 + (e1 - e0) * (double)j / (double)(SB_J - 1);
    const double cf = (pow(10.0, cf_erb / 21.4) - 1.0) / 4.37 * 1000.0;
    const double a = 36.0 / (M_PI * 720.0 * 0.015625);  // (3!)^2 / (pi * 6! * 2^-6)
    const double bw = a * 24.7 * (4.37 * cf / 1000.0 + 1.0);
    double t = 0.0;
    if (q < SB_NBIN) {
        const double f = 16000.0 * (double)q / 400.0;
        const double d = bw * bw + (f - cf) * (f - cf);
        t = 1.0 / (d * d);  // order/2 = 2
    }
    const double mx = block_max(t, red);
    if (q < SB_NBIN) {
        const double g = t / mx;
        g2[j * SB_NBIN + q] = g * g;
        g2t[q * SB_J + j] = g * g;
    }
}

// frame f of the tiled signal: samples x[(200 f + j) mod L], j < 400 (zero beyond M*L: intel.py:28-31 pads)
// win = the hann400 table siib_g2_kernel has written (ws.tab + 800: the same values, a load instead of a float64 cospi per sample)
__device__ __forceinline__ double frame_db(const float* __restrict__ x, int L, long long total, int f, int lane, const double* __restrict__ win) {
    double s = 0.0;
    const long long p0 = (long long)SB_SHIFT * f;
    int q = (int)(p0 % L);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int j = lane + 64 * i;
        if (j < SB_WLEN) {
            int qq = q + j;
            while (qq >= L) qq -= L;
            const double v = (p0 + j < total) ? (double)x[qq] * win[j] : 0.0;
            s += v * v;
        }
    }
    s = wave_sum(s);
    return 10.0 * log10(s / (double)SB_WLEN + SB_EPS);
}

__device__ __forceinline__ int nframes_of(long long total) {
    if (total < SB_WLEN + 1) total = SB_WLEN + 1;
    return (int)((total - SB_WLEN + SB_SHIFT - 1) / SB_SHIFT);
}

__device__ __forceinline__ int round_half_even_pos(double v) { return (int)rint(v); }

// Threshold of intel.get_vad: the value at sorted position ind = round(n*0.999)-1 (ascending).
// r = n-1-ind values are strictly above it in sorted order; walk down from the maximum counting duplicates.
__device__ double kth_largest(const double* __restrict__ v, int n, int r, double* red, int* ired) {
    double cur = 1e300;
    int remaining = r;
    for (int it = 0; it <= r; ++it) {
        double m = -1e300;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const double a = v[i];
            if (a < cur) m = fmax(m, a);
        }
        m = block_max(m, red);
        int cnt = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) cnt += (v[i] == m) ? 1 : 0;
        // integer block sum through the double scratch (exact for these counts)
        const double c = block_sum((double)cnt, red);
        const int ci = (int)c;
        if (remaining < ci) return m;
        remaining -= ci;
        cur = m;
    }
    return cur;
}

// s1a / s2a: frame power (dB) of the base (tiled = 0) or M-times tiled (tiled = 1) clean signal.
// grid (ceil(frames / 4), B), block 256 (one wave per frame).  The tiled pass only computes what the base pass has not: frames
// f < n1 of the tiled signal ARE the base frames (no wrap-around, same samples, same arithmetic), and frames f >= Pf repeat frame
// f - Pf (siib_compact_kernel fills them in), so it covers n1 <= f < min(n_tiled, Pf) - two frames when L is a multiple of 200.
__global__ __launch_bounds__(256) void siib_db_kernel(const float* __restrict__ x, int L, SiibWs ws, int tiled, int Pf) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int Lb = sb_len(ws, b, L);
    long long total = Lb;
    int nf = nframes_of(Lb);
    if (tiled) {
        f += nf;                                           // first frame the base pass did not produce
        total = (long long)ws.info[4 * b] * Lb;
        nf = min(ws.info[4 * b + 1], Pf);
    }
    if (f >= nf || f >= ws.NT) return;
    const double e = frame_db(x + (size_t)b * L, Lb, total, f, lane, ws.tab + 2 * SB_WLEN);
    if (lane == 0) ws.xdb[(size_t)b * ws.NT + f] = e;
}

// s1b: VAD on the base signal -> active duration -> replication factor M (intel.py:84-97). grid B, block 256
__global__ __launch_bounds__(256) void siib_m_kernel(int L, SiibWs ws) {
    __shared__ double red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* xdb = ws.xdb + (size_t)b * ws.NT;
    L = sb_len(ws, b, L);
    const int n1 = nframes_of(L);
    const int ind = round_half_even_pos((double)n1 * 0.999) - 1;
    const double thr = kth_largest(xdb, n1, n1 - 1 - ind, red, nullptr) - 40.0;
    int cnt = 0;
    for (int f = tid; f < n1; f += 256) cnt += (xdb[f] > thr) ? 1 : 0;
    const int nact1 = (int)block_sum((double)cnt, red);
    int M = 1;
    const double dur = (double)nact1 / 80.0;
    if (dur < 20.0) M = (int)floor(25.0 / dur);
    int status = 0;
    if (M > SB_MMAX) { M = SB_MMAX; status = 1; }
    if (tid == 0) {
        int* info = ws.info + 4 * b;
        info[0] = M; info[1] = nframes_of((long long)M * L); info[2] = 0; info[3] = status;
    }
}

// s2b: VAD on the tiled signal + ordered compaction of the active frames. grid B, block 256
__global__ __launch_bounds__(256) void siib_compact_kernel(SiibWs ws, int Pf) {
    __shared__ double red[8];
    __shared__ int scan[256];
    __shared__ int base;
    const int b = blockIdx.x, tid = threadIdx.x;
    double* xdb = ws.xdb + (size_t)b * ws.NT;
    int* info = ws.info + 4 * b;
    int n2 = info[1];
    int status = info[3];
    if (n2 > ws.NT) { n2 = ws.NT; status |= 2; }
    if (Pf < n2)
        for (int f = Pf + tid; f < n2; f += 256) xdb[f] = xdb[f % Pf];   // frame f repeats frame f - Pf (sources are all < Pf: no hazard)
    __syncthreads();
    const int ind = round_half_even_pos((double)n2 * 0.999) - 1;
    const double thr = kth_largest(xdb, n2, n2 - 1 - ind, red, nullptr) - 40.0;
    if (tid == 0) base = 0;
    __syncthreads();
    int cprim = 0;
    for (int f0 = 0; f0 < n2; f0 += 256) {
        const int f = f0 + tid;
        const int k = (f < n2 && xdb[f] > thr) ? 1 : 0;
        cprim += (k && f < Pf) ? 1 : 0;
        scan[tid] = k;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int v = (tid >= o) ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const int pos = base + scan[tid] - 1;
        if (k && pos < ws.NA) ws.list[(size_t)b * ws.NA + pos] = f;
        __syncthreads();
        if (tid == 255) base += scan[255];
        __syncthreads();
    }
    const int nprim = (int)block_sum((double)cprim, red);
    if (tid == 0) {
        int na = base;
        if (na > ws.NA) { na = ws.NA; status |= 4; }
        if (na < SB_K + 1) status |= 8;  // not enough active frames
        info[2] = na; info[3] = status;
        ws.nprim[b] = min(nprim, na);
    }
}

// orders the LDS accesses of ONE wave (its DS instructions execute in issue order; the compiler only has to keep them in place)
__device__ __forceinline__ void sb_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// s3: 400-point spectra of the active frames -> 28 log band energies.  grid (ceil(NA / 6), B), block 128: six frames per workgroup.
// 400 = 20 x 20 (n = 20 n1 + n2, k = k1 + 20 k2).  Both 20-point stages run from REGISTERS with compile-time twiddles: a thread
// loads its 20 inputs once and produces all its outputs (stage 1: thread = (frame, n2), real input, k1 = 0..10 + conjugate
// symmetry; stage 2: thread = (frame, k1), k2 = 0..10 since only bins <= 200 are used).  LDS only carries the windowed frame, the
// 20 x 20 intermediate and |X|^2 - about 6x less LDS traffic than a thread-per-output formulation, which was LDS-bound.
#define SP_F 6
struct Tw20 { double c[20], s[20]; };
__device__ constexpr Tw20 make_tw20() {
    // cos / sin (2 pi t / 20), exact symmetries written out so that the table is a compile-time constant
    Tw20 t{};
    const double c1 = 0.95105651629515357212, c2 = 0.80901699437494742410, c3 = 0.58778525229247312917, c4 = 0.30901699437494742410;
    const double cc[20] = {1.0, c1, c2, c3, c4, 0.0, -c4, -c3, -c2, -c1, -1.0, -c1, -c2, -c3, -c4, 0.0, c4, c3, c2, c1};
    const double ss[20] = {0.0, c4, c3, c2, c1, 1.0, c1, c2, c3, c4, 0.0, -c4, -c3, -c2, -c1, -1.0, -c1, -c2, -c3, -c4};
    for (int i = 0; i < 20; ++i) { t.c[i] = cc[i]; t.s[i] = ss[i]; }
    return t;
}
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(128) void siib_spec_kernel(const float* __restrict__ x, const float* __restrict__ y, int L, SiibWs ws, int sig0,
                                                        int sig1) {   // signals sig0..sig1 (0 = clean x, 1 = degraded y)
    constexpr Tw20 tw = make_tw20();
    __shared__ double sx[SP_F][SB_WLEN];              // windowed frames; reused as |X|^2 [SP_F][201]
    __shared__ double2 Aq[SP_F][SB_WLEN];             // stage-1 output [n2][k1]
    const int b = blockIdx.y, k0 = blockIdx.x * SP_F, tid = threadIdx.x;
    const int* info = ws.info + 4 * b;
    const int na = ws.nprim[b];                        // frames of the first period only: siib_spread_kernel copies the repeats
    if (k0 >= na) return;
    const int Lrow = L;                                 // row stride of the padded buffers
    L = sb_len(ws, b, L);
    const long long total = (long long)info[0] * L;
    const int fs = tid / 20, lane20 = tid - fs * 20;   // frame slot, n2 (stage 1) / k1 (stage 2)
    __shared__ double tcs[SB_WLEN], tsn[SB_WLEN];     // W400 twiddles (LDS copy of ws.tab: 31 dependent-latency reads per thread otherwise)
    for (int e = tid; e < SB_WLEN; e += 128) { tcs[e] = ws.tab[e]; tsn[e] = ws.tab[SB_WLEN + e]; }
    __shared__ int fq0[SP_F], fvalid[SP_F];           // first sample (mod L) of each frame or -1; samples before the tiled signal ends
    if (tid < SP_F) {
        int q0 = -1, nv = 0;
        if (k0 + tid < na) {
            const long long p0 = (long long)SB_SHIFT * ws.list[(size_t)b * ws.NA + k0 + tid];
            q0 = (int)(p0 % L);
            const long long left = total - p0;
            nv = left >= SB_WLEN ? SB_WLEN : (left > 0 ? (int)left : 0);
        }
        fq0[tid] = q0; fvalid[tid] = nv;
    }
    const bool act = fs < SP_F && k0 + fs < na;
    for (int sig = sig0; sig <= sig1; ++sig) {
        const float* sb = (sig ? y : x) + (size_t)b * Lrow;
        __syncthreads();
        for (int e = tid; e < SP_F * SB_WLEN; e += 128) {
            const int f_ = e / SB_WLEN, j = e - f_ * SB_WLEN;
            double v = 0.0;
            if (fq0[f_] >= 0 && j < fvalid[f_]) {
                int q = fq0[f_] + j;
                if (q >= L) q -= L;                         // L >= 400 (checked on the host): one wrap at most
                v = (double)sb[q] * ws.tab[2 * SB_WLEN + j];
            }
            sx[f_][j] = v;
        }
        __syncthreads();
        if (act) {      // stage 1: a[k1] = sum_n1 v[n1] W20^(n1 k1), then * W400^(n2 k1)
            const int n2 = lane20;
            double v[20];
#pragma unroll
            for (int n1 = 0; n1 < 20; ++n1) v[n1] = sx[fs][20 * n1 + n2];
#pragma unroll
            for (int k1 = 0; k1 <= 10; ++k1) {
                double ar = 0.0, ai = 0.0;
#pragma unroll
                for (int n1 = 0; n1 < 20; ++n1) {
                    ar += v[n1] * tw.c[(n1 * k1) % 20];
                    ai -= v[n1] * tw.s[(n1 * k1) % 20];
                }
                {
                    const double c2 = tcs[n2 * k1], s2 = tsn[n2 * k1];
                    Aq[fs][n2 * 20 + k1] = make_double2(ar * c2 + ai * s2, ai * c2 - ar * s2);
                }
                if (k1 >= 1 && k1 <= 9) {              // a[20 - k1] = conj(a[k1])
                    const int kc = 20 - k1;
                    const double c2 = tcs[n2 * kc], s2 = tsn[n2 * kc];
                    Aq[fs][n2 * 20 + kc] = make_double2(ar * c2 - ai * s2, -ai * c2 - ar * s2);
                }
            }
        }
        __syncthreads();
        if (act) {      // stage 2: X[k1 + 20 k2] = sum_n2 A[n2][k1] W20^(n2 k2); |X|^2 into sx (the frames are dead)
            const int k1 = lane20;
            double2 a[20];
#pragma unroll
            for (int n2 = 0; n2 < 20; ++n2) a[n2] = Aq[fs][n2 * 20 + k1];
#pragma unroll
            for (int k2 = 0; k2 <= 10; ++k2) {
                double xr = 0.0, xi = 0.0;
#pragma unroll
                for (int n2 = 0; n2 < 20; ++n2) {
                    const double c = tw.c[(n2 * k2) % 20], s_ = tw.s[(n2 * k2) % 20];
                    xr += a[n2].x * c + a[n2].y * s_;
                    xi += a[n2].y * c - a[n2].x * s_;
                }
                const int kk = k1 + 20 * k2;
                if (kk < SB_NBIN) sx[fs][kk] = xr * xr + xi * xi;
            }
        }
        __syncthreads();
        // band energies: out[f][j] = sum_q |X_f[q]|^2 g2[j][q].  thread = (band j, quarter of the bins), all six frames per thread:
        // one coalesced read of the transposed filter row per bin, the six spectra are LDS broadcasts
        {
            double (*bpart)[SP_F][SB_J] = reinterpret_cast<double (*)[SP_F][SB_J]>(&Aq[0][0]);   // [4][SP_F][28], Aq is dead here
            if (tid < 4 * SB_J) {
                const int j = tid % SB_J, qp = tid / SB_J;
                const int q0 = qp * 51, q1 = min(SB_NBIN, q0 + 51);
                double acc[SP_F];
#pragma unroll
                for (int f_ = 0; f_ < SP_F; ++f_) acc[f_] = 0.0;
#pragma unroll 4
                for (int q = q0; q < q1; ++q) {
                    const double gq = ws.g2t[q * SB_J + j];
#pragma unroll
                    for (int f_ = 0; f_ < SP_F; ++f_) acc[f_] += gq * sx[f_][q];
                }
#pragma unroll
                for (int f_ = 0; f_ < SP_F; ++f_) bpart[qp][f_][j] = acc[f_];
            }
            __syncthreads();
            for (int o = tid; o < SP_F * SB_J; o += 128) {
                const int f_ = o / SB_J, j = o - f_ * SB_J;
                if (k0 + f_ < na) {
                    const double a = (bpart[0][f_][j] + bpart[1][f_][j]) + (bpart[2][f_][j] + bpart[3][f_][j]);
                    ws.XL[(((size_t)b * 2 + sig) * SB_J + j) * ws.NA + k0 + f_] = log(a + SB_EPS);
                }
            }
        }
    }
}
#endif  // NELE_AB

// s3, wave-autonomous form (round 3, third session; NELE_SIIB_SPECW=0 = the kernel above).  The kernel above runs at one wave per SIMD
// (502 registers: with the 20-point twiddles as literals the compiler pre-multiplies and keeps hundreds of products live; 64 KB of
// LDS per 2-wave workgroup) with four workgroup barriers per signal, a staging loop of dependent global loads and a band stage that
// waits for 51 filter rows four at a time: 22 us per workgroup for 3.5 us of float64 arithmetic - 2.9 ms per signal at B = 256 on a
// non-periodic length, the largest kernel of that step.  Here a WAVE owns three frames from the samples to the band energies:
//  * a lane loads its 20 samples straight into registers (thread = (frame, n2): for each n1 the 20 lanes of a frame read 80
//    contiguous bytes), one (group, signal) pass AHEAD of their use - no staged frame in LDS, no load latency on the chain;
//  * both 20-point stages split radix-2 first (W20^(10 k) = (-1)^k: even outputs from sums, odd ones from differences; stage 2 in two
//    passes over its LDS operands, 40 registers of operands each); the four twiddle magnitudes are wave-uniform scalars, their signs /
//    zeros / ones resolved at compile time from the table index: 850 float64 operations per lane and frame, 256 registers, two waves
//    per SIMD; window values and the W400 twiddles are re-read from L1 per pass (laundered pointer) instead of living in 120 registers;
//  * LDS only carries the stage-1 output (6.4 KB per frame; |X|^2 overlays it once the stage-2 operands are in registers): 38 KB per
//    workgroup, four workgroups = eight waves per CU;
//  * the stages are ordered by wave-level fences only (the three frames of a wave never meet another wave's);
//  * band stage: lane = (band, half of the bins), filter rows from L1 / L2 in blocks of 25 with all loads of a block in flight (the
//    45 KB table does not fit L1: a block costs an L2 latency), the halves meet through a shuffle.
// Same transform in another association (radix-2 split, 2 partial band sums instead of 4): scores agree with the kernel above to the
// last float32 bit or two (tests/test_metrics_gpu.py).  Measured at B = 256, L = 63 871, both signals in one launch: 5.73 -> 2.79 ms
// (what is left by diagnostic builds: band stage 1.1 ms, sample / table loads 0.5 ms, the transform itself 1.3 ms, instruction-issue
// bound - every wave64 instruction takes four cycles and only 40 % of them are float64 arithmetic).
#ifndef SPW_NG
#define SPW_NG 1         // groups of SP_F frames a workgroup walks: 1 2.43, 2 2.46, 4 2.52, 8 2.59, 16 2.73 ms per call (B = 256, L = 63 871)
#endif
#ifndef SPW_BLK
#define SPW_BLK 10        // filter rows of the band stage in flight per block (divides 100): 2 2.87, 4 2.62, 5 2.61, 10 2.53, 20 2.62, 25 2.82, 50 2.78 ms
#endif                   // per call at B = 256, L = 63 871 (both signals; tools/r4_sp.sh) - the stage is not waiting for L2, more loads in flight only cost registers
// cos / sin (2 pi t / 20): the kernel reads entries 1 .. 4 (the four magnitudes) as scalars
__constant__ double c_dft20c[20] = {1.0, 0.95105651629515357212, 0.80901699437494742410, 0.58778525229247312917, 0.30901699437494742410, 0.0,
                                    -0.30901699437494742410, -0.58778525229247312917, -0.80901699437494742410, -0.95105651629515357212, -1.0,
                                    -0.95105651629515357212, -0.80901699437494742410, -0.58778525229247312917, -0.30901699437494742410, 0.0,
                                    0.30901699437494742410, 0.58778525229247312917, 0.80901699437494742410, 0.95105651629515357212};
__global__ __launch_bounds__(128, 2) void siib_spec_wave_kernel(const float* __restrict__ x, const float* __restrict__ y, int L, SiibWs ws, int sig0,
                                                             int sig1) {
    __shared__ __attribute__((aligned(16))) double2 Aq[SP_F][SB_WLEN];   // stage-1 output [n2][k1]; then |X|^2 in its first 201 doubles
    const int b = blockIdx.y, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    // the four magnitudes of cos / sin (2 pi t / 20) as wave-uniform scalars (8 SGPRs for the whole kernel); signs, zeros and ones are
    // resolved at compile time from the table index, so multiplications by 0 and +-1 disappear and nothing waits for a table load
    const double M1 = c_dft20c[1], M2 = c_dft20c[2], M3 = c_dft20c[3], M4 = c_dft20c[4];
#define TWM(m) ((m) == 0 ? 1.0 : (m) == 1 ? M1 : (m) == 2 ? M2 : (m) == 3 ? M3 : (m) == 4 ? M4 : 0.0)
#define TWC(t) ((t) <= 5 ? TWM(t) : (t) <= 10 ? -TWM(10 - (t)) : (t) <= 15 ? -TWM((t) - 10) : TWM(20 - (t)))
#define TWS(t) ((t) <= 5 ? TWM(5 - (t)) : (t) <= 10 ? TWM((t) - 5) : (t) <= 15 ? -TWM(15 - (t)) : -TWM((t) - 15))
    const int* info = ws.info + 4 * b;
    const int na = ws.nprim[b];                        // frames of the first period only: siib_spread_kernel copies the repeats
    const int kbase = blockIdx.x * (SP_F * SPW_NG);
    if (kbase >= na) return;
    const int Lrow = L;
    L = sb_len(ws, b, L);
    const long long total = (long long)info[0] * L;
    const int fl = lane / 20, l20 = lane - 20 * fl;    // frame of the wave (3 = idle lanes 60..63), n2 (stage 1) / k1 (stage 2)
    const int fs = min(3 * wv + fl, SP_F - 1);
    const bool lact = fl < 3;
    double2* Af = Aq[fs];
    double* Pf = reinterpret_cast<double*>(Af);
    const int bj = lane % SB_J, bpart = lane / SB_J;   // band stage: band, half of the bins (lanes 56..63 idle: they shadow part 0)
    const int bp = bpart < 2 ? bpart : 0;
    const double* gp = ws.g2t + (bp * 100) * SB_J + bj;
    const double* P0 = reinterpret_cast<const double*>(Aq[3 * wv]) + bp * 100;
    const double* P1 = reinterpret_cast<const double*>(Aq[3 * wv + 1]) + bp * 100;
    const double* P2 = reinterpret_cast<const double*>(Aq[3 * wv + 2]) + bp * 100;
    // frame bookkeeping of group g for this lane: first sample (mod L) and the samples before the tiled signal ends
    auto frame_of = [&](int g, int& q0, int& nv) {
        q0 = 0; nv = 0;
        const int k0 = kbase + SP_F * g;
        if (g < SPW_NG && lact && k0 + fs < na) {
            const long long p0 = (long long)SB_SHIFT * ws.list[(size_t)b * ws.NA + k0 + fs];
            q0 = (int)(p0 % L);
            const long long left = total - p0;
            nv = left >= SB_WLEN ? SB_WLEN : (left > 0 ? (int)left : 0);
        }
    };
    // the lane's 20 raw samples of one (group, signal) pass: 20 independent loads, issued one pass AHEAD of their use
    auto fetch = [&](const float* __restrict__ sb, int q0, int nv, float (&r)[20]) {
#pragma unroll
        for (int n1 = 0; n1 < 20; ++n1) {
            const int j = 20 * n1 + l20;
            int q = q0 + j;
            if (q >= L) q -= L;                             // L >= 400 (checked on the host): one wrap at most
            r[n1] = (j < nv) ? sb[q] : 0.f;
        }
    };
    const float* xb = x + (size_t)b * Lrow;
    const float* yb = y + (size_t)b * Lrow;
    int q0, nv;
    float cur[20];
    frame_of(0, q0, nv);
    fetch(sig0 ? yb : xb, q0, nv, cur);
    for (int g = 0; g < SPW_NG; ++g) {
        const int k0 = kbase + SP_F * g;
        if (k0 >= na) break;
        const bool act = nv > 0 || (lact && k0 + fs < na);
        for (int sig = sig0; sig <= sig1; ++sig) {
            float nxt[20];
            int q0n = q0, nvn = nv;
            if (sig == sig1) frame_of(g + 1, q0n, nvn);
            fetch((sig == sig1 ? sig0 : sig + 1) ? yb : xb, q0n, nvn, nxt);
            const double* tab = ws.tab;                     // laundered per pass: the per-lane table values are re-read from L1 instead of
            asm volatile("" : "+s"(tab));                   // living in 120 registers across the whole kernel
            if (act) {      // stage 1: a[k1] = sum_n1 v[n1] W20^(n1 k1), then * W400^(n2 k1); radix 2 first: W20^(10 k1) = (-1)^k1
                double ve[10], vo[10];
#pragma unroll
                for (int n1 = 0; n1 < 10; ++n1) {
                    const int j0 = 20 * n1 + l20, j1 = j0 + 200;
                    const double va = (j0 < nv) ? (double)cur[n1] * tab[2 * SB_WLEN + j0] : 0.0;
                    const double vb = (j1 < nv) ? (double)cur[n1 + 10] * tab[2 * SB_WLEN + j1] : 0.0;
                    ve[n1] = va + vb; vo[n1] = va - vb;
                }
#pragma unroll
                for (int k1 = 0; k1 <= 10; ++k1) {
                    double ar = 0.0, ai = 0.0;
#pragma unroll
                    for (int n1 = 0; n1 < 10; ++n1) {
                        const double vv = (k1 & 1) ? vo[n1] : ve[n1];
                        ar += vv * TWC((n1 * k1) % 20);
                        ai -= vv * TWS((n1 * k1) % 20);
                    }
                    {
                        const double c2 = tab[l20 * k1], s2 = tab[SB_WLEN + l20 * k1];
                        Af[l20 * 20 + k1] = make_double2(ar * c2 + ai * s2, ai * c2 - ar * s2);
                    }
                    if (k1 >= 1 && k1 <= 9) {              // a[20 - k1] = conj(a[k1])
                        const int kc = 20 - k1;
                        const double c2 = tab[l20 * kc], s2 = tab[SB_WLEN + l20 * kc];
                        Af[l20 * 20 + kc] = make_double2(ar * c2 - ai * s2, -ai * c2 - ar * s2);
                    }
                }
            }
            sb_wave_sync();
            // stage 2: X[k1 + 20 k2] = sum_n2 A[n2][k1] W20^(n2 k2), k2 = 0..10 (only bins <= 200 are used); even k2 from the sums
            // A[n2] + A[n2 + 10], odd k2 from the differences - two passes over the LDS operands, 40 registers of operands each
            double pw[11];
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                if (act) {
                    double2 h[10];
#pragma unroll
                    for (int n2 = 0; n2 < 10; ++n2) {
                        const double2 u = Af[n2 * 20 + l20], w = Af[(n2 + 10) * 20 + l20];
                        h[n2] = par ? make_double2(u.x - w.x, u.y - w.y) : make_double2(u.x + w.x, u.y + w.y);
                    }
#pragma unroll
                    for (int k2 = par; k2 <= 10; k2 += 2) {
                        double xr = 0.0, xi = 0.0;
#pragma unroll
                        for (int n2 = 0; n2 < 10; ++n2) {
                            const double c = TWC((n2 * k2) % 20), s_ = TWS((n2 * k2) % 20);
                            xr += h[n2].x * c + h[n2].y * s_;
                            xi += h[n2].y * c - h[n2].x * s_;
                        }
                        pw[k2] = xr * xr + xi * xi;
                    }
                }
            }
            sb_wave_sync();                                 // every lane has read its operands: the frame's region is free for |X|^2
            if (act) {
#pragma unroll
                for (int k2 = 0; k2 <= 10; ++k2) {
                    const int kk = l20 + 20 * k2;
                    if (kk < SB_NBIN) Pf[kk] = pw[k2];
                }
            }
            sb_wave_sync();
            // band energies of the wave's three frames: out[f][j] = sum_q |X_f[q]|^2 g2[j][q]
            double e0 = 0.0, e1 = 0.0, e2 = 0.0;
            // bins 100 bp .. 100 bp + 99 in blocks of SPW_BLK filter rows, all loads of a block in flight; bin 200 belongs to half 1
            const double g200 = ws.g2t[200 * SB_J + bj];
#pragma unroll 1
            for (int c = 0; c < 100 / SPW_BLK; ++c) {
                double gq[SPW_BLK];
#pragma unroll
                for (int u = 0; u < SPW_BLK; ++u) gq[u] = gp[(SPW_BLK * c + u) * SB_J];
#pragma unroll
                for (int u = 0; u < SPW_BLK; ++u) {
                    const int q = SPW_BLK * c + u;
                    e0 += gq[u] * P0[q]; e1 += gq[u] * P1[q]; e2 += gq[u] * P2[q];
                }
            }
            if (bpart == 1) { e0 += g200 * P0[100]; e1 += g200 * P1[100]; e2 += g200 * P2[100]; }
            e0 += __shfl(e0, lane + SB_J, 64); e1 += __shfl(e1, lane + SB_J, 64); e2 += __shfl(e2, lane + SB_J, 64);
            if (lane < SB_J) {
                double* o = ws.XL + (((size_t)b * 2 + sig) * SB_J + bj) * ws.NA + k0 + 3 * wv;
                if (k0 + 3 * wv < na) o[0] = log(e0 + SB_EPS);
                if (k0 + 3 * wv + 1 < na) o[1] = log(e1 + SB_EPS);
                if (k0 + 3 * wv + 2 < na) o[2] = log(e2 + SB_EPS);
            }
            sb_wave_sync();                                 // the regions are rewritten by the next signal / group
#pragma unroll
            for (int n1 = 0; n1 < 20; ++n1) cur[n1] = nxt[n1];
            q0 = q0n; nv = nvn;
        }
    }
}
#undef TWM
#undef TWC
#undef TWS

// s3b: the tiled signal repeats every Pf = L / gcd(L, 200) frames (frame f starts at sample 200 f mod L; every frame lies wholly inside
// the tiled signal), so the band energies of an active frame f >= Pf are those of frame f mod Pf, which is active too (same samples,
// same arithmetic, same dB value).  One thread per active frame beyond the first period: binary search of f mod Pf in the sorted list
// of first-period frames, then copy the 28 band values of each signal.  grid (ceil(NA / 256), B), block 256
__global__ __launch_bounds__(256) void siib_spread_kernel(SiibWs ws, int Pf, int sig0, int sig1) {
    const int b = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    int* info = ws.info + 4 * b;
    const int na = info[2], np_ = ws.nprim[b];
    if (a < np_ || a >= na) return;
    const int* list = ws.list + (size_t)b * ws.NA;
    const int fr = list[a] % Pf;
    int lo = 0, hi = np_ - 1, src = -1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1, v = list[mid];
        if (v == fr) { src = mid; break; }
        if (v < fr) lo = mid + 1; else hi = mid - 1;
    }
    if (src != a % np_) { atomicOr(&info[3], 16); return; }   // cannot happen (see above); the score becomes NaN if it does
    for (int sig = sig0; sig <= sig1; ++sig) {
        double* row = ws.XL + ((size_t)b * 2 + sig) * SB_J * ws.NA;
#pragma unroll 4
        for (int j = 0; j < SB_J; ++j) row[(size_t)j * ws.NA + a] = row[(size_t)j * ws.NA + src];
    }
}

// s4a: band minima before masking.  grid (28, B, signals), block 256
__global__ __launch_bounds__(256) void siib_rowmin_kernel(SiibWs ws, int sig0) {
    __shared__ double red[8];
    const int j = blockIdx.x, b = blockIdx.y, sig = sig0 + blockIdx.z, tid = threadIdx.x;
    const int na = ws.info[4 * b + 2];
    const double* row = ws.XL + (((size_t)b * 2 + sig) * SB_J + j) * ws.NA;
    double m = 1e300;
    for (int i = tid; i < na; i += 256) m = fmin(m, row[i]);
    m = -block_max(-m, red);
    if (tid == 0) ws.rowstat[(((size_t)b * 2 + sig) * SB_J + j) * 2] = m;
}

// s4b: forward masking.  grid (B, signals), block 64: lanes 0..27 = band rows of one signal.  The recurrence is serial over frames
// per row; rows are streamed through LDS in chunks of 64 frames (coalesced loads/stores), the next chunk's loads are in flight
// while the current one is processed, and 8 frames' inputs are read from LDS ahead of the dependent chain.  The row means of
// the masked values go to rowstat; siib_stack_kernel subtracts them on the fly (same arithmetic as a separate pass).
#define SB_CH 64
// fmax() on doubles compiles to two canonicalising v_max_f64 plus the max itself (IEEE sNaN quieting); the values here are never NaN
__device__ __forceinline__ double max_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__global__ __launch_bounds__(64) void siib_mask_kernel(SiibWs ws, int sig0) {
    __shared__ double buf[SB_J][SB_CH + 1];
    const int b = blockIdx.x, sig = sig0 + blockIdx.y, tid = threadIdx.x;
    const int na = ws.info[4 * b + 2];
    if (na < 1) return;
    // Periodic input (siib_spread_kernel: XL[a] = XL[a mod npr] when npr < na).  The recurrence's whole state is pend[]; when the
    // state at a period boundary equals the state one period earlier, every later period repeats the last one bit for bit, so the
    // serial recurrence stops there and the rest is a copy (the row sum still accumulates in frame order).
    const int npr = ws.nprim[b];
    const bool periodic = npr < na && npr >= SB_CH;
    double* base = ws.XL + ((size_t)b * 2 + sig) * SB_J * ws.NA;
    double* rstat = ws.rowstat + ((size_t)b * 2 + sig) * SB_J * 2;
    const double eX = (tid < SB_J) ? rstat[2 * tid] : 0.0;
    double lt[SB_TF];
#pragma unroll
    for (int m = 0; m < SB_TF; ++m) lt[m] = log((double)(m + 1)) / log((double)SB_TF);
    // pend[m] = masking level already imposed on frame (i + 1 + m) by frames <= i
    double pend[SB_TF - 1], snap[SB_TF - 1];
#pragma unroll
    for (int m = 0; m < SB_TF - 1; ++m) { pend[m] = -1e300; snap[m] = 0.0; }
    int nextb = periodic ? npr : 0x7fffffff;   // next period boundary (frame index)
    int steady = 0, bsteady = 0;               // this row's state repeated at boundary bsteady
    double sum = 0.0;
    double t[SB_J];
#pragma unroll
    for (int r = 0; r < SB_J; ++r) t[r] = base[(size_t)r * ws.NA + min(tid, ws.NA - 1)];
    int c0 = 0;
    for (; c0 < na; c0 += SB_CH) {
        const int n = min(SB_CH, na - c0);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SB_J; ++r) buf[r][tid] = t[r];
        __syncthreads();
        if (c0 + SB_CH < na) {
#pragma unroll
            for (int r = 0; r < SB_J; ++r) t[r] = base[(size_t)r * ws.NA + min(c0 + SB_CH + tid, ws.NA - 1)];
        }
        if (tid < SB_J) {
            for (int i0 = 0; i0 < n; i0 += 8) {
                double xs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) xs[u] = buf[tid][min(i0 + u, SB_CH - 1)];
                if (c0 + i0 + 8 > nextb) {     // a period boundary inside these 8 frames: handle it frame by frame
                    for (int u = 0; u < 8; ++u) {
                        if (i0 + u < n) {
                            if (c0 + i0 + u == nextb) {
                                int eq = 1;
#pragma unroll
                                for (int m = 0; m < SB_TF - 1; ++m) { eq &= (pend[m] == snap[m]) ? 1 : 0; snap[m] = pend[m]; }
                                if (eq && nextb > npr && !steady) { steady = 1; bsteady = nextb; }
                                nextb += npr;
                            }
                            const double v = max_f64(xs[u], pend[0]);
#pragma unroll
                            for (int m = 1; m < SB_TF - 1; ++m) pend[m - 1] = max_f64(pend[m], v - (v - eX) * lt[m]);
                            pend[SB_TF - 2] = v - (v - eX) * lt[SB_TF - 1];
                            buf[tid][i0 + u] = v;
                            sum += v;
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (i0 + u < n) {
                        const double v = max_f64(xs[u], pend[0]);
#pragma unroll
                        for (int m = 1; m < SB_TF - 1; ++m) pend[m - 1] = max_f64(pend[m], v - (v - eX) * lt[m]);
                        pend[SB_TF - 2] = v - (v - eX) * lt[SB_TF - 1];
                        buf[tid][i0 + u] = v;
                        sum += v;
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int r = 0; r < SB_J; ++r)
            if (tid < n) base[(size_t)r * ws.NA + c0 + tid] = buf[r][tid];
        if (periodic && __all(tid >= SB_J || steady)) { c0 += SB_CH; break; }
    }
    if (c0 < na) {
        // every row repeats with period npr from its boundary on; the latest boundary bq <= c0 serves all rows: frame i >= bq equals
        // frame bq - npr + (i - bq) mod npr, all of which are final in memory (written before the last barrier)
        int bq = bsteady;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bq = max(bq, __shfl_xor(bq, o, 64));
        __syncthreads();
        auto src = [&](int i) { return bq - npr + (i - bq) % npr; };
#pragma unroll
        for (int r = 0; r < SB_J; ++r) t[r] = base[(size_t)r * ws.NA + src(min(c0 + tid, na - 1))];
        for (; c0 < na; c0 += SB_CH) {
            const int n = min(SB_CH, na - c0);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < SB_J; ++r) {
                buf[r][tid] = t[r];
                if (tid < n) base[(size_t)r * ws.NA + c0 + tid] = t[r];
            }
            __syncthreads();
            if (c0 + SB_CH < na) {
#pragma unroll
                for (int r = 0; r < SB_J; ++r) t[r] = base[(size_t)r * ws.NA + src(min(c0 + SB_CH + tid, na - 1))];
            }
            if (tid < SB_J)
                for (int i = 0; i < n; ++i) sum += buf[tid][i];
        }
    }
    if (tid < SB_J) rstat[2 * tid + 1] = sum / (double)na;
}

// grid (28 bands, B, signals), block 256, dynamic LDS NA doubles: s5.  The 15 stacked rows of a band are shifted copies of ONE row of
// the masked spectrum: it is read once into LDS (the version with one block per stacked row read every band row 30 times), the 15
// window means are summed from there in the same thread-strided order as before (bit-identical), and the 15 rows go out.
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(256) void siib_stack_kernel(SiibWs ws, int sig0) {
    extern __shared__ double sk_v[];                     // v[t] = masked band value - its row mean
    __shared__ double red[8];
    __shared__ double mus[SB_K];
    const int j = blockIdx.x, b = blockIdx.y, sig = sig0 + blockIdx.z, tid = threadIdx.x;
    const int na = ws.info[4 * b + 2];
    const int ncols = na - SB_K + 1;
    double* dst0 = ws.Xs + (((size_t)b * 2 + sig) * SB_D + j) * ws.NA;               // row a = k * 28 + j
    if (ncols < 2) {
        for (int k = 0; k < SB_K; ++k)
            for (int t = tid; t < ws.NA; t += 256) dst0[(size_t)k * SB_J * ws.NA + t] = 0.0;
        return;
    }
    const double* src = ws.XL + (((size_t)b * 2 + sig) * SB_J + j) * ws.NA;
    const double mr = ws.rowstat[(((size_t)b * 2 + sig) * SB_J + j) * 2 + 1];     // row mean of the masked band (removed here)
    for (int t = tid; t < na; t += 256) sk_v[t] = src[t] - mr;
    __syncthreads();
    for (int k = 0; k < SB_K; ++k) {
        double sm = 0.0;
        for (int t = tid; t < ncols; t += 256) sm += sk_v[t + k];
        const double mu = block_sum(sm, red) / (double)ncols;
        if (tid == 0) mus[k] = mu;
    }
    __syncthreads();
    for (int k = 0; k < SB_K; ++k) {
        const double mu = mus[k];
        double* dst = dst0 + (size_t)k * SB_J * ws.NA;
        for (int t = tid; t < ws.NA; t += 256) dst[t] = (t < ncols) ? sk_v[t + k] - mu : 0.0;
    }
}
#endif  // NELE_AB

// ---------------------------------------------------------------- float64 MFMA GEMMs
// v_mfma_f64_16x16x4_f64: A lane l = A[row l&15][k l>>4], B lane l = B[k l>>4][col l&15], D reg q of lane l = D[(l>>4) + 4q][l&15].
// One double per lane and operand feeds 2048 flops, so LDS traffic is negligible next to the vector-FMA formulation (which
// was LDS-bound at ~15 TFLOP/s); tiles are double-buffered in LDS with the next tile's global loads in flight.
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define SG_LD 68    // LDS row stride (doubles) of a [16][64] k-major tile

// C[b][i][j] = scale_b * sum_t Xs[b][0][i][t] * Xs[b][0][j][t]   (s6).  grid (7, 7, B): lower-triangle 64x64 tiles only, mirrored
// on store.  Waves 2x2, 32x32 per wave.  Columns >= n_cols of Xs are zero padded and NA is a multiple of 64, so K needs no guard;
// rows >= 420 are clamped (they only reach outputs that are never stored).
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(256) void siib_cov_kernel(SiibWs ws) {
    __shared__ __attribute__((aligned(16))) double As[2][16][SG_LD], Bs[2][16][SG_LD];
    // 1-D grid, XCD-aware (workgroup id w runs on XCD w % 8): all tiles of an utterance get ids of one XCD, so that its L2 serves the row
    // tiles of Xs that they share (with the natural 3-D grid the 28 tiles of an utterance sat on all eight XCDs: 4.7 x the operand bytes
    // from HBM, and the kernel ran at HBM speed)
    const int xw = (int)blockIdx.x, slot = xw >> 3;
    const int b = (slot / 49) * 8 + (xw & 7), tile = slot % 49, bi = tile / 7, bj = tile % 7;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (b >= ws.Bn || bj > bi) return;
    const int ti = bi * 64, tj = bj * 64;
    const int na = ws.info[4 * b + 2];
    const int ncols = na - SB_K + 1;
    const double* X = ws.Xs + (size_t)b * 2 * SB_D * ws.NA;
    const int kmax = (ncols > 0) ? ((ncols + 15) & ~15) : 0;
    const int lr = tid >> 2, lq = (tid & 3) * 4;
    const double* pa = X + (size_t)min(ti + lr, SB_D - 1) * ws.NA + lq;
    const double* pb = X + (size_t)min(tj + lr, SB_D - 1) * ws.NA + lq;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32, li = lane & 15, lk = lane >> 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    double4 ra, rb;
    if (kmax > 0) { ra = *reinterpret_cast<const double4*>(pa); rb = *reinterpret_cast<const double4*>(pb); }
    for (int k0 = 0, it = 0; k0 < kmax; k0 += 16, ++it) {
        const int buf = it & 1;
        As[buf][lq][lr] = ra.x; As[buf][lq + 1][lr] = ra.y; As[buf][lq + 2][lr] = ra.z; As[buf][lq + 3][lr] = ra.w;
        Bs[buf][lq][lr] = rb.x; Bs[buf][lq + 1][lr] = rb.y; Bs[buf][lq + 2][lr] = rb.z; Bs[buf][lq + 3][lr] = rb.w;
        __syncthreads();
        if (k0 + 16 < kmax) {
            ra = *reinterpret_cast<const double4*>(pa + k0 + 16);
            rb = *reinterpret_cast<const double4*>(pb + k0 + 16);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = 4 * kk + lk;
            const double a0 = As[buf][k][wr + li], a1 = As[buf][k][wr + 16 + li];
            const double b0 = Bs[buf][k][wc + li], b1 = Bs[buf][k][wc + 16 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    const double scale = (ncols > 1) ? 1.0 / (double)(ncols - 1) : 0.0;
    double* C = ws.C + (size_t)b * SB_D * SB_D;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gi = ti + wr + 16 * i + lk + 4 * q, gj = tj + wc + 16 * j + li;
                if (gi < SB_D && gj < SB_D) {
                    const double v = acc[i][j][q] * scale;
                    C[(size_t)gi * SB_D + gj] = v;
                    if (bi != bj) C[(size_t)gj * SB_D + gi] = v;
                }
            }
}
#endif  // NELE_AB

// s8: P = U X for both signals (U rows = eigenvectors, [420][420]; X [420][NA]); per 64x64 tile of P emit the row-wise partial
// sums of Xp^2, Yp^2, Xp*Yp.  grid (NTL, 7, B).  Waves 4x1: a wave owns 16 eigenvectors x 64 frames of both signals, so the row
// sums stay inside the wave (in-lane over the 4 column tiles, then a DPP reduction over the 16 lanes of a row).
// MODE 0: both signals (one-shot call).  MODE 1: clean signal only - its projections are kept (ws.px, accumulator order) together
// with the sum of squares; MODE 2: degraded signal only, the clean projections are read back.  1 + 2 perform exactly the MFMA
// sequences of 0, so the split is bit-identical; it takes half of the projection off the path that waits for the enhanced signal.
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
template <int MODE>
__global__ __launch_bounds__(256) void siib_proj_kernel(SiibWs ws) {
    __shared__ __attribute__((aligned(32))) double Us[2][16][SG_LD], Xt[2][16][SG_LD], Yt[2][16][SG_LD];
    // 1-D grid, XCD-aware like siib_cov_kernel: the 7 x NTL tiles of an utterance share U and the column tiles of Xs in one L2
    const int xw = (int)blockIdx.x, slot = xw >> 3, per_b = 7 * ws.NTL;
    const int b = (slot / per_b) * 8 + (xw & 7), tl = slot % per_b, by = tl % 7, bx = tl / 7;   // the 7 row tiles of a column tile are adjacent
    const int ti = by * 64, t0 = bx * 64, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    if (b >= ws.Bn) return;
    if (t0 >= ws.info[4 * b + 2] - SB_K + 1) {      // tile beyond n_cols: all zero padding
        if (tid < 192) {
            const int q = tid / 64, r = tid - q * 64;
            const bool mine = (MODE == 0) || (MODE == 1 && q == 0) || (MODE == 2 && q != 0);
            if (mine && ti + r < SB_D) ws.part[(((size_t)b * SB_D + ti + r) * ws.NTL + bx) * 3 + q] = 0.0;
        }
        return;
    }
    const double* U = ws.U + (size_t)b * SB_D * SB_D;
    const double* X = ws.Xs + (size_t)b * 2 * SB_D * ws.NA;
    const double* Y = X + (size_t)SB_D * ws.NA;
    const int ur = tid >> 2, uq = (tid & 3) * 4;                 // U tile: 64 rows x 16 k, one double4 per thread
    const double* pu = U + (size_t)min(ti + ur, SB_D - 1) * SB_D + uq;
    const int xc = tid >> 4, xq = (tid & 15) * 4;                // X / Y tiles: 16 k x 64 frames, one double4 per thread
    const double* px = X + (size_t)xc * ws.NA + t0 + xq;
    const double* py = Y + (size_t)xc * ws.NA + t0 + xq;
    f64x4 ax[4], ay[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ax[j] = (f64x4){0.0, 0.0, 0.0, 0.0}; ay[j] = (f64x4){0.0, 0.0, 0.0, 0.0}; }
    const double4 z4 = make_double4(0.0, 0.0, 0.0, 0.0);
    auto gload = [&](int k0, double4& ru, double4& rx, double4& ry) {
        ru = (k0 + uq < SB_D) ? *reinterpret_cast<const double4*>(pu + k0) : z4;          // 420 = 4 * 105: quads are all-in or all-out
        const bool kin = k0 + xc < SB_D;
        rx = (kin && MODE != 2) ? *reinterpret_cast<const double4*>(px + (size_t)k0 * ws.NA) : z4;
        ry = (kin && MODE != 1) ? *reinterpret_cast<const double4*>(py + (size_t)k0 * ws.NA) : z4;
    };
    double4 ru, rx, ry;
    gload(0, ru, rx, ry);
    for (int k0 = 0, it = 0; k0 < SB_D; k0 += 16, ++it) {
        const int buf = it & 1;
        Us[buf][uq][ur] = ru.x; Us[buf][uq + 1][ur] = ru.y; Us[buf][uq + 2][ur] = ru.z; Us[buf][uq + 3][ur] = ru.w;
        if (MODE != 2) *reinterpret_cast<double4*>(&Xt[buf][xc][xq]) = rx;
        if (MODE != 1) *reinterpret_cast<double4*>(&Yt[buf][xc][xq]) = ry;
        __syncthreads();
        if (k0 + 16 < SB_D) gload(k0 + 16, ru, rx, ry);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = 4 * kk + lk;
            const double a = Us[buf][k][16 * w + li];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE != 2) ax[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Xt[buf][k][16 * j + li], ax[j], 0, 0, 0);
                if (MODE != 1) ay[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Yt[buf][k][16 * j + li], ay[j], 0, 0, 0);
            }
        }
    }
    double* pxs = ws.px + ((((size_t)b * 7 + by) * ws.NTL + bx) * 16) * 256 + tid;
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) pxs[(4 * j + q) * 256] = ax[j][q];
    } else if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) ax[j][q] = pxs[(4 * j + q) * 256];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double sxx = 0.0, syy = 0.0, sxy = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { sxx += ax[j][q] * ax[j][q]; syy += ay[j][q] * ay[j][q]; sxy += ax[j][q] * ay[j][q]; }
        sxx = row16_sum_dpp(sxx); syy = row16_sum_dpp(syy); sxy = row16_sum_dpp(sxy);
        const int gi = ti + 16 * w + lk + 4 * q;
        if (li == 0 && gi < SB_D) {
            double* dst = ws.part + (((size_t)b * SB_D + gi) * ws.NTL + bx) * 3;
            if (MODE != 2) dst[0] = sxx;
            if (MODE != 1) { dst[1] = syy; dst[2] = sxy; }
        }
    }
}
#endif  // NELE_AB


// ---------------------------------------------------------------- second moments of the stacked frames from LAG PRODUCTS (round 3)
// The 420-dimensional stacked frame is 15 consecutive 28-band frames (oracle/siib.py stack), so every entry of Xs Xs^T, Ys Ys^T and
// Xs Ys^T is a lag product of two band rows,
//     S_ab[(k1, j1), (k2, j2)] = sum_{t < ncols} a_j1[t + k1] b_j2[t + k2] - ncols mu^a_{k1 j1} mu^b_{k2 j2}
//                              = L_ab(j1, j2, k2 - k1) - (at most 14 head and 14 tail products) - ncols mu mu',
//     L_ab(j1, j2, D) = sum_s a_j1[s] b_j2[s + D]            (all s for which both indices are frames),
// with a, b the masked, row-mean-removed log band energies: 28 x 28 x 29 lag products of length n instead of 420 x 420 inner products -
// 8 x fewer multiply-adds for the covariance.  And the per-component sums the score needs (oracle/siib.py: vx, vy, cxy of the
// KLT-projected frames) are quadratic forms of the same matrices, sum_t (u_k^T xs_t)(u_k^T ys_t) = u_k^T (Xs Ys^T) u_k: two 420^3
// products (siib_quad_kernel) replace the two 420 x 420 x n projections, 4.5 x fewer flops, and the 3 GB of stacked frames are never
// written.  vx_k = (ncols - 1) lambda_k comes from the eigenvalue.  Same mathematics, different summation order: scores agree with
// the projection path to 1e-9 relative (tests/test_metrics_gpu.py), both are kept (NELE_SIIB_LAG=0: projections).
#define SL_ND 29                   // lags -14 .. 14
#define SL_CH 58                   // frames per LDS chunk (a multiple of SL_ND: the rotating register window keeps its phase)
// window means of the stacked rows: mu[k][j] = mean_t v_j[t + k], t < ncols.  grid (B, signals), block 256
__global__ __launch_bounds__(256) void siib_mu_kernel(SiibWs ws, int sig0) {
    __shared__ double red[8];
    __shared__ double tot[SB_J];
    const int b = blockIdx.x, sig = sig0 + blockIdx.y, tid = threadIdx.x;
    const int na = ws.info[4 * b + 2], ncols = na - SB_K + 1;
    double* mu = ws.mu + ((size_t)b * 2 + sig) * SB_D;
    if (ncols < 2) { for (int a = tid; a < SB_D; a += 256) mu[a] = 0.0; return; }
    const double* XL = ws.XL + ((size_t)b * 2 + sig) * SB_J * ws.NA;
    const double* rs = ws.rowstat + ((size_t)b * 2 + sig) * SB_J * 2;
    for (int j = 0; j < SB_J; ++j) {
        const double mr = rs[2 * j + 1];
        double sm = 0.0;
        for (int t = tid; t < na; t += 256) sm += XL[(size_t)j * ws.NA + t] - mr;
        const double T = block_sum(sm, red);
        if (tid == 0) tot[j] = T;
    }
    __syncthreads();
    for (int a = tid; a < SB_D; a += 256) {
        const int k = a / SB_J, j = a - k * SB_J;
        const double mr = rs[2 * j + 1];
        const double* row = XL + (size_t)j * ws.NA;
        double edge = 0.0;
        for (int q = 0; q < k; ++q) edge += row[q] - mr;                          // frames before the window
        for (int q = k + ncols; q < na; ++q) edge += row[q] - mr;                 // frames behind it
        mu[a] = (tot[j] - edge) / (double)ncols;
    }
}

// The lag products of an utterance are ALWAYS summed as SL_NSEG = 4 frame segments (boundaries from the utterance's own active-frame count),
// each from zero, and combined as (s0 + s1) + (s2 + s3) by siib_assemble_kernel: how many workgroups share the segments (ws.nseg = 4, 2, 1 by
// batch size) changes who computes a segment, not the association - an utterance's covariance, and so its SIIB score, does not depend on the
// batch it is scored in.  (Until round 5 the segment boundaries followed ws.nseg: 1e-6 relative between batch-size classes.)
#define SL_NSEG 4
// L_ab partials.  grid (4, nseg, B), block 256: thread p = 256 blockIdx.x + tid < 784 owns the band pair (j1, j2) = (p / 28, p % 28)
// and ND consecutive lags D0 .. D0 + ND - 1: b_j2[s + D0 .. s + D0 + ND - 1] lives in a ROTATING register window (slot = frame index
// mod ND; the step loop is unrolled ND times so that every slot index is a compile-time constant), so a frame costs ND multiply-adds,
// one new window element and one a value from LDS ([frame][band] tiles: a wave's read is one 224-byte run).  Frames outside [0, na)
// are staged as zeros: the full-lag sums need no bounds in the loop.  The symmetric matrices (Sxx, Syy) only need the lags 0 .. 14
// (the other half is the mirror image); the cross term needs all 29.
template <int SIGA, int SIGB, int D0, int ND>
__global__ __launch_bounds__(256, (ND > 15 ? 2 : 4)) void siib_lag_kernel(SiibWs ws, int pair) {
    constexpr int CH = (SL_CH / ND) * ND;                    // frames per LDS chunk: whole window periods
    __shared__ double sa[CH][SB_J], sb[CH + ND - 1][SB_J];
    const int b = blockIdx.z, tid = threadIdx.x, p = blockIdx.x * 256 + tid;
    const int na = ws.info[4 * b + 2];
    const int j1 = min(p, SB_J * SB_J - 1) / SB_J, j2 = min(p, SB_J * SB_J - 1) % SB_J;
    const int per = ((na + SL_NSEG - 1) / SL_NSEG + CH - 1) / CH * CH;                   // frames per segment: whole chunks
    const int spg = SL_NSEG / ws.nseg;                                                   // segments this workgroup walks, one after the other
    const double* A = ws.XL + ((size_t)b * 2 + SIGA) * SB_J * ws.NA;
    const double* Bm = ws.XL + ((size_t)b * 2 + SIGB) * SB_J * ws.NA;
    const double* ra = ws.rowstat + ((size_t)b * 2 + SIGA) * SB_J * 2;
    const double* rb = ws.rowstat + ((size_t)b * 2 + SIGB) * SB_J * 2;
    double acc[ND], w[ND];
    // Staging, software-pipelined (third session of round 3): a chunk's multiply-adds take 3 us, its staging loop - two dependent
    // global loads per element, 16 rounds - took 10 (PMC: the waves of this kernel waited 55 % of their time at two workgroups per
    // CU).  The raw values of chunk c + 1 are requested into registers before the multiply-adds of chunk c and stored (minus the row
    // means, which sit in LDS) after them: one barrier pair per chunk as before, the load latency under the arithmetic.
    __shared__ double rmean[2][SB_J];
    if (tid < 2 * SB_J) rmean[tid / SB_J][tid % SB_J] = (tid < SB_J ? ra : rb)[2 * (tid % SB_J) + 1];
    constexpr int NEA = (CH * SB_J + 255) / 256, NEB = ((CH + ND - 1) * SB_J + 255) / 256;
    double pa[NEA], pb[NEB];
    auto fetch = [&](int c0) {                               // thread -> (frame, band), coalesced along frames per band row
#pragma unroll
        for (int i = 0; i < NEA; ++i) {
            const int e = tid + 256 * i, j = e / CH, r = e - j * CH, sidx = c0 + r;
            pa[i] = (e < CH * SB_J && sidx < na) ? A[(size_t)j * ws.NA + sidx] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NEB; ++i) {
            const int e = tid + 256 * i, j = e / (CH + ND - 1), r = e - j * (CH + ND - 1), sidx = c0 + D0 + r;
            pb[i] = (e < (CH + ND - 1) * SB_J && sidx >= 0 && sidx < na) ? Bm[(size_t)j * ws.NA + sidx] : 0.0;
        }
    };
    auto commit = [&](int c0) {                              // a[c0 .. c0 + CH) and b[c0 + D0 .. c0 + CH + D0 + ND - 1), zeros outside [0, na)
#pragma unroll
        for (int i = 0; i < NEA; ++i) {
            const int e = tid + 256 * i, j = e / CH, r = e - j * CH, sidx = c0 + r;
            if (e < CH * SB_J) sa[r][j] = (sidx < na) ? pa[i] - rmean[0][j] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NEB; ++i) {
            const int e = tid + 256 * i, j = e / (CH + ND - 1), r = e - j * (CH + ND - 1), sidx = c0 + D0 + r;
            if (e < (CH + ND - 1) * SB_J) sb[r][j] = (sidx >= 0 && sidx < na) ? pb[i] - rmean[1][j] : 0.0;
        }
    };
    // (only the 29-lag instantiation is pipelined: the 15-lag ones run four workgroups per CU at 98 registers and hide the staging behind
    //  each other - with the prefetch registers they drop to three and take 540 instead of 455 us)
    constexpr bool PIPE = ND > 15;
    for (int seg = blockIdx.y * spg; seg < (int)(blockIdx.y + 1) * spg; ++seg) {
    const int s0 = seg * per, s1 = min(na, s0 + per);
#pragma unroll
    for (int d = 0; d < ND; ++d) { acc[d] = 0.0; w[d] = 0.0; }
    if (PIPE && s0 < s1) fetch(s0);
    for (int c0 = s0; c0 < s1; c0 += CH) {
        __syncthreads();                                    // the previous chunk's reads of sa / sb are done (and rmean is visible)
        if constexpr (PIPE) commit(c0);
        else {                                              // element by element: a[c0 .. c0 + CH) and b[c0 + D0 .. c0 + CH + D0 + ND - 1)
            for (int e = tid; e < CH * SB_J; e += 256) {
                const int j = e / CH, r = e - j * CH, sidx = c0 + r;
                sa[r][j] = (sidx < na) ? A[(size_t)j * ws.NA + sidx] - rmean[0][j] : 0.0;
            }
            for (int e = tid; e < (CH + ND - 1) * SB_J; e += 256) {
                const int j = e / (CH + ND - 1), r = e - j * (CH + ND - 1), sidx = c0 + D0 + r;
                sb[r][j] = (sidx >= 0 && sidx < na) ? Bm[(size_t)j * ws.NA + sidx] - rmean[1][j] : 0.0;
            }
        }
        __syncthreads();
        if (PIPE && c0 + CH < s1) fetch(c0 + CH);
        if (c0 == s0) {                                     // window = b[s0 + D0 .. s0 + D0 + ND - 2] in slots 0 .. ND - 2
#pragma unroll
            for (int q = 0; q < ND - 1; ++q) w[q] = sb[q][j2];
        }
#pragma unroll 1
        for (int r0 = 0; r0 < CH; r0 += ND) {
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                const double a = sa[r0 + u][j1];
                w[(u + ND - 1) % ND] = sb[r0 + u + ND - 1][j2];
#pragma unroll
                for (int d = 0; d < ND; ++d) acc[d] = fma(a, w[(u + d) % ND], acc[d]);
            }
        }
    }
    if (p < SB_J * SB_J) {                                   // slot D + 14 of the 29 lag slots
        double* out = ws.lagp + ((((size_t)b * 3 + pair) * SL_NSEG + seg) * SL_ND + (D0 + 14)) * (SB_J * SB_J) + p;
#pragma unroll
        for (int d = 0; d < ND; ++d) out[(size_t)d * (SB_J * SB_J)] = acc[d];
    }
    }
}

// Matrices from the lag products.  MODE 0: C = Sxx / (ncols - 1) (the covariance the eigensolver takes);  MODE 1: S2[0] = Syy,
// S2[1] = (Sxy + Sxy^T) / 2, unscaled.  grid (15, B), block 256: block D = lag k2 - k1 >= 0; a thread owns the band pair (j1, j2) and
// walks down the diagonal k1 = 0 .. 14 - D with the running sum  P(k1 + 1) = P(k1) - a[k1] b[k2] + a[k1 + ncols] b[k2 + ncols],
// P(0) = L(D) - (the 14 - D products behind the window); the first and last 29 frames of every band row it touches sit in LDS.
// Each value is written to (a1, a2) and (a2, a1): the matrices are exactly symmetric.
template <int MODE>
__global__ __launch_bounds__(256) void siib_assemble_kernel(SiibWs ws) {
    __shared__ double eh[2][SB_J][SL_ND], et[2][SB_J][SL_ND], mus[2][SB_D];
    const int b = blockIdx.y, D = blockIdx.x, tid = threadIdx.x;
    const int na = ws.info[4 * b + 2], ncols = na - SB_K + 1;
    double* C = ws.C + (size_t)b * SB_D * SB_D;
    double* S = MODE ? ws.S2 + (size_t)b * 2 * SB_D * SB_D : nullptr;
    if (ncols < 2) {                                         // not enough active frames (status bit set by the front end): zeros
        for (int e = D * 256 + tid; e < SB_D * SB_D; e += 15 * 256) {
            if (MODE == 0) C[e] = 0.0; else { S[e] = 0.0; S[(size_t)SB_D * SB_D + e] = 0.0; }
        }
        return;
    }
    for (int e = tid; e < 2 * SB_J * SL_ND; e += 256) {
        const int sig = e / (SB_J * SL_ND), r = e - sig * (SB_J * SL_ND), j = r / SL_ND, q = r - j * SL_ND;
        const double* row = ws.XL + (((size_t)b * 2 + sig) * SB_J + j) * ws.NA;
        const double mr = ws.rowstat[(((size_t)b * 2 + sig) * SB_J + j) * 2 + 1];
        eh[sig][j][q] = (q < na) ? row[q] - mr : 0.0;                             // frames 0 .. 28
        et[sig][j][q] = (na - SL_ND + q >= 0) ? row[na - SL_ND + q] - mr : 0.0;   // frames na - 29 .. na - 1
    }
    for (int e = tid; e < 2 * SB_D; e += 256) mus[e / SB_D][e % SB_D] = ws.mu[(size_t)b * 2 * SB_D + e];
    __syncthreads();
    const double nc = (double)ncols, scale = 1.0 / (double)(ncols - 1);
    auto lag = [&](int pair, int ja, int jb, int dd) {       // L_pair(ja, jb, dd) = (s0 + s1) + (s2 + s3), whatever the batch size
        const double* lp = ws.lagp + (((size_t)b * 3 + pair) * SL_NSEG * SL_ND + (dd + 14)) * (SB_J * SB_J) + ja * SB_J + jb;
        const size_t st = (size_t)SL_ND * (SB_J * SB_J);
        return (lp[0] + lp[st]) + (lp[2 * st] + lp[3 * st]);
    };
    for (int p = tid; p < SB_J * SB_J; p += 256) {
        const int j1 = p / SB_J, j2 = p - j1 * SB_J;
        if (D == 0 && j2 > j1) continue;
        // walk(sa, ja, sb, jb): sum_t a_ja[t + k1] b_jb[t + k1 + D] for k1 = 0 .. 14 - D
        const int sA = MODE ? 1 : 0;
        double P = lag(MODE ? 1 : 0, j1, j2, D);             // (x, x) or (y, y)
        for (int i = 0; i < 14 - D; ++i) P -= et[sA][j1][15 + i] * et[sA][j2][15 + i + D];
        double P1 = 0.0, Q = 0.0;
        if (MODE) {
            P1 = lag(2, j1, j2, D);                          // sum_t x_j1[t + k1] y_j2[t + k2]
            Q = lag(2, j2, j1, -D);                          // sum_t x_j2[t + k2] y_j1[t + k1]
            for (int i = 0; i < 14 - D; ++i) {
                P1 -= et[0][j1][15 + i] * et[1][j2][15 + i + D];
                Q -= et[0][j2][15 + D + i] * et[1][j1][15 + i];
            }
        }
        for (int k1 = 0; k1 + D < SB_K; ++k1) {
            const int k2 = k1 + D, a1 = k1 * SB_J + j1, a2 = k2 * SB_J + j2;
            const size_t o12 = (size_t)a1 * SB_D + a2, o21 = (size_t)a2 * SB_D + a1;
            if (MODE == 0) {
                const double v = (P - nc * mus[0][a1] * mus[0][a2]) * scale;
                C[o12] = v; C[o21] = v;
            } else {
                const double vy = P - nc * mus[1][a1] * mus[1][a2];
                const double vxy = 0.5 * ((P1 - nc * mus[0][a1] * mus[1][a2]) + (Q - nc * mus[0][a2] * mus[1][a1]));
                // siib_quad_kernel, the only reader, takes the upper triangle (rows <= columns) and discards the rest by a select: the
                // mirror entries - one row per lane, uncoalesced - are not written (0.72 GB of writes per call at B = 256 -> 0.36 GB)
                // (D > 0: a1 < a2, the coalesced o12; D = 0 runs the pairs j2 <= j1: their upper entry is o21)
                const size_t ou = a1 <= a2 ? o12 : o21;
                S[ou] = vy; S[(size_t)SB_D * SB_D + ou] = vxy;
            }
            if (k2 + 1 >= SB_K) break;
            // the window moves one frame: frames k1 / k2 leave, frames k1 + ncols = na - 14 + k1 / k2 + ncols enter
            P += et[sA][j1][15 + k1] * et[sA][j2][15 + k2] - eh[sA][j1][k1] * eh[sA][j2][k2];
            if (MODE) {
                P1 += et[0][j1][15 + k1] * et[1][j2][15 + k2] - eh[0][j1][k1] * eh[1][j2][k2];
                Q += et[0][j2][15 + k2] * et[1][j1][15 + k1] - eh[0][j2][k2] * eh[1][j1][k1];
            }
        }
    }
}

// Quadratic forms q_k = u_k^T S u_k for S = Syy and sym(Sxy): P = U S on the f64 matrix cores exactly as siib_proj_kernel computes
// U X (U rows = eigenvectors), then the row-wise sums of P[k][j] U[k][j] over the tile's 64 columns.  1-D XCD-aware grid:
// 14 column tiles (7 of Syy, 7 of sym(Sxy)) x 7 row tiles per utterance.
// S is symmetric (siib_assemble_kernel stores its upper triangle only), so q_k = 2 sum_{i<j} u_i S_ij u_j + sum_i S_ii u_i^2:
// the product only runs over the rows i <= j of a column tile (S staged as its strict upper triangle + half the diagonal, the row sums
// doubled) - 60 % of the multiply-adds of the full product.
__global__ __launch_bounds__(256) void siib_quad_kernel(SiibWs ws) {
    __shared__ __attribute__((aligned(32))) double Us[2][16][SG_LD], Mt[2][16][SG_LD];
    const int xw = (int)blockIdx.x, slot = xw >> 3, per_b = 7 * 14;
    const int b = (slot / per_b) * 8 + (xw & 7), tl = slot % per_b, by = tl % 7, bx = tl / 7;
    if (b >= ws.Bn) return;
    const int which = bx / 7, ti = by * 64, t0 = (bx % 7) * 64, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const double* U = ws.U + (size_t)b * SB_D * SB_D;
    const double* M = ws.S2 + ((size_t)b * 2 + which) * SB_D * SB_D;
    const int ur = tid >> 2, uq = (tid & 3) * 4;                 // U tile: 64 rows x 16 k
    const double* pu = U + (size_t)min(ti + ur, SB_D - 1) * SB_D + uq;
    const int xc = tid >> 4, xq = (tid & 15) * 4;                // M tile: 16 k x 64 columns
    const bool cin = t0 + xq < SB_D;                             // 420 = 4 * 105: column quads are all-in or all-out
    const double* pm = M + (size_t)xc * SB_D + (cin ? t0 + xq : 0);
    f64x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const double4 z4 = make_double4(0.0, 0.0, 0.0, 0.0);
    const int kend = min(SB_D, t0 + 64);                         // rows i of S beyond the tile's last column contribute nothing
    auto gload = [&](int k0, double4& ru, double4& rm) {
        ru = (k0 + uq < SB_D) ? *reinterpret_cast<const double4*>(pu + k0) : z4;
        rm = (cin && k0 + xc < SB_D) ? *reinterpret_cast<const double4*>(pm + (size_t)k0 * SB_D) : z4;
        if (k0 + 15 >= t0) {                                     // the chunk reaches the diagonal block: keep i < j, halve i == j
            const int i = k0 + xc, j = t0 + xq;
            rm.x = i < j ? rm.x : i == j ? 0.5 * rm.x : 0.0;
            rm.y = i < j + 1 ? rm.y : i == j + 1 ? 0.5 * rm.y : 0.0;
            rm.z = i < j + 2 ? rm.z : i == j + 2 ? 0.5 * rm.z : 0.0;
            rm.w = i < j + 3 ? rm.w : i == j + 3 ? 0.5 * rm.w : 0.0;
        }
    };
    double4 ru, rm;
    gload(0, ru, rm);
    for (int k0 = 0, it = 0; k0 < kend; k0 += 16, ++it) {
        const int buf = it & 1;
        Us[buf][uq][ur] = ru.x; Us[buf][uq + 1][ur] = ru.y; Us[buf][uq + 2][ur] = ru.z; Us[buf][uq + 3][ur] = ru.w;
        *reinterpret_cast<double4*>(&Mt[buf][xc][xq]) = rm;
        __syncthreads();
        if (k0 + 16 < kend) gload(k0 + 16, ru, rm);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = 4 * kk + lk;
            const double a = Us[buf][k][16 * w + li];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Mt[buf][k][16 * j + li], acc[j], 0, 0, 0);
        }
    }
    // acc[j][q] = P[row ti + 16 w + lk + 4 q][column t0 + 16 j + li]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int gi = ti + 16 * w + lk + 4 * q;
        double sm = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = t0 + 16 * j + li;
            sm += (gi < SB_D && c < SB_D) ? acc[j][q] * U[(size_t)gi * SB_D + c] : 0.0;
        }
        sm = row16_sum_dpp(sm);
        if (li == 0 && gi < SB_D) ws.qpart[((size_t)b * SB_D + gi) * 14 + bx] = 2.0 * sm;
    }
}

// score from the eigenvalues and the quadratic forms (lag path); same decisions as siib_final_kernel
__global__ __launch_bounds__(512) void siib_final_lag_kernel(SiibWs ws, float* __restrict__ raw, float* __restrict__ mapped, const int* __restrict__ eigflag) {
    __shared__ double red[8];
    const int b = blockIdx.x, j = threadIdx.x;
    int* info = ws.info + 4 * b;
    const double* lam = ws.lam + (size_t)b * SB_D;
    const int ncols = info[2] - SB_K + 1;
    double lmax = -1e300;
    if (j < SB_D) lmax = lam[j];
    lmax = block_max(lmax, red);
    double I = 0.0;
    if (j < SB_D && lam[j] > 1e-10 * lmax) {
        const double* q = ws.qpart + ((size_t)b * SB_D + j) * 14;
        double vy = 0.0, cxy = 0.0;
        for (int t = 0; t < 7; ++t) { vy += q[t]; cxy += q[7 + t]; }
        const double vx = lam[j] * (double)(ncols - 1);
        const double rho = cxy / sqrt(vx * vy);
        I = -0.5 * log2(1.0 - 0.5625 * rho * rho);
    }
    const double tot = block_sum(I, red);
    if (j == 0) {
        double v = fmax(0.0, 80.0 / 15.0 * tot);
        if (info[3] & (8 | 16)) v = nan("");
        if (raw) raw[b] = (float)v;
        if (mapped) mapped[b] = (float)(1.0 / (1.0 + exp(-0.06 * (v - 32.0))));
        // status bit 32: this utterance's covariance went through the eigensolver's repair path (its cluster tridiagonalisation gave up:
        // the score is as accurate as ever, the step was slower) - counted by the training loop without a synchronisation
        if (eigflag && eigflag[b] != 0) info[3] |= 32;
    }
}

// s9: one block (512 threads >= 420) per utterance
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(512) void siib_final_kernel(SiibWs ws, float* __restrict__ raw, float* __restrict__ mapped) {
    __shared__ double red[8];
    const int b = blockIdx.x, j = threadIdx.x;
    const int* info = ws.info + 4 * b;
    const double* lam = ws.lam + (size_t)b * SB_D;
    double lmax = -1e300;
    if (j < SB_D) lmax = lam[j];
    lmax = block_max(lmax, red);
    double I = 0.0;
    if (j < SB_D && lam[j] > 1e-10 * lmax) {
        double vx = 0, vy = 0, cxy = 0;
        const double* p = ws.part + ((size_t)b * SB_D + j) * ws.NTL * 3;
        for (int t = 0; t < ws.NTL; ++t) { vx += p[3 * t]; vy += p[3 * t + 1]; cxy += p[3 * t + 2]; }
        const double rho = cxy / sqrt(vx * vy);
        I = -0.5 * log2(1.0 - 0.5625 * rho * rho);
    }
    const double tot = block_sum(I, red);
    if (j == 0) {
        double v = fmax(0.0, 80.0 / 15.0 * tot);
        if (info[3] & (8 | 16)) v = nan("");  // reference raises: not enough active frames (16: internal inconsistency)
        if (raw) raw[b] = (float)v;
        if (mapped) mapped[b] = (float)(1.0 / (1.0 + exp(-0.06 * (v - 32.0))));
    }
}
#endif  // NELE_AB

// ------------------------------------------------------------------------------------------ C ABI
static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

static void siib_dims(int L, int* NT, int* NA, int* NTL) {
    long long total = (long long)SB_MMAX * L;
    if (total < SB_WLEN + 1) total = SB_WLEN + 1;
    *NT = (int)((total - SB_WLEN + SB_SHIFT - 1) / SB_SHIFT) + 8;
    const int n1 = (int)((((long long)L < SB_WLEN + 1 ? SB_WLEN + 1 : L) - SB_WLEN + SB_SHIFT - 1) / SB_SHIFT);
    // active frames: <= 25 s * 80 fps (+ one base period of slack) when tiled, <= n1 otherwise
    int na = 2000 + n1 + 2 * SB_MMAX;
    if (na < n1) na = n1;
    na = (na + 63) / 64 * 64;
    *NA = na;
    *NTL = na / 64;
}

// NELE_SIIB_LAG=0: covariance and projections from the stacked frames (the round-2 kernels; A/B switch)
static bool siib_lag_path() {
    const bool on = NELE_SWITCH_INT("NELE_SIIB_LAG", 1) != 0;
    return on;
}

static size_t siib_layout(int B, int L, SiibWs* w, char* base) {
    int NT, NA, NTL;
    siib_dims(L, &NT, &NA, &NTL);
    size_t o = 0;
#define TAKE(field, type, count) do { if (w) w->field = (type*)(base + o); o += al(sizeof(type) * (size_t)(count)); } while (0)
    TAKE(g2, double, SB_J * SB_NBIN);
    TAKE(tab, double, 3 * SB_WLEN);
    TAKE(g2t, double, SB_NBIN * SB_J);
    TAKE(rowstat, double, (size_t)B * 2 * SB_J * 2);
    TAKE(xdb, double, (size_t)B * NT);
    TAKE(list, int, (size_t)B * NA);
    TAKE(info, int, (size_t)B * 4);
    TAKE(nprim, int, (size_t)B);
    TAKE(XL, double, (size_t)B * 2 * SB_J * NA);
    TAKE(C, double, (size_t)B * SB_D * SB_D);
    TAKE(lam, double, (size_t)B * SB_D);
    TAKE(U, double, (size_t)B * SB_D * SB_D);
    TAKE(eigws, char, (size_t)nele_eigh_workspace_bytes(B, SB_D));
    TAKE(part, double, (size_t)B * SB_D * NTL * 3);
    // Frame segments of the lag-product kernel: more workgroups at small batches.  NOTE (documented non-invariance): the partial sums
    // of an utterance are added in segment order, so the association of its covariance sums - and with it the last bits of its score -
    // depends on the batch-size class (B < 48 / < 128 / >= 128) it is scored in; the products and their order inside a segment do not.
    // Folding the segments inside one workgroup (the same association for every B) costs the 29-lag kernel 58 more registers on top
    // of 256 + 20 spilled; always writing four partials costs 0.84 GB of traffic per B = 256 call.  tests/test_metrics_gpu.py compares
    // B = 1, 50 and 130 (equal at the float32 precision of the returned score).
    const int nseg = B >= 128 ? 1 : (B >= 48 ? 2 : 4);
    if (siib_lag_path()) {
        // the lag path needs neither the stacked frames nor the kept clean projections: Xs / px shrink to nothing
        TAKE(lagp, double, (size_t)B * 3 * SL_NSEG * SL_ND * SB_J * SB_J);
        TAKE(mu, double, (size_t)B * 2 * SB_D);
        TAKE(S2, double, (size_t)B * 2 * SB_D * SB_D);
        TAKE(qpart, double, (size_t)B * SB_D * 14);
        if (w) { w->Xs = nullptr; w->px = nullptr; }
    } else {
        TAKE(Xs, double, (size_t)B * 2 * SB_D * NA);
        TAKE(px, double, (size_t)B * 7 * NTL * 16 * 256);
        if (w) { w->lagp = nullptr; w->mu = nullptr; w->S2 = nullptr; w->qpart = nullptr; }
    }
#undef TAKE
    if (w) { w->NT = NT; w->NA = NA; w->NTL = NTL; w->Bn = B; w->nseg = nseg; }
    return o;
}

extern "C" long long nele_metric_siib_workspace_bytes(int B, int L) { return (long long)siib_layout(B, L, nullptr, nullptr); }

// What phase 3 (the clean-signal half) leaves in a workspace for phase 4, as byte ranges of the workspace: out[3 k .. 3 k + 2] =
// {offset, stride, bytes}.  stride > 0: a per-utterance section - utterance b's state is the first `bytes` bytes at offset + b * stride;
// stride == 0: a table every utterance shares (gammatone responses, DFT / window tables).  The clean-signal state of an utterance - the
// active-frame list, the clean log spectra and row statistics, the window means, the KLT eigenvalues and eigenvectors, the eigensolver's
// repair flag - is a pure function of the clean waveform (and of the padded length L, which fixes the strides): a training loop that
// scores the same clean files every epoch (train_nele.py:35-38,119,318-340) may copy these ranges out after phase 3 and, in a later call
// with the same L, copy them into the workspace INSTEAD of running phase 3 (utterances may sit at other rows b, B may differ).
// Returns the number of sections (<= max_sections), or a negative status.
extern "C" int nele_metric_siib_clean_sections(int B, int L, long long* out, int max_sections) {
    NELE_CHECK_ARG(B > 0 && out && max_sections >= 13, "nele_metric_siib_clean_sections: bad arguments (13 sections)");
    if (!siib_lag_path()) return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_metric_siib_clean_sections: lag-product path only");
    SiibWs w;
    char* base = reinterpret_cast<char*>((uintptr_t)1 << 20);
    siib_layout(B, L, &w, base);
    const int* fl = nele_eigh_flags(w.eigws, B, SB_D);
    int k = 0;
    auto put = [&](const void* p, long long stride, long long bytes) {
        out[3 * k] = (long long)(reinterpret_cast<const char*>(p) - base); out[3 * k + 1] = stride; out[3 * k + 2] = bytes; ++k;
    };
    put(w.g2, 0, sizeof(double) * SB_J * SB_NBIN);
    put(w.tab, 0, sizeof(double) * 3 * SB_WLEN);
    put(w.g2t, 0, sizeof(double) * SB_NBIN * SB_J);
    put(w.rowstat, sizeof(double) * 2 * SB_J * 2, sizeof(double) * 2 * SB_J * 2);
    put(w.xdb, sizeof(double) * (size_t)w.NT, sizeof(double) * (size_t)w.NT);
    put(w.list, sizeof(int) * (size_t)w.NA, sizeof(int) * (size_t)w.NA);
    put(w.info, sizeof(int) * 4, sizeof(int) * 4);
    put(w.nprim, sizeof(int), sizeof(int));
    put(w.XL, sizeof(double) * 2 * SB_J * (size_t)w.NA, sizeof(double) * SB_J * (size_t)w.NA);        // the clean half of [2][28][NA]
    put(w.lam, sizeof(double) * SB_D, sizeof(double) * SB_D);
    put(w.U, sizeof(double) * SB_D * SB_D, sizeof(double) * SB_D * SB_D);
    put(w.mu, sizeof(double) * 2 * SB_D, sizeof(double) * SB_D);                                       // the clean half of [2][420]
    put(fl, sizeof(int), sizeof(int));
    return k;
}

// phase: 0 = everything, 1 = front only (VAD .. covariance), 2 = back only (eigenvectors .. score): the split lets the caller put
// independent work between the wide front kernels and the latency-bound eigen-decomposition.
// 3 = everything that depends on the CLEAN signal only (VAD, active-frame list, x spectra / masking / stacking, covariance,
// eigen-decomposition; y is not read and may be null), 4 = the rest (y spectra / masking / stacking, projections, score) on the
// same workspace: the Karhunen-Loeve basis of SIIB is that of the clean signal, so phase 3 can run before the degraded signal
// exists (GanTrainer runs it beside the G-step).
extern "C" int nele_metric_siib_var(const float* x, const float* y, const int* lengths, int B, int L, void* workspace, long long workspace_bytes,
                                    float* raw, float* mapped, int* info_out, int phase, void* stream) {
    NELE_CHECK_ARG(x && workspace && (raw || mapped) && B > 0, "nele_metric_siib: bad arguments");
    NELE_CHECK_ARG(phase >= 0 && phase <= 4, "nele_metric_siib: phase must be 0..4");
    NELE_CHECK_ARG(y || phase == 3, "nele_metric_siib: degraded signal missing");
    if (L < SB_WLEN + SB_SHIFT * (SB_K + 1)) return nele_set_error(NELE_ERR_SIGNAL, "nele_metric_siib: L=%d too short", L);
    if (workspace_bytes < nele_metric_siib_workspace_bytes(B, L))
        return nele_set_error(NELE_ERR_WORKSPACE, "nele_metric_siib: workspace too small");
    SiibWs ws;
    const size_t used = siib_layout(B, L, &ws, (char*)workspace);
    (void)used;
    ws.lens = lengths;
    hipStream_t s = as_stream(stream);
    int g = L, r = SB_SHIFT;
    while (r) { const int t = g % r; g = r; r = t; }
    const bool dedup = NELE_SWITCH_INT("NELE_SIIB_DEDUP", 1) != 0;
    // frame period of the tiled signal (NELE_SIIB_DEDUP=0: A/B switch, every frame computed; per-utterance lengths: every row has its
    // own period, the shortcut is not taken)
    const int Pf = (dedup && !lengths) ? L / g : 0x7fffffff;
    const int n1 = (int)(((long long)L - SB_WLEN + SB_SHIFT - 1) / SB_SHIFT);   // frames of the un-tiled signal (L > 400 checked above)
    const bool vad = (phase == 0 || phase == 1 || phase == 3);
    const bool sx = vad, sy = (phase == 0 || phase == 1 || phase == 4);
    const bool eig = (phase == 0 || phase == 2 || phase == 3);
    const bool fin = (phase == 0 || phase == 2 || phase == 4);
    if (vad) {
        hipLaunchKernelGGL(siib_g2_kernel, dim3(SB_J), dim3(256), 0, s, ws.g2, ws.tab, ws.g2t);
        hipLaunchKernelGGL(siib_db_kernel, dim3((n1 + 3) / 4, B), dim3(256), 0, s, x, L, ws, 0, Pf);
        hipLaunchKernelGGL(siib_m_kernel, dim3(B), dim3(256), 0, s, L, ws);
        {
            const int last = Pf < ws.NT ? Pf : ws.NT;            // tiled frames the base pass and the period do not give
            const int first = lengths ? 0 : n1;                  // shorter rows have fewer base frames: the grid starts at each row's own count
            if (last > first) hipLaunchKernelGGL(siib_db_kernel, dim3((last - first + 3) / 4, B), dim3(256), 0, s, x, L, ws, 1, Pf);
        }
        hipLaunchKernelGGL(siib_compact_kernel, dim3(B), dim3(256), 0, s, ws, Pf);
    }
    if (sx || sy) {
        const int sig0 = sx ? 0 : 1, sig1 = sy ? 1 : 0, nsig = sig1 - sig0 + 1;
        const bool specw = NELE_SWITCH_INT("NELE_SIIB_SPECW", 1) != 0;
        if (specw)
            hipLaunchKernelGGL(siib_spec_wave_kernel, dim3((ws.NA + SP_F * SPW_NG - 1) / (SP_F * SPW_NG), B), dim3(128), 0, s, x, y, L, ws, sig0, sig1);
        else {
            NELE_AB_ONLY(hipLaunchKernelGGL(siib_spec_kernel, dim3((ws.NA + SP_F - 1) / SP_F, B), dim3(128), 0, s, x, y, L, ws, sig0, sig1);)
        }
        hipLaunchKernelGGL(siib_spread_kernel, dim3((ws.NA + 255) / 256, B), dim3(256), 0, s, ws, Pf, sig0, sig1);
        hipLaunchKernelGGL(siib_rowmin_kernel, dim3(SB_J, B, nsig), dim3(256), 0, s, ws, sig0);
        hipLaunchKernelGGL(siib_mask_kernel, dim3(B, nsig), dim3(64), 0, s, ws, sig0);
        NELE_AB_ONLY(NELE_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(siib_stack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));)
        if (siib_lag_path()) {
            hipLaunchKernelGGL(siib_mu_kernel, dim3(B, nsig), dim3(256), 0, s, ws, sig0);
            const dim3 lg(4, ws.nseg, B);
            if (sx) {
                hipLaunchKernelGGL((siib_lag_kernel<0, 0, 0, 15>), lg, dim3(256), 0, s, ws, 0);
                hipLaunchKernelGGL(siib_assemble_kernel<0>, dim3(15, B), dim3(256), 0, s, ws);
            }
            if (sy) {
                hipLaunchKernelGGL((siib_lag_kernel<1, 1, 0, 15>), lg, dim3(256), 0, s, ws, 1);
                hipLaunchKernelGGL((siib_lag_kernel<0, 1, -14, 29>), lg, dim3(256), 0, s, ws, 2);
                hipLaunchKernelGGL(siib_assemble_kernel<1>, dim3(15, B), dim3(256), 0, s, ws);
            }
        } else {
            NELE_AB_ONLY(NELE_CHECK_ARG((size_t)ws.NA * sizeof(double) <= 152 * 1024, "nele_metric_siib: signal too long (%d active frames)", ws.NA);
                         hipLaunchKernelGGL(siib_stack_kernel, dim3(SB_J, B, nsig), dim3(256), sizeof(double) * (size_t)ws.NA, s, ws, sig0);
                         if (sx) hipLaunchKernelGGL(siib_cov_kernel, dim3(49 * 8 * ((B + 7) / 8)), dim3(256), 0, s, ws);)
        }
        NELE_CHECK_LAUNCH("nele_metric_siib(front)");
    }
    if (eig) {
        // phase 3 runs beside the G-step: 32 matrices per cluster launch leave half of the chip to it (eigh.hip)
        int st = nele_eigh_sym_batched_ex(ws.C, SB_D, B, ws.lam, ws.U, ws.eigws, nele_eigh_workspace_bytes(B, SB_D), stream, (phase == 3) ? 32 : 64);
        if (st) return st;
        if (phase == 3 && !siib_lag_path()) {               // clean-signal half of the projections, beside whatever the caller overlaps
            NELE_AB_ONLY(hipLaunchKernelGGL(siib_proj_kernel<1>, dim3(7 * ws.NTL * 8 * ((B + 7) / 8)), dim3(256), 0, s, ws);)
            NELE_CHECK_LAUNCH("nele_metric_siib(clean projections)");
        }
    }
    if (fin && siib_lag_path()) {
        hipLaunchKernelGGL(siib_quad_kernel, dim3(7 * 14 * 8 * ((B + 7) / 8)), dim3(256), 0, s, ws);
        hipLaunchKernelGGL(siib_final_lag_kernel, dim3(B), dim3(512), 0, s, ws, raw, mapped, nele_eigh_flags(ws.eigws, B, SB_D));
        if (info_out) (void)hipMemcpyAsync(info_out, ws.info, sizeof(int) * 4 * (size_t)B, hipMemcpyDeviceToDevice, s);
        NELE_CHECK_LAUNCH("nele_metric_siib(back, lag path)");
    } else if (fin) {
        NELE_AB_ONLY(if (phase == 4) hipLaunchKernelGGL(siib_proj_kernel<2>, dim3(7 * ws.NTL * 8 * ((B + 7) / 8)), dim3(256), 0, s, ws);
                     else hipLaunchKernelGGL(siib_proj_kernel<0>, dim3(7 * ws.NTL * 8 * ((B + 7) / 8)), dim3(256), 0, s, ws);
                     hipLaunchKernelGGL(siib_final_kernel, dim3(B), dim3(512), 0, s, ws, raw, mapped);)
        if (info_out) (void)hipMemcpyAsync(info_out, ws.info, sizeof(int) * 4 * (size_t)B, hipMemcpyDeviceToDevice, s);
        NELE_CHECK_LAUNCH("nele_metric_siib(back)");
    }
    return NELE_OK;
}

extern "C" int nele_metric_siib_phase(const float* x, const float* y, int B, int L, void* workspace, long long workspace_bytes, float* raw,
                                      float* mapped, int* info_out, int phase, void* stream) {
    return nele_metric_siib_var(x, y, nullptr, B, L, workspace, workspace_bytes, raw, mapped, info_out, phase, stream);
}

extern "C" int nele_metric_siib(const float* x, const float* y, int B, int L, void* workspace, long long workspace_bytes, float* raw,
                                float* mapped, int* info_out, void* stream) {
    return nele_metric_siib_var(x, y, nullptr, B, L, workspace, workspace_bytes, raw, mapped, info_out, 0, stream);
}
