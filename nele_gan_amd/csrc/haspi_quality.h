// HASPI version 1 (`haspi`, pyhaspi2.py:109-157) and HASQI v2 (`hasqi_v2`, pyhaspi2.py:32-74): the two remaining entry points of the
// reference's pyHASPI module, normal-hearing case HL = 0 (for which the intelligibility ear model, itype 0, and the
// quality one, itype 2, coincide: pyhaspi2.py:1162-1165, :1176).  Included at the end of haspi.hip: the ear model up to the dB-SL
// envelope is haspi_chain() itself, run with the BM switch; from there
//   q1  hq_ihc_bm_kernel   IHC adaptation pass 2 (pyhaspi2.py:1028-1078) writing the adapted envelope AND the basilar-membrane motion
//   q2  hq_segment_kernel  eb_EnvSmooth (:674-703): 16 ms raised-cosine segments, 50 % overlap, after the group-delay shift, and
//   q3                     eb_BMcovary (:550-657): windowed, mean-removed BM segments, cross-covariance over |lag| <= 24, mean squares
//   q4  hq_loud_kernel     segment loudness (band average of 10^(dB/20), back in dB) for the silence gates of eb_melcor / the covariance stages
//   q5  hq_final_kernel    eb_melcor (:706-751), eb_3LevelCovary (:416-547), eb_AveCovary2 (:160-220), eb_aveSL (:1135-1152), eb_SpectDiff
//                          (:222-251) and the two score formulas, one block per utterance
// The BM motion that reaches eb_BMcovary is bm gain_c gain_SL gain_IHC (:997, :1087, :1076) with gain_SL = (sl + 1e-30) / (c + 1e-30) and
// gain_IHC = (out + 1e-30) / (sl + 1e-30), c = gain_c |u| the compressed envelope: the product is (out + 1e-30) cos(phase) c / (c + 1e-30)
// with cos(phase) = bm / |u|.  The filter-bank pass stores cos(phase) (float32, like the envelopes: see hp_env_t) and q1 multiplies it by
// the adapted envelope; c / (c + 1e-30) differs from 1 by less than 1e-16 unless |u| < 1e-14, i.e. digital silence, where bm = 0 anyway.
// eb_BMaddnoise (:1091-1095) adds N(0, 10^((-10 - 65)/20)) = 1.8e-4 rms to every BM sample from numpy's global generator; here a
// counter-based generator (seed argument) does, or none (noise = 0: deterministic, what the parity tests compare with the oracle).
#define HQ_NWIN 384
#define HQ_NHALF 192
#define HQ_MAXLAG 24
#define HQ_NOUT 12
#define HQ_MAXBINS 2048

struct QualWs {
    double* sm;      // [B][2][nseg][32]  smoothed dB-SL envelopes
    double* cov;     // [B][nseg][32]     segment cross-covariance, clipped to [0, 1]
    double* msx;     // [B][nseg][32]     2 x mean square of the reference BM segments
    double* msy;     // [B][nseg][32]
    double* corr;    // [2][64]           1 / xcorr(window, window, 24) for the full and the half window
    double* segl;    // [B][2][nseg]      segment loudness: of the smoothed reference envelope, of sqrt(msx)
    int* qinfo;      // [B][4]            {segments above threshold (eb_melcor), status bits, above threshold (covariance), histogram bins}
    int nseg;        // of the longest row
    unsigned long long seed;
    int noise;
};

__device__ __forceinline__ int hq_nseg(int n24) { return 1 + n24 / HQ_NWIN + (n24 - HQ_NHALF) / HQ_NWIN; }
__device__ __forceinline__ double hq_win(int k) { return 0.5 - 0.5 * cospi(2.0 * (double)k / (double)(HQ_NWIN - 1)); }   // np.hanning(384)[k]

// Standard normal from a counter: two splitmix64 outputs -> Box-Muller in float32 (the noise is 1.8e-4 rms on signals of order 1..100)
__device__ __forceinline__ unsigned long long hq_mix(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float hq_gauss(unsigned long long seed, unsigned long long idx) {
    const unsigned long long r = hq_mix(seed ^ hq_mix(idx));
    const float u1 = ((float)(unsigned)(r >> 40) + 1.0f) * (1.0f / 16777216.0f);          // (0, 1]
    const float u2 = (float)(unsigned)((r >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);      // [0, 1)
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

// ---- q1: grid (ceil(chunks / 8), rows), block 256 = 8 chunks of GL_N samples x 32 channels.  In place: env <- adapted envelope,
// cphi <- BM motion.
__global__ __launch_bounds__(256) void hq_ihc_bm_kernel(HaspiWs ws, QualWs q, int sig0, int nsig) {
    const int tid = threadIdx.x, ch = tid & 31, row = hp_row(blockIdx.y, sig0, nsig);
    const int chunk = blockIdx.x * 8 + (tid >> 5), n0 = chunk * ws.lcg;
    const int n24 = hp_n24(ws, row >> 1);
    if (n0 >= n24) return;
    const int n1 = min(n0 + ws.lcg, (n24 + GL_U - 1) / GL_U * GL_U);      // whole groups: the buffers are padded to n24p (multiple of 32)
    const int ncg = (ws.n24p + ws.lcg - 1) / ws.lcg;
    const IhcC k = hp_ihc_consts();
    const double* ihe = ws.ihe + ((size_t)row * ncg + chunk) * 64 + ch;
    double V1 = ihe[0], V2 = ihe[32];
    hp_env_t* e = ws.env + ((size_t)row * ws.n24p) * HP_NCH + ch;
    float* c = ws.cphi + ((size_t)row * ws.n24p) * HP_NCH + ch;
    const float gn = q.noise ? 1.7782794100389227e-4f : 0.f;          // 10^((-10 - 65) / 20)
    const unsigned long long base = ((unsigned long long)row * HP_NCH + ch) * (unsigned long long)ws.n24p;
    for (int nb = n0; nb < n1; nb += GL_U) {
        float ev[GL_U], cv[GL_U];
#pragma unroll
        for (int u = 0; u < GL_U; ++u) { ev[u] = e[(size_t)(nb + u) * HP_NCH]; cv[u] = c[(size_t)(nb + u) * HP_NCH]; }
#pragma unroll
        for (int u = 0; u < GL_U; ++u) {
            const double V0 = (double)ev[u];
            hp_ihc_step(k, V0, V1, V2);
            double out = (V0 - V1) * k.R1inv;
            out = out < 0.0 ? 0.0 : out;
            ev[u] = (float)out;
            float bm = (float)((out + 1.0e-30) * (double)cv[u]);
            if (q.noise) bm += gn * hq_gauss(q.seed, base + (unsigned long long)(nb + u));
            cv[u] = bm;
        }
#pragma unroll
        for (int u = 0; u < GL_U; ++u) { e[(size_t)(nb + u) * HP_NCH] = ev[u]; c[(size_t)(nb + u) * HP_NCH] = cv[u]; }
    }
}

// Segment n of a row with nseg segments: samples [st, st + len), window taps from woff (pyhaspi2.py:689-699, :578-640)
__device__ __forceinline__ void hq_segment(int seg, int nseg, int& st, int& len, int& woff) {
    st = seg * HQ_NHALF;
    if (seg == 0) { len = HQ_NHALF; woff = HQ_NHALF; }
    else if (seg == nseg - 1) { len = HQ_NHALF; woff = 0; }
    else { len = HQ_NWIN; woff = 0; }
}

// 1 / xcorr(w, w, 24) of the full window and of its second half (the literal tables of pyhaspi2.py:563, :570).  grid 1, block 128
__global__ void hq_corr_kernel(QualWs q) {
    const int t = threadIdx.x, half = t >> 6, i = t & 63;
    if (i > 2 * HQ_MAXLAG) return;
    const int l = i - HQ_MAXLAG, N = half ? HQ_NHALF : HQ_NWIN, off = half ? HQ_NHALF : 0;
    double s = 0.0;
    for (int n = 0; n < N; ++n) {
        const int m = n + l;
        if (m >= 0 && m < N) s += hq_win(off + m) * hq_win(off + n);
    }
    q.corr[half * 64 + i] = 1.0 / s;
}

// ---- q2 + q3: one block per (segment, group of 8 channels, utterance): eb_EnvSmooth of both envelopes and eb_BMcovary.
// grid (nseg, 4, B), block 128: thread = (k16 = tid >> 3, c = tid & 7).
// The group-delay compensation (pyhaspi2.py:1098-1131, :1239-1242) delays channel ch by shift[ch] samples, up to 433 between the lowest
// and the highest band: a lane that reads its own shifted column straight from the [sample][32] arrays touches a different cache
// line than every other lane (the first version did: 50 ms for the smoothing alone at B = 256).  Here the rows [st - max shift, st + len -
// min shift) of the group's 8 columns are loaded row by row (32 bytes = two float4 lanes per row) into an LDS tile, and the lanes
// pick their shifted columns from there.  The four arrays (envelope and BM motion of x and y) go through the same tile one after
// the other.  The windowed, mean-removed BM segments are kept in LDS as float32 (they were float32 in memory), all sums are float64.
// Lag loop: thread (lag group lg < 13, c) accumulates the 4 consecutive lags -24 + 4 lg ... with a sliding register window over x:
// 2 LDS reads per 4 multiply-adds.
#define HQ_XR (HQ_NWIN + 2 * HQ_MAXLAG + 4)
#define HQ_TROWS 640                 // tile rows: 384 + the largest shift spread inside a group of 8 neighbouring bands that still uses the tile
__global__ __launch_bounds__(128) void hq_segment_kernel(HaspiWs ws, QualWs q) {
    __shared__ __attribute__((aligned(16))) float tile[HQ_TROWS][8];
    __shared__ float xs[HQ_XR][8];
    __shared__ float ys[HQ_NWIN][8];
    __shared__ double wtab[HQ_NWIN];
    __shared__ double part[16][8][2];
    __shared__ double mx[16][8];
    const int tid = threadIdx.x, c = tid & 7, k16 = tid >> 3, b = blockIdx.z, ch0 = blockIdx.y * 8, ch = ch0 + c;
    const int n24 = hp_n24(ws, b), nseg = hq_nseg(n24), seg = blockIdx.x;
    if (seg >= nseg) return;
    int st, N, woff;
    hq_segment(seg, nseg, st, N, woff);
    for (int k = tid; k < N; k += 128) wtab[k] = hq_win(woff + k);
    const int sh = ws.shift[(size_t)b * HP_NCH + ch];
    int smin = sh, smax = sh;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { smin = min(smin, __shfl_xor(smin, o, 64)); smax = max(smax, __shfl_xor(smax, o, 64)); }
    const int base = st - smax;                              // first row of the tile (may be negative: zeros, as the reference prepends)
    const int rows = min(N + smax - smin, HQ_TROWS);
    const int roff = smax - sh;                              // the lane's segment starts at tile row roff
    for (int r = k16; r < HQ_XR; r += 16) xs[r][c] = 0.f;
    double res[2] = {0.0, 0.0};
    double MSx = 0.0, MSy = 0.0;
    // the tile rows of array a + 1 are loaded into registers before array a is processed (one global round trip per array and workgroup was
    // most of this kernel's time: 12 barriers and 4 exposed load latencies for 61 KB of input)
    constexpr int HQ_NLD = HQ_TROWS / 64;                    // float4 per thread and array
    float4 pre[HQ_NLD];
    auto srcof = [&](int a) { return ((a < 2) ? (const float*)ws.env : ws.cphi) + ((size_t)(2 * b + (a & 1)) * ws.n24p) * HP_NCH + ch0; };
    auto gload = [&](int a) {
        const float* src = srcof(a);
#pragma unroll
        for (int u = 0; u < HQ_NLD; ++u) {
            const int r = (tid >> 1) + 64 * u, m = base + r;
            pre[u] = (r < rows && m >= 0 && m < n24) ? *reinterpret_cast<const float4*>(src + (size_t)m * HP_NCH + 4 * (tid & 1)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    gload(0);
    for (int a = 0; a < 4; ++a) {                            // 0, 1: envelopes of x, y; 2, 3: BM motion of x, y
        const float* src = srcof(a);
        __syncthreads();                                     // the previous array's readers are done with the tile
#pragma unroll
        for (int u = 0; u < HQ_NLD; ++u) {
            const int r = (tid >> 1) + 64 * u;
            if (r < rows) *reinterpret_cast<float4*>(&tile[r][4 * (tid & 1)]) = pre[u];
        }
        if (a + 1 < 4) gload(a + 1);
        __syncthreads();
        double s = 0.0;
        if (a < 2) {
            for (int k = k16; k < N; k += 16) {
                const int r = roff + k, m = base + r;
                const float v = r < HQ_TROWS ? tile[r][c] : ((m >= 0 && m < n24) ? src[(size_t)m * HP_NCH + c] : 0.f);
                s += (double)v * wtab[k];
            }
            part[k16][c][0] = s;
            __syncthreads();
            if (k16 == 0) {
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < 16; ++j) t += part[j][c][0];
                // sum(np.hanning(384)) = 191.5, of one half 95.75
                q.sm[((size_t)(2 * b + a) * q.nseg + seg) * HP_NCH + ch] = t / (N == HQ_NWIN ? 191.5 : 95.75);
            }
        } else {
            float* dst = (a == 2) ? &xs[HQ_MAXLAG][0] : &ys[0][0];
            for (int k = k16; k < N; k += 16) {
                const int r = roff + k, m = base + r;
                const float v = r < HQ_TROWS ? tile[r][c] : ((m >= 0 && m < n24) ? src[(size_t)m * HP_NCH + c] : 0.f);
                s += (double)v * wtab[k];
            }
            part[k16][c][0] = s;
            __syncthreads();
            double mean = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) mean += part[j][c][0];
            mean /= (double)N;
            double s2 = 0.0;
            for (int k = k16; k < N; k += 16) {
                const int r = roff + k, m = base + r;
                const float v = r < HQ_TROWS ? tile[r][c] : ((m >= 0 && m < n24) ? src[(size_t)m * HP_NCH + c] : 0.f);
                const float d = (float)((double)v * wtab[k] - mean);
                dst[k * 8 + c] = d;
                s2 += (double)d * (double)d;
            }
            part[k16][c][1] = s2;
            __syncthreads();
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) t += part[j][c][1];
            if (a == 2) MSx = t; else MSy = t;
        }
    }
    __syncthreads();
    double best = 0.0;
    if (k16 < 13) {
        const int l0 = -HQ_MAXLAG + 4 * k16;                 // lags l0 .. l0 + 3; x row of (n, l) = n + l + 24
        const double* corr = q.corr + (N == HQ_NWIN ? 0 : 64);
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        double x0 = (double)xs[l0 + HQ_MAXLAG][c], x1 = (double)xs[l0 + HQ_MAXLAG + 1][c], x2 = (double)xs[l0 + HQ_MAXLAG + 2][c];
#pragma unroll 4
        for (int n = 0; n < N; ++n) {
            const double x3 = (double)xs[n + l0 + HQ_MAXLAG + 3][c], y = (double)ys[n][c];
            a0 += x0 * y; a1 += x1 * y; a2 += x2 * y; a3 += x3 * y;
            x0 = x1; x1 = x2; x2 = x3;
        }
        const double av[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (l0 + j <= HQ_MAXLAG) best = fmax(best, fabs(av[j] * corr[l0 + j + HQ_MAXLAG]));
    }
    mx[k16][c] = best;
    __syncthreads();
    if (k16 == 0) {
        double Mxy = 0.0;
#pragma unroll
        for (int j = 0; j < 13; ++j) Mxy = fmax(Mxy, mx[j][c]);
        const double norm2 = 1.0 / (N == HQ_NWIN ? 143.625 : 71.8125);     // sum(np.hanning(384)^2) = 3 (N - 1) / 8; the second half holds half of it
        MSx *= norm2; MSy *= norm2;
        double cv = (MSx > 1.0e-30 && MSy > 1.0e-30) ? Mxy / sqrt(MSx * MSy) : 0.0;
        cv = fmin(fmax(cv, 0.0), 1.0);
        const size_t o = ((size_t)b * q.nseg + seg) * HP_NCH + ch;
        q.cov[o] = cv; q.msx[o] = 2.0 * MSx; q.msy[o] = 2.0 * MSy;
    }
}

// ---- q4: grid (ceil(nseg / 64), B), block 64: thread = segment
__global__ void hq_loud_kernel(HaspiWs ws, QualWs q) {
    const int b = blockIdx.y, seg = blockIdx.x * 64 + threadIdx.x;
    const int nseg = hq_nseg(hp_n24(ws, b));
    if (seg >= nseg) return;
    const double* sm = q.sm + ((size_t)(2 * b) * q.nseg + seg) * HP_NCH;
    const double* ms = q.msx + ((size_t)b * q.nseg + seg) * HP_NCH;
    const double LN10_20 = 0.11512925464970228;              // ln(10) / 20
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < HP_NCH; ++c) { s1 += exp(sm[c] * LN10_20); s2 += exp(sqrt(ms[c]) * LN10_20); }
    q.segl[((size_t)b * 2 + 0) * q.nseg + seg] = 20.0 * log10(s1 / (double)HP_NCH);
    q.segl[((size_t)b * 2 + 1) * q.nseg + seg] = 20.0 * log10(s2 / (double)HP_NCH);
}

// Sum over the block's 256 threads, returned to every thread
__device__ __forceinline__ double hq_block_sum(double v, double* sh4) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}

// ---- q5: grid B, block 256.  out [B][HQ_NOUT] = {HASPI v1 Intel, CepCorr, cov3 low / mid / high, HASQI Combined, Nonlin, Linear, BMsync5,
// Dloud, Dslope, avecov}.  Where the reference raises ('Signal below threshold', pyhaspi2.py:723-724, :427-428; eb_AveCovary2's (0, 0)
// return makes hasqi_v2 fail at syncov[4]) the values are NaN and the status bits say which: 1 eb_melcor, 2 the covariance stages.
__global__ __launch_bounds__(256) void hq_final_kernel(HaspiWs ws, QualWs q, double alpha, double* __restrict__ out) {
    __shared__ double cepm[HP_NCH][HP_NBASIS];
    __shared__ double sh4[4];
    __shared__ double accs[8][HP_NCH][8];
    __shared__ int hist[HQ_MAXBINS];
    __shared__ double edges[2];
    __shared__ double SL[2][HP_NCH];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n24 = hp_n24(ws, b), nseg = hq_nseg(n24);
    const double thr = 2.5, small = 1.0e-30, NaN = nan("");
    int status = 0;
    // ---------------- eb_melcor
    if (tid < HP_NCH * HP_NBASIS) {
        const int c = tid / HP_NBASIS, k = tid % HP_NBASIS;
        cepm[c][k] = cospi((double)k * (double)c / (double)(HP_NCH - 1));
    }
    __syncthreads();
    if (tid < HP_NBASIS) {
        double s = 0.0;
        for (int c = 0; c < HP_NCH; ++c) s += cepm[c][tid] * cepm[c][tid];
        s = 1.0 / sqrt(s);
        for (int c = 0; c < HP_NCH; ++c) cepm[c][tid] *= s;
    }
    __syncthreads();
    const double* lm = q.segl + ((size_t)b * 2 + 0) * q.nseg;
    const double* lc = q.segl + ((size_t)b * 2 + 1) * q.nseg;
    const double* smx = q.sm + ((size_t)(2 * b) * q.nseg) * HP_NCH;
    const double* smy = q.sm + ((size_t)(2 * b + 1) * q.nseg) * HP_NCH;
    double mean[2][HP_NBASIS];
    double cnt = 0.0;
    {
        double a[2][HP_NBASIS] = {};
        for (int seg = tid; seg < nseg; seg += 256) {
            if (!(lm[seg] > thr)) continue;
            cnt += 1.0;
            for (int c = 0; c < HP_NCH; ++c) {
                const double vx = smx[(size_t)seg * HP_NCH + c], vy = smy[(size_t)seg * HP_NCH + c];
#pragma unroll
                for (int k = 0; k < HP_NBASIS; ++k) { a[0][k] += vx * cepm[c][k]; a[1][k] += vy * cepm[c][k]; }
            }
        }
        cnt = hq_block_sum(cnt, sh4);
        for (int k = 0; k < HP_NBASIS; ++k) { mean[0][k] = hq_block_sum(a[0][k], sh4) / cnt; mean[1][k] = hq_block_sum(a[1][k], sh4) / cnt; }
    }
    const int n_mel = (int)cnt;
    double CepCorr = NaN;
    {
        double sxx[HP_NBASIS] = {}, syy[HP_NBASIS] = {}, sxy[HP_NBASIS] = {};
        for (int seg = tid; seg < nseg; seg += 256) {
            if (!(lm[seg] > thr)) continue;
            double cx[HP_NBASIS] = {}, cy[HP_NBASIS] = {};
            for (int c = 0; c < HP_NCH; ++c) {
                const double vx = smx[(size_t)seg * HP_NCH + c], vy = smy[(size_t)seg * HP_NCH + c];
#pragma unroll
                for (int k = 0; k < HP_NBASIS; ++k) { cx[k] += vx * cepm[c][k]; cy[k] += vy * cepm[c][k]; }
            }
#pragma unroll
            for (int k = 0; k < HP_NBASIS; ++k) {
                const double dx = cx[k] - mean[0][k], dy = cy[k] - mean[1][k];
                sxx[k] += dx * dx; syy[k] += dy * dy; sxy[k] += dx * dy;
            }
        }
        double m1 = 0.0;
        for (int k = 0; k < HP_NBASIS; ++k) {
            const double xx = hq_block_sum(sxx[k], sh4), yy = hq_block_sum(syy[k], sh4), xy = hq_block_sum(sxy[k], sh4);
            const double r = (xx < small || yy < small) ? 0.0 : fabs(xy / sqrt(xx * yy));
            if (k >= 1) m1 += r;
        }
        if (n_mel <= 1) status |= 1; else CepCorr = m1 / (double)(HP_NBASIS - 1);
    }
    // ---------------- segments above threshold for the covariance stages; 0.5 dB histogram of their loudness
    double lo = 1e300, hi = -1e300, ncov = 0.0;
    for (int seg = tid; seg < nseg; seg += 256) {
        const double v = lc[seg];
        if (v > thr) { lo = fmin(lo, v); hi = fmax(hi, v); ncov += 1.0; }
    }
    ncov = hq_block_sum(ncov, sh4);
    for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
    __syncthreads();
    if ((tid & 63) == 0) { accs[0][0][tid >> 6] = lo; accs[0][1][tid >> 6] = hi; }
    __syncthreads();
    lo = fmin(fmin(accs[0][0][0], accs[0][0][1]), fmin(accs[0][0][2], accs[0][0][3]));
    hi = fmax(fmax(accs[0][1][0], accs[0][1][1]), fmax(accs[0][1][2], accs[0][1][3]));
    __syncthreads();
    const int n_cov = (int)ncov;
    int nbins = 0;
    double cov3[3] = {NaN, NaN, NaN}, avecov = NaN, sync5 = NaN;
    if (n_cov <= 1) {
        status |= 2;
    } else {
        nbins = (int)ceil(((hi + 0.5) - lo) / 0.5);                     // len(np.arange(dBmin, dBmax + 0.5, 0.5))
        if (nbins > HQ_MAXBINS) { status |= 4; nbins = HQ_MAXBINS; }
        for (int i = tid; i < nbins; i += 256) hist[i] = 0;
        __syncthreads();
        // np.histogram over the mid-points between the bin centres lo + 0.5 i: bin i = [edge(i), edge(i + 1)), open at both ends
        for (int seg = tid; seg < nseg; seg += 256) {
            const double v = lc[seg];
            if (!(v > thr)) continue;
            int i = (int)floor((v - lo) / 0.5 + 0.5);
            i = max(0, min(i, nbins - 1));
            while (i < nbins - 1 && v >= ((lo + 0.5 * (double)i) + (lo + 0.5 * (double)(i + 1))) / 2.0) ++i;
            while (i > 0 && v < ((lo + 0.5 * (double)(i - 1)) + (lo + 0.5 * (double)i)) / 2.0) --i;
            atomicAdd(&hist[i], 1);
        }
        __syncthreads();
        if (tid == 0) {                                                  // pyhaspi2.py:459-472
            double e0 = 0.0, e1 = 0.0, cum = 0.0;
            for (int n = 0; n < nbins; ++n) {
                cum += (double)hist[n];
                const double xc = cum / ncov;
                if (xc < 0.333) e0 = lo + 0.5 * (double)n;
                if (xc < 0.667) e1 = lo + 0.5 * (double)n;
            }
            edges[0] = e0; edges[1] = e1;
        }
        __syncthreads();
        const double e0 = edges[0], e1 = edges[1];
        // per band: sums of weight * covariance and of the weights, by loudness third
        const int sg = tid >> 5, ch = tid & 31;
        double S[3] = {0, 0, 0}, W[3] = {0, 0, 0};
        for (int seg = sg; seg < nseg; seg += 8) {
            const double v = lc[seg];
            if (!(v > thr)) continue;
            const int g = v < e0 ? 0 : (v < e1 ? 1 : 2);
            const size_t o = ((size_t)b * q.nseg + seg) * HP_NCH + ch;
            const bool w = sqrt(q.msx[o]) > thr;
            const double cv = q.cov[o];
#pragma unroll
            for (int j = 0; j < 3; ++j) { S[j] += (w && g == j) ? cv : 0.0; W[j] += (w && g == j) ? 1.0 : 0.0; }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) { accs[sg][ch][j] = S[j]; accs[sg][ch][3 + j] = W[j]; }
        __syncthreads();
        if (tid < 64) {                                                  // wave 0; lanes 0..31 hold the bands
            double St[3] = {0, 0, 0}, Wt[3] = {0, 0, 0};
            if (tid < HP_NCH) {
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int j = 0; j < 3; ++j) { St[j] += accs[s][tid][j]; Wt[j] += accs[s][tid][3 + j]; }
            }
            for (int j = 0; j < 3; ++j) {                                // pyhaspi2.py:486-545
                double ave = (tid < HP_NCH && Wt[j] != 0.0) ? St[j] / Wt[j] : 0.0;
                double nc = (tid < HP_NCH && Wt[j] != 0.0) ? 1.0 : 0.0;
                for (int o = 32; o > 0; o >>= 1) { ave += __shfl_xor(ave, o, 64); nc += __shfl_xor(nc, o, 64); }
                cov3[j] = ave / nc;                                      // 0 / 0 = NaN where the reference's division warns and yields nan
            }
            // eb_AveCovary2: all above-threshold cells; synchrony weighting 5: fcut 3.5 kHz, order p = 5
            double C = (St[0] + St[1]) + St[2], Wa = (Wt[0] + Wt[1]) + Wt[2];
            double fs5 = 0.0;
            if (tid < HP_NCH) {
                const double fc = pow(3500.0, 10.0), f = pow(hp_cfreq(tid), 10.0);
                fs5 = sqrt(fc / (fc + f));
            }
            double fC = fs5 * C, fW = fs5 * Wa;
            for (int o = 32; o > 0; o >>= 1) {
                C += __shfl_xor(C, o, 64); Wa += __shfl_xor(Wa, o, 64); fC += __shfl_xor(fC, o, 64); fW += __shfl_xor(fW, o, 64);
            }
            avecov = Wa < 1.0 ? 0.0 : C / Wa;
            sync5 = fC / fW;
        }
    }
    // ---------------- eb_aveSL + eb_SpectDiff (wave 0)
    double Dloud = NaN, Dslope = NaN;
    if (tid < 64) {
        const int ch = tid & 31, sig = tid >> 5, row = 2 * b + sig;
        const int nchk = (n24 + ws.lc - 1) / ws.lc;
        double se = 0.0, sc = 0.0;
        for (int c = 0; c < nchk; ++c) {
            se += ws.sse[((size_t)row * ws.nchunk + c) * HP_NCH + ch];
            sc += ws.ssp[((size_t)row * ws.nchunk + c) * HP_NCH + ch];
        }
        const double cf = hp_cfreq(ch);
        const double sg_ = hp_gt(ws.bw[(size_t)row * HP_NCH + ch], cf).gain, cg_ = hp_gt(hp_bw1(ch), cf).gain;
        const double ave = sg_ * sqrt(se / (double)n24), cave = cg_ * sqrt(sc / (double)n24);
        const HpLoss lo = hp_loss(ws, sig, ch);                          // eb_aveSL (pyhaspi2.py:1135-1152)
        double le = HP_LEVEL + 20.0 * log10(fmax(cave, small));
        le = fmin(fmax(le, lo.lowknee), 100.0);
        const double gain = -lo.attnOHC - (le - lo.lowknee) * (1.0 - (1.0 / lo.CR));
        double ls = HP_LEVEL + 20.0 * log10(fmax(ave, small));
        ls = fmax(ls, 0.0);
        SL[sig][ch] = fmax(ls + gain - lo.attnIHC, 0.0);
    }
    __syncthreads();
    if (tid < 64) {
        const int ch = tid & 31;                                         // both half-waves compute the same
        const double LN10_20 = 0.11512925464970228;
        double x = exp(SL[0][ch] * LN10_20), y = exp(SL[1][ch] * LN10_20);
        double sx = x, sy = y;
        for (int o = 16; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); }
        x /= sx; y /= sy;
        // dloud[1] = nbands * np.std(x - y)
        const double d = x - y;
        double m = d;
        for (int o = 16; o > 0; o >>= 1) m += __shfl_xor(m, o, 64);
        m /= (double)HP_NCH;
        double v = (d - m) * (d - m);
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        const double dl = (double)HP_NCH * sqrt(v / (double)HP_NCH);
        // dslope[1] = nbands * np.std((x[1:] - x[:-1]) - (y[1:] - y[:-1])): 31 differences
        const double xn = __shfl(x, (tid & 32) + min(ch + 1, HP_NCH - 1), 64), yn = __shfl(y, (tid & 32) + min(ch + 1, HP_NCH - 1), 64);
        const double ds = (ch < HP_NCH - 1) ? (xn - x) - (yn - y) : 0.0;
        double ms_ = ds;
        for (int o = 16; o > 0; o >>= 1) ms_ += __shfl_xor(ms_, o, 64);
        ms_ /= (double)(HP_NCH - 1);
        double vs = (ch < HP_NCH - 1) ? (ds - ms_) * (ds - ms_) : 0.0;
        for (int o = 16; o > 0; o >>= 1) vs += __shfl_xor(vs, o, 64);
        const double dsl = (double)HP_NCH * sqrt(vs / (double)(HP_NCH - 1));
        Dloud = fmin(fmax(1.0 - dl / 2.5, 0.0), 1.0);
        Dslope = fmin(fmax(1.0 - dsl, 0.0), 1.0);
    }
    if (tid == 0) {
        double* o = out + (size_t)b * HQ_NOUT;
        const double arg = -9.047 + 14.816 * CepCorr + ((0.0 * cov3[0] + 0.0 * cov3[1]) + 4.616 * cov3[2]);     // pyhaspi2.py:147-150
        o[0] = 1.0 / (1.0 + exp(alpha * arg));
        o[1] = CepCorr; o[2] = cov3[0]; o[3] = cov3[1]; o[4] = cov3[2];
        const double Nonlin = (CepCorr * CepCorr) * sync5, Linear = 0.579 * Dloud + 0.421 * Dslope;                 // pyhaspi2.py:69-71
        o[5] = Nonlin * Linear; o[6] = Nonlin; o[7] = Linear; o[8] = sync5; o[9] = Dloud; o[10] = Dslope; o[11] = avecov;
        int* qi = q.qinfo + 4 * b;
        qi[0] = n_mel; qi[1] = status; qi[2] = n_cov; qi[3] = nbins;
    }
}

static size_t quality_layout(int B, int L, int fs_in, HaspiWs* w, QualWs* q, char* base) {
    size_t o = haspi_layout(B, L, fs_in, w, base);
    const int n24 = hp_n24_of(L, fs_in);
    const int n24p = (n24 + 31) / 32 * 32;
    const int nseg = 1 + n24 / HQ_NWIN + (n24 - HQ_NHALF) / HQ_NWIN;
    int lc = GS_LC;
    while ((n24p + lc - 1) / lc > GS_MAXC) lc += GS_LC;
    const int nchunk = (n24p + lc - 1) / lc;
#define TAKEQ(ptr, type, count) do { if (w) ptr = (type*)(base + o); o += al(sizeof(type) * (size_t)(count)); } while (0)
    TAKEQ(w->cphi, float, (size_t)B * 2 * n24p * HP_NCH);
    TAKEQ(w->sse, double, (size_t)B * 2 * nchunk * HP_NCH);
    TAKEQ(q->sm, double, (size_t)B * 2 * nseg * HP_NCH);
    TAKEQ(q->cov, double, (size_t)B * nseg * HP_NCH);
    TAKEQ(q->msx, double, (size_t)B * nseg * HP_NCH);
    TAKEQ(q->msy, double, (size_t)B * nseg * HP_NCH);
    TAKEQ(q->corr, double, 128);
    TAKEQ(q->segl, double, (size_t)B * 2 * nseg);
    TAKEQ(q->qinfo, int, (size_t)B * 4);
#undef TAKEQ
    if (q) q->nseg = nseg;
    return o;
}

extern "C" long long nele_metric_haspi_quality_workspace_bytes(int B, int L, int fs_in) {
    return (long long)quality_layout(B, L, fs_in, nullptr, nullptr, nullptr);
}

// x, y [B][L] float32 (reference, processed); lengths [B] or NULL; noise != 0: eb_BMaddnoise with the counter-based generator seeded
// by `seed`; alpha: the logistic slope of `haspi` (pyhaspi2.py:109, default -1).  out [B][12] float64 (see hq_final_kernel),
// info_out [B][4] int or NULL.
static int haspi_quality_impl(const float* x, const float* y, const int* lengths, int B, int L, int fs_in, int noise, unsigned long long seed, double alpha,
                              const double* hl6, int itype, void* workspace, long long workspace_bytes, double* out, int* info_out, void* stream);

extern "C" int nele_metric_haspi_quality(const float* x, const float* y, const int* lengths, int B, int L, int fs_in, int noise,
                                         unsigned long long seed, double alpha, void* workspace, long long workspace_bytes, double* out,
                                         int* info_out, void* stream) {
    return haspi_quality_impl(x, y, lengths, B, L, fs_in, noise, seed, alpha, nullptr, 0, workspace, workspace_bytes, out, info_out, stream);
}

// The same for a hearing-impaired listener: hl6 (HOST pointer, 6 doubles) = audiogram at 250 .. 6000 Hz; itype 0 = `haspi` (reference
// signal heard with normal hearing: columns 0-4 of `out` are that call's results), itype 2 = `hasqi_v2` (both signals with the loss:
// columns 5-11).  With a loss the two models differ in the reference's ear model (pyhaspi2.py:1162-1166), so one call serves one of them.
extern "C" int nele_metric_haspi_quality_hl(const float* x, const float* y, const int* lengths, int B, int L, int fs_in, int noise,
                                            unsigned long long seed, double alpha, const double* hl6_host, int itype, void* workspace,
                                            long long workspace_bytes, double* out, int* info_out, void* stream) {
    return haspi_quality_impl(x, y, lengths, B, L, fs_in, noise, seed, alpha, hl6_host, itype, workspace, workspace_bytes, out, info_out, stream);
}

static int haspi_quality_impl(const float* x, const float* y, const int* lengths, int B, int L, int fs_in, int noise, unsigned long long seed, double alpha,
                              const double* hl6, int itype, void* workspace, long long workspace_bytes, double* out, int* info_out, void* stream) {
    NELE_CHECK_ARG(x && y && out && workspace && B > 0, "nele_metric_haspi_quality: bad arguments");
    HpHL hl;
    { const int st_ = haspi_hl_table(hl6, itype, &hl); if (st_) return st_; }
    NELE_CHECK_ARG(fs_in >= 1000 && fs_in <= 24000, "nele_metric_haspi_quality: fs must be in [1000, 24000] Hz (got %d; the reference has no downsampler)", fs_in);
    if (L < 2400) return nele_set_error(NELE_ERR_SIGNAL, "nele_metric_haspi_quality: L=%d too short", L);
    if (workspace_bytes < nele_metric_haspi_quality_workspace_bytes(B, L, fs_in))
        return nele_set_error(NELE_ERR_WORKSPACE, "nele_metric_haspi_quality: workspace too small");
    HaspiWs ws;
    QualWs q;
    quality_layout(B, L, fs_in, &ws, &q, (char*)workspace);
    ws.lens = lengths;
    q.seed = seed; q.noise = noise;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(haspi_loss_kernel, dim3(1), dim3(64), 0, s, ws, hl);
    if (fs_in != 24000) haspi_build_window(ws, s);
    haspi_chain(x, y, B, L, fs_in, ws, 0, 2, s, true);       // ear model of both signals up to the IHC prefix states
    const int rows = 2 * B;
    hipLaunchKernelGGL(hq_ihc_bm_kernel, dim3(((ws.n24p + GL_N - 1) / GL_N + 7) / 8, rows), dim3(256), 0, s, ws, q, 0, 2);
    hipLaunchKernelGGL(hq_corr_kernel, dim3(1), dim3(128), 0, s, q);
    hipLaunchKernelGGL(hq_segment_kernel, dim3(q.nseg, 4, B), dim3(128), 0, s, ws, q);
    hipLaunchKernelGGL(hq_loud_kernel, dim3((q.nseg + 63) / 64, B), dim3(64), 0, s, ws, q);
    hipLaunchKernelGGL(hq_final_kernel, dim3(B), dim3(256), 0, s, ws, q, alpha, out);
    if (info_out) (void)hipMemcpyAsync(info_out, q.qinfo, sizeof(int) * 4 * (size_t)B, hipMemcpyDeviceToDevice, s);
    NELE_CHECK_LAUNCH("nele_metric_haspi_quality");
    return NELE_OK;
}
