// Batched symmetric eigen-decomposition (float64) for the SIIB Karhunen-Loeve transform (oracle/siib.py:
// np.linalg.eigh(cov)); replaces the rocSOLVER stop-gap.  n <= 512 (SIIB: n = 420), B matrices, row-major.
//
//   e1  Householder tridiagonalisation, one 1024-thread workgroup per matrix (LAPACK dsytd2 recurrences;
//       matrix-vector product wave-per-row, rank-2 update thread-per-element; reflectors kept in the rows of A)
//   e2  eigenvalues of the tridiagonal matrix by Sturm-sequence bisection, one thread per eigenvalue
//       (LAPACK dstebz recurrences: embarrassingly parallel, every eigenvalue to ~1 ulp of ||T||)
//   e3  eigenvectors of the tridiagonal matrix by inverse iteration, one thread per eigenvector
//       (LAPACK dlagtf / dlagts / dstein recurrences; work arrays laid out [step][thread] so that the
//       threads of a wave touch one coalesced row per step)
//   e4  back-transformation with the Householder reflectors, a slab of 30 eigenvectors per workgroup held in LDS
//
// Close eigenvalues are separated by dstein's perturbation (10 eps |lambda|); vectors are NOT re-orthogonalised
// against each other: for spectra without (numerically) repeated eigenvalues inverse iteration already delivers
// eigenvectors orthogonal to ~eps ||T|| / gap, and for repeated eigenvalues the individual vectors are arbitrary in any
// solver (SIIB drops the only systematic such cluster, the rank-deficient null space, by its eigenvalue tolerance).
#include "common.h"
#include <cstdlib>

#define EG_MAXN 512
#define EG_XCH 8        // tagged exchange slot blocks per matrix: 2 parities x up to 4 kinds (the symmetric first stage uses all)

typedef double f64x4_t __attribute__((ext_vector_type(4)));
struct EighWs {
    double* d;      // [B][n]
    double* e;      // [B][n]
    double* tau;    // [B][n]
    double* lamp;   // [B][n] perturbed eigenvalues used as shifts
    double* zt;     // [B][n][EG_MAXN]  zt[i][j] = component i of eigenvector j (of T, then of A)
    double* lu;     // [B][6][(n+2)/2][EG_MAXN][2]
    int* pin;       // [B][n][EG_MAXN]
    uint4* xch;     // [B][2][2][EG_MAXN] cluster tridiagonalisation: tagged exchange slots (zeroed per call)
    int* flag;      // [B + 1] cluster tridiagonalisation gave up on matrix b (its workgroups were not co-resident within the spin limit):
                    //         the single-workgroup kernels skip it and eigh_tridiag_repair_kernel redoes it; [B] = matrices repaired (zeroed per call)
};

// ------------------------------------------------------------------------------------------ e1
// Step k: Householder vector v_k of column k, p = tau A22 v, w = p - (tau/2)(p.v) v, A22 -= v w^T + w v^T.
// The trailing matrix is streamed ONCE per step: while a wave writes back the updated row i it also accumulates
// that row's dot product with the NEXT Householder vector (known as soon as the first updated row is), which is the
// next step's matrix-vector product.  Per-CU L2 bandwidth bounds this kernel (sum_k 2 m_k^2 * 8 B per matrix).
__device__ __forceinline__ void eigh_house(double* v, int m, double* red, double* s_tau, double* s_scale, double* s_beta) {
    // v[0..m) holds x; on exit v = Householder vector (v[0] = 1), *s_tau = tau, *s_beta = beta
    double ss = 0.0;
    for (int i = 1 + threadIdx.x; i < m; i += blockDim.x) ss += v[i] * v[i];
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) {
        const double alpha = v[0];
        double t = 0.0, beta = alpha, sc = 0.0;
        if (ss > 0.0) {
            const double nrm = sqrt(alpha * alpha + ss);
            beta = alpha >= 0.0 ? -nrm : nrm;
            t = (beta - alpha) / beta;
            sc = 1.0 / (alpha - beta);
        }
        *s_tau = t; *s_scale = sc; *s_beta = beta;
    }
    __syncthreads();
    const double sc = *s_scale;
    for (int i = threadIdx.x; i < m; i += blockDim.x) v[i] = (i == 0) ? 1.0 : v[i] * sc;
    __syncthreads();
}

// Six workgroup barriers per step: (1,2) reduce p.v, (3,4) reduce ||x'||^2 of the next pivot row, (5) publish the next
// Householder vector, (6) end of the fused pass.  Vectors ping-pong between two LDS slots; the two row groups of the pass
// leave their partial matrix-vector products in separate arrays that the next step adds on the fly.
// NT threads (the stand-alone kernels run it with a workgroup of a thousand and twenty-four, the repair path inside
// eigh_tridiag_midx_kernel with 512).  sh: EG_TRI_SH doubles of LDS scratch.
#define EG_TRI_SH (5 * EG_MAXN + 24)
template <int NT>
__device__ __forceinline__ void eigh_tridiag_one_t(double* __restrict__ A, int n, int lda, double* __restrict__ d, double* __restrict__ e,
                                                   double* __restrict__ tau, double* __restrict__ sh) {
    constexpr int HALF = NT / 2;                           // threads per row group of the fused pass
    double (*vb)[EG_MAXN] = reinterpret_cast<double (*)[EG_MAXN]>(sh);
    double* wv = sh + 2 * EG_MAXN;
    double* pa = sh + 3 * EG_MAXN;
    double* pb2 = sh + 4 * EG_MAXN;
    double* red = sh + 5 * EG_MAXN;
    double& s_tau = sh[5 * EG_MAXN + 16];
    double& s_scale = sh[5 * EG_MAXN + 17];
    double& s_beta = sh[5 * EG_MAXN + 18];
    const int tid = threadIdx.x;
    // step 0: v from row 0, p = A22 v by a plain pass (two row groups -> pa, pb2)
    {
        const int m = n - 1;
        double* v = vb[0];
        for (int i = tid; i < m; i += NT) v[i] = A[1 + i];
        __syncthreads();
        eigh_house(v, m, red, &s_tau, &s_scale, &s_beta);
        double* A22 = A + (size_t)lda + 1;
        const int grp = tid / HALF;
        for (int c = tid % HALF; c < m; c += HALF) {
            double acc = 0.0;
            for (int r = grp; r < m; r += 2) acc += A22[(size_t)r * lda + c] * v[r];
            (grp ? pb2 : pa)[c] = acc;
        }
        __syncthreads();
    }
    double tk = s_tau, betak = s_beta;     // every thread carries tau_k / beta_k in registers
    for (int k = 0; k < n - 1; ++k) {
        const int m = n - k - 1, cur = k & 1;
        double* v = vb[cur];
        double* vn = vb[cur ^ 1];
        double* rowk = A + (size_t)k * lda + k + 1;
        double* A22 = A + (size_t)(k + 1) * lda + (k + 1);
        if (tid == 0) { d[k] = A[(size_t)k * lda + k]; e[k] = betak; tau[k] = tk; }
        // (1,2) p.v with p = tau * (pa + pb2)
        double pv = 0.0;
        for (int i = tid; i < m; i += NT) {
            const double vi = v[i];
            rowk[i] = vi;                                  // keep the reflector in row k
            pv += tk * (pa[i] + pb2[i]) * vi;
        }
        pv = block_sum(pv, red);
        const double al = -0.5 * tk * pv;
        const double w0 = tk * (pa[0] + pb2[0]) + al * v[0];
        if (m == 1) {                                      // last step: 1x1 trailing block
            if (tid == 0) A22[0] -= 2.0 * v[0] * w0;
            __syncthreads();
            break;
        }
        // w, first updated row -> x' (next pivot row), ||x'[1:]||^2
        const int m2 = m - 1;
        const double v0 = v[0];
        double ss = 0.0;
        for (int i = tid; i < m; i += NT) {
            const double wi = tk * (pa[i] + pb2[i]) + al * v[i];
            wv[i] = wi;
            if (i >= 1) {
                const double x = A22[i] - v0 * wi - w0 * v[i];
                vn[i - 1] = x;
                if (i >= 2) ss += x * x;
            }
        }
        if (tid == 0) A22[0] -= 2.0 * v0 * w0;
        ss = block_sum(ss, red);                           // (3,4); also publishes wv and the raw vn
        double tn_ = 0.0, betan = vn[0], scn = 0.0;
        {
            const double alpha = vn[0];
            if (ss > 0.0) {
                const double nrm = sqrt(alpha * alpha + ss);
                betan = alpha >= 0.0 ? -nrm : nrm;
                tn_ = (betan - alpha) / betan;
                scn = 1.0 / (alpha - betan);
            }
        }
        __syncthreads();                                   // everyone has read vn[0] before it is overwritten
        for (int i = tid; i < m2; i += NT) vn[i] = (i == 0) ? 1.0 : vn[i] * scn;
        __syncthreads();                                   // (5)
        // fused pass: thread = column c, two row groups; x = A22[r][c] - v_r w_c - w_r v_c; acc += x * vn[r-1]
        for (int c = tid % HALF; c < m; c += HALF) {
            const int grp = tid / HALF;
            double acc = 0.0;
            if (c >= 1) {
                const double vc = v[c], wc = wv[c];
                if (tk != 0.0) {
                    int r = 1 + grp;
                    for (; r + 6 < m; r += 8) {
                        double x0 = A22[(size_t)r * lda + c], x1 = A22[(size_t)(r + 2) * lda + c], x2 = A22[(size_t)(r + 4) * lda + c],
                               x3 = A22[(size_t)(r + 6) * lda + c];
                        x0 -= v[r] * wc + wv[r] * vc; x1 -= v[r + 2] * wc + wv[r + 2] * vc;
                        x2 -= v[r + 4] * wc + wv[r + 4] * vc; x3 -= v[r + 6] * wc + wv[r + 6] * vc;
                        A22[(size_t)r * lda + c] = x0; A22[(size_t)(r + 2) * lda + c] = x1; A22[(size_t)(r + 4) * lda + c] = x2;
                        A22[(size_t)(r + 6) * lda + c] = x3;
                        acc += x0 * vn[r - 1] + x1 * vn[r + 1] + x2 * vn[r + 3] + x3 * vn[r + 5];
                    }
                    for (; r < m; r += 2) {
                        double x0 = A22[(size_t)r * lda + c];
                        x0 -= v[r] * wc + wv[r] * vc;
                        A22[(size_t)r * lda + c] = x0;
                        acc += x0 * vn[r - 1];
                    }
                } else {
                    for (int r = 1 + grp; r < m; r += 2) acc += A22[(size_t)r * lda + c] * vn[r - 1];
                }
                (grp ? pb2 : pa)[c - 1] = acc;
            }
        }
        __syncthreads();                                   // (6)
        tk = tn_;
        betak = betan;
    }
    if (tid == 0) { d[n - 1] = A[(size_t)(n - 1) * lda + (n - 1)]; e[n - 1] = 0.0; tau[n - 1] = 0.0; }
}
__device__ __forceinline__ void eigh_tridiag_one(double* __restrict__ A, int n, int lda, double* __restrict__ d, double* __restrict__ e,
                                                 double* __restrict__ tau) {
    __shared__ double tri_sh[EG_TRI_SH];
    eigh_tridiag_one_t<1024>(A, n, lda, d, e, tau, tri_sh);
}

__global__ __launch_bounds__(1024) void eigh_tridiag_kernel(double* __restrict__ Aall, int n, EighWs ws) {
    const int b = blockIdx.x;
    eigh_tridiag_one(Aall + (size_t)b * n * n, n, n, ws.d + (size_t)b * n, ws.e + (size_t)b * n, ws.tau + (size_t)b * n);
}

// Repair of the matrices the cluster kernels gave up on (ws.flag[b] != 0: some of a matrix's workgroups were not scheduled within the
// spin limit - another process on the GPU, a debugger, a launch larger than the chip).  The cluster kernels keep the trailing matrix in
// registers and only ever write reflector rows into the strict UPPER triangle of A before their hand-over, which a matrix that gave up
// never reaches: its lower triangle and diagonal still hold the original matrix.  One workgroup mirrors them back and runs the
// memory-streaming single-workgroup tridiagonalisation (slow: ~3 ms for one 420 x 420 matrix; it never runs on a GPU the launch fits on).
// grid B, block 1024; unflagged matrices return at once.
template <int NT>
__device__ __forceinline__ void eigh_repair1(double* __restrict__ A, int n, EighWs ws, int b, int B, double* __restrict__ sh) {
    for (int idx = threadIdx.x; idx < n * n; idx += NT) {
        const int r = idx / n, c = idx - r * n;
        if (c > r) A[idx] = A[(size_t)c * n + r];
    }
    __syncthreads();
    eigh_tridiag_one_t<NT>(A, n, n, ws.d + (size_t)b * n, ws.e + (size_t)b * n, ws.tau + (size_t)b * n, sh);
    if (threadIdx.x == 0) atomicAdd(&ws.flag[B], 1);
}
__global__ __launch_bounds__(1024) void eigh_tridiag_repair_kernel(double* __restrict__ Aall, int n, EighWs ws, int B) {
    __shared__ double tri_sh[EG_TRI_SH];
    const int b = blockIdx.x;
    if (ws.flag[b] != 1) return;
    eigh_repair1<1024>(Aall + (size_t)b * n * n, n, ws, b, B, tri_sh);
}

// The same for a give-up inside the SECOND cluster stage (ws.flag[b] == 2).  The block has been updated in place by then, so there is
// no original matrix to go back to - but the first stage's hand-over is still there: the trailing block (rows / columns >= base =
// s_first + 1, updated through reflector s_first - 1) sits in A, where the second stage has only written reflector rows into its strict
// UPPER triangle, and ws.zt holds the pending reflector s_first (vector v, tau) with its matrix-vector product p.  One workgroup
// mirrors the block's lower triangle back, applies the pending reflector (w = tau p - (tau / 2)(tau p . v) v, A -= v w^T + w v^T) and
// tridiagonalises what is left - an ordinary symmetric m x m problem with leading dimension n - with the memory-streaming kernel;
// d / e / tau / the reflector rows from base on are rewritten.  sh: EG_REPAIR_SH doubles.
#define EG_REPAIR_SH (EG_TRI_SH + 2 * EG_MAXN + 16)
template <int NT>
__device__ __forceinline__ void eigh_repair2(double* __restrict__ A, int n, EighWs ws, int b, int B, int s_first, double* __restrict__ sh) {
    double* vv = sh + EG_TRI_SH;
    double* ww = vv + EG_MAXN;
    double* red2 = ww + EG_MAXN;
    const int tid = threadIdx.x;
    const int base = s_first + 1, m = n - base;
    double* Ab = A + (size_t)base * n + base;              // the trailing block, leading dimension n
    const double* st = ws.zt + (size_t)b * n * EG_MAXN;
    const double tk = st[3 * EG_MAXN];
    double pv = 0.0;
    for (int i = tid; i < m; i += NT) {
        const double v = st[base + i], pp = st[EG_MAXN + base + i];
        vv[i] = v;
        ww[i] = tk * pp;
        pv += tk * pp * v;
    }
    for (int idx = tid; idx < m * m; idx += NT) {           // lower -> upper (the second stage's reflector rows go)
        const int r = idx / m, c = idx - r * m;
        if (c > r) Ab[(size_t)r * n + c] = Ab[(size_t)c * n + r];
    }
    pv = block_sum(pv, red2);                              // (barriers inside: vv / ww / the mirrored block are visible behind it)
    const double al = -0.5 * tk * pv;
    for (int i = tid; i < m; i += NT) ww[i] += al * vv[i];
    __syncthreads();
    for (int idx = tid; idx < m * m; idx += NT) {
        const int r = idx / m, c = idx - r * m;
        Ab[(size_t)r * n + c] -= vv[r] * ww[c] + ww[r] * vv[c];
    }
    __syncthreads();
    eigh_tridiag_one_t<NT>(Ab, m, n, ws.d + (size_t)b * n + base, ws.e + (size_t)b * n + base, ws.tau + (size_t)b * n + base, sh);
    if (tid == 0) atomicAdd(&ws.flag[B], 1);
}
__global__ __launch_bounds__(1024) void eigh_tridiag_repair2_kernel(double* __restrict__ Aall, int n, EighWs ws, int B, int s_first) {
    __shared__ double rep_sh[EG_REPAIR_SH];
    const int b = blockIdx.x;
    if (ws.flag[b] != 2) return;
    eigh_repair2<1024>(Aall + (size_t)b * n * n, n, ws, b, B, s_first, rep_sh);
}


// ------------------------------------------------------------------------------------------ e1, cluster variant
// EC_P workgroups per matrix keep the whole trailing matrix in REGISTERS (column-cyclic over the workgroups: workgroup
// p owns columns c = p + 8 * lane; wave rg of a workgroup owns rows r = rg + 16 * ri), so the per-step pass never touches
// memory: the single-workgroup kernel above streams sum_k 2 m_k^2 * 8 B per matrix through one CU's L2 port.
// Per step the workgroups exchange 2n doubles (each contributes the matrix-vector products of its own columns; the owner
// of the next pivot column publishes it); the O(n) vector work of a step is done redundantly by every workgroup.
// The exchange is a data-flow handshake without a separate barrier: every double travels as one 16-byte slot
// {lo32, tag, hi32, tag} written with a single cache-bypassing store and polled by its consumer until both tags carry the
// step number (each 8-byte half is self-validating, as in the LL protocol of the collectives libraries), so a step
// costs one store -> load propagation instead of store / atomic / poll / load round trips.  Slots are double-buffered by
// step parity: a workgroup can only publish step s+2 after it consumed step s+1 from every peer, which in turn published
// it after consuming step s.  All EC_P * (matrices per launch) workgroups must be co-resident: the host sizes the launch
// from the device's CU count.  blockIdx -> (XCD, matrix, p) keeps the workgroups of one matrix on one XCD.
#define EC_P 8
#define EC_RI (EG_MAXN / 16)
#define EC_SPIN_LIMIT (1u << 22)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Exchange stores.  `sc1` (write-through, the line is DROPPED from the XCD's L2) is visible to every CU of the device; `sc0` keeps the
// line in the XCD's L2, where the consumer's L1-bypassing poll finds it without a trip through the fabric - the consumer of a slot is
// always a workgroup of the SAME matrix, which the blockIdx mapping puts on the same XCD (workgroup g runs on XCD g % 8).  That
// placement is a property of the dispatcher, not of the ISA, so it is PROBED once per device (eigh_xch_probe_kernel: pairs of
// workgroups laid out like the real launches ping-pong tagged slots with `sc0` stores); only if every pair got through does the
// device-side flag ec_store_keep switch the cluster kernels over.  Measured: 6.93 -> 6.60 ms per 256 matrices (cluster stages
// 2.31 + 0.88 -> 2.10 + 0.78 ms); a step costs two dependent exchanges.
__device__ int ec_store_keep = 0;
__device__ int ec_probe_ticket = 0, ec_probe_ok = 0;
__device__ __forceinline__ void st_tagged(int keep, u32x4* p, double v, unsigned tag) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    u32x4 q;
    q.x = (unsigned)u; q.y = tag; q.z = (unsigned)(u >> 32); q.w = tag;
    if (keep) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(q) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ double tagged_value(u32x4 q) {
    return __longlong_as_double((long long)(((unsigned long long)q.z << 32) | q.x));
}
__device__ __forceinline__ int ec_keep_flag() { return __builtin_amdgcn_readfirstlane(*(volatile int*)&ec_store_keep); }
// grid 16 * pairs / 8 (the two-workgroup kernels' layout: workgroup g = 16 (m >> 3) + 8 p + (m & 7) is half p of pair m), block 64.
// Every pair plays `rounds` rounds of ping-pong through two slots with the store flavour under test; the consumer has polled a slot's
// previous value before the next one is stored, so a copy that another L2 kept would be stale for good and run into the spin limit.
__global__ __launch_bounds__(64) void eigh_xch_probe_kernel(u32x4* slots, int pairs, int rounds) {
    const int g = blockIdx.x, slot = g >> 3, p = slot & 1, m = (g & 7) + 8 * (slot >> 1);
    if (m >= pairs || threadIdx.x != 0) return;
    u32x4* mine = slots + 2 * m + p;                 // written by this half
    const u32x4* theirs = slots + 2 * m + (1 - p);
    bool ok = true;
    for (int r = 1; r <= rounds && ok; ++r) {
        if (p == 0) st_tagged(1, mine, (double)r, (unsigned)r);
        unsigned spins = 0;
        u32x4 q;
        do {
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(theirs) : "memory");
        } while ((q.y != (unsigned)r || q.w != (unsigned)r) && ++spins < (1u << 16));
        ok = (q.y == (unsigned)r && q.w == (unsigned)r && tagged_value(q) == (double)r);
        if (p == 1 && ok) st_tagged(1, mine, (double)r, (unsigned)r);
    }
    if (ok) atomicAdd(&ec_probe_ok, 1);
    __threadfence();
    if (atomicAdd(&ec_probe_ticket, 1) == 2 * pairs - 1) {         // the last half to finish decides
        __threadfence();
        ec_store_keep = (atomicAdd(&ec_probe_ok, 0) == 2 * pairs) ? 1 : 0;
    }
}
__device__ __forceinline__ double block_sum1(double v, double* slot) {   // one barrier; slot[16] is not reused before 2 more barriers
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += slot[i];
    return t;
}

__global__ __launch_bounds__(512) void eigh_tridiag_cluster_kernel(double* __restrict__ Aall, int n, int b0, int Bc, EighWs ws) {
    const int keep_ = ec_keep_flag();              // exchange stores may stay in the XCD's L2 (probed once per device)
    __shared__ __attribute__((aligned(16))) double vperm[3][EG_MAXN];   // v, w, v_next at [(r & 15) * 32 + (r >> 4)]
    __shared__ double vnat[EG_MAXN], wnat[EG_MAXN];
    __shared__ double accb[16][64];
    __shared__ double red0[8], red1[8];
    __shared__ double s_alpha, s_ppiv;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int slot = g >> 3, p = slot & 7, mloc = (g & 7) + 8 * (slot >> 3);
    if (mloc >= Bc) return;
    const int b = b0 + mloc;
    double* A = Aall + (size_t)b * n * n;
    double* d = ws.d + (size_t)b * n;
    double* e = ws.e + (size_t)b * n;
    double* tau = ws.tau + (size_t)b * n;
    u32x4* xch = reinterpret_cast<u32x4*>(ws.xch) + (size_t)b * EG_XCH * EG_MAXN;   // [parity][0: A v | 1: raw pivot column][EG_MAXN]

    // thread tile: 2 columns x 32 rows.  column slots cl0 = lane & 31, cl0 + 32 (c = p + 8 * slot); row slot rs = 2 * wave + (lane >> 5),
    // rows r = rs + 16 * ri: the 32 lanes of a half-wave share their row values (LDS broadcast), each row value feeds 2 columns.
    const int cl0 = lane & 31, rs = 2 * wv + (lane >> 5);
    const int c0 = p + EC_P * cl0, c1 = c0 + EC_P * 32;
    double a0[EC_RI], a1[EC_RI];
#pragma unroll
    for (int ri = 0; ri < EC_RI; ++ri) {
        const int r = rs + 16 * ri;
        a0[ri] = (r < n && c0 < n) ? A[(size_t)r * n + c0] : 0.0;
        a1[ri] = (r < n && c1 < n) ? A[(size_t)r * n + c1] : 0.0;
    }
    const int i = tid;                                     // vector element owned in the O(n) phase
    const int pi = (i & 15) * 32 + (i >> 4);
    double v_i = 0.0, tk = 0.0, p_i = 0.0;
    double col_i = (i < n) ? A[i] : 0.0;                   // column 0 (= row 0)
#ifdef EC_PROF
    long long tacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = clock64(), t1;
#define EC_T(j) do { t1 = clock64(); tacc[j] += t1 - t0; t0 = t1; } while (0)
#else
#define EC_T(j)
#endif
    for (int s = -1; s <= n - 2; ++s) {
        // ---- O(n) phase: w_s, pivot column s+1 of the updated matrix, Householder vector s+1
        const bool in = (i >= s + 1) && (i < n);
        double w_i = 0.0, wpiv = 0.0;
        if (s >= 0) {
            if (i == s + 1) s_ppiv = p_i;
            double pv = in ? tk * p_i * v_i : 0.0;
            pv = wave_sum_dpp(pv);
            if (lane == 0) red0[wv] = pv;
            __syncthreads();
            pv = ((red0[0] + red0[1]) + (red0[2] + red0[3])) + ((red0[4] + red0[5]) + (red0[6] + red0[7]));
            EC_T(0);
            const double al = -0.5 * tk * pv;
            w_i = in ? tk * p_i + al * v_i : 0.0;
            wpiv = tk * s_ppiv + al;                       // v[s+1] = 1
        }
        const double x_i = in ? col_i - v_i * wpiv - w_i : 0.0;
        double tn = 0.0, betan = 0.0, vn_i = 0.0;
        if (s + 2 <= n - 1) {
            if (i == s + 2) s_alpha = x_i;
            double ss = (i >= s + 3 && i < n) ? x_i * x_i : 0.0;
            ss = wave_sum_dpp(ss);
            if (lane == 0) red1[wv] = ss;
            __syncthreads();
            ss = ((red1[0] + red1[1]) + (red1[2] + red1[3])) + ((red1[4] + red1[5]) + (red1[6] + red1[7]));
            EC_T(1);
            const double alpha = s_alpha;
            double scn = 0.0;
            betan = alpha;
            if (ss > 0.0) {
                // beta = -sign(alpha) ||x||, tau = (beta - alpha) / beta = 1 + |alpha| / ||x||, scale = 1 / (alpha - beta) =
                // sign(alpha) / (|alpha| + ||x||): one rsqrt and one rcp with two Newton steps each instead of sqrt + two divisions
                // (this sits on the serial critical path of every step; the results agree with the divisions to ~1 ulp)
                const double s2 = alpha * alpha + ss, aa = fabs(alpha);
                double y = __builtin_amdgcn_rsq(s2);
                y = y * (1.5 - 0.5 * s2 * y * y);
                y = y * (1.5 - 0.5 * s2 * y * y);
                const double nrm = s2 * y;
                betan = alpha >= 0.0 ? -nrm : nrm;
                tn = 1.0 + aa * y;
                const double dd = aa + nrm;
                double r = __builtin_amdgcn_rcp(dd);
                r = r * (2.0 - dd * r);
                r = r * (2.0 - dd * r);
                scn = alpha >= 0.0 ? r : -r;
            }
            vn_i = (i == s + 2) ? 1.0 : ((i > s + 2 && i < n) ? x_i * scn : 0.0);
        }
        EC_T(6);
        // Step -1's record (row 0 of A becomes reflector 0) is deferred until after this step's exchange: a peer workgroup that is
        // scheduled late (the launch is only co-resident on an otherwise idle GPU) still has to read row 0 / column 0 of the ORIGINAL
        // matrix in its prologue; every peer's first publication proves that it has.  (Found in round 2: wrong - not NaN - eigenvalues
        // whenever another stream's kernels delayed some workgroups of a matrix.)
        if (s >= 0 && p == ((s + 1) & 7)) {                          // one workgroup records the step
            if (i == s + 1) { d[s + 1] = x_i; e[s + 1] = betan; tau[s + 1] = tn; }
            if (i >= s + 2 && i < n) A[(size_t)(s + 1) * n + i] = vn_i;
        }
        EC_T(7);
        if (s == n - 2) break;
        vnat[i] = v_i; wnat[i] = w_i;
        vperm[0][pi] = v_i; vperm[1][pi] = w_i; vperm[2][pi] = vn_i;
        EC_T(8);
        __syncthreads();
        EC_T(2);
        // ---- fused pass over the registers: x = a - v_r w_c - w_r v_c ; acc_c += x * vnext_r
        double acc0 = 0.0, acc1 = 0.0;
        if (c0 >= s + 2 - EC_P * 32 && c0 < n) {           // at least one of the two columns is live (dead ones only see zeros)
            const double vc0 = vnat[c0], wc0 = wnat[c0];
            const double vc1 = (c1 < n) ? vnat[c1 & (EG_MAXN - 1)] : 0.0, wc1 = (c1 < n) ? wnat[c1 & (EG_MAXN - 1)] : 0.0;
            const int ri0 = (s + 2 - rs + 15) >> 4;        // first row slot with r >= s + 2 (may be <= 0)
            const double2* pv0 = reinterpret_cast<const double2*>(&vperm[0][rs * 32]);
            const double2* pv1 = reinterpret_cast<const double2*>(&vperm[1][rs * 32]);
            const double2* pv2 = reinterpret_cast<const double2*>(&vperm[2][rs * 32]);
#pragma unroll
            for (int q = 0; q < EC_RI / 8; ++q) {
                if (8 * q + 7 >= ri0) {
                    double2 vr[4], wr[4], nr[4];
#pragma unroll
                    for (int h = 0; h < 4; ++h) { vr[h] = pv0[4 * q + h]; wr[h] = pv1[4 * q + h]; nr[h] = pv2[4 * q + h]; }
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const int k0 = 8 * q + 2 * h;
                        double x0 = a0[k0], x1 = a0[k0 + 1], y0 = a1[k0], y1 = a1[k0 + 1];
                        x0 -= vr[h].x * wc0 + wr[h].x * vc0;
                        x1 -= vr[h].y * wc0 + wr[h].y * vc0;
                        y0 -= vr[h].x * wc1 + wr[h].x * vc1;
                        y1 -= vr[h].y * wc1 + wr[h].y * vc1;
                        a0[k0] = x0; a0[k0 + 1] = x1; a1[k0] = y0; a1[k0 + 1] = y1;
                        acc0 += x0 * nr[h].x + x1 * nr[h].y;
                        acc1 += y0 * nr[h].x + y1 * nr[h].y;
                    }
                }
            }
        }
        EC_T(3);
        accb[rs][cl0] = acc0;
        accb[rs][cl0 + 32] = acc1;
        const unsigned tag = (unsigned)(s + 3);
        u32x4* xp = xch + (size_t)((s + 1) & 1) * 2 * EG_MAXN;
        if (rs == ((s + 2) & 15)) {                        // every workgroup publishes its part of pivot row s+2 (= column, by symmetry)
            double r0 = 0.0, r1 = 0.0;
            switch ((s + 2) >> 4) {
#define EC_CASE(k) case k: r0 = a0[k]; r1 = a1[k]; break;
                EC_CASE(0) EC_CASE(1) EC_CASE(2) EC_CASE(3) EC_CASE(4) EC_CASE(5) EC_CASE(6) EC_CASE(7)
                EC_CASE(8) EC_CASE(9) EC_CASE(10) EC_CASE(11) EC_CASE(12) EC_CASE(13) EC_CASE(14) EC_CASE(15)
                EC_CASE(16) EC_CASE(17) EC_CASE(18) EC_CASE(19) EC_CASE(20) EC_CASE(21) EC_CASE(22) EC_CASE(23)
                EC_CASE(24) EC_CASE(25) EC_CASE(26) EC_CASE(27) EC_CASE(28) EC_CASE(29) EC_CASE(30) EC_CASE(31)
#undef EC_CASE
            }
            if (c0 >= s + 2 && c0 < n) st_tagged(keep_, &xp[EG_MAXN + c0], r0, tag);
            if (c1 >= s + 2 && c1 < n) st_tagged(keep_, &xp[EG_MAXN + c1], r1, tag);
        }
        __syncthreads();
        if (tid < 64) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += accb[q][tid];
            const int cc = p + EC_P * tid;
            if (cc >= s + 2 && cc < n) st_tagged(keep_, &xp[cc], t, tag);
        }
        EC_T(4);
        // ---- consume the peers' slots for step s + 1
        const bool need = (i >= s + 2) && (i < n);
        if (__any(need)) {
            u32x4 qp, qc;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                if (need) {
                    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(qp), "=&v"(qc) : "v"(&xp[i]), "v"(&xp[EG_MAXN + i]) : "memory");
                    ok = (qp.y == tag) && (qp.w == tag) && (qc.y == tag) && (qc.w == tag);
                }
#ifdef EC_NOPOLL
                break;
#endif
                if (__all(ok)) break;
                if (++spins > EC_SPIN_LIMIT) {             // never hang the device: poison the output instead
                    if (i < n) d[i] = __builtin_nan("");       // (overwritten by eigh_tridiag_repair_kernel, which redoes this matrix)
                    if (tid == 0) ws.flag[b] = 1;
                    return;
                }
#ifndef EC_NOSLEEP
                __builtin_amdgcn_s_sleep(1);
#endif
            }
            if (need) { p_i = tagged_value(qp); col_i = tagged_value(qc); }
        }
        if (s == -1 && p == 0) {                             // deferred record of step -1 (see above)
            if (i == 0) { d[0] = x_i; e[0] = betan; tau[0] = tn; }
            if (i >= 1 && i < n) A[i] = vn_i;
        }
        v_i = vn_i;
        tk = tn;
        EC_T(5);
    }
#ifdef EC_PROF
    if (tid == 0 && g == 0) for (int j = 0; j < 9; ++j) ws.lamp[j] = (double)tacc[j];
#endif
}

// ------------------------------------------------------------------------------------------ e1, cluster variant with 4 workgroups
// Same algorithm as eigh_tridiag_cluster_kernel with FOUR workgroups per matrix (n <= 448): a spinning workgroup owns its CU, and
// the step time is dominated by latencies that do not depend on the share of the matrix a workgroup holds, so halving the
// workgroups per matrix halves the CU time of the whole tridiagonalisation and lets 32 matrices run in one launch on half of the
// chip.  A thread holds 3 columns x 28 rows in registers; the 4th column slot of lanes 0..15 (112 = 3.5 x 32 column slots per
// workgroup) lives in LDS.
#define E4_P 4
#define E4_RI 28
__global__ __launch_bounds__(512) void eigh_tridiag_cluster4_kernel(double* __restrict__ Aall, int n, int b0, int Bc, EighWs ws, int s_stop, int fail_every) {
    const int keep_ = ec_keep_flag();              // exchange stores may stay in the XCD's L2 (probed once per device)
    __shared__ __attribute__((aligned(16))) double vperm[3][EG_MAXN];   // v, w, v_next at [(r & 15) * 32 + (r >> 4)]
    __shared__ double vnat[EG_MAXN], wnat[EG_MAXN];
    __shared__ double accb[16][128];
    __shared__ double aL[E4_RI][16][16];                   // 4th column slot of lanes 0..15: [ri][row slot][lane]
    __shared__ double red0[8], red1[8];
    __shared__ double s_alpha, s_ppiv;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int slot = g >> 3, p = slot & 3, mloc = (g & 7) + 8 * (slot >> 2);
    if (mloc >= Bc) return;
    const int b = b0 + mloc;
    if (fail_every > 0 && b % fail_every == 0) {           // test hook (NELE_EIGH_FAIL_EVERY=k > 0; -k: the second stage): behave as if this matrix's workgroups had given up
        if (tid == 0 && p == 0) ws.flag[b] = 1;
        return;
    }
    double* A = Aall + (size_t)b * n * n;
    double* d = ws.d + (size_t)b * n;
    double* e = ws.e + (size_t)b * n;
    double* tau = ws.tau + (size_t)b * n;
    u32x4* xch = reinterpret_cast<u32x4*>(ws.xch) + (size_t)b * EG_XCH * EG_MAXN;

    const int cl0 = lane & 31, rs = 2 * wv + (lane >> 5);
    const int c0 = p + E4_P * cl0, c1 = c0 + E4_P * 32, c2 = c0 + E4_P * 64, c3 = c0 + E4_P * 96;
    const bool has3 = cl0 < 16;
    double a0[E4_RI], a1[E4_RI], a2[E4_RI];
#pragma unroll
    for (int ri = 0; ri < E4_RI; ++ri) {
        const int r = rs + 16 * ri;
        a0[ri] = (r < n && c0 < n) ? A[(size_t)r * n + c0] : 0.0;
        a1[ri] = (r < n && c1 < n) ? A[(size_t)r * n + c1] : 0.0;
        a2[ri] = (r < n && c2 < n) ? A[(size_t)r * n + c2] : 0.0;
        if (has3) aL[ri][rs][cl0] = (r < n && c3 < n) ? A[(size_t)r * n + c3] : 0.0;
    }
    const int i = tid;
    const int pi = (i & 15) * 32 + (i >> 4);
    double v_i = 0.0, tk = 0.0, p_i = 0.0;
    double col_i = (i < n) ? A[i] : 0.0;
#ifdef EC_PROF
    long long tacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = clock64(), t1;
#endif
    for (int s = -1; s <= n - 2; ++s) {
        const bool in = (i >= s + 1) && (i < n);
        double w_i = 0.0, wpiv = 0.0;
        if (s >= 0) {
            if (i == s + 1) s_ppiv = p_i;
            double pv = in ? tk * p_i * v_i : 0.0;
            pv = wave_sum_dpp(pv);
            if (lane == 0) red0[wv] = pv;
            __syncthreads();
            pv = ((red0[0] + red0[1]) + (red0[2] + red0[3])) + ((red0[4] + red0[5]) + (red0[6] + red0[7]));
            EC_T(0);
            const double al = -0.5 * tk * pv;
            w_i = in ? tk * p_i + al * v_i : 0.0;
            wpiv = tk * s_ppiv + al;
        }
        const double x_i = in ? col_i - v_i * wpiv - w_i : 0.0;
        double tn = 0.0, betan = 0.0, vn_i = 0.0;
        if (s + 2 <= n - 1) {
            if (i == s + 2) s_alpha = x_i;
            double ss = (i >= s + 3 && i < n) ? x_i * x_i : 0.0;
            ss = wave_sum_dpp(ss);
            if (lane == 0) red1[wv] = ss;
            __syncthreads();
            ss = ((red1[0] + red1[1]) + (red1[2] + red1[3])) + ((red1[4] + red1[5]) + (red1[6] + red1[7]));
            EC_T(1);
            const double alpha = s_alpha;
            double scn = 0.0;
            betan = alpha;
            if (ss > 0.0) {
                const double s2 = alpha * alpha + ss, aa = fabs(alpha);
                double y = __builtin_amdgcn_rsq(s2);
                y = y * (1.5 - 0.5 * s2 * y * y);
                y = y * (1.5 - 0.5 * s2 * y * y);
                const double nrm = s2 * y;
                betan = alpha >= 0.0 ? -nrm : nrm;
                tn = 1.0 + aa * y;
                const double dd = aa + nrm;
                double r = __builtin_amdgcn_rcp(dd);
                r = r * (2.0 - dd * r);
                r = r * (2.0 - dd * r);
                scn = alpha >= 0.0 ? r : -r;
            }
            vn_i = (i == s + 2) ? 1.0 : ((i > s + 2 && i < n) ? x_i * scn : 0.0);
        }
        // Step -1's record (row 0 of A becomes reflector 0) is deferred until after this step's exchange: a peer workgroup that is
        // scheduled late (the launch is only co-resident on an otherwise idle GPU) still has to read row 0 / column 0 of the ORIGINAL
        // matrix in its prologue; every peer's first publication proves that it has.  (Found in round 2: wrong - not NaN - eigenvalues
        // whenever another stream's kernels delayed some workgroups of a matrix.)
        if (s >= 0 && p == ((s + 1) & 3)) {
            if (i == s + 1) { d[s + 1] = x_i; e[s + 1] = betan; tau[s + 1] = tn; }
            if (i >= s + 2 && i < n) A[(size_t)(s + 1) * n + i] = vn_i;
        }
        if (s == n - 2) break;
        vnat[i] = v_i; wnat[i] = w_i;
        vperm[0][pi] = v_i; vperm[1][pi] = w_i; vperm[2][pi] = vn_i;
        __syncthreads();
        EC_T(2);
        // ---- fused pass: x = a - v_r w_c - w_r v_c ; acc_c += x * vnext_r  (3 register columns + 1 LDS column)
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        if (c0 < n) {
            const double vc0 = vnat[c0], wc0 = wnat[c0];
            const double vc1 = (c1 < n) ? vnat[c1 & (EG_MAXN - 1)] : 0.0, wc1 = (c1 < n) ? wnat[c1 & (EG_MAXN - 1)] : 0.0;
            const double vc2 = (c2 < n) ? vnat[c2 & (EG_MAXN - 1)] : 0.0, wc2 = (c2 < n) ? wnat[c2 & (EG_MAXN - 1)] : 0.0;
            const bool l3 = has3 && c3 < n;
            const double vc3 = l3 ? vnat[c3 & (EG_MAXN - 1)] : 0.0, wc3 = l3 ? wnat[c3 & (EG_MAXN - 1)] : 0.0;
            const int ri0 = (s + 2 - rs + 15) >> 4;
            const double2* pv0 = reinterpret_cast<const double2*>(&vperm[0][rs * 32]);
            const double2* pv1 = reinterpret_cast<const double2*>(&vperm[1][rs * 32]);
            const double2* pv2 = reinterpret_cast<const double2*>(&vperm[2][rs * 32]);
#pragma unroll
            for (int q = 0; q < E4_RI / 2; ++q) {
                if (2 * q + 1 >= ri0) {                    // rows rs + 16 (2q), rs + 16 (2q + 1)
                    const double2 vr = pv0[q], wr = pv1[q], nr = pv2[q];
                    const int k0 = 2 * q;
                    double x0 = a0[k0], x1 = a0[k0 + 1];
                    x0 -= vr.x * wc0 + wr.x * vc0;
                    x1 -= vr.y * wc0 + wr.y * vc0;
                    a0[k0] = x0; a0[k0 + 1] = x1;
                    acc0 += x0 * nr.x + x1 * nr.y;
                    double y0 = a1[k0], y1 = a1[k0 + 1];
                    y0 -= vr.x * wc1 + wr.x * vc1;
                    y1 -= vr.y * wc1 + wr.y * vc1;
                    a1[k0] = y0; a1[k0 + 1] = y1;
                    acc1 += y0 * nr.x + y1 * nr.y;
                    double z0 = a2[k0], z1 = a2[k0 + 1];
                    z0 -= vr.x * wc2 + wr.x * vc2;
                    z1 -= vr.y * wc2 + wr.y * vc2;
                    a2[k0] = z0; a2[k0 + 1] = z1;
                    acc2 += z0 * nr.x + z1 * nr.y;
                    if (l3) {
                        double u0 = aL[k0][rs][cl0], u1 = aL[k0 + 1][rs][cl0];
                        u0 -= vr.x * wc3 + wr.x * vc3;
                        u1 -= vr.y * wc3 + wr.y * vc3;
                        aL[k0][rs][cl0] = u0; aL[k0 + 1][rs][cl0] = u1;
                        acc3 += u0 * nr.x + u1 * nr.y;
                    }
                }
            }
        }
        EC_T(3);
        accb[rs][cl0] = acc0;
        accb[rs][cl0 + 32] = acc1;
        accb[rs][cl0 + 64] = acc2;
        accb[rs][cl0 + 96] = acc3;
        const unsigned tag = (unsigned)(s + 3);
        u32x4* xp = xch + (size_t)((s + 1) & 1) * 2 * EG_MAXN;
        if (rs == ((s + 2) & 15)) {                        // every workgroup publishes its part of pivot row s+2
            double r0 = 0.0, r1 = 0.0, r2 = 0.0;
            const int rq = (s + 2) >> 4;
            switch (rq) {
#define EC_CASE(k) case k: r0 = a0[k]; r1 = a1[k]; r2 = a2[k]; break;
                EC_CASE(0) EC_CASE(1) EC_CASE(2) EC_CASE(3) EC_CASE(4) EC_CASE(5) EC_CASE(6) EC_CASE(7)
                EC_CASE(8) EC_CASE(9) EC_CASE(10) EC_CASE(11) EC_CASE(12) EC_CASE(13) EC_CASE(14) EC_CASE(15)
                EC_CASE(16) EC_CASE(17) EC_CASE(18) EC_CASE(19) EC_CASE(20) EC_CASE(21) EC_CASE(22) EC_CASE(23)
                EC_CASE(24) EC_CASE(25) EC_CASE(26) EC_CASE(27)
#undef EC_CASE
            }
            if (c0 >= s + 2 && c0 < n) st_tagged(keep_, &xp[EG_MAXN + c0], r0, tag);
            if (c1 >= s + 2 && c1 < n) st_tagged(keep_, &xp[EG_MAXN + c1], r1, tag);
            if (c2 >= s + 2 && c2 < n) st_tagged(keep_, &xp[EG_MAXN + c2], r2, tag);
            if (has3 && c3 >= s + 2 && c3 < n) st_tagged(keep_, &xp[EG_MAXN + c3], aL[rq][rs][cl0], tag);
        }
        __syncthreads();
        if (tid < 128) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += accb[q][tid];
            const int cc = p + E4_P * tid;
            if (tid < 112 && cc >= s + 2 && cc < n) st_tagged(keep_, &xp[cc], t, tag);
        }
        EC_T(4);
        const bool need = (i >= s + 2) && (i < n);
        if (__any(need)) {
            u32x4 qp, qc;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                if (need) {
                    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(qp), "=&v"(qc) : "v"(&xp[i]), "v"(&xp[EG_MAXN + i]) : "memory");
                    ok = (qp.y == tag) && (qp.w == tag) && (qc.y == tag) && (qc.w == tag);
                }
                if (__all(ok)) break;
                if (++spins > EC_SPIN_LIMIT) {
                    if (i < n) d[i] = __builtin_nan("");       // (overwritten by eigh_tridiag_repair_kernel, which redoes this matrix)
                    if (tid == 0) ws.flag[b] = 1;
                    return;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (need) { p_i = tagged_value(qp); col_i = tagged_value(qc); }
        }
        EC_T(5);
        if (s == -1 && p == 0) {                             // deferred record of step -1 (see above)
            if (i == 0) { d[0] = x_i; e[0] = betan; tau[0] = tn; }
            if (i >= 1 && i < n) A[i] = vn_i;
        }
        v_i = vn_i;
        tk = tn;
        if (s == s_stop) {
            // hand-over to eigh_tridiag_tail_kernel: the trailing block (rows / columns >= s + 2, updated through reflector s) goes back
            // to A in place - those rows only receive reflector rows later - and the per-element state of the next iteration to ws.zt
            const int base = s + 2;
#pragma unroll
            for (int ri = 0; ri < E4_RI; ++ri) {
                const int r = rs + 16 * ri;
                if (r >= base && r < n) {
                    if (c0 >= base && c0 < n) A[(size_t)r * n + c0] = a0[ri];
                    if (c1 >= base && c1 < n) A[(size_t)r * n + c1] = a1[ri];
                    if (c2 >= base && c2 < n) A[(size_t)r * n + c2] = a2[ri];
                    if (has3 && c3 >= base && c3 < n) A[(size_t)r * n + c3] = aL[ri][rs][cl0];
                }
            }
            if (p == 0 && i < n) {
                double* st = ws.zt + (size_t)b * n * EG_MAXN;
                st[i] = v_i; st[EG_MAXN + i] = p_i; st[2 * EG_MAXN + i] = col_i;
                if (i == 0) st[3 * EG_MAXN] = tk;
            }
            break;
        }
    }
#ifdef EC_PROF
    if (tid == 0 && g == 0) for (int j = 0; j < 9; ++j) ws.lamp[j] = (double)tacc[j];
#endif
}

// ------------------------------------------------------------------------------------------ e1, first cluster stage on the LOWER TRIANGLE: 2 workgroups per matrix
// Round 4.  eigh_tridiag_cluster4_kernel holds the full 420 x 420 trailing matrix in the registers of four CUs - 64 matrices fill the
// chip.  The matrix is symmetric: its lower triangle (incl. the diagonal) fits TWO register files + 108 KB of LDS, so 128 matrices run
// per launch and a batch of 256 takes two rounds of the latency chain instead of four.  Same algorithm, exchange protocol, records
// and hand-over as the four-workgroup kernel; what changes is the fused pass:
//   * a thread (row class rs = 16 classes, column class cl0 = 32 classes; workgroup p owns the columns c = p + 2 (cl0 + 32 j)) stores
//     element (r, c) only for r >= c: column slot j holds the row slots ri = 4 j .. 26 (r = rs + 16 ri), 105 slots per thread - 78 in
//     registers (j = 1 .. 6), the 27 of j = 0 in LDS (those columns are eliminated first: the LDS traffic ends after 64 steps); the
//     (at most four) slots of a column that lie above the diagonal are kept at exactly zero;
//   * y = A v_next needs BOTH directions now: the column sums sum_{r >= c} x_rc n_r as before (over a thread's row slots, then over the
//     16 row classes through LDS) and the row sums sum_{c < r} x_rc n_c - over a thread's 7 column slots, then over the 32 column
//     classes of a half wave by a reduce-scatter butterfly (lane xor 16 through the LDS crossbar, then row_ror:8 / row_half_mirror /
//     two quad permutations: 31 exchanged values per lane), after which lane cl0 holds the complete row sum of row slot ri = cl0;
//   * the exchange carries four kinds of tagged slots per step: column partial of the owner, row partial of each workgroup, and the
//     next pivot COLUMN (the lower triangle holds column s + 2 from the diagonal down: it equals the pivot row), published by the 16
//     threads of the owning workgroup that hold it.  p_i = (column partial + row partial 0) + row partial 1, the same on both sides.
#define ECS_P 2
#define ECS_RI 27
#define ECS_LD 28                       // row-vector stride per row class (even: double2 reads)
#define ECS_NJ 7
#define ECS_NREG 78                     // slots of j = 1 .. 6
__host__ __device__ constexpr int ecs_off(int j) { return j == 1 ? 0 : j == 2 ? 23 : j == 3 ? 42 : j == 4 ? 57 : j == 5 ? 68 : 75; }
__host__ __device__ constexpr int ecs_len(int j) { return ECS_RI - 4 * j; }

#ifdef ECS_PROF
__device__ unsigned long long ecs_prof[8];     // shader clocks of thread 0 of workgroup 0 per phase, summed over the steps (diagnostic build)
#define ECS_T(j) do { const unsigned long long t1_ = __builtin_readcyclecounter(); pacc_[j] += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define ECS_T(j)
#endif
__global__ __launch_bounds__(512) void eigh_tridiag_clusters_kernel(double* __restrict__ Aall, int n, int b0, int Bc, EighWs ws, int s_stop, int fail_every) {
    const int keep_ = ec_keep_flag();              // exchange stores may stay in the XCD's L2 (probed once per device)
    extern __shared__ double ecs_col[];                    // aL[ECS_RI][16][32]: column slot j = 0 of every lane
    __shared__ __attribute__((aligned(16))) double vperm[3][16 * ECS_LD];   // v, w, v_next at [(r & 15) * ECS_LD + (r >> 4)]
    __shared__ double accb[16][32 * ECS_NJ];
    __shared__ double prow[16 * ECS_LD];                   // the next pivot column on its way from its 16 holders to one publishing thread per row
    __shared__ double red0[8], red1[8];
    __shared__ double s_alpha, s_ppiv;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int slot = g >> 3, p = slot & 1, mloc = (g & 7) + 8 * (slot >> 1);
    if (mloc >= Bc) return;
    const int b = b0 + mloc;
    if (fail_every > 0 && b % fail_every == 0) {           // test hook (see eigh_tridiag_cluster4_kernel)
        if (tid == 0 && p == 0) ws.flag[b] = 1;
        return;
    }
    double* A = Aall + (size_t)b * n * n;
    double* d = ws.d + (size_t)b * n;
    double* e = ws.e + (size_t)b * n;
    double* tau = ws.tau + (size_t)b * n;
    u32x4* xch = reinterpret_cast<u32x4*>(ws.xch) + (size_t)b * EG_XCH * EG_MAXN;
    double (*aL)[16][32] = reinterpret_cast<double (*)[16][32]>(ecs_col);

    const int cl0 = lane & 31, rs = 2 * wv + (lane >> 5);
    const int t0 = p + 2 * cl0 - rs;                       // slot k of a column is on / below the diagonal iff 16 k >= t0
    const int kmin = t0 <= 0 ? 0 : (t0 + 15) >> 4;         // 0 .. 4
    const int kdiag = (t0 >= 0 && (t0 & 15) == 0) ? (t0 >> 4) : -1;   // the slot that IS the diagonal element (every column alike)
    int cc[ECS_NJ];
#pragma unroll
    for (int j = 0; j < ECS_NJ; ++j) cc[j] = p + ECS_P * (cl0 + 32 * j);
    double a[ECS_NREG];
    // ---- load the lower triangle (r >= c)
#pragma unroll
    for (int k = 0; k < ECS_RI; ++k) {
        const int r = rs + 16 * k;
        aL[k][rs][cl0] = (k >= kmin && r < n && cc[0] < n) ? A[(size_t)r * n + cc[0]] : 0.0;
    }
#pragma unroll
    for (int j = 1; j < ECS_NJ; ++j)
#pragma unroll
        for (int k = 0; k < ecs_len(j); ++k) {
            const int r = rs + 16 * (4 * j + k);
            a[ecs_off(j) + k] = (k >= kmin && r < n && cc[j] < n) ? A[(size_t)r * n + cc[j]] : 0.0;
        }
    const int i = tid;
    const int pi = (i & 15) * ECS_LD + (i >> 4);           // permuted slot of vector element i (i < 16 * ECS_LD = 448)
    double v_i = 0.0, tk = 0.0, p_i = 0.0;
    double col_i = (i < n) ? A[i] : 0.0;                   // column 0 (= row 0)
    int s_done = -2;
#ifdef ECS_PROF
    unsigned long long pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_readcyclecounter();
#endif
    for (int s = -1; s <= n - 2; ++s) {
        const bool in = (i >= s + 1) && (i < n);
        double w_i = 0.0, wpiv = 0.0;
        if (s >= 0) {
            if (i == s + 1) s_ppiv = p_i;
            double pv = in ? tk * p_i * v_i : 0.0;
            pv = wave_sum_dpp(pv);
            if (lane == 0) red0[wv] = pv;
            __syncthreads();
            pv = ((red0[0] + red0[1]) + (red0[2] + red0[3])) + ((red0[4] + red0[5]) + (red0[6] + red0[7]));
            const double al = -0.5 * tk * pv;
            w_i = in ? tk * p_i + al * v_i : 0.0;
            wpiv = tk * s_ppiv + al;
        }
        const double x_i = in ? col_i - v_i * wpiv - w_i : 0.0;
        double tn = 0.0, betan = 0.0, vn_i = 0.0;
        if (s + 2 <= n - 1) {
            if (i == s + 2) s_alpha = x_i;
            double ss = (i >= s + 3 && i < n) ? x_i * x_i : 0.0;
            ss = wave_sum_dpp(ss);
            if (lane == 0) red1[wv] = ss;
            __syncthreads();
            ss = ((red1[0] + red1[1]) + (red1[2] + red1[3])) + ((red1[4] + red1[5]) + (red1[6] + red1[7]));
            const double alpha = s_alpha;
            double scn = 0.0;
            betan = alpha;
            if (ss > 0.0) {
                const double s2 = alpha * alpha + ss, aa = fabs(alpha);
                double y = __builtin_amdgcn_rsq(s2);
                y = y * (1.5 - 0.5 * s2 * y * y);
                y = y * (1.5 - 0.5 * s2 * y * y);
                const double nrm = s2 * y;
                betan = alpha >= 0.0 ? -nrm : nrm;
                tn = 1.0 + aa * y;
                const double dd = aa + nrm;
                double r = __builtin_amdgcn_rcp(dd);
                r = r * (2.0 - dd * r);
                r = r * (2.0 - dd * r);
                scn = alpha >= 0.0 ? r : -r;
            }
            vn_i = (i == s + 2) ? 1.0 : ((i > s + 2 && i < n) ? x_i * scn : 0.0);
        }
        // (the record of step -1 is deferred until after this step's exchange: see eigh_tridiag_cluster4_kernel)
        if (s >= 0 && p == ((s + 1) & 1)) {
            if (i == s + 1) { d[s + 1] = x_i; e[s + 1] = betan; tau[s + 1] = tn; }
            if (i >= s + 2 && i < n) A[(size_t)(s + 1) * n + i] = vn_i;
        }
        if (s == n - 2) break;
        if (i < 16 * ECS_LD) { vperm[0][pi] = v_i; vperm[1][pi] = w_i; vperm[2][pi] = vn_i; }
        __syncthreads();
        ECS_T(0);
        const int lo = s + 2;                              // first live row / column
        const int ri0 = (lo - rs + 15) >> 4;               // first live row slot of this row class
        const double* pv0 = &vperm[0][rs * ECS_LD];
        const double* pv1 = &vperm[1][rs * ECS_LD];
        const double* pv2 = &vperm[2][rs * ECS_LD];
        const double2* pq0 = reinterpret_cast<const double2*>(pv0);
        const double2* pq1 = reinterpret_cast<const double2*>(pv1);
        const double2* pq2 = reinterpret_cast<const double2*>(pv2);
        double nc[ECS_NJ];                                 // v_next at this lane's columns (row-sum sweep)
#pragma unroll
        for (int j = 0; j < ECS_NJ; ++j) nc[j] = (cc[j] < n) ? vperm[2][(cc[j] & 15) * ECS_LD + (cc[j] >> 4)] : 0.0;
        const bool live0 = cc[0] >= lo && cc[0] < n;       // the LDS column of this lane is still part of the trailing matrix
        // ---- sweep 1: x = a - v_r w_c - w_r v_c (slots above the diagonal stay zero); column sums acc_c += x * vnext_r.  Row vectors as
        // 16-byte pairs, read once per column pair (the kernel is bound by its LDS instruction count: 390 per thread and step with 8-byte
        // reads and a crossbar shuffle for lane xor 16 - 18 us per step; a single rows-outer sweep needs 270 registers and spilled 164).
        {   // the LDS column (j = 0); skipped once this lane's column is eliminated (c < lo: no update reaches it, its v_next is zero)
            double acc = 0.0;
            if (live0) {
                const int pc = (cc[0] & 15) * ECS_LD + (cc[0] >> 4);
                const double vc = vperm[0][pc], wc = vperm[1][pc];
#pragma unroll
                for (int q = 0; q < (ECS_RI + 1) / 2; ++q) {
                    if (2 * q + 1 >= ri0) {
                        const double2 vr = pq0[q], wr = pq1[q], nr = pq2[q];
                        {
                            const int k = 2 * q;
                            double x = aL[k][rs][cl0];
                            x -= vr.x * wc + wr.x * vc;
                            if (k < 4 && k < kmin) x = 0.0;
                            aL[k][rs][cl0] = x;
                            acc += x * nr.x;
                        }
                        if (2 * q + 1 < ECS_RI) {
                            const int k = 2 * q + 1;
                            double x = aL[k < ECS_RI ? k : 0][rs][cl0];
                            x -= vr.y * wc + wr.y * vc;
                            if (k < 4 && k < kmin) x = 0.0;
                            aL[k < ECS_RI ? k : 0][rs][cl0] = x;
                            acc += x * nr.y;
                        }
                    }
                }
            }
            accb[rs][cl0] = acc;
        }
#pragma unroll
        for (int h = 0; h < 3; ++h) {                      // register columns in pairs (2h + 1, 2h + 2)
            const int j1 = 2 * h + 1, j2 = 2 * h + 2;
            const bool ok1 = cc[j1] < n, ok2 = cc[j2] < n;
            const int pc1 = ok1 ? (cc[j1] & 15) * ECS_LD + (cc[j1] >> 4) : 0, pc2 = ok2 ? (cc[j2] & 15) * ECS_LD + (cc[j2] >> 4) : 0;
            const double vc1 = ok1 ? vperm[0][pc1] : 0.0, wc1 = ok1 ? vperm[1][pc1] : 0.0;
            const double vc2 = ok2 ? vperm[0][pc2] : 0.0, wc2 = ok2 ? vperm[1][pc2] : 0.0;
            double acc1 = 0.0, acc2 = 0.0;
#pragma unroll
            for (int q = 2 * j1; q < (ECS_RI + 1) / 2; ++q) {   // row pairs (2q, 2q + 1) from row slot 4 j1 on
                if (2 * q + 1 >= ri0) {
                    const double2 vr = pq0[q], wr = pq1[q], nr = pq2[q];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int ri = 2 * q + hh, k = ri - 4 * j1;
                        if (ri < ECS_RI) {
                            const double vrr = hh ? vr.y : vr.x, wrr = hh ? wr.y : wr.x, nrr = hh ? nr.y : nr.x;
                            double x = a[ecs_off(j1) + (k < ecs_len(j1) ? k : 0)];
                            x -= vrr * wc1 + wrr * vc1;
                            if (k < 4 && k < kmin) x = 0.0;
                            a[ecs_off(j1) + (k < ecs_len(j1) ? k : 0)] = x;
                            acc1 += x * nrr;
                            if (k >= 4) {                  // the same row in column j2 (its slot k - 4)
                                const int k2 = k - 4 < 0 ? 0 : k - 4;
                                double y = a[ecs_off(j2) + k2];
                                y -= vrr * wc2 + wrr * vc2;
                                if (k2 < 4 && k2 < kmin) y = 0.0;
                                a[ecs_off(j2) + k2] = y;
                                acc2 += y * nrr;
                            }
                        }
                    }
                }
            }
            accb[rs][cl0 + 32 * j1] = acc1;
            accb[rs][cl0 + 32 * j2] = acc2;
        }
        ECS_T(1);
        // ---- the next pivot column (column lo from the diagonal down = pivot row of the full matrix) is held by 16 threads of the owning
        // workgroup, 27 entries each: they hand it to LDS, one thread per row publishes it behind the barrier below (27 tagged stores per
        // holder took 6.7 k of the step's 39 k clocks)
        const unsigned tag = (unsigned)(s + 3);
        u32x4* xp = xch + (size_t)((s + 1) & 1) * 4 * EG_MAXN;
        const bool owner = p == (lo & 1);
        if (owner && cl0 == (((lo - p) >> 1) & 31)) {
            const int js = ((lo - p) >> 1) >> 5;           // 0 or 1 while lo < 128 (the host hands over long before)
            if (js == 0) {
#pragma unroll
                for (int k = 0; k < ECS_RI; ++k) prow[rs + 16 * k] = aL[k][rs][cl0];
            } else {
#pragma unroll
                for (int k = 0; k < ecs_len(1); ++k) prow[rs + 16 * (4 + k)] = a[ecs_off(1) + k];
            }
        }
        ECS_T(2);
        // ---- sweep 2: row sums over this lane's columns (the diagonal element belongs to the column sum only), reduce-scattered over the
        // 32 column classes of the half wave as they appear: stage A pairs row slots (t, t + 16) across lane xor 16 (v_permlane16_swap), B
        // (t, t + 8) across xor 8 (row_ror:8), C / D / E the two mirrors and the quad swap - depth-first, one pending value per level
        // (all sixteen stage-A values alive at once: 189 spilled registers, 40 us per step)
        double u1;
        {
            const bool hi16 = (cl0 & 16) != 0, b3 = (cl0 & 8) != 0, b2 = (cl0 & 4) != 0, b1 = (cl0 & 2) != 0, b0_ = (cl0 & 1) != 0;
            auto rowsum = [&](int ri) {                    // ri is a compile-time constant at every call site (full unroll)
                double r = 0.0;
#pragma unroll
                for (int j = 0; j < ECS_NJ; ++j) {
                    const int k = ri - 4 * j;
                    if (ri < ECS_RI && k >= 0 && k < ecs_len(j)) {
                        const int kk = k < 0 ? 0 : k;
                        const double x = (j == 0) ? (live0 ? aL[kk][rs][cl0] : 0.0) : a[(j == 0 ? 0 : ecs_off(j)) + kk];
                        const double term = x * nc[j];
                        r += (kk < 4 && kk == kdiag) ? 0.0 : term;
                    }
                }
                return r;
            };
            double u2[2];
#pragma unroll
            for (int e_ = 0; e_ < 2; ++e_) {
                double u4[2];
#pragma unroll
                for (int d_ = 0; d_ < 2; ++d_) {
                    double u8[2];
#pragma unroll
                    for (int c_ = 0; c_ < 2; ++c_) {
                        double u16[2];
#pragma unroll
                        for (int b_ = 0; b_ < 2; ++b_) {
                            const int t = e_ + 2 * d_ + 4 * c_ + 8 * b_;
                            const double r1 = rowsum(t), r2 = rowsum(t + 16);
                            const double send = hi16 ? r1 : r2, keep = hi16 ? r2 : r1;
                            u16[b_] = keep + lane_xor16(send);
                        }
                        const double send = b3 ? u16[0] : u16[1], keep = b3 ? u16[1] : u16[0];
                        u8[c_] = keep + dpp_move<0x128>(send);                                          // row_ror:8
                    }
                    const double send = b2 ? u8[0] : u8[1], keep = b2 ? u8[1] : u8[0];
                    u4[d_] = keep + dpp_move<0x141>(send);                                              // row_half_mirror
                }
                const double send = b1 ? u4[0] : u4[1], keep = b1 ? u4[1] : u4[0];
                u2[e_] = keep + dpp_move<0x1B>(send);                                                   // quad_perm [3,2,1,0]
            }
            const double send = b0_ ? u2[0] : u2[1], keep = b0_ ? u2[1] : u2[0];
            u1 = keep + dpp_move<0xB1>(send);                                                           // quad_perm [1,0,3,2]
        }
        ECS_T(3);
        // lane cl0 now holds this workgroup's row sum of row slot ri = cl0
        {
            const int r = rs + 16 * cl0;
            if (cl0 < ECS_RI && r >= lo && r < n) st_tagged(keep_, &xp[(1 + p) * EG_MAXN + r], u1, tag);
        }
        __syncthreads();
        if (tid < 32 * ECS_NJ) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += accb[q][tid];
            const int c = p + ECS_P * tid;
            if (c >= lo && c < n) st_tagged(keep_, &xp[c], t, tag);
        }
        if (owner && i >= lo && i < n) st_tagged(keep_, &xp[3 * EG_MAXN + i], prow[i], tag);
        ECS_T(4);
        const bool need = (i >= lo) && (i < n);
        if (__any(need)) {
            u32x4 q0, q1, q2, q3;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                if (need) {
                    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
                                 : "v"(&xp[i]), "v"(&xp[EG_MAXN + i]), "v"(&xp[2 * EG_MAXN + i]), "v"(&xp[3 * EG_MAXN + i]) : "memory");
                    ok = (q0.y == tag) && (q0.w == tag) && (q1.y == tag) && (q1.w == tag) && (q2.y == tag) && (q2.w == tag) && (q3.y == tag) && (q3.w == tag);
                }
                if (__all(ok)) break;
                if (++spins > EC_SPIN_LIMIT) {             // never hang the device; eigh_tridiag_repair_kernel redoes this matrix
                    if (i < n) d[i] = __builtin_nan("");
                    if (tid == 0) ws.flag[b] = 1;
                    return;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (need) { p_i = (tagged_value(q0) + tagged_value(q1)) + tagged_value(q2); col_i = tagged_value(q3); }
        }
        ECS_T(5);
        if (s == -1 && p == 0) {                             // deferred record of step -1
            if (i == 0) { d[0] = x_i; e[0] = betan; tau[0] = tn; }
            if (i >= 1 && i < n) A[i] = vn_i;
        }
        v_i = vn_i;
        tk = tn;
        if (s == s_stop) { s_done = s; break; }
    }
#ifdef ECS_PROF
    if (tid == 0 && g == 0) for (int q_ = 0; q_ < 8; ++q_) ecs_prof[q_] = pacc_[q_];
#endif
    if (s_done >= -1 && s_done == s_stop) {
        // hand-over to the second stage (full storage): the trailing block (rows / columns >= s_stop + 2) back to A, BOTH triangles; state to ws.zt
        int lbv = s_stop + 2;
        asm volatile("" : "+v"(lbv));                      // (not an invariant to hoist above the step loop: see eigh_tridiag_cluster2_kernel)
        const int lb = lbv;
#pragma unroll
        for (int k = 0; k < ECS_RI; ++k) {
            const int r = rs + 16 * k, c = cc[0];
            if (k >= kmin && r >= lb && r < n && c >= lb && c < n) {
                const double x = aL[k][rs][cl0];
                A[(size_t)r * n + c] = x;
                A[(size_t)c * n + r] = x;
            }
        }
#pragma unroll
        for (int j = 1; j < ECS_NJ; ++j)
#pragma unroll
            for (int k = 0; k < ecs_len(j); ++k) {
                const int r = rs + 16 * (4 * j + k), c = cc[j];
                if (k >= kmin && r >= lb && r < n && c >= lb && c < n) {
                    const double x = a[ecs_off(j) + k];
                    A[(size_t)r * n + c] = x;
                    A[(size_t)c * n + r] = x;
                }
            }
        if (p == 0 && i < n) {
            double* st = ws.zt + (size_t)b * n * EG_MAXN;
            st[i] = v_i; st[EG_MAXN + i] = p_i; st[2 * EG_MAXN + i] = col_i;
            if (i == 0) st[3 * EG_MAXN] = tk;
        }
    }
}

// ------------------------------------------------------------------------------------------ e1, second cluster stage: 2 workgroups per matrix
// Round 4.  The four-workgroup kernel above needs a quarter of a CU's registers per 105 columns of a 420 x 420 matrix, whatever is left
// of it: 64 matrices fill the chip, a batch of 256 takes four rounds of the same latency chain.  Once the trailing block has shrunk
// to EC2_M = 320 rows it fits TWO register files (column-cyclic over two workgroups: 160 column slots = 4 register columns + 1 LDS
// column per lane, 20 row slots of 16): 128 matrices per launch, half as many rounds for the steps from 320 down to the hand-over to
// the single-workgroup kernel at 256.  Same algorithm, exchange protocol and hand-over state as eigh_tridiag_cluster4_kernel; indices are
// LOCAL to the trailing block (global row = base + local row), the state comes from the previous stage through ws.zt / A exactly as
// eigh_tridiag_mid_kernel takes it.  Tags continue the step numbering of the first stage (the exchange slots are zeroed once per call).
#define EC2_P 2
#define EC2_RI 20
#define EC2_NR 4
#define EC2_M (16 * EC2_RI)
__global__ __launch_bounds__(512) void eigh_tridiag_cluster2_kernel(double* __restrict__ Aall, int n, int b0, int Bc, EighWs ws, int s_first, int s_stop,
                                                                    int fail_every) {
    const int keep_ = ec_keep_flag();              // exchange stores may stay in the XCD's L2 (probed once per device)
    extern __shared__ double ec2_col[];                    // aL[EC2_RI][16][32]: the fifth column slot of every lane
    __shared__ __attribute__((aligned(16))) double vperm[3][EC2_M + 64];   // v, w, v_next at [(r & 15) * EC2_RI + (r >> 4)]
    __shared__ double vnat[EC2_M], wnat[EC2_M];
    __shared__ double accb[16][32 * (EC2_NR + 1)];
    __shared__ double red0[8], red1[8];
    __shared__ double s_alpha, s_ppiv;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int slot = g >> 3, p = slot & 1, mloc = (g & 7) + 8 * (slot >> 1);
    if (mloc >= Bc) return;
    const int b = b0 + mloc;
    if (ws.flag[b] != 0) return;                           // the first stage gave up on this matrix
    if (fail_every < 0 && b % (-fail_every) == 0) {        // test hook (NELE_EIGH_FAIL_EVERY=-k): this stage gives up on every k-th matrix
        if (tid == 0 && p == 0) ws.flag[b] = 2;
        return;
    }
    const int base = s_first + 1, m = n - base;            // trailing block = rows / columns base .. n-1, m <= EC2_M
    double* A = Aall + (size_t)b * n * n;
    double* d = ws.d + (size_t)b * n;
    double* e = ws.e + (size_t)b * n;
    double* tau = ws.tau + (size_t)b * n;
    u32x4* xch = reinterpret_cast<u32x4*>(ws.xch) + (size_t)b * EG_XCH * EG_MAXN;
    double (*aL)[16][32] = reinterpret_cast<double (*)[16][32]>(ec2_col);

    const int cl0 = lane & 31, rs = 2 * wv + (lane >> 5);
    int cc[EC2_NR + 1];                                    // local columns of this lane: p + 2 (cl0 + 32 j)
#pragma unroll
    for (int j = 0; j <= EC2_NR; ++j) cc[j] = p + EC2_P * (cl0 + 32 * j);
    double a[EC2_NR][EC2_RI];
#pragma unroll
    for (int ri = 0; ri < EC2_RI; ++ri) {
        const int r = rs + 16 * ri;
#pragma unroll
        for (int j = 0; j < EC2_NR; ++j) a[j][ri] = (r < m && cc[j] < m) ? A[(size_t)(base + r) * n + base + cc[j]] : 0.0;
        aL[ri][rs][cl0] = (r < m && cc[EC2_NR] < m) ? A[(size_t)(base + r) * n + base + cc[EC2_NR]] : 0.0;
    }
    const int il = tid, i = base + tid;                    // vector element owned by this thread: local / global index
    const bool own = il < m;
    const int pi = (il & 15) * EC2_RI + (il >> 4);         // permuted slot (il < EC2_M)
    double v_i = 0.0, tk = 0.0, p_i = 0.0, col_i = 0.0;
    {
        const double* st = ws.zt + (size_t)b * n * EG_MAXN;
        if (own) { v_i = st[i]; p_i = st[EG_MAXN + i]; col_i = st[2 * EG_MAXN + i]; }
        tk = st[3 * EG_MAXN];
    }
    __syncthreads();
    for (int s = s_first; s <= n - 2; ++s) {
        const bool in = own && (i >= s + 1);
        double w_i = 0.0, wpiv = 0.0;
        {
            if (own && i == s + 1) s_ppiv = p_i;
            double pv = in ? tk * p_i * v_i : 0.0;
            pv = wave_sum_dpp(pv);
            if (lane == 0) red0[wv] = pv;
            __syncthreads();
            pv = ((red0[0] + red0[1]) + (red0[2] + red0[3])) + ((red0[4] + red0[5]) + (red0[6] + red0[7]));
            const double al = -0.5 * tk * pv;
            w_i = in ? tk * p_i + al * v_i : 0.0;
            wpiv = tk * s_ppiv + al;
        }
        const double x_i = in ? col_i - v_i * wpiv - w_i : 0.0;
        double tn = 0.0, betan = 0.0, vn_i = 0.0;
        if (s + 2 <= n - 1) {
            if (own && i == s + 2) s_alpha = x_i;
            double ss = (own && i >= s + 3) ? x_i * x_i : 0.0;
            ss = wave_sum_dpp(ss);
            if (lane == 0) red1[wv] = ss;
            __syncthreads();
            ss = ((red1[0] + red1[1]) + (red1[2] + red1[3])) + ((red1[4] + red1[5]) + (red1[6] + red1[7]));
            const double alpha = s_alpha;
            double scn = 0.0;
            betan = alpha;
            if (ss > 0.0) {
                const double s2 = alpha * alpha + ss, aa = fabs(alpha);
                double y = __builtin_amdgcn_rsq(s2);
                y = y * (1.5 - 0.5 * s2 * y * y);
                y = y * (1.5 - 0.5 * s2 * y * y);
                const double nrm = s2 * y;
                betan = alpha >= 0.0 ? -nrm : nrm;
                tn = 1.0 + aa * y;
                const double dd = aa + nrm;
                double r = __builtin_amdgcn_rcp(dd);
                r = r * (2.0 - dd * r);
                r = r * (2.0 - dd * r);
                scn = alpha >= 0.0 ? r : -r;
            }
            vn_i = (own && i == s + 2) ? 1.0 : ((own && i > s + 2) ? x_i * scn : 0.0);
        }
        if (p == ((s + 1) & 1)) {                          // one of the two workgroups records the step (both hold the same vectors)
            if (own && i == s + 1) { d[s + 1] = x_i; e[s + 1] = betan; tau[s + 1] = tn; }
            if (own && i >= s + 2) A[(size_t)(s + 1) * n + i] = vn_i;
        }
        if (s == n - 2) break;
        if (il < EC2_M) {
            vnat[il] = v_i; wnat[il] = w_i;
            vperm[0][pi] = v_i; vperm[1][pi] = w_i; vperm[2][pi] = vn_i;
        }
        __syncthreads();
        // ---- fused pass: x = a - v_r w_c - w_r v_c ; acc_c += x * vnext_r  (4 register columns + 1 LDS column; dead rows / columns are updated harmlessly)
        const int lo = s + 2 - base;                       // first live local index
        {
            // the columns go in groups of two (as in eigh_tridiag_mid_kernel): v_c, w_c and acc_c of all five columns beside the 80-double block
            // spilled 111 registers; the row vectors are re-read from LDS per group
            const int ri0 = (lo - rs + 15) >> 4;
            const double2* pv0 = reinterpret_cast<const double2*>(&vperm[0][rs * EC2_RI]);
            const double2* pv1 = reinterpret_cast<const double2*>(&vperm[1][rs * EC2_RI]);
            const double2* pv2 = reinterpret_cast<const double2*>(&vperm[2][rs * EC2_RI]);
#pragma unroll
            for (int h = 0; h < EC2_NR / 2; ++h) {
                double acc[2], vc[2], wc[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = 2 * h + u;
                    const bool ok = cc[j] < m;
                    acc[u] = 0.0;
                    vc[u] = ok ? vnat[min(cc[j], EC2_M - 1)] : 0.0;
                    wc[u] = ok ? wnat[min(cc[j], EC2_M - 1)] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < EC2_RI / 2; ++q) {
                    if (2 * q + 1 >= ri0) {                // rows rs + 16 (2q), rs + 16 (2q + 1)
                        const double2 vr = pv0[q], wr = pv1[q], nr = pv2[q];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int j = 2 * h + u;
                            double x0 = a[j][2 * q], x1 = a[j][2 * q + 1];
                            x0 -= vr.x * wc[u] + wr.x * vc[u];
                            x1 -= vr.y * wc[u] + wr.y * vc[u];
                            a[j][2 * q] = x0; a[j][2 * q + 1] = x1;
                            acc[u] += x0 * nr.x + x1 * nr.y;
                        }
                    }
                }
                accb[rs][cl0 + 32 * (2 * h)] = acc[0];
                accb[rs][cl0 + 32 * (2 * h + 1)] = acc[1];
            }
            {                                              // the LDS column
                const bool ok = cc[EC2_NR] < m;
                const double vcl = ok ? vnat[min(cc[EC2_NR], EC2_M - 1)] : 0.0, wcl = ok ? wnat[min(cc[EC2_NR], EC2_M - 1)] : 0.0;
                double accl = 0.0;
#pragma unroll
                for (int q = 0; q < EC2_RI / 2; ++q) {
                    if (2 * q + 1 >= ri0) {
                        const double2 vr = pv0[q], wr = pv1[q], nr = pv2[q];
                        double u0 = aL[2 * q][rs][cl0], u1 = aL[2 * q + 1][rs][cl0];
                        u0 -= vr.x * wcl + wr.x * vcl;
                        u1 -= vr.y * wcl + wr.y * vcl;
                        aL[2 * q][rs][cl0] = u0; aL[2 * q + 1][rs][cl0] = u1;
                        accl += u0 * nr.x + u1 * nr.y;
                    }
                }
                accb[rs][cl0 + 32 * EC2_NR] = accl;
            }
        }
        const unsigned tag = (unsigned)(s + 3);
        u32x4* xp = xch + (size_t)((s + 1) & 1) * 2 * EG_MAXN;
        if (rs == (lo & 15)) {                             // every workgroup publishes its part of pivot row s + 2 (local row lo)
            const int rq = lo >> 4;
            double rv[EC2_NR] = {0.0, 0.0, 0.0, 0.0};
            switch (rq) {
#define EC2_CASE(k) case k: _Pragma("unroll") for (int j = 0; j < EC2_NR; ++j) rv[j] = a[j][k]; break;
                EC2_CASE(0) EC2_CASE(1) EC2_CASE(2) EC2_CASE(3) EC2_CASE(4) EC2_CASE(5) EC2_CASE(6) EC2_CASE(7) EC2_CASE(8) EC2_CASE(9)
                EC2_CASE(10) EC2_CASE(11) EC2_CASE(12) EC2_CASE(13) EC2_CASE(14) EC2_CASE(15) EC2_CASE(16) EC2_CASE(17) EC2_CASE(18) EC2_CASE(19)
#undef EC2_CASE
            }
#pragma unroll
            for (int j = 0; j < EC2_NR; ++j)
                if (cc[j] >= lo && cc[j] < m) st_tagged(keep_, &xp[EG_MAXN + cc[j]], rv[j], tag);
            if (cc[EC2_NR] >= lo && cc[EC2_NR] < m) st_tagged(keep_, &xp[EG_MAXN + cc[EC2_NR]], aL[rq][rs][cl0], tag);
        }
        __syncthreads();
        if (tid < 32 * (EC2_NR + 1)) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += accb[q][tid];
            const int c = p + EC2_P * tid;                 // accb column tid = cl0 + 32 j  <->  local column p + 2 (cl0 + 32 j)
            if (c >= lo && c < m) st_tagged(keep_, &xp[c], t, tag);
        }
        const bool need = own && (il >= lo);
        if (__any(need)) {
            u32x4 qp, qc;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                if (need) {
                    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(qp), "=&v"(qc) : "v"(&xp[il]), "v"(&xp[EG_MAXN + il]) : "memory");
                    ok = (qp.y == tag) && (qp.w == tag) && (qc.y == tag) && (qc.w == tag);
                }
                if (__all(ok)) break;
                if (++spins > EC_SPIN_LIMIT) {             // never hang the device (see eigh_tridiag_cluster4_kernel): give up, and
                    if (own) d[i] = __builtin_nan("");     // eigh_tridiag_repair2_kernel restarts this matrix from the first stage's hand-over
                    if (tid == 0) ws.flag[b] = 2;
                    return;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (need) { p_i = tagged_value(qp); col_i = tagged_value(qc); }
        }
        v_i = vn_i;
        tk = tn;
        if (s == s_stop) break;
    }
    if (s_stop >= s_first && s_stop < n - 2) {
        // hand-over to the single-workgroup kernel (outside the step loop: inside it the compiler kept the 100 store addresses live across
        // the loop - 92 spilled registers): trailing block (rows / columns >= s_stop + 2) back to A in place, per-element state to ws.zt
        int lbv = s_stop + 2 - base;
        asm volatile("" : "+v"(lbv));                      // (not a loop invariant the optimiser may hoist above the loop)
        const int lb = lbv;
#pragma unroll
        for (int ri = 0; ri < EC2_RI; ++ri) {
            const int r = rs + 16 * ri;
            if (r >= lb && r < m) {
#pragma unroll
                for (int j = 0; j < EC2_NR; ++j)
                    if (cc[j] >= lb && cc[j] < m) A[(size_t)(base + r) * n + base + cc[j]] = a[j][ri];
                if (cc[EC2_NR] >= lb && cc[EC2_NR] < m) A[(size_t)(base + r) * n + base + cc[EC2_NR]] = aL[ri][rs][cl0];
            }
        }
        if (p == 0 && own) {
            double* st = ws.zt + (size_t)b * n * EG_MAXN;
            st[i] = v_i; st[EG_MAXN + i] = p_i; st[2 * EG_MAXN + i] = col_i;
            if (il == 0) st[3 * EG_MAXN] = tk;
        }
    }
}

// ------------------------------------------------------------------------------------------ e1, tail
// The last ET_M = 128 steps of the tridiagonalisation (and whole matrices with n <= 128) in ONE workgroup per matrix with the trailing
// block in LDS: the cluster kernel's step costs 5 us whatever is left of the matrix (exchange, peer wait), and a 128 x 128 block needs
// neither.  Same recurrences as the cluster kernel; state handed over through ws.zt (v, p = A v, pivot column, tau) when s_first >= 0.
// grid B, block 512, dynamic LDS m * m doubles.
#define ET_M 128
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's outstanding GLOBAL stores (vmcnt(0)); in the
// step loop below those are the reflector row and d / e / tau, which nothing in the kernel reads back - waiting for their
// acknowledgement at every barrier is a store round trip per step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(512) void eigh_tridiag_tail_kernel(double* __restrict__ Aall, int n, EighWs ws, int s_first) {
    extern __shared__ double et_sm[];                    // Am[m][m]
    __shared__ double vs[ET_M], wv[ET_M], vn[ET_M], part[8][ET_M];
    __shared__ double red0[8], red1[8];
    __shared__ double s_alpha, s_ppiv;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wvi = tid >> 6;
    if (ws.flag[b] != 0) return;                         // the cluster kernel gave up on this matrix: eigh_tridiag_repair_kernel redoes it
    const int base = s_first + 1, m = n - base;          // trailing block = rows / columns base .. n-1
    double* A = Aall + (size_t)b * n * n;
    double* d = ws.d + (size_t)b * n;
    double* e = ws.e + (size_t)b * n;
    double* tau = ws.tau + (size_t)b * n;
    double* Am = et_sm;
    for (int idx = tid; idx < m * m; idx += 512) {
        const int r = idx / m, c = idx - r * m;
        Am[idx] = A[(size_t)(base + r) * n + base + c];
    }
    const int i = base + tid;                            // vector element owned by this thread (tid < m)
    const bool own = tid < m;
    double v_i = 0.0, tk = 0.0, p_i = 0.0, col_i = 0.0;
    if (s_first >= 0) {
        const double* st = ws.zt + (size_t)b * n * EG_MAXN;
        if (own) { v_i = st[i]; p_i = st[EG_MAXN + i]; col_i = st[2 * EG_MAXN + i]; }
        tk = st[3 * EG_MAXN];
    } else if (own) col_i = A[i];                        // whole matrix: column 0
    __syncthreads();
    const int c2 = (tid & 63) * 2, q = tid >> 6;         // pass: columns c2, c2 + 1 (m = ET_M is even), row group q
#ifdef EC_PROF
    long long tacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = clock64(), t1;
#endif
    for (int s = s_first; s <= n - 2; ++s) {
        const bool in = own && (i >= s + 1);
        double w_i = 0.0, wpiv = 0.0;
        if (s >= 0) {
            if (own && i == s + 1) s_ppiv = p_i;
            double pv = in ? tk * p_i * v_i : 0.0;
            pv = wave_sum_dpp(pv);
            if (lane == 0) red0[wvi] = pv;
            lds_barrier();
            pv = ((red0[0] + red0[1]) + (red0[2] + red0[3])) + ((red0[4] + red0[5]) + (red0[6] + red0[7]));
            EC_T(0);
            const double al = -0.5 * tk * pv;
            w_i = in ? tk * p_i + al * v_i : 0.0;
            wpiv = tk * s_ppiv + al;
        }
        const double x_i = in ? col_i - v_i * wpiv - w_i : 0.0;
        double tn = 0.0, betan = 0.0, vn_i = 0.0;
        if (s + 2 <= n - 1) {
            if (own && i == s + 2) s_alpha = x_i;
            double ss = (own && i >= s + 3) ? x_i * x_i : 0.0;
            ss = wave_sum_dpp(ss);
            if (lane == 0) red1[wvi] = ss;
            lds_barrier();
            ss = ((red1[0] + red1[1]) + (red1[2] + red1[3])) + ((red1[4] + red1[5]) + (red1[6] + red1[7]));
            EC_T(1);
            const double alpha = s_alpha;
            double scn = 0.0;
            betan = alpha;
            if (ss > 0.0) {
                const double s2 = alpha * alpha + ss, aa = fabs(alpha);
                double y = __builtin_amdgcn_rsq(s2);
                y = y * (1.5 - 0.5 * s2 * y * y);
                y = y * (1.5 - 0.5 * s2 * y * y);
                const double nrm = s2 * y;
                betan = alpha >= 0.0 ? -nrm : nrm;
                tn = 1.0 + aa * y;
                const double dd = aa + nrm;
                double r = __builtin_amdgcn_rcp(dd);
                r = r * (2.0 - dd * r);
                r = r * (2.0 - dd * r);
                scn = alpha >= 0.0 ? r : -r;
            }
            vn_i = (own && i == s + 2) ? 1.0 : ((own && i > s + 2) ? x_i * scn : 0.0);
        }
        if (own && i == s + 1) { d[s + 1] = x_i; e[s + 1] = betan; tau[s + 1] = tn; }
        if (own && i >= s + 2) A[(size_t)(s + 1) * n + i] = vn_i;
        if (s == n - 2) break;
        if (own) { vs[tid] = v_i; wv[tid] = w_i; vn[tid] = vn_i; }
        lds_barrier();
        EC_T(2);
        // ---- fused pass on the LDS block: x = a - v_r w_c - w_r v_c ; acc_c += x * vnext_r   (rows / columns >= s + 2)
        const int lo = s + 2 - base;                     // first live local index (>= 0)
        // thread = (column pair c2, row group q of 8): rows lo + q, lo + q + 8, ..., eight at a time; one 16-byte LDS access serves two
        // columns and the row's three vector values are shared by them (the pass is bound by LDS instruction issue)
        double acc0 = 0.0, acc1 = 0.0;
        if (c2 + 1 >= lo) {
            const double vc0 = vs[c2], wc0 = wv[c2], vc1 = vs[c2 + 1], wc1 = wv[c2 + 1];
            for (int r = lo + q; r < m; r += 64) {
                double2 a[8];
                double vr[8], wr[8], nr[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int rr = min(r + 8 * u, m - 1);
                    a[u] = *reinterpret_cast<const double2*>(&Am[rr * m + c2]); vr[u] = vs[rr]; wr[u] = wv[rr]; nr[u] = vn[rr];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (r + 8 * u < m) {
                        const double x0 = a[u].x - (vr[u] * wc0 + wr[u] * vc0), x1 = a[u].y - (vr[u] * wc1 + wr[u] * vc1);
                        *reinterpret_cast<double2*>(&Am[(r + 8 * u) * m + c2]) = make_double2(x0, x1);
                        acc0 += x0 * nr[u];
                        acc1 += x1 * nr[u];
                    }
                }
            }
        }
        EC_T(3);
        part[q][c2] = acc0;
        part[q][c2 + 1] = acc1;
        lds_barrier();
        EC_T(4);
        if (own) {
            p_i = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + ((part[4][tid] + part[5][tid]) + (part[6][tid] + part[7][tid]));
            col_i = (tid >= lo) ? Am[lo * m + tid] : 0.0;          // pivot row s + 2 of the updated block
        }
        EC_T(5);
        v_i = vn_i;
        tk = tn;
        // the next iteration's first barrier orders these reads against its LDS writes (s_ppiv / red0 are written before it, but
        // nothing reads them after this point of the current iteration)
    }
#ifdef EC_PROF
    if (tid == 0 && b == 0) for (int j = 0; j < 9; ++j) ws.lamp[j] = (double)tacc[j];
#endif
}
#endif  // NELE_AB

// ------------------------------------------------------------------------------------------ e1, middle + tail in registers
// The last EM_M = 224 steps in ONE workgroup per matrix with the trailing block in REGISTERS (a thread holds 7 columns x 14 rows): no
// cross-workgroup exchange, so a step costs the tail kernel's 2.5 us on one CU instead of the cluster kernel's 4.6 us on four - the
// cluster kernel hands over as soon as what is left of the matrix fits one register file (after 194 of 418 steps at n = 420).  Same
// recurrences and the same hand-over state as eigh_tridiag_tail_kernel; the fused pass is the cluster kernel's (rows rs + 16 ri, columns
// cl + 32 cj, vectors in the permuted LDS layout so that a lane's rows are contiguous).  grid B, block 512.
#define EM_RI 14
#define EM_NC 7
#define EM_M (16 * EM_RI)
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(512) void eigh_tridiag_mid_kernel(double* __restrict__ Aall, int n, EighWs ws, int s_first) {
    __shared__ __attribute__((aligned(16))) double vperm[3][EM_M + 64];   // v, w, v_next at [(r & 15) * (EM_RI) + (r >> 4)]
    __shared__ double vnat[EM_M], wnat[EM_M];
    __shared__ double accb[16][EM_M];
    __shared__ double prow[EM_M];
    __shared__ double red0[8], red1[8];
    __shared__ double s_alpha, s_ppiv;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wvi = tid >> 6;
    if (ws.flag[b] != 0) return;                         // the cluster kernel gave up on this matrix: eigh_tridiag_repair_kernel redoes it
    const int base = s_first + 1, m = n - base;          // trailing block = rows / columns base .. n-1, m <= EM_M
    double* A = Aall + (size_t)b * n * n;
    double* d = ws.d + (size_t)b * n;
    double* e = ws.e + (size_t)b * n;
    double* tau = ws.tau + (size_t)b * n;
    const int cl0 = lane & 31, rs = 2 * wvi + (lane >> 5);
    double a[EM_NC][EM_RI];
#pragma unroll
    for (int cj = 0; cj < EM_NC; ++cj)
#pragma unroll
        for (int ri = 0; ri < EM_RI; ++ri) {
            const int r = rs + 16 * ri, c = cl0 + 32 * cj;
            a[cj][ri] = (r < m && c < m) ? A[(size_t)(base + r) * n + base + c] : 0.0;
        }
    const int i = base + tid;                            // vector element owned by this thread (tid < m)
    const bool own = tid < m;
    const int pi = (tid & 15) * EM_RI + (tid >> 4);      // permuted slot of local row tid (tid < EM_M)
    double v_i = 0.0, tk = 0.0, p_i = 0.0, col_i = 0.0;
    if (s_first >= 0) {
        const double* st = ws.zt + (size_t)b * n * EG_MAXN;
        if (own) { v_i = st[i]; p_i = st[EG_MAXN + i]; col_i = st[2 * EG_MAXN + i]; }
        tk = st[3 * EG_MAXN];
    } else if (own) col_i = A[i];                        // whole matrix: column 0
    __syncthreads();
    for (int s = s_first; s <= n - 2; ++s) {
        const bool in = own && (i >= s + 1);
        double w_i = 0.0, wpiv = 0.0;
        if (s >= 0) {
            if (own && i == s + 1) s_ppiv = p_i;
            double pv = in ? tk * p_i * v_i : 0.0;
            pv = wave_sum_dpp(pv);
            if (lane == 0) red0[wvi] = pv;
            lds_barrier();
            pv = ((red0[0] + red0[1]) + (red0[2] + red0[3])) + ((red0[4] + red0[5]) + (red0[6] + red0[7]));
            const double al = -0.5 * tk * pv;
            w_i = in ? tk * p_i + al * v_i : 0.0;
            wpiv = tk * s_ppiv + al;
        }
        const double x_i = in ? col_i - v_i * wpiv - w_i : 0.0;
        double tn = 0.0, betan = 0.0, vn_i = 0.0;
        if (s + 2 <= n - 1) {
            if (own && i == s + 2) s_alpha = x_i;
            double ss = (own && i >= s + 3) ? x_i * x_i : 0.0;
            ss = wave_sum_dpp(ss);
            if (lane == 0) red1[wvi] = ss;
            lds_barrier();
            ss = ((red1[0] + red1[1]) + (red1[2] + red1[3])) + ((red1[4] + red1[5]) + (red1[6] + red1[7]));
            const double alpha = s_alpha;
            double scn = 0.0;
            betan = alpha;
            if (ss > 0.0) {
                const double s2 = alpha * alpha + ss, aa = fabs(alpha);
                double y = __builtin_amdgcn_rsq(s2);
                y = y * (1.5 - 0.5 * s2 * y * y);
                y = y * (1.5 - 0.5 * s2 * y * y);
                const double nrm = s2 * y;
                betan = alpha >= 0.0 ? -nrm : nrm;
                tn = 1.0 + aa * y;
                const double dd = aa + nrm;
                double r = __builtin_amdgcn_rcp(dd);
                r = r * (2.0 - dd * r);
                r = r * (2.0 - dd * r);
                scn = alpha >= 0.0 ? r : -r;
            }
            vn_i = (own && i == s + 2) ? 1.0 : ((own && i > s + 2) ? x_i * scn : 0.0);
        }
        if (own && i == s + 1) { d[s + 1] = x_i; e[s + 1] = betan; tau[s + 1] = tn; }
        if (own && i >= s + 2) A[(size_t)(s + 1) * n + i] = vn_i;
        if (s == n - 2) break;
        if (tid < EM_M) {
            vnat[tid] = v_i; wnat[tid] = w_i;
            vperm[0][pi] = v_i; vperm[1][pi] = w_i; vperm[2][pi] = vn_i;
        }
        lds_barrier();
        // ---- fused pass: x = a - v_r w_c - w_r v_c ; acc_c += x * vnext_r   (rows >= s + 2; dead rows / columns are updated harmlessly)
        const int lo = s + 2 - base;                     // first live local index (>= 0)
        // the columns go in groups of two: 2 x 3 x EM_NC more live doubles (v_c, w_c, acc_c of every column) do not fit beside the block
        {
            const int ri0 = (lo - rs + 15) >> 4;
            const double2* pv0 = reinterpret_cast<const double2*>(&vperm[0][rs * EM_RI]);
            const double2* pv1 = reinterpret_cast<const double2*>(&vperm[1][rs * EM_RI]);
            const double2* pv2 = reinterpret_cast<const double2*>(&vperm[2][rs * EM_RI]);
            constexpr int CH = 2, NH = (EM_NC + CH - 1) / CH;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                double acc[CH], vc[CH], wc[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int cj = h * CH + u;
                    acc[u] = 0.0;
                    vc[u] = (cj < EM_NC) ? vnat[cl0 + 32 * (cj < EM_NC ? cj : 0)] : 0.0;
                    wc[u] = (cj < EM_NC) ? wnat[cl0 + 32 * (cj < EM_NC ? cj : 0)] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < EM_RI / 2; ++q) {
                    if (2 * q + 1 >= ri0) {              // rows rs + 16 (2q), rs + 16 (2q + 1)
                        const double2 vr = pv0[q], wr = pv1[q], nr = pv2[q];
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const int cj = h * CH + u;
                            if (cj < EM_NC) {
                                double x0 = a[cj][2 * q], x1 = a[cj][2 * q + 1];
                                x0 -= vr.x * wc[u] + wr.x * vc[u];
                                x1 -= vr.y * wc[u] + wr.y * vc[u];
                                a[cj][2 * q] = x0; a[cj][2 * q + 1] = x1;
                                acc[u] += x0 * nr.x + x1 * nr.y;
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (h * CH + u < EM_NC) accb[rs][cl0 + 32 * (h * CH + u)] = acc[u];
            }
        }
        if (rs == (lo & 15)) {                           // the lanes that hold pivot row s + 2 of the updated block publish it
            switch (lo >> 4) {
#define EM_CASE(k) case k: _Pragma("unroll") for (int cj = 0; cj < EM_NC; ++cj) prow[cl0 + 32 * cj] = a[cj][k]; break;
                EM_CASE(0) EM_CASE(1) EM_CASE(2) EM_CASE(3) EM_CASE(4) EM_CASE(5) EM_CASE(6) EM_CASE(7) EM_CASE(8) EM_CASE(9) EM_CASE(10) EM_CASE(11)
                EM_CASE(12) EM_CASE(13)
#undef EM_CASE
            }
        }
        lds_barrier();
        if (tid < EM_M) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += accb[q][tid];
            p_i = t;
            col_i = (tid >= lo) ? prow[tid] : 0.0;       // pivot row s + 2 of the updated block = the next pivot column
        }
        v_i = vn_i;
        tk = tn;
        // the next iteration's first barrier orders these reads against its LDS writes
    }
}
#endif  // NELE_AB

// The same with up to EX_E = 32 more rows / columns: the trailing block's FIRST E = m - 224 rows (the ones eliminated first) sit in an LDS
// strip S[E][256] (full rows: the E x E corner is stored twice, the off-diagonal block once), the other 224 x 224 in registers as above.
// While the strip is alive (the first E steps) a step also updates it in place - pass A: thread per column, column sums; pass B: thread
// group per row, the row sums that stand in for the transposed off-diagonal block - and takes the next pivot column from it.  The
// cluster kernel, whose step costs 7.2 us on FOUR CUs that no other kernel can share (registers full), hands over 32 steps earlier
// (LDS: 64 KB of strip + 52 KB of vectors and partial sums).
#ifndef EX_E
#define EX_E 32                       // (48 rows x 272 columns, 102 KB, measured: the step gains nothing more - 44.9 / 44.9 / 45.4 against 44.9 / 45.2 / 45.2 ms)
#define EX_LD 256
#endif
__global__ __launch_bounds__(512) void eigh_tridiag_midx_kernel(double* __restrict__ Aall, int n, EighWs ws, int s_first, int B, int c2_first) {
    extern __shared__ double em_strip[];                  // S[E][EX_LD]: rows 0 .. E-1 of the trailing block, all its columns
    __shared__ __attribute__((aligned(16))) double vperm[3][EM_M + 64];   // v, w, v_next of the register block at [(r & 15) * (EM_RI) + (r >> 4)]
    __shared__ double vL[EX_LD], wL[EX_LD], nL[EX_LD];    // the same vectors over the whole trailing block, natural order (strip passes)
    __shared__ double accA[2][EX_LD], accB[EX_E];
    __shared__ double vnat[EM_M], wnat[EM_M];
    __shared__ double accb[16][EM_M];
    __shared__ double prow[EM_M];
    __shared__ double red0[8], red1[8];
    __shared__ double s_alpha, s_ppiv;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wvi = tid >> 6;
    {
        // A matrix the cluster kernels gave up on (its workgroups were not co-resident within the spin limit - never on a GPU the launch
        // fits on) is redone HERE, by this workgroup, from what the cluster stage left intact (eigh_repair1 / eigh_repair2; the strip's
        // 64 KB are the scratch).  Until round 5 two fat repair kernels (B workgroups of 16 waves each) were launched for every batch just
        // to test this flag: 4.8 us alone, but 0.2 - 1.3 ms inside a training step, where such workgroups wait for whole CUs.
        const int fl = ws.flag[b];
        if (fl != 0) {
            static_assert(EG_REPAIR_SH <= EX_E * EX_LD, "the repair scratch must fit into the strip");
            if (fl == 1) eigh_repair1<512>(Aall + (size_t)b * n * n, n, ws, b, B, em_strip);
            else if (fl == 2 && c2_first >= 0) eigh_repair2<512>(Aall + (size_t)b * n * n, n, ws, b, B, c2_first, em_strip);
            return;
        }
    }
    const int base = s_first + 1, m = n - base;          // trailing block = rows / columns base .. n-1, m <= EM_M + EX_E
    const int E = max(m - EM_M, 0);                      // its first E rows / columns live in the LDS strip, the rest in registers
    double* S = em_strip;
    double* A = Aall + (size_t)b * n * n;
    double* d = ws.d + (size_t)b * n;
    double* e = ws.e + (size_t)b * n;
    double* tau = ws.tau + (size_t)b * n;
    const int cl0 = lane & 31, rs = 2 * wvi + (lane >> 5);
    double a[EM_NC][EM_RI];
#pragma unroll
    for (int cj = 0; cj < EM_NC; ++cj)
#pragma unroll
        for (int ri = 0; ri < EM_RI; ++ri) {
            const int r = E + rs + 16 * ri, c = E + cl0 + 32 * cj;
            a[cj][ri] = (r < m && c < m) ? A[(size_t)(base + r) * n + base + c] : 0.0;
        }
    for (int q = tid; q < E * EX_LD; q += 512) {
        const int er = q / EX_LD, ec = q - er * EX_LD;
        S[q] = (ec < m) ? A[(size_t)(base + er) * n + base + ec] : 0.0;
    }
    const int i = base + tid;                            // vector element owned by this thread (tid < m)
    const bool own = tid < m;
    const int rr = tid - E;                              // index inside the register block (0 <= rr < EM_M for its owners)
    const bool rown = rr >= 0 && rr < EM_M;
    const int pi = (rr & 15) * EM_RI + (rr >> 4);        // permuted slot of register-block row rr
    double v_i = 0.0, tk = 0.0, p_i = 0.0, col_i = 0.0;
    if (s_first >= 0) {
        const double* st = ws.zt + (size_t)b * n * EG_MAXN;
        if (own) { v_i = st[i]; p_i = st[EG_MAXN + i]; col_i = st[2 * EG_MAXN + i]; }
        tk = st[3 * EG_MAXN];
    } else if (own) col_i = A[i];                        // whole matrix: column 0
    __syncthreads();
    for (int s = s_first; s <= n - 2; ++s) {
        const bool in = own && (i >= s + 1);
        double w_i = 0.0, wpiv = 0.0;
        if (s >= 0) {
            if (own && i == s + 1) s_ppiv = p_i;
            double pv = in ? tk * p_i * v_i : 0.0;
            pv = wave_sum_dpp(pv);
            if (lane == 0) red0[wvi] = pv;
            lds_barrier();
            pv = ((red0[0] + red0[1]) + (red0[2] + red0[3])) + ((red0[4] + red0[5]) + (red0[6] + red0[7]));
            const double al = -0.5 * tk * pv;
            w_i = in ? tk * p_i + al * v_i : 0.0;
            wpiv = tk * s_ppiv + al;
        }
        const double x_i = in ? col_i - v_i * wpiv - w_i : 0.0;
        double tn = 0.0, betan = 0.0, vn_i = 0.0;
        if (s + 2 <= n - 1) {
            if (own && i == s + 2) s_alpha = x_i;
            double ss = (own && i >= s + 3) ? x_i * x_i : 0.0;
            ss = wave_sum_dpp(ss);
            if (lane == 0) red1[wvi] = ss;
            lds_barrier();
            ss = ((red1[0] + red1[1]) + (red1[2] + red1[3])) + ((red1[4] + red1[5]) + (red1[6] + red1[7]));
            const double alpha = s_alpha;
            double scn = 0.0;
            betan = alpha;
            if (ss > 0.0) {
                const double s2 = alpha * alpha + ss, aa = fabs(alpha);
                double y = __builtin_amdgcn_rsq(s2);
                y = y * (1.5 - 0.5 * s2 * y * y);
                y = y * (1.5 - 0.5 * s2 * y * y);
                const double nrm = s2 * y;
                betan = alpha >= 0.0 ? -nrm : nrm;
                tn = 1.0 + aa * y;
                const double dd = aa + nrm;
                double r = __builtin_amdgcn_rcp(dd);
                r = r * (2.0 - dd * r);
                r = r * (2.0 - dd * r);
                scn = alpha >= 0.0 ? r : -r;
            }
            vn_i = (own && i == s + 2) ? 1.0 : ((own && i > s + 2) ? x_i * scn : 0.0);
        }
        if (own && i == s + 1) { d[s + 1] = x_i; e[s + 1] = betan; tau[s + 1] = tn; }
        if (own && i >= s + 2) A[(size_t)(s + 1) * n + i] = vn_i;
        if (s == n - 2) break;
        if (rown) {
            vnat[rr] = v_i; wnat[rr] = w_i;
            vperm[0][pi] = v_i; vperm[1][pi] = w_i; vperm[2][pi] = vn_i;
        }
        if (E > 0 && tid < EX_LD) { vL[tid] = v_i; wL[tid] = w_i; nL[tid] = vn_i; }
        lds_barrier();
        // ---- fused pass: x = a - v_r w_c - w_r v_c ; acc_c += x * vnext_r   (rows >= s + 2; dead rows / columns are updated harmlessly)
        const int lo = s + 2 - base;                     // first live local index (>= 0)
        const int lor = max(lo - E, 0);                  // ... inside the register block
        // ---- strip pass A (rows lo .. E-1 of the strip, all columns): the same update, column sums acc_c += x * vnext_row.
        // thread = (column c, row parity); the E x E corner is stored in full, so its column sums cover both of its triangles
        if (lo < E) {
            // 512 threads for EX_LD columns x 2 row parities (EX_LD = 256: one thread per column and parity; wider strips: the columns
            // beyond 512 - EX_LD have one thread for all rows)
            const int par = tid >= EX_LD ? 1 : 0, c = tid - par * EX_LD;
            const int stp = (c < 512 - EX_LD) ? 2 : 1;
            double accc = 0.0;
            if (c < m) {
                const double vc_ = vL[c], wc_ = wL[c];
                for (int er = lo + (stp == 2 ? par : 0); er < E; er += stp) {
                    const double x = S[er * EX_LD + c] - (vL[er] * wc_ + wL[er] * vc_);
                    S[er * EX_LD + c] = x;
                    accc += x * nL[er];
                }
            }
            accA[par][c] = accc;
            if (par == 0 && stp == 1) accA[1][c] = 0.0;
        }
        // the columns go in groups of two: 2 x 3 x EM_NC more live doubles (v_c, w_c, acc_c of every column) do not fit beside the block
        {
            const int ri0 = (lor - rs + 15) >> 4;
            const double2* pv0 = reinterpret_cast<const double2*>(&vperm[0][rs * EM_RI]);
            const double2* pv1 = reinterpret_cast<const double2*>(&vperm[1][rs * EM_RI]);
            const double2* pv2 = reinterpret_cast<const double2*>(&vperm[2][rs * EM_RI]);
            constexpr int CH = 2, NH = (EM_NC + CH - 1) / CH;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                double acc[CH], vc[CH], wc[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int cj = h * CH + u;
                    acc[u] = 0.0;
                    vc[u] = (cj < EM_NC) ? vnat[cl0 + 32 * (cj < EM_NC ? cj : 0)] : 0.0;
                    wc[u] = (cj < EM_NC) ? wnat[cl0 + 32 * (cj < EM_NC ? cj : 0)] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < EM_RI / 2; ++q) {
                    if (2 * q + 1 >= ri0) {              // rows rs + 16 (2q), rs + 16 (2q + 1)
                        const double2 vr = pv0[q], wr = pv1[q], nr = pv2[q];
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const int cj = h * CH + u;
                            if (cj < EM_NC) {
                                double x0 = a[cj][2 * q], x1 = a[cj][2 * q + 1];
                                x0 -= vr.x * wc[u] + wr.x * vc[u];
                                x1 -= vr.y * wc[u] + wr.y * vc[u];
                                a[cj][2 * q] = x0; a[cj][2 * q + 1] = x1;
                                acc[u] += x0 * nr.x + x1 * nr.y;
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (h * CH + u < EM_NC) accb[rs][cl0 + 32 * (h * CH + u)] = acc[u];
            }
        }
        if (lo >= E && rs == (lor & 15)) {               // the lanes that hold pivot row s + 2 of the updated block publish it
            switch (lor >> 4) {
#define EM_CASE(k) case k: _Pragma("unroll") for (int cj = 0; cj < EM_NC; ++cj) prow[cl0 + 32 * cj] = a[cj][k]; break;
                EM_CASE(0) EM_CASE(1) EM_CASE(2) EM_CASE(3) EM_CASE(4) EM_CASE(5) EM_CASE(6) EM_CASE(7) EM_CASE(8) EM_CASE(9) EM_CASE(10) EM_CASE(11)
                EM_CASE(12) EM_CASE(13)
#undef EM_CASE
            }
        }
        lds_barrier();
        // ---- strip pass B: row sums of the updated strip over the register block's columns (the transposed half of the off-diagonal
        // block, which is not stored): thread = (row er, 16 threads per row), 16-lane reduction
        if (lo < E) {
            const int j = tid & 15;
            for (int er = tid >> 4; er < E; er += 32) {      // (uniform trip count per 16-lane group)
                double t = 0.0;
                if (er >= lo)
                    for (int c = E + j; c < m; c += 16) t += S[er * EX_LD + c] * nL[c];
                t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 1, 64);
                if (j == 0) accB[er] = t;
            }
            lds_barrier();
        }
        if (tid < EX_LD) {
            double t = 0.0;
            if (rown) {
#pragma unroll
                for (int q = 0; q < 16; ++q) t += accb[q][rr];
            }
            if (lo < E) t += (accA[0][tid] + accA[1][tid]) + ((tid < E) ? accB[tid] : 0.0);
            p_i = t;
            // pivot row s + 2 of the updated block = the next pivot column: from the strip while it is alive, then from the register block
            col_i = (tid >= lo && tid < m) ? ((lo < E) ? S[lo * EX_LD + tid] : (rown ? prow[rr] : 0.0)) : 0.0;
        }
        v_i = vn_i;
        tk = tn;
        // the next iteration's first barrier orders these reads against its LDS writes
    }
}


// ------------------------------------------------------------------------------------------ e2
// grid (ceil(n / (256/NL)), B), block 256.  NL lanes -> one eigenvalue (j-th smallest) by (NL+1)-section on the Sturm count: every
// round the NL lanes evaluate the count at NL interior points of the current interval, so ~23 rounds at NL = 4 (instead of 53
// bisection steps) of the n-step serial recurrence reach 1 ulp.  NL trades the serial depth (rounds) against the total work
// (NL x rounds sweeps per eigenvalue): the kernel is issue-bound at NL = 16 and latency-bound at NL = 4 for 32 x 420 eigenvalues.
// Count: division-free three-term recurrence p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2} (sign changes = eigenvalues below x).
// Measured (tools/eigh_nl_ab.sh, kernel time per call, 256 / 32 matrices of order 420): NL = 8 1296 / 250 us, NL = 4 925 / 226 us, NL = 2
// 801 / 331 us, NL = 1 (plain bisection) 780 / 487 us - since the loads left the dependent chain the kernel is issue-bound at every batch
// size, and fewer lanes per eigenvalue mean fewer sweeps in total (136, 92, 68, 53 per eigenvalue); below NL = 4 the small batch runs out
// of waves.  One value for every batch size: the converged midpoint depends on NL, and a matrix's result must not depend on its batch.
// Round 4 (tools/r4_q.sh, kernel time under rocprofv3, 256 / 32 matrices): sign history + exponent pre-check (see sturm below) 946 -> 788 us;
// shared grids: none 788 / 231, one level 733 / 212, two 726 / 226, three 776 / 257, four 842 / 277 us; scalar loads 691 / 243;
// with them NL = 2 645 / 331 (one level 630 / 306), NL = 8 1000 / 236 - NL stays 4.
#ifndef EG_NL
#define EG_NL 4      // lanes per eigenvalue: (EG_NL + 1)-section per round
#endif
#ifndef EG_GRID
#define EG_GRID 2    // levels of the workgroup's shared 512-point grid ahead of the per-eigenvalue search (0 = none)
#endif
// eigh_sde_kernel: grid (B), block 256.  {d_i, e_{i-1}^2} pairs of one matrix + its Gershgorin interval, norm and pivmin, written to the head
// of the matrix's (still unused) inverse-iteration work area: the Sturm sweeps read the pairs with wave-uniform indices, i.e. by SCALAR
// loads (two s_load_dwordx16 per eight steps) instead of one 16-byte LDS broadcast read per step and wave - 1 KB of LDS return
// bandwidth per wave and step, 8 clocks of the CU's one LDS pipe beside 5 clocks of the wave's share of its SIMD.  Measured: 726 -> 691 us
// per 256 matrices (243 against 226 us at 32 matrices, where the unprefetched scalar loads show); the prologue of the seven
// workgroups per matrix (loads, four block reductions) is gone as well.
#define EG_SDE_STRIDE(n) ((size_t)6 * ((n) + 2) * EG_MAXN)        // doubles per matrix in ws.lu
__global__ __launch_bounds__(256) void eigh_sde_kernel(int n, EighWs ws) {
    __shared__ double red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* d = ws.d + (size_t)b * n;
    const double* e = ws.e + (size_t)b * n;
    double2* out = reinterpret_cast<double2*>(ws.lu + (size_t)b * EG_SDE_STRIDE(n));
    double gl = 1e300, gu = -1e300, tn = 0.0, e2m = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double di = d[i];
        const double ej = (i < n - 1) ? e[i] : 0.0;
        const double em = (i > 0) ? e[i - 1] : 0.0;
        out[i] = make_double2(di, em * em);
        const double r = fabs(ej) + fabs(em);
        gl = fmin(gl, di - r);
        gu = fmax(gu, di + r);
        tn = fmax(tn, fabs(di) + r);
        e2m = fmax(e2m, ej * ej);
    }
    const double glo = -block_max(-gl, red), ghi = block_max(gu, red), tnorm = block_max(tn, red);
    const double safemn = 2.2250738585072014e-308;
    const double pivmin = fmax(safemn, safemn * block_max(e2m, red));
    if (tid == 0) { out[n] = make_double2(glo, ghi); out[n + 1] = make_double2(tnorm, pivmin); }
}

__global__ __launch_bounds__(256) void eigh_bisect_kernel(int n, EighWs ws, double* __restrict__ lam_out) {
    constexpr int NL = EG_NL, PER = 256 / NL;
    constexpr double inv = 1.0 / (double)(NL + 1);
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, t = lane & (NL - 1);
    const int j = blockIdx.x * PER + tid / NL;
    const double2* __restrict__ sde = reinterpret_cast<const double2*>(ws.lu + (size_t)b * EG_SDE_STRIDE(n));   // uniform indices below: scalar loads
    const double glo = sde[n].x, ghi = sde[n].y, tnorm = sde[n + 1].x, pivmin = sde[n + 1].y;
    const double eps = 2.220446049250313e-16;
    double lo = glo - 2.0 * tnorm * eps * n - 2.0 * pivmin, hi = ghi + 2.0 * tnorm * eps * n + 2.0 * pivmin;
    // Sturm count at x.  Per step three float64 operations and ONE integer operation: the sign bit of every p_i is shifted into a
    // history word (v_alignbit) and the sign changes are counted once per eight steps (popcount of history ^ history >> 1) - the
    // xor / shift / add per step of the first version were 2.5 of its 5.5 issue slots per step.  The range check of the running pair
    // looks at the exponent field first (one bit-field extract, one compare, one wave-uniform branch): the float64 comparisons against
    // 1e+-100 and the rescaling only run when some lane is outside 2^+-331, and decide exactly as before (same counts, same bits)
    auto sturm = [&](double x) -> int {
        double p0 = 1.0, p1 = sde[0].x - x;
        if (p1 == 0.0) p1 = -pivmin;
        unsigned hist = (unsigned)__double2hiint(p1) >> 31;        // bit 0 = sign of the newest p
        int cnt = (int)hist;                                       // p_{-1} = 1 > 0
        int i = 1;
        for (; i + 8 <= n; i += 8) {
            double2 de[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) de[u] = sde[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                // an exact zero needs no repair here: the recurrence continues with p3 = -e^2 p1 and its sign bit counts as positive, which can
                // only misplace the count AT a point that is exactly an eigenvalue of a leading block - the interval still closes around it
                const double p2 = (de[u].x - x) * p1 - de[u].y * p0;
                hist = __builtin_amdgcn_alignbit(hist, (unsigned)__double2hiint(p2), 31);     // (hist << 1) | sign(p2)
                p0 = p1;
                p1 = p2;
            }
            cnt += __popc((hist ^ (hist >> 1)) & 0xffu);           // changes between the nine newest signs
            const unsigned ex = ((unsigned)__double2hiint(p1) >> 20) & 0x7ffu;
            if (__any((unsigned)(ex - (1023u - 331u)) > 662u)) {   // |d - x| + e^2 grows a term by < 1e8 per step here: 8 steps are safe
                const double ap = fabs(p1);
                if (ap > 1e100) { p0 *= 1e-100; p1 *= 1e-100; }
                else if (ap < 1e-100) { p0 *= 1e100; p1 *= 1e100; }
            }
        }
        for (; i < n; ++i) {
            const double2 de = sde[i];
            const double p2 = (de.x - x) * p1 - de.y * p0;
            cnt += (int)((unsigned)(__double2hiint(p2) ^ __double2hiint(p1)) >> 31);
            p0 = p1;
            p1 = p2;
        }
        return cnt;
    };
    bool active = j < n;
#if EG_GRID > 0
    // ---- shared grids ahead of the per-eigenvalue search.  The first rounds of all eigenvalues of a matrix look at the same interval:
    // the workgroup counts once at 2 x 256 points of the range that still holds ITS 64 eigenvalues (level 0: the Gershgorin interval;
    // then the hull of their brackets, which for a clustered spectrum is a sliver of it) and every eigenvalue takes the grid cell whose
    // ends bracket its index - two sweeps per thread buy log5(513) = 3.9 multisection rounds of four sweeps on the first level and
    // 1.2 (evenly spread spectrum) to 3.9 (cluster) on the following ones.  A bracket found by binary search over the counts is valid
    // whether or not rounding left them monotone: the search keeps count[a] <= j < count[b] and ends at b = a + 1.
    {
        constexpr int NG = 512;
        __shared__ int gcnt[NG];
        __shared__ double ghull[2][4];
        double rl = lo, rh = hi;
        for (int lev = 0; lev < EG_GRID; ++lev) {
            const double gw = rh - rl;
            if (!(gw > 0.0)) break;                                // (uniform)
            constexpr double ginv = 1.0 / (double)(NG + 1);
#pragma unroll 1
            for (int k = tid; k < NG; k += 256) gcnt[k] = sturm(rl + gw * ((double)(k + 1) * ginv));
            __syncthreads();
            if (j < n) {
                int a = -1, bnd = NG;                              // virtual ends: count(rl) <= j < count(rh)
                while (bnd - a > 1) {
                    const int mid = (a + bnd) >> 1;
                    if (gcnt[mid] <= j) a = mid; else bnd = mid;
                }
                lo = (a < 0) ? rl : rl + gw * ((double)(a + 1) * ginv);
                hi = (bnd >= NG) ? rh : rl + gw * ((double)(bnd + 1) * ginv);
            }
            if (lev + 1 < EG_GRID) {                               // hull of this workgroup's brackets
                double hl = (j < n) ? lo : 1e300, hh = (j < n) ? hi : -1e300;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) { hl = fmin(hl, __shfl_xor(hl, o, 64)); hh = fmax(hh, __shfl_xor(hh, o, 64)); }
                if (lane == 0) { ghull[0][tid >> 6] = hl; ghull[1][tid >> 6] = hh; }
                __syncthreads();
                rl = fmin(fmin(ghull[0][0], ghull[0][1]), fmin(ghull[0][2], ghull[0][3]));
                rh = fmax(fmax(ghull[1][0], ghull[1][1]), fmax(ghull[1][2], ghull[1][3]));
            }
            __syncthreads();
        }
    }
#endif
    for (int it = 0; it < 96; ++it) {
        const double w = hi - lo;
        const double x = lo + w * (double)(t + 1) * inv;
        const double x1 = lo + w * inv, xl = lo + w * (double)NL * inv;
        if (!(w > fmax(eps * tnorm, 2.0 * eps * fmax(fabs(lo), fabs(hi))) + 2.0 * pivmin) || x1 <= lo || xl >= hi) active = false;
        if (!__any(active)) break;
        const int cnt = sturm(x);
        const unsigned long long bal = __ballot(cnt <= j);
        const int m = __popc((unsigned)((bal >> (lane & (64 - NL))) & ((1ull << NL) - 1ull)));   // points of this group with count <= j
        if (active) {
            const double nlo = (m == 0) ? lo : lo + w * (double)m * inv;
            const double nhi = (m == NL) ? hi : lo + w * (double)(m + 1) * inv;
            lo = nlo;
            hi = nhi;
        }
    }
    if (j < n && t == 0) lam_out[(size_t)b * n + j] = 0.5 * (lo + hi);
}

// ------------------------------------------------------------------------------------------ e3
// grid (ceil(n/64), B), block 64.  Thread -> one eigenvector of T by inverse iteration (dlagtf / dlagts / dstein
// recurrences).  The factors and the iterate live in global work arrays laid out [step pair][thread][2]: a thread's two
// consecutive steps are one 16-byte load and a wave's loads are contiguous.  Every sweep walks the arrays in chunks whose
// operands are loaded into registers one chunk ahead, so the serial recurrence runs from registers and the HBM latency
// of a chunk hides behind the previous chunk.  Row interchanges are one 32-bit mask per 32 steps; 1/pivot is stored at
// factorisation time so that the back-substitution chain is two FMAs and a multiply (the reference's tiny-pivot perturbation
// runs on a slow path when that product is not finite or too large).
// arrays: 0 - (the pivots a: no longer stored), 1 b, 2 c (stored at step+1, next to the iterate element it meets in the forward sweep), 3 d2, 4 x, 5 1/a
#ifndef IV_F
#define IV_F 32     // forward-sweep chunk (steps; 32 or 16: the interchange masks are one word per 32 steps)
#endif
#ifndef IV_B
#define IV_B 16     // backward-sweep chunk (steps)
#endif
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
__global__ __launch_bounds__(64) void eigh_invit_kernel(int n, EighWs ws, const double* __restrict__ lam_in, int extra) {
    __shared__ double d[EG_MAXN], e[EG_MAXN];
    const int b = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
    for (int i = threadIdx.x; i < n; i += 64) { d[i] = ws.d[(size_t)b * n + i]; e[i] = ws.e[(size_t)b * n + i]; }
    __syncthreads();
    if (j >= n) return;
    const int npairs = (n + 2) >> 1;
    double2* lub = reinterpret_cast<double2*>(ws.lu + (size_t)b * 6 * (n + 2) * EG_MAXN);
    int* pinm = ws.pin + (size_t)b * n * EG_MAXN;               // [chunk of 32 steps][thread] interchange masks
    double* z = ws.zt + (size_t)b * n * EG_MAXN + j;
#define LUP(arr, pr) lub[((size_t)(arr) * npairs + (pr)) * EG_MAXN + j]
#define LUE(arr, i) (reinterpret_cast<double*>(&LUP(arr, (i) >> 1))[(i) & 1])
    const double eps = 2.220446049250313e-16, sfmin = 2.2250738585072014e-308, bignum = 1.0 / sfmin;
    double xj;
    {   // dstein: shifts closer than 10 eps |x| to their (already separated) predecessor are pushed apart
        const double* lam = lam_in + (size_t)b * n;
        int s0 = j;
        while (s0 > 0 && lam[s0] - lam[s0 - 1] < 10.0 * fabs(eps * lam[s0]) && j - s0 < 64) --s0;
        double prev = lam[s0];
        for (int i = s0 + 1; i <= j; ++i) {
            double x = lam[i];
            const double pertol = 10.0 * fabs(eps * x);
            if (x - prev < pertol) x = prev + pertol;
            prev = x;
        }
        xj = prev;
    }
    double onenrm = fabs(d[0]) + (n > 1 ? fabs(e[0]) : 0.0);
    if (n > 1) onenrm = fmax(onenrm, fabs(d[n - 1]) + fabs(e[n - 2]));
    for (int i = 1; i < n - 1; ++i) onenrm = fmax(onenrm, fabs(d[i]) + fabs(e[i - 1]) + fabs(e[i]));
    const double dtpcrt = sqrt(0.1 / (double)n);
    // ---- start vector: deterministic pseudo-random in (-1, 1), never stored.  The first forward elimination rides inside the
    // factorisation below (its multiplier and interchange of step k are at hand the moment they are computed) on the UNSCALED vector -
    // the elimination is linear, and dstein's scale factor needs the last pivot - so the first sweep reads nothing and writes only the
    // eliminated right-hand side: 14 KB less HBM traffic per eigenvector than storing the vector and sweeping over it (82 KB before).
    const unsigned int rs0 = 0x9E3779B9u * (unsigned)(j + 1) + 12345u;
    auto lcg = [](unsigned int& rs) {
        rs = rs * 1664525u + 1013904223u;
        return ((double)(rs >> 8) / 8388608.0) - 1.0;
    };
    double asum = 0.0, s2 = 0.0;
    {
        unsigned int rs = rs0;
        for (int i = 0; i < n; ++i) asum += fabs(lcg(rs));
    }
    // ---- dlagtf: LU of T - xj I with partial pivoting; the recurrence values live in registers, every output element is
    // stored once.  The pivot test |c|/scale2 <= |a|/scale1 is evaluated cross-multiplied (no divisions on the chain).
    double tol = 0.0, alast;
    {
        double acur = d[0] - xj;                        // a[k] as modified by step k-1
        double bcur = (n > 1) ? e[0] : 0.0;             // b[k] as modified by step k-1
        double scale1 = fabs(acur) + fabs(bcur);
        unsigned mask = 0;
        unsigned int rs = rs0;
        double yprev = lcg(rs);                         // forward elimination of the (unscaled) start vector
        for (int k = 0; k < n - 1; ++k) {
            const double ak = acur, bk = bcur, ak1 = d[k + 1] - xj, ck = e[k];
            const double bk1 = (k < n - 2) ? e[k + 1] : 0.0;
            const double scale2 = fabs(ck) + fabs(ak1) + fabs(bk1);
            double a_out = ak, b_out = bk, c_out = ck, d2_out = 0.0, a_next = ak1, b_next = bk1, r;
            bool swapped = false;
            if (ck == 0.0) {
                scale1 = scale2;
                r = 1.0 / ak;
            } else if (ak != 0.0 && fabs(ck) * scale1 <= fabs(ak) * scale2) {
                scale1 = scale2;
                r = 1.0 / ak;
                c_out = ck * r;
                a_next = ak1 - c_out * bk;
            } else {
                mask |= 1u << ((k + 1) & 31);
                swapped = true;
                r = 1.0 / ck;
                const double mult = ak * r;
                a_out = ck;
                a_next = bk - mult * ak1;
                d2_out = bk1;
                b_next = -mult * bk1;
                b_out = ak1;
                c_out = mult;
            }
            LUE(1, k) = b_out;
            LUE(2, k + 1) = c_out;
            // (the second superdiagonal of U is not stored: d2[k] = e[k + 1] where step k interchanged rows, else 0 - the back substitution
            //  rebuilds it from the interchange masks and the LDS copy of e: 15 % less HBM traffic in a kernel that is bound by it)
            LUE(5, k) = r;
            {                                           // element t = k + 1 of the start vector meets c[t] (see the forward sweep below)
                const double y = lcg(rs);
                if (!swapped) { LUE(4, k) = yprev; yprev = y - c_out * yprev; }
                else { LUE(4, k) = y; yprev = yprev - c_out * y; }
            }
            tol = fmax(fmax(tol, fabs(a_out)), fmax(fabs(b_out), fabs(d2_out)));
            if (((k + 1) & 31) == 31 || k == n - 2) { pinm[(size_t)((k + 1) >> 5) * EG_MAXN + j] = (int)mask; mask = 0; }
            acur = a_next;
            bcur = b_next;
        }
        LUE(5, n - 1) = 1.0 / acur;
        LUE(4, n - 1) = yprev;
        alast = fabs(acur);
        tol = fmax(tol, alast) * eps;
        if (tol == 0.0) tol = eps;
    }
    const int nf = (n + IV_F - 1) / IV_F;             // forward chunks over t = step + 1 in [1, n)
    const int cbtop = (n - 1) / IV_B;                 // backward chunks over k, cb = cbtop .. 0
    int nrmchk = 0;
    for (int its = 0; its < 8; ++its) {
        // scale: ||x||_1 -> n * onenrm * max(eps, |a_n|)   (folded into the loads of the forward sweep)
        const double scl = (double)n * onenrm * fmax(eps, alast) / asum;
        // ---- forward elimination with the recorded row interchanges; element t = step + 1 meets c[t] (= c of step t-1)
        // (the first one was done during the factorisation, unscaled: its scale factor is applied by the back substitution)
        if (its > 0) {
            double yprev = scl * LUE(4, 0);
            auto fload = [&](int ci, double2 (&yk)[IV_F / 2], double2 (&ck)[IV_F / 2], int& m) {
#pragma unroll
                for (int u = 0; u < IV_F / 2; ++u) {
                    const int pr = min((IV_F / 2) * ci + u, npairs - 1);
                    yk[u] = LUP(4, pr);
                    ck[u] = LUP(2, pr);
                }
                m = pinm[(size_t)((IV_F * ci) >> 5) * EG_MAXN + j] >> ((IV_F * ci) & 31);
            };
            auto fstep = [&](int t, double yraw, double c, int bit) {
                if (t >= 1 && t <= n - 1) {
                    const double y = scl * yraw;
                    if (bit == 0) { LUE(4, t - 1) = yprev; yprev = y - c * yprev; }
                    else { LUE(4, t - 1) = y; yprev = yprev - c * y; }
                }
            };
            auto fproc = [&](int ci, const double2 (&yk)[IV_F / 2], const double2 (&ck)[IV_F / 2], int m) {
#pragma unroll
                for (int u = 0; u < IV_F / 2; ++u) {
                    const int t = IV_F * ci + 2 * u;
                    fstep(t, yk[u].x, ck[u].x, (m >> (2 * u)) & 1);
                    fstep(t + 1, yk[u].y, ck[u].y, (m >> (2 * u + 1)) & 1);
                }
            };
            double2 yA[IV_F / 2], cA[IV_F / 2], yB[IV_F / 2], cB[IV_F / 2];
            int mA = 0, mB = 0;
            fload(0, yA, cA, mA);
            for (int ci = 0; ci < nf; ci += 2) {
                if (ci + 1 < nf) fload(ci + 1, yB, cB, mB);
                fproc(ci, yA, cA, mA);
                if (ci + 1 < nf) {
                    if (ci + 2 < nf) fload(ci + 2, yA, cA, mA);
                    fproc(ci + 1, yB, cB, mB);
                }
            }
            LUE(4, n - 1) = yprev;
        }
        // ---- back substitution, perturbing tiny pivots (job = -1)
        const double sclb = (its == 0) ? scl : 1.0;
        double y1 = 0.0, y2 = 0.0;   // x[k+1], x[k+2]
        double nrm = 0.0;
        asum = 0.0;
        s2 = 0.0;
        {
            auto bload = [&](int cb, double2 (&xr)[IV_B / 2], double2 (&rr)[IV_B / 2], double2 (&br)[IV_B / 2], unsigned& mk) {
#pragma unroll
                for (int u = 0; u < IV_B / 2; ++u) {
                    const int pr = min((IV_B / 2) * cb + u, npairs - 1);
                    xr[u] = LUP(4, pr);
                    rr[u] = LUP(5, pr);
                    br[u] = LUP(1, pr);
                }
                // interchange bits of steps k = IV_B cb .. IV_B cb + IV_B - 1 (bit k + 1 of the mask words: may straddle two of them)
                const int p0 = IV_B * cb + 1, wd = p0 >> 5, sh = p0 & 31;
                const unsigned lo = (unsigned)pinm[(size_t)wd * EG_MAXN + j], hi = (unsigned)pinm[(size_t)(wd + 1) * EG_MAXN + j];
                mk = (unsigned)((((unsigned long long)hi << 32) | lo) >> sh);
            };
            auto bstep = [&](int k, double xv, double rv, double bv, unsigned bit) {
                if (k <= n - 1) {
                    const double dv = (k <= n - 3 && bit) ? e[k + 1] : 0.0;
                    double temp = sclb * xv - ((k <= n - 2) ? bv : 0.0) * y1 - ((k <= n - 3) ? dv : 0.0) * y2;
                    double xk = temp * rv;
                    if (!(fabs(xk) <= bignum)) {                   // slow path: the reference's pivot perturbation
                        double ak = 1.0 / rv;              // (the pivot itself is not stored: only this slow path wanted it)
                        double pert = copysign(tol, ak);
                        for (int guard = 0; guard < 200; ++guard) {
                            const double absak = fabs(ak);
                            if (absak < 1.0) {
                                if (absak < sfmin) {
                                    if (absak == 0.0 || fabs(temp) * sfmin > absak) { ak += pert; pert *= 2.0; continue; }
                                    temp *= bignum;
                                    ak *= bignum;
                                } else if (fabs(temp) > absak * bignum) { ak += pert; pert *= 2.0; continue; }
                            }
                            break;
                        }
                        xk = temp / ak;
                    }
                    LUE(4, k) = xk;
                    y2 = y1;
                    y1 = xk;
                    nrm = fmax(nrm, fabs(xk));
                    asum += fabs(xk);
                    s2 += xk * xk;
                }
            };
            auto bproc = [&](int cb, const double2 (&xr)[IV_B / 2], const double2 (&rr)[IV_B / 2], const double2 (&br)[IV_B / 2], unsigned mk) {
#pragma unroll
                for (int u = IV_B / 2 - 1; u >= 0; --u) {
                    const int k = IV_B * cb + 2 * u;
                    bstep(k + 1, xr[u].y, rr[u].y, br[u].y, (mk >> (2 * u + 1)) & 1u);
                    bstep(k, xr[u].x, rr[u].x, br[u].x, (mk >> (2 * u)) & 1u);
                }
            };
            double2 xA[IV_B / 2], rA[IV_B / 2], bA[IV_B / 2], xB[IV_B / 2], rB[IV_B / 2], bB[IV_B / 2];
            unsigned mA = 0, mB = 0;
            bload(cbtop, xA, rA, bA, mA);
            for (int cb = cbtop; cb >= 0; cb -= 2) {
                if (cb - 1 >= 0) bload(cb - 1, xB, rB, bB, mB);
                bproc(cb, xA, rA, bA, mA);
                if (cb - 1 >= 0) {
                    if (cb - 2 >= 0) bload(cb - 2, xA, rA, bA, mA);
                    bproc(cb - 1, xB, rB, bB, mB);
                }
            }
        }
        if (nrm < dtpcrt) continue;
        if (++nrmchk < extra + 1) continue;   // dstein: EXTRA = 2 more sweeps after the growth criterion is met
        break;
    }
    const double inv = 1.0 / sqrt(s2);
    for (int p0 = 0; p0 < npairs; p0 += 8) {
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = LUP(4, min(p0 + u, npairs - 1));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = 2 * (p0 + u);
            if (i < n) z[(size_t)i * EG_MAXN] = t[u].x * inv;
            if (i + 1 < n) z[(size_t)(i + 1) * EG_MAXN] = t[u].y * inv;
        }
    }
#undef LUP
#undef LUE
}
#endif  // NELE_AB

// ------------------------------------------------------------------------------------------ e3, round 4: factors recomputed, not stored
// eigh_invit_kernel above keeps the LU factors of T - x I (b, the multipliers c, 1 / pivot) in global work arrays and moves 57 KB per
// eigenvector at the HBM rate the chip sustains (6.2 GB per 256 matrices, 1.39 ms): the kernel is bound by exactly that traffic while its
// arithmetic units idle.  The factorisation is a three-value forward recurrence (a, b, scale1) over d and e, which sit in LDS - a few
// multiply-adds, a compare and one division per step.  This kernel stores NOTHING of it but a checkpoint (a, b, scale1) every IV2_C
// steps (53 doubles per eigenvector where the arrays took 1260): the forward elimination of a later sweep recomputes the factors in
// lock-step, the back substitution recomputes each chunk of IV2_C steps forward from its checkpoint into registers and then walks it
// backwards.  Same operations on the same operands in the same order as the stored form - bit-identical eigenvectors (A/B test against
// eigh_invit_kernel, which stays in the test library) - with the iterate as the only array that travels: 30 KB per eigenvector.
#ifndef IV2_C
#define IV2_C 16           // steps per checkpoint / back-substitution chunk (registers: two factor values per step)
#endif
#ifndef IV2_OCC
#define IV2_OCC 2          // waves per SIMD the register budget is cut for (measured: 16 / 2 868 us, 8 / 3 949, 8 / 4 1061, 32 / 1 1348 per 256 matrices)
#endif
#ifndef IV2_FC
#define IV2_FC 8           // iterate elements per prefetch chunk of the forward sweep
#endif
__global__ __launch_bounds__(64, IV2_OCC) void eigh_invit2_kernel(int n, EighWs ws, const double* __restrict__ lam_in, int extra) {
    __shared__ double d[EG_MAXN], e[EG_MAXN];
    const int b = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
    for (int i = threadIdx.x; i < n; i += 64) { d[i] = ws.d[(size_t)b * n + i]; e[i] = ws.e[(size_t)b * n + i]; }
    __syncthreads();
    if (j >= n) return;
    const int npairs = (n + 2) >> 1;
    double2* lub = reinterpret_cast<double2*>(ws.lu + (size_t)b * 6 * (n + 2) * EG_MAXN);
    double* z = ws.zt + (size_t)b * n * EG_MAXN + j;
    // the iterate: array 4 of the old layout ([pair][thread][2]); checkpoints: arrays 0 .. 2 reused as [chunk][3][thread]
#define X2P(pr) lub[((size_t)4 * npairs + (pr)) * EG_MAXN + j]
#define X2E(i) (reinterpret_cast<double*>(&X2P((i) >> 1))[(i) & 1])
    double* ckp = reinterpret_cast<double*>(lub) + j;
#define CK2(c, q) ckp[((size_t)(c) * 3 + (q)) * EG_MAXN]
    const double eps = 2.220446049250313e-16, sfmin = 2.2250738585072014e-308, bignum = 1.0 / sfmin;
    double xj;
    {   // dstein: shifts closer than 10 eps |x| to their (already separated) predecessor are pushed apart
        const double* lam = lam_in + (size_t)b * n;
        int s0 = j;
        while (s0 > 0 && lam[s0] - lam[s0 - 1] < 10.0 * fabs(eps * lam[s0]) && j - s0 < 64) --s0;
        double prev = lam[s0];
        for (int i = s0 + 1; i <= j; ++i) {
            double x = lam[i];
            const double pertol = 10.0 * fabs(eps * x);
            if (x - prev < pertol) x = prev + pertol;
            prev = x;
        }
        xj = prev;
    }
    double onenrm = fabs(d[0]) + (n > 1 ? fabs(e[0]) : 0.0);
    if (n > 1) onenrm = fmax(onenrm, fabs(d[n - 1]) + fabs(e[n - 2]));
    for (int i = 1; i < n - 1; ++i) onenrm = fmax(onenrm, fabs(d[i]) + fabs(e[i - 1]) + fabs(e[i]));
    const double dtpcrt = sqrt(0.1 / (double)n);
    const unsigned int rs0 = 0x9E3779B9u * (unsigned)(j + 1) + 12345u;
    auto lcg = [](unsigned int& rs) {
        rs = rs * 1664525u + 1013904223u;
        return ((double)(rs >> 8) / 8388608.0) - 1.0;
    };
    // one step of dlagtf (LU of T - xj I with partial pivoting): (acur, bcur, scale1) -> the same after step k; outputs of the step
    struct Fst { double acur, bcur, scale1; };
    auto fact = [&](int k, Fst& st, double& a_out, double& b_out, double& c_out, double& d2_out, double& r, bool& swapped) {
        // branch-free form of eigh_invit_kernel's three-way step (same expressions, selected): the lanes of a wave take different
        // branches at almost every step, and one IEEE division per step instead of one per taken branch is what this kernel is bound by
        const double ak = st.acur, bk = st.bcur, ak1 = d[k + 1] - xj, ck = e[k];
        const double bk1 = (k < n - 2) ? e[k + 1] : 0.0;
        const double scale2 = fabs(ck) + fabs(ak1) + fabs(bk1);
        const bool czero = (ck == 0.0);
        swapped = !czero && !(ak != 0.0 && fabs(ck) * st.scale1 <= fabs(ak) * scale2);
#ifdef IV2_FASTDIV
        {   // reciprocal by v_rcp_f64 + two Newton steps (<= 1 ulp, not correctly rounded: NOT bit-identical with the stored-factor kernel)
            const double pv_ = swapped ? ck : ak;
            double r_ = __builtin_amdgcn_rcp(pv_);
            r_ = r_ * (2.0 - pv_ * r_);
            r_ = r_ * (2.0 - pv_ * r_);
            r = r_;
        }
#else
        r = 1.0 / (swapped ? ck : ak);
#endif
        const double cn = czero ? ck : ck * r;              // no interchange: multiplier (ck itself, i.e. zero, when ck == 0)
        const double an = czero ? ak1 : ak1 - cn * bk;
        const double mult = ak * r;                          // interchange
        const double as = bk - mult * ak1, bs = -mult * bk1;
        a_out = swapped ? ck : ak;
        b_out = swapped ? ak1 : bk;
        c_out = swapped ? mult : cn;
        d2_out = swapped ? bk1 : 0.0;
        st.scale1 = swapped ? st.scale1 : scale2;
        st.acur = swapped ? as : an;
        st.bcur = swapped ? bs : bk1;
    };
    const Fst st0 = {d[0] - xj, (n > 1) ? e[0] : 0.0, fabs(d[0] - xj) + ((n > 1) ? fabs(e[0]) : 0.0)};
    double asum = 0.0, s2 = 0.0;
    {
        unsigned int rs = rs0;
        for (int i = 0; i < n; ++i) asum += fabs(lcg(rs));
    }
    // ---- factorisation + the first forward elimination (on the unscaled start vector: see eigh_invit_kernel); checkpoints, masks
    double tol = 0.0, alast, rlast;
    {
        Fst st = st0;
        unsigned int rs = rs0;
        double yprev = lcg(rs);
        for (int k = 0; k < n - 1; ++k) {
            if ((k & (IV2_C - 1)) == 0) { CK2(k / IV2_C, 0) = st.acur; CK2(k / IV2_C, 1) = st.bcur; CK2(k / IV2_C, 2) = st.scale1; }
            double a_out, b_out, c_out, d2_out, r;
            bool swapped;
            fact(k, st, a_out, b_out, c_out, d2_out, r, swapped);
            const double y = lcg(rs);
            if (!swapped) { X2E(k) = yprev; yprev = y - c_out * yprev; }
            else { X2E(k) = y; yprev = yprev - c_out * y; }
            tol = fmax(fmax(tol, fabs(a_out)), fmax(fabs(b_out), fabs(d2_out)));
        }
        if (((n - 1) & (IV2_C - 1)) == 0) { CK2((n - 1) / IV2_C, 0) = st.acur; CK2((n - 1) / IV2_C, 1) = st.bcur; CK2((n - 1) / IV2_C, 2) = st.scale1; }
        rlast = 1.0 / st.acur;
        X2E(n - 1) = yprev;
        alast = fabs(st.acur);
        tol = fmax(tol, alast) * eps;
        if (tol == 0.0) tol = eps;
    }
    const int cbtop = (n - 1) / IV2_C;                 // backward chunks over k, cb = cbtop .. 0
    int nrmchk = 0;
    for (int its = 0; its < 8; ++its) {
        const double scl = (double)n * onenrm * fmax(eps, alast) / asum;
        // ---- forward elimination: the factors are recomputed in lock-step (element t meets the multiplier of step t - 1)
        if (its > 0) {
            Fst st = st0;
            double yprev = scl * X2E(0);
            constexpr int FC = IV2_FC;
            const int nfc = (n + FC - 1) / FC;
            auto fload = [&](int ci, double2 (&yk)[FC / 2]) {
#pragma unroll
                for (int u = 0; u < FC / 2; ++u) yk[u] = X2P(min((FC / 2) * ci + u, npairs - 1));
            };
            auto fproc = [&](int ci, const double2 (&yk)[FC / 2]) {
#pragma unroll
                for (int u = 0; u < FC; ++u) {
                    const int t = FC * ci + u;
                    if (t >= 1 && t <= n - 1) {
                        double a_out, b_out, c_out, d2_out, r;
                        bool swapped;
                        fact(t - 1, st, a_out, b_out, c_out, d2_out, r, swapped);
                        const double y = scl * ((u & 1) ? yk[u >> 1].y : yk[u >> 1].x);
                        if (!swapped) { X2E(t - 1) = yprev; yprev = y - c_out * yprev; }
                        else { X2E(t - 1) = y; yprev = yprev - c_out * y; }
                    }
                }
            };
            double2 yA[FC / 2], yB[FC / 2];
            fload(0, yA);
            for (int ci = 0; ci < nfc; ci += 2) {
                if (ci + 1 < nfc) fload(ci + 1, yB);
                fproc(ci, yA);
                if (ci + 1 < nfc) {
                    if (ci + 2 < nfc) fload(ci + 2, yA);
                    fproc(ci + 1, yB);
                }
            }
            X2E(n - 1) = yprev;
        }
        // ---- back substitution, perturbing tiny pivots (job = -1): per chunk, factors forward from the checkpoint, then backwards
        const double sclb = (its == 0) ? scl : 1.0;
        double y1 = 0.0, y2 = 0.0;   // x[k+1], x[k+2]
        double nrm = 0.0;
        asum = 0.0;
        s2 = 0.0;
        {
            auto bload = [&](int cb, double2 (&xr)[IV2_C / 2], Fst& ck) {
#pragma unroll
                for (int u = 0; u < IV2_C / 2; ++u) xr[u] = X2P(min((IV2_C / 2) * cb + u, npairs - 1));
                ck.acur = CK2(cb, 0); ck.bcur = CK2(cb, 1); ck.scale1 = CK2(cb, 2);
            };
            auto bproc = [&](int cb, const double2 (&xr)[IV2_C / 2], Fst st) {
                double rr[IV2_C], bb[IV2_C];
                unsigned sw = 0;
#pragma unroll
                for (int u = 0; u < IV2_C; ++u) {
                    const int k = IV2_C * cb + u;
                    rr[u] = 0.0; bb[u] = 0.0;
                    if (k <= n - 2) {
                        double a_out, c_out, d2_out;
                        bool swapped;
                        fact(k, st, a_out, bb[u], c_out, d2_out, rr[u], swapped);
                        if (swapped) sw |= 1u << u;
                    } else if (k == n - 1) rr[u] = rlast;
                }
#pragma unroll
                for (int u = IV2_C - 1; u >= 0; --u) {
                    const int k = IV2_C * cb + u;
                    if (k <= n - 1) {
                        const double xv = (u & 1) ? xr[u >> 1].y : xr[u >> 1].x;
                        const double rv = rr[u], bv = bb[u];
                        const double dv = (k <= n - 3 && ((sw >> u) & 1u)) ? e[k + 1] : 0.0;
                        double temp = sclb * xv - ((k <= n - 2) ? bv : 0.0) * y1 - ((k <= n - 3) ? dv : 0.0) * y2;
                        double xk = temp * rv;
                        if (!(fabs(xk) <= bignum)) {                   // slow path: the reference's pivot perturbation
                            double ak = 1.0 / rv;
                            double pert = copysign(tol, ak);
                            for (int guard = 0; guard < 200; ++guard) {
                                const double absak = fabs(ak);
                                if (absak < 1.0) {
                                    if (absak < sfmin) {
                                        if (absak == 0.0 || fabs(temp) * sfmin > absak) { ak += pert; pert *= 2.0; continue; }
                                        temp *= bignum;
                                        ak *= bignum;
                                    } else if (fabs(temp) > absak * bignum) { ak += pert; pert *= 2.0; continue; }
                                }
                                break;
                            }
                            xk = temp / ak;
                        }
                        X2E(k) = xk;
                        y2 = y1;
                        y1 = xk;
                        nrm = fmax(nrm, fabs(xk));
                        asum += fabs(xk);
                        s2 += xk * xk;
                    }
                }
            };
            double2 xA[IV2_C / 2], xB[IV2_C / 2];
            Fst cA, cB;
            bload(cbtop, xA, cA);
            for (int cb = cbtop; cb >= 0; cb -= 2) {
                if (cb - 1 >= 0) bload(cb - 1, xB, cB);
                bproc(cb, xA, cA);
                if (cb - 1 >= 0) {
                    if (cb - 2 >= 0) bload(cb - 2, xA, cA);
                    bproc(cb - 1, xB, cB);
                }
            }
        }
        if (nrm < dtpcrt) continue;
        if (++nrmchk < extra + 1) continue;   // dstein: EXTRA more sweeps after the growth criterion is met
        break;
    }
    const double inv = 1.0 / sqrt(s2);
    for (int p0 = 0; p0 < npairs; p0 += 8) {
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = X2P(min(p0 + u, npairs - 1));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = 2 * (p0 + u);
            if (i < n) z[(size_t)i * EG_MAXN] = t[u].x * inv;
            if (i + 1 < n) z[(size_t)(i + 1) * EG_MAXN] = t[u].y * inv;
        }
    }
#undef X2P
#undef X2E
#undef CK2
}

// ------------------------------------------------------------------------------------------ e4
// Eigenvectors of A = Q z, Q = H_0 H_1 ... H_{n-2}, H_k = I - tau_k v_k v_k^T acting on rows k+1..n-1.
// grid (ceil(n/32), B), block 256.  Eigenvectors live in REGISTERS: a 16-lane row holds two eigenvectors, lane t the rows
// t + 16 q, so a reflector costs 2 x 2 FMAs per held row, a 4-step DPP row reduction and no barrier.  Reflectors are staged
// through LDS eight at a time (zeroed below their first row, next chunk's loads in flight during the compute), so there
// is one barrier per 8 reflectors.  The result is written as U[j][i] (row j = eigenvector j), the layout the SIIB
// projection reads.
#define BT_CH 8
#ifdef NELE_AB                                          // superseded variant: only in the test library (libnele_hip_ab.so)
template <int BT_Q>     // 16-row groups held per lane: 28 covers n <= 448 (SIIB: 420), 32 covers EG_MAXN
__global__ __launch_bounds__(256, 2) void eigh_backtransform_kernel(const double* __restrict__ Aall, int n, EighWs ws, double* __restrict__ U) {
    // reflector chunk in LDS, PERMUTED: element i of reflector r at r * RS + (i & 15) * BT_Q + (i >> 4), so the BT_Q values a lane needs
    // (rows t, t + 16, ...) are contiguous and go out as 16-byte reads (the kernel is bound by LDS instruction issue: 2 x BT_Q reads per
    // reflector and wave in the plain layout)
    extern __shared__ double bt_sm[];         // vch[2][BT_CH][16][BT_Q], tch[2][BT_CH]
    const int nld = (n + 15) & ~15;
    constexpr int RS = 16 * BT_Q;
    double* vch = bt_sm;
    double* tch = bt_sm + 2 * BT_CH * RS;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, t = lane & 15, g = lane >> 4;
    const int j0 = blockIdx.x * 32 + w * 8 + g * 2;
    const double* A = Aall + (size_t)b * n * n;
    const double* tau = ws.tau + (size_t)b * n;
    const double* zt = ws.zt + (size_t)b * n * EG_MAXN;
    double z0[BT_Q], z1[BT_Q];
#pragma unroll
    for (int q = 0; q < BT_Q; ++q) {
        const int i = t + 16 * q;
        z0[q] = (i < n && j0 < n) ? zt[(size_t)i * EG_MAXN + j0] : 0.0;
        z1[q] = (i < n && j0 + 1 < n) ? zt[(size_t)i * EG_MAXN + j0 + 1] : 0.0;
    }
    const int nchunks = (n - 1 + BT_CH - 1) / BT_CH;
    const int per = BT_CH * nld;              // <= 8 * 512 = 16 * 256
    constexpr int NPRE = BT_Q / 2;            // 8 * 16 * BT_Q staged values / 256 threads
    double pre[NPRE];
    double pret = 0.0;
    auto gload = [&](int c) {
        const int kc = n - 2 - BT_CH * c;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int idx = tid + 256 * u;
            const int r = idx / nld, i = idx - r * nld, k = kc - r;
            pre[u] = (idx < per && k >= 0 && i > k && i < n) ? A[(size_t)k * n + i] : 0.0;
        }
        if (tid < BT_CH) pret = (kc - tid >= 0) ? tau[kc - tid] : 0.0;
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int idx = tid + 256 * u;
            const int r = idx / nld, i = idx - r * nld;
            if (idx < per) vch[buf * BT_CH * RS + r * RS + (i & 15) * BT_Q + (i >> 4)] = pre[u];
        }
        if (tid < BT_CH) tch[buf * BT_CH + tid] = pret;
    };
    for (int e_ = tid; e_ < 2 * BT_CH * RS; e_ += 256) vch[e_] = 0.0;      // slots of rows >= nld are never staged: they must read as zero
    __syncthreads();
    gload(0);
    lstore(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1, kc = n - 2 - BT_CH * c;
        if (c + 1 < nchunks) gload(c + 1);
        for (int r = 0; r < BT_CH; ++r) {
            const int k = kc - r;
            if (k < 0) break;
            const double tk = tch[buf * BT_CH + r];
            if (tk == 0.0) continue;
            const double2* vr2 = reinterpret_cast<const double2*>(vch + buf * BT_CH * RS + r * RS + t * BT_Q);
            const int q0 = (k + 1) >> 4;      // rows below 16 q0 are zero in this reflector
            double d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int qq = 0; qq < BT_Q / 4; ++qq) {
                if (4 * qq + 3 >= q0) {
                    const double2 va = vr2[2 * qq], vb = vr2[2 * qq + 1];
                    d0 += va.x * z0[4 * qq] + va.y * z0[4 * qq + 1] + vb.x * z0[4 * qq + 2] + vb.y * z0[4 * qq + 3];
                    d1 += va.x * z1[4 * qq] + va.y * z1[4 * qq + 1] + vb.x * z1[4 * qq + 2] + vb.y * z1[4 * qq + 3];
                }
            }
            d0 = tk * row16_sum_dpp(d0);
            d1 = tk * row16_sum_dpp(d1);
#pragma unroll
            for (int qq = 0; qq < BT_Q / 4; ++qq) {
                if (4 * qq + 3 >= q0) {
                    const double2 va = vr2[2 * qq], vb = vr2[2 * qq + 1];
                    z0[4 * qq] -= d0 * va.x; z0[4 * qq + 1] -= d0 * va.y; z0[4 * qq + 2] -= d0 * vb.x; z0[4 * qq + 3] -= d0 * vb.y;
                    z1[4 * qq] -= d1 * va.x; z1[4 * qq + 1] -= d1 * va.y; z1[4 * qq + 2] -= d1 * vb.x; z1[4 * qq + 3] -= d1 * vb.y;
                }
            }
        }
        if (c + 1 < nchunks) lstore(buf ^ 1);
        __syncthreads();
    }
    double* Ub = U + (size_t)b * n * n;
#pragma unroll
    for (int q = 0; q < BT_Q; ++q) {
        const int i = t + 16 * q;
        if (i < n && j0 < n) Ub[(size_t)j0 * n + i] = z0[q];
        if (i < n && j0 + 1 < n) Ub[(size_t)(j0 + 1) * n + i] = z1[q];
    }
}
#endif  // NELE_AB

// ------------------------------------------------------------------------------------------ e4, blocked (compact WY) on the matrix cores
// The reflector-by-reflector kernel above is bound by LDS instruction issue (2 x BT_Q reads per reflector and wave for 4 x BT_Q FMAs):
// 3.0 ms for 256 matrices of 420.  Here 16 consecutive reflectors H_klo ... H_khi are applied at once as I - V T V^T (LAPACK dlarft
// "forward, columnwise": T upper triangular, T_jj = tau_j, T_(0:j, j) = -tau_j T_(0:j, 0:j) V_(:, 0:j)^T v_j):
//     W = V^T Z (16 x 16 per wave),  W' = -T W,  Z += V W'
// with v_mfma_f64_16x16x4_f64.  A wave keeps 16 eigenvectors in the MFMA accumulator layout (reg q of lane l = row (l >> 4) + 4 q of the
// 16-row tile, column l & 15) for the whole kernel: a k-step t of the first product takes rows 4 t + (l >> 4), which IS register t of the
// tile, and the W / W' tiles feed the next product the same way - no shuffles, one 8-byte LDS read (the V or T operand) per MFMA.
// eigh_wy_t_kernel: grid (ceil(blocks of 16 reflectors / 4), B), block 256 = one wave per block: -T per block -> ws.lu (free once the inverse iteration is done).
#define WY_NB 16
#define WY_LD 17
// Round 4, second session: one WAVE per block of 16 reflectors, four blocks per workgroup, no staged copy of V.  The Gram matrix G = V V^T is a
// chain of v_mfma_f64_16x16x4 whose A and B operands are the SAME register (lane l: V[l & 15][4 t + (l >> 4)], read straight from A's
// reflector rows), four independent accumulators; the 16-column recurrence for T runs with a row of T per lane in registers.  The first
// version staged V in 57 KB of LDS per workgroup and spent two LDS reads per multiply-add on the Gram matrix: 283 us per 256 matrices
// for 54 KB of input per block.
__global__ __launch_bounds__(256) void eigh_wy_t_kernel(const double* __restrict__ Aall, int n, EighWs ws) {
    __shared__ double Gs[4][WY_NB][WY_LD];
    const int b = blockIdx.y, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int nblk = (n - 1 + WY_NB - 1) / WY_NB;
    const int c = 4 * blockIdx.x + wv;
    if (c >= nblk) return;                                              // (wave-uniform; no workgroup barrier below)
    const int khi = n - 2 - WY_NB * c, klo = khi - (WY_NB - 1);          // reflectors klo .. khi (those with k < 0 do not exist: tau = 0)
    const double* A = Aall + (size_t)b * n * n;
    const int lo = max(klo + 1, 0);                                    // rows <= klo are zero in every reflector of the block
    const int k = klo + li;                                            // this lane's reflector
    const double* vrow = A + (size_t)max(k, 0) * n;
    const int nt = (n - lo + 3) >> 2;
    f64x4_t acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    // 32 k-steps per batch with all 32 loads in flight (the reflector rows come from HBM: a batch costs one memory latency whatever its
    // size); steps behind the last row are masked, not peeled - a scalar remainder loop would pay that latency per step
    for (int t = 0; t < nt; t += 32) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int i = lo + 4 * (t + u) + g;
            v[u] = (k >= 0 && i > k && i < n) ? vrow[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u], v[u], acc[u & 3], 0, 0, 0);
    }
    double (*G)[WY_LD] = Gs[wv];
#pragma unroll
    for (int q = 0; q < 4; ++q)                                         // accumulator register q of lane l: row (l >> 4) + 4 q, column l & 15
        G[g + 4 * q][li] = (acc[0][q] + acc[1][q]) + (acc[2][q] + acc[3][q]);
    const double tau_l = (lane < WY_NB && klo + lane >= 0) ? ws.tau[(size_t)b * n + klo + lane] : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // T column by column: T[:j, j] = -tau_j T[:j, :j] G[:j, j].  Lane r keeps ROW r of T in registers (entries right of the columns done
    // so far are still zero, entries left of the diagonal stay zero), so column j is 120 multiply-adds over compile-time indices with
    // G[m][j] as LDS broadcast reads - no exchange between the lanes at all (the first version went through LDS and a wave barrier per column)
    double Trow[WY_NB];
#pragma unroll
    for (int j = 0; j < WY_NB; ++j) {
        const double tau_j = __shfl(tau_l, j, 64);
        double tt = 0.0;
#pragma unroll
        for (int m = 0; m < j; ++m) tt += Trow[m] * G[m][j];
        Trow[j] = (lane < j) ? -tau_j * tt : ((lane == j) ? tau_j : 0.0);
    }
    if (lane < WY_NB) {
        double* out = ws.lu + (size_t)b * 6 * (n + 2) * EG_MAXN + (size_t)c * 256 + lane * WY_NB;
#pragma unroll
        for (int m = 0; m < WY_NB; ++m) out[m] = -Trow[m];
    }
}

// grid (ceil(n / 64), B), block 256: wave w holds eigenvectors 64 blockIdx.x + 16 w ... + 15.  One workgroup per CU (the two V buffers
// take 117 KB of LDS), so a wave may use the whole register file: WY_RT tiles x 4 doubles of Z per lane.
template <int WY_RT>     // 16-row tiles held per lane: 27 covers n <= 432 (SIIB: 420), 32 covers EG_MAXN
__global__ __launch_bounds__(256, 1) void eigh_backtransform_wy_kernel(const double* __restrict__ Aall, int n, EighWs ws, double* __restrict__ U) {
    extern __shared__ double wy_sm[];                       // Vs[2][16 WY_RT][WY_LD], Ts[2][16][WY_LD]
    constexpr int NR = 16 * WY_RT;
    double* Vs = wy_sm;
    double* Ts = wy_sm + 2 * NR * WY_LD;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
    const int j = blockIdx.x * 64 + 16 * w + li;            // this lane's eigenvector
    const double* A = Aall + (size_t)b * n * n;
    const double* zt = ws.zt + (size_t)b * n * EG_MAXN;
    const double* Tall = ws.lu + (size_t)b * 6 * (n + 2) * EG_MAXN;
    f64x4_t Z[WY_RT];
#pragma unroll
    for (int rt = 0; rt < WY_RT; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 16 * rt + g + 4 * q;
            Z[rt][q] = (i < n && j < n) ? zt[(size_t)i * EG_MAXN + j] : 0.0;
        }
    const int nblk = (n - 1 + WY_NB - 1) / WY_NB;
    // staging: thread = (reflector r = tid >> 4, row slot ii = tid & 15), rows ii + 16 u: 128-byte runs of a reflector row
    double pre[WY_RT], pret = 0.0;
    const int sr = tid >> 4, sii = tid & 15;
    constexpr int GT = (WY_RT % 3 == 0) ? 3 : 4;             // tiles per group (see the MFMA loops)
    int sg0 = 0;                                            // first tile group of the staged block that is ever read
    auto gload = [&](int c) {
        const int klo = n - 2 - WY_NB * c - (WY_NB - 1), k = klo + sr;
        const double* src = A + (size_t)max(k, 0) * n + sii;
        sg0 = (max(klo + 1, 0) >> 4) / GT;
#pragma unroll
        for (int gi = 0; gi < WY_RT / GT; ++gi) {
            if (gi >= sg0) {
#pragma unroll
                for (int u = GT * gi; u < GT * gi + GT; ++u) {
                    const int i = sii + 16 * u;
                    pre[u] = (k >= 0 && i > k && i < n) ? src[16 * u] : 0.0;
                }
            }
        }
        pret = Tall[(size_t)c * 256 + tid];
    };
    auto lstore = [&](int buf) {
        double* dst = Vs + ((size_t)buf * NR + sii) * WY_LD + sr;
#pragma unroll
        for (int gi = 0; gi < WY_RT / GT; ++gi) {
            if (gi >= sg0) {
#pragma unroll
                for (int u = GT * gi; u < GT * gi + GT; ++u) dst[16 * u * WY_LD] = pre[u];
            }
        }
        Ts[(buf * 16 + sr) * WY_LD + sii] = pret;
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int c = 0; c < nblk; ++c) {
        const int buf = c & 1, klo = n - 2 - WY_NB * c - (WY_NB - 1);
        if (c + 1 < nblk) gload(c + 1);
        const int rt0 = max(klo + 1, 0) >> 4;               // rows below 16 rt0 are zero in every reflector of the block
        const double* vb = Vs + (size_t)buf * NR * WY_LD;
        // Tiles are handled in groups of GT: one uniform branch per group (tiles above the block's first row are all zero and skipped),
        // the group's 4 GT operand reads issued together ahead of its 4 GT MFMAs - a guard and an LDS round trip per MFMA left the matrix
        // pipe idle most of the time.  Four independent accumulators (one per k-step t) for W.
        f64x4_t Wa[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int gi = 0; gi < WY_RT / GT; ++gi) {
            if (GT * gi + GT - 1 >= rt0) {
                double a[GT][4];
#pragma unroll
                for (int u = 0; u < GT; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) a[u][t] = vb[(16 * (GT * gi + u) + 4 * t + g) * WY_LD + li];   // A = V^T: [reflector li][row]
#pragma unroll
                for (int u = 0; u < GT; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) Wa[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][t], Z[GT * gi + u][t], Wa[t], 0, 0, 0);
            }
        }
        const f64x4_t W = (Wa[0] + Wa[1]) + (Wa[2] + Wa[3]);
        f64x4_t Wn = {0.0, 0.0, 0.0, 0.0};
        {
            double a[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = Ts[(buf * 16 + li) * WY_LD + g + 4 * t];                       // A = -T: [row li][column g + 4 t]
#pragma unroll
            for (int t = 0; t < 4; ++t) Wn = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], W[t], Wn, 0, 0, 0);
        }
#pragma unroll
        for (int gi = 0; gi < WY_RT / GT; ++gi) {
            if (GT * gi + GT - 1 >= rt0) {
                double a[GT][4];
#pragma unroll
                for (int u = 0; u < GT; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) a[u][t] = vb[(16 * (GT * gi + u) + li) * WY_LD + g + 4 * t];   // A = V: [row][reflector g + 4 t]
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < GT; ++u)
                        Z[GT * gi + u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][t], Wn[t], Z[GT * gi + u], 0, 0, 0);
            }
        }
        if (c + 1 < nblk) lstore(buf ^ 1);
        __syncthreads();
    }
    double* Ub = U + (size_t)b * n * n;
#pragma unroll
    for (int rt = 0; rt < WY_RT; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 16 * rt + g + 4 * q;
            if (i < n && j < n) Ub[(size_t)j * n + i] = Z[rt][q];
        }
}

// ------------------------------------------------------------------------------------------ C ABI
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t eigh_layout(int B, int n, EighWs* w, char* base) {
    size_t o = 0;
#define TAKE(field, type, count) do { if (w) w->field = (type*)(base + o); o += al256(sizeof(type) * (size_t)(count)); } while (0)
    TAKE(d, double, (size_t)B * n);
    TAKE(e, double, (size_t)B * n);
    TAKE(tau, double, (size_t)B * n);
    TAKE(lamp, double, (size_t)B * n);
    TAKE(zt, double, (size_t)B * n * EG_MAXN);
    TAKE(lu, double, (size_t)B * 6 * (n + 2) * EG_MAXN);
    TAKE(pin, int, (size_t)B * n * EG_MAXN);
    TAKE(xch, uint4, (size_t)B * EG_XCH * EG_MAXN);
    TAKE(flag, int, (size_t)B + 64);
#undef TAKE
    return o;
}

#ifdef ECS_PROF
extern "C" int nele_ecs_prof_read(unsigned long long* out8) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(ecs_prof), sizeof(unsigned long long) * 8) != hipSuccess;
}
#endif
extern "C" long long nele_eigh_workspace_bytes(int B, int n) { return (long long)eigh_layout(B, n, nullptr, nullptr); }

// Matrices of the LAST nele_eigh_sym_batched call on this workspace that the cluster tridiagonalisation gave up on and the single-workgroup
// repair kernel redid (synchronises the device; 0 on a GPU the launch fits on).  -1 on error.
extern "C" int nele_eigh_repaired(void* workspace, int B, int n) {
    if (!workspace || B <= 0 || n < 2) return -1;
    EighWs ws;
    eigh_layout(B, n, &ws, (char*)workspace);
    int v = -1;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, ws.flag + B, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}

// Device pointer to the per-matrix give-up flags of a workspace ([B] ints, non-zero = the matrix took the repair path in the last call;
// [B] = how many).  Internal (SIIB folds it into its status word so that a training loop can count repairs without a synchronisation).
const int* nele_eigh_flags(void* workspace, int B, int n) {
    EighWs ws;
    eigh_layout(B, n, &ws, (char*)workspace);
    return ws.flag;
}

// A [B][n][n] symmetric (destroyed: holds the Householder reflectors on exit) -> lam [B][n] ascending,
// U [B][n][n] with row j = eigenvector j.  U may alias A? No: U must be a different buffer.
extern "C" int nele_eigh_sym_batched(double* A, int n, int B, double* lam, double* U, void* workspace, long long workspace_bytes,
                                     void* stream) {
    return nele_eigh_sym_batched_ex(A, n, B, lam, U, workspace, workspace_bytes, stream, 0);
}

// Internal entry point (not exported): `cluster_batch` = matrices per cluster launch the caller wants (8 .. 64, 0 = no preference: 64).
// An explicit argument - it used to be a mutable global "hint" set around the call, which two host threads could read crosswise.
int nele_eigh_sym_batched_ex(double* A, int n, int B, double* lam, double* U, void* workspace, long long workspace_bytes, void* stream,
                             int cluster_batch) {
    NELE_CHECK_ARG(A && lam && U && workspace && B > 0 && n >= 2, "nele_eigh_sym_batched: bad arguments");
    NELE_CHECK_ARG(A != U, "nele_eigh_sym_batched: U must not alias A");
    if (n > EG_MAXN) return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_eigh_sym_batched: n=%d > %d", n, EG_MAXN);
    if (workspace_bytes < nele_eigh_workspace_bytes(B, n)) return nele_set_error(NELE_ERR_WORKSPACE, "nele_eigh_sym_batched: workspace too small");
    EighWs ws;
    eigh_layout(B, n, &ws, (char*)workspace);
    hipStream_t s = as_stream(stream);
    // cluster tridiagonalisation when at least 8 matrices' worth of workgroups can be co-resident, else one workgroup per matrix
    static int cluster_cap = -1;                           // matrices per cluster launch (multiple of 8), 0 = unavailable
    if (cluster_cap < 0) {
        int dev = 0, ncu = 0, occ = 0;
        if (!NELE_SWITCH_INT("NELE_EIGH_CLUSTER", 1)) cluster_cap = 0;
        else if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                 hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(eigh_tridiag_cluster_kernel), 512, 0) == hipSuccess)
            cluster_cap = ((long long)ncu * (occ > 0 ? 1 : 0) / (8 * EC_P)) * 8;   // one workgroup per CU: never rely on sharing a CU
        else cluster_cap = 0;
        if (cluster_cap > 32) cluster_cap = 32;
        const int capenv = NELE_SWITCH_INT("NELE_EIGH_CLUSTER_CAP", 0);    // matrices per launch (multiple of 8): fewer leaves CUs to other streams
        if (capenv >= 8 && capenv < cluster_cap) cluster_cap = capenv / 8 * 8;
    }
    int c2_first = -1;                                     // first step of the second cluster stage when it runs (its give-ups are repaired from there)
    bool repaired_inline = false;                          // eigh_tridiag_midx_kernel redoes flagged matrices itself
    if (cluster_cap >= 8) {
        // once per device: may the exchange stores stay in the XCD's L2 (st_tagged)?  128 pairs of workgroups = one per CU, laid out like
        // the two-workgroup launches, 64 rounds of ping-pong each; the device-side flag stays 0 (write-through stores) unless all pass.
        // NELE_EIGH_XCH_KEEP=0 (test library) skips the probe.
        if (NELE_SWITCH_INT("NELE_EIGH_XCH_KEEP", 1)) {
            NELE_ONCE_PER_DEVICE({
                if (hipMemsetAsync(ws.xch, 0, sizeof(uint4) * 256, s) == hipSuccess)
                    hipLaunchKernelGGL(eigh_xch_probe_kernel, dim3(256), dim3(64), 0, s, reinterpret_cast<u32x4*>(ws.xch), 128, 64);
            });
        }
        if (hipMemsetAsync(ws.xch, 0, sizeof(uint4) * (size_t)B * EG_XCH * EG_MAXN, s) != hipSuccess) return nele_set_error(NELE_ERR_HIP, "nele_eigh_sym_batched: memset failed");
        if (hipMemsetAsync(ws.flag, 0, sizeof(int) * ((size_t)B + 1), s) != hipSuccess) return nele_set_error(NELE_ERR_HIP, "nele_eigh_sym_batched: memset failed");
        // a launch owns 8 CUs per matrix for ~2 ms whatever the count (the kernel is latency-bound per matrix): small batches go in
        // two half-size launches, which leaves half of the CUs to the other streams (measured 2 % on the whole step at B = 32)
        const int p4_on = NELE_SWITCH_INT("NELE_EIGH_P4", 1);
        if (p4_on && n <= 16 * E4_RI && cluster_cap >= 32) {
            // four workgroups per matrix: 64 matrices per launch would fill the chip; 32 use half of it.  The last ET_M steps run in
            // eigh_tridiag_tail_kernel (one workgroup per matrix, block in LDS): s_stop = last iteration of the cluster kernel
            const int tail_on = NELE_SWITCH_INT("NELE_EIGH_TAIL", 1);
            NELE_AB_ONLY(NELE_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(eigh_tridiag_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                        (int)(sizeof(double) * ET_M * ET_M)));)
            const int mid_on = NELE_SWITCH_INT("NELE_EIGH_MID", 1);                        // NELE_EIGH_MID=0: hand over to the LDS tail kernel at 128 instead (A/B diagnostic)
            const int midx_on = NELE_SWITCH_INT("NELE_EIGH_MIDX", 1);   // =0: hand over at 224 (registers only) instead of 256 (registers + LDS strip)
            NELE_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(eigh_tridiag_midx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
            const int mhand = (tail_on != 0 && mid_on != 0) ? (midx_on ? EM_M + EX_E : EM_M) : ET_M;
            const int s_stop = (tail_on && n > mhand + 2) ? n - mhand - 2 : -2;
            // second cluster stage (round 4): once 320 rows are left, two workgroups per matrix hold the block - twice the matrices per launch
            // for the steps from 320 down to the single-workgroup hand-over (NELE_EIGH_C2=0: the four-workgroup kernel runs them all)
            const int c2_on = NELE_SWITCH_INT("NELE_EIGH_C2", 1);
            NELE_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(eigh_tridiag_cluster2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                           (int)(sizeof(double) * EC2_RI * 16 * 32)));
            const bool two_stage = c2_on && s_stop >= 0 && mhand < EC2_M && n > EC2_M + 8;
            const int s_stop1 = two_stage ? n - EC2_M - 2 : s_stop;
            // first stage on the lower triangle with two workgroups per matrix (round 4; NELE_EIGH_SYM=0: the four-workgroup full-storage kernel)
            const int sym_on = NELE_SWITCH_INT("NELE_EIGH_SYM", 1);
            NELE_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void*>(eigh_tridiag_clusters_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                           (int)(sizeof(double) * ECS_RI * 16 * 32)));
            const bool sym_stage = sym_on && two_stage && n <= 16 * ECS_RI && s_stop1 + 2 < 128;
            // NELE_EIGH_P4_BATCH: matrices per launch (32 = half of the chip inside a training step, 64 = all of it otherwise).  The spinning workgroups own
            // their CU - registers full, issue slots mostly idle - so 32 per launch (twice the launches) leaves room for the step's other
            // streams while the chain's own time doubles (8 x 1.2 instead of 4 x 1.2 ms per 256 matrices).  Measured twice on the
            // B = 256 step, alternating on one box: 45.4 / 45.5 / 46.0 against 46.1 / 46.2 / 46.3 ms (second session), 42.8 / 42.5
            // against 43.7 / 43.3 ms (third session; 48 per launch: 42.9 / 42.4) - the default since the second measurement.
            // The caller says which: SIIB's clean-signal phase inside a training step (nele_metric_siib phase 3, which runs beside the
            // G-step) asks for 32 through nele_eigh_sym_batched_ex's argument; a stand-alone call (one-shot SIIB, nele_eigh_sym_batched by
            // itself) has nothing to share the chip with and takes 64 - one SIIB call at B = 256 is 17 ms that way and 21.5 ms with 32.
            const int p4_raw = NELE_SWITCH_INT("NELE_EIGH_P4_BATCH", 0);
            const int p4_env = (p4_raw >= 8 && p4_raw <= 64) ? p4_raw / 8 * 8 : 0;
            const int hint = cluster_batch;
            const int p4_batch = p4_env ? p4_env : (hint >= 8 && hint <= 64 ? hint / 8 * 8 : 64);
            const int fail_every = NELE_SWITCH_INT("NELE_EIGH_FAIL_EVERY", 0);                     // NELE_EIGH_FAIL_EVERY=k (tests): every k-th matrix takes the give-up / repair path
            if (sym_stage) {
                const int cs_batch = 2 * p4_batch;
                for (int b0 = 0; b0 < B; b0 += cs_batch) {
                    const int Bc = (B - b0 < cs_batch) ? B - b0 : cs_batch;
                    NELE_PROF("eigh_tridiag_cluster", s,
                              hipLaunchKernelGGL(eigh_tridiag_clusters_kernel, dim3(16 * ((Bc + 7) / 8)), dim3(512), sizeof(double) * ECS_RI * 16 * 32, s, A, n, b0, Bc,
                                                 ws, s_stop1, fail_every));
                }
            } else
            for (int b0 = 0; b0 < B; b0 += p4_batch) {
                const int Bc = (B - b0 < p4_batch) ? B - b0 : p4_batch;
                NELE_PROF("eigh_tridiag_cluster", s,
                          hipLaunchKernelGGL(eigh_tridiag_cluster4_kernel, dim3(32 * ((Bc + 7) / 8)), dim3(512), 0, s, A, n, b0, Bc, ws, s_stop1, fail_every));
            }
            if (two_stage) {
                c2_first = s_stop1 + 1;
                const int c2_batch = 2 * p4_batch;
                for (int b0 = 0; b0 < B; b0 += c2_batch) {
                    const int Bc = (B - b0 < c2_batch) ? B - b0 : c2_batch;
                    NELE_PROF("eigh_tridiag_cluster", s,
                              hipLaunchKernelGGL(eigh_tridiag_cluster2_kernel, dim3(16 * ((Bc + 7) / 8)), dim3(512), sizeof(double) * EC2_RI * 16 * 32, s, A, n, b0,
                                                 Bc, ws, s_stop1 + 1, s_stop, fail_every));
                }
            }
            if (s_stop >= -1) {
                const int mt = n - (s_stop + 2);
                if (mhand == EM_M + EX_E) { hipLaunchKernelGGL(eigh_tridiag_midx_kernel, dim3(B), dim3(512), sizeof(double) * EX_E * EX_LD, s, A, n, ws, s_stop + 1, B, c2_first); repaired_inline = true; }
                NELE_AB_ONLY(else if (mhand == EM_M) hipLaunchKernelGGL(eigh_tridiag_mid_kernel, dim3(B), dim3(512), 0, s, A, n, ws, s_stop + 1);
                             else hipLaunchKernelGGL(eigh_tridiag_tail_kernel, dim3(B), dim3(512), sizeof(double) * (size_t)mt * mt, s, A, n, ws, s_stop + 1);)
            }
        } else {
        const int split_small = NELE_SWITCH_INT("NELE_EIGH_SPLIT", 1);
        const int per = (split_small && B <= 32 && cluster_cap >= 32) ? 16 : cluster_cap;
        for (int b0 = 0; b0 < B; b0 += per) {
            const int Bc = (B - b0 < per) ? B - b0 : per;
            hipLaunchKernelGGL(eigh_tridiag_cluster_kernel, dim3(64 * ((Bc + 7) / 8)), dim3(512), 0, s, A, n, b0, Bc, ws);
        }
        }
        // matrices a cluster launch gave up on (never on a GPU the launch fits on): redone by one workgroup each instead of NaN results -
        // inside eigh_tridiag_midx_kernel where that kernel runs (the two-stage cluster path: SIIB's n = 420), else by these two
        if (!repaired_inline) {
            hipLaunchKernelGGL(eigh_tridiag_repair_kernel, dim3(B), dim3(1024), 0, s, A, n, ws, B);
            if (c2_first >= 0) hipLaunchKernelGGL(eigh_tridiag_repair2_kernel, dim3(B), dim3(1024), 0, s, A, n, ws, B, c2_first);
        }
    } else {
        hipLaunchKernelGGL(eigh_tridiag_kernel, dim3(B), dim3(1024), 0, s, A, n, ws);
    }
    hipLaunchKernelGGL(eigh_sde_kernel, dim3(B), dim3(256), 0, s, n, ws);
    hipLaunchKernelGGL(eigh_bisect_kernel, dim3((n + 256 / EG_NL - 1) / (256 / EG_NL), B), dim3(256), 0, s, n, ws, lam);
    // sweeps after the growth criterion is met: LAPACK's dstein uses EXTRA = 2 and documents "should be at least 1"; with the eigenvalues
    // bisected to 1 ulp one is enough for every test matrix (eigenvalues 1e-13, residual 1e-11, orthogonality 1e-8, SIIB unchanged to 17
    // digits) and saves a quarter of this kernel's HBM traffic (its work arrays: 108 KB per eigenvector); 0 fails the residual test
    const int iv_extra = NELE_SWITCH_INT("NELE_EIGH_INVIT_EXTRA", 1);
    // NELE_EIGH_INVIT_STORE=1 (test library): the kernel that stores the LU factors instead of recomputing them (bit-identical; 6.2 GB per 256 matrices)
    if (NELE_SWITCH_INT("NELE_EIGH_INVIT_STORE", 0)) { NELE_AB_ONLY(hipLaunchKernelGGL(eigh_invit_kernel, dim3((n + 63) / 64, B), dim3(64), 0, s, n, ws, lam, iv_extra);) }
    else hipLaunchKernelGGL(eigh_invit2_kernel, dim3((n + 63) / 64, B), dim3(64), 0, s, n, ws, lam, iv_extra);
    const size_t lds = sizeof(double) * (2 * (size_t)BT_CH * 16 * (n <= 448 ? 28 : 32) + 2 * BT_CH);
    const int wy_on = NELE_SWITCH_INT("NELE_EIGH_WY", 1);   // =0: the reflector-by-reflector back-transformation (A/B diagnostic)
    NELE_ONCE_PER_DEVICE({
        NELE_AB_ONLY((void)hipFuncSetAttribute(reinterpret_cast<const void*>(eigh_backtransform_kernel<28>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
                     (void)hipFuncSetAttribute(reinterpret_cast<const void*>(eigh_backtransform_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(eigh_backtransform_wy_kernel<27>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(eigh_backtransform_wy_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    });
    if (wy_on) {
        const int nblk = (n - 1 + WY_NB - 1) / WY_NB;
        hipLaunchKernelGGL(eigh_wy_t_kernel, dim3((nblk + 3) / 4, B), dim3(256), 0, s, A, n, ws);
        if (n <= 432) hipLaunchKernelGGL(eigh_backtransform_wy_kernel<27>, dim3((n + 63) / 64, B), dim3(256), sizeof(double) * (2 * 432 * WY_LD + 2 * 16 * WY_LD), s, A, n, ws, U);
        else hipLaunchKernelGGL(eigh_backtransform_wy_kernel<32>, dim3((n + 63) / 64, B), dim3(256), sizeof(double) * (2 * 512 * WY_LD + 2 * 16 * WY_LD), s, A, n, ws, U);
    }
    NELE_AB_ONLY(else if (n <= 448) hipLaunchKernelGGL(eigh_backtransform_kernel<28>, dim3((n + 31) / 32, B), dim3(256), lds, s, A, n, ws, U);
                 else hipLaunchKernelGGL(eigh_backtransform_kernel<32>, dim3((n + 31) / 32, B), dim3(256), lds, s, A, n, ws, U);)
    NELE_CHECK_LAUNCH("nele_eigh_sym_batched");
    return NELE_OK;
}
