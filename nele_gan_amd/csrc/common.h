// Shared host/device helpers for libnele_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#define NELE_OK 0
#define NELE_ERR_INVALID_ARG (-1)
#define NELE_ERR_UNSUPPORTED (-2)
#define NELE_ERR_SIGNAL (-3)   // signal below threshold / too short (reference raises)
#define NELE_ERR_HIP (-4)
#define NELE_ERR_WORKSPACE (-5)

#define NELE_NFFT 512
#define NELE_HOP 256
#define NELE_NBINS 257
#define NELE_NBANDS 64

int nele_set_error(int code, const char* fmt, ...);

// csrc/eigh.hip (also part of the public C ABI)
__attribute__((visibility("hidden"))) int nele_eigh_sym_batched_ex(double* A, int n, int B, double* lam, double* U, void* workspace,
                                                                   long long workspace_bytes, void* stream, int cluster_batch);
__attribute__((visibility("hidden"))) const int* nele_eigh_flags(void* workspace, int B, int n);
extern "C" long long nele_eigh_workspace_bytes(int B, int n);
extern "C" int nele_eigh_sym_batched(double* A, int n, int B, double* lam, double* U, void* workspace, long long workspace_bytes,
                                     void* stream);

#define NELE_CHECK_ARG(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) return nele_set_error(NELE_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)

#define NELE_CHECK_LAUNCH(name)                                                      \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess)                                                        \
            return nele_set_error(NELE_ERR_HIP, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- A/B switches.  The PRODUCT library (libnele_hip.so) has ONE path per operation: NELE_SWITCH_INT(name, default) is the constant
// `default` there, so the dispatch of every superseded variant folds away at compile time.  The TEST library libnele_hip_ab.so (same
// sources and ABI, built with -DNELE_AB by the same Makefile) reads the environment once per process; the A/B tests under tests/ and
// the scripts under tools/ load it through NELE_LIB.  Nothing in the product path reads the environment.
#ifdef NELE_AB
int nele_env_int(const char* name, int dflt);              // csrc/capi.hip: atoi(getenv(name)) or dflt
#define NELE_SWITCH_INT(name, dflt) ([]() -> int { static const int v_ = nele_env_int(name, dflt); return v_; }())
#define NELE_AB_ONLY(...) __VA_ARGS__                      // launches of superseded kernels (their definitions sit in #ifdef NELE_AB)
#else
#define NELE_SWITCH_INT(name, dflt) (dflt)
#define NELE_AB_ONLY(...)
#endif

// csrc/capi.hip: true the first time it is called with this mask on the CURRENT device (one-time hipFuncSetAttribute blocks: a process
// that switches devices must set the attribute on each of them)
bool nele_first_use_on_device(unsigned long long* mask);
#define NELE_ONCE_PER_DEVICE(body)                                   \
    do {                                                             \
        static unsigned long long once_mask_ = 0;                    \
        if (nele_first_use_on_device(&once_mask_)) { body; }         \
    } while (0)

// csrc/capi.hip: HIP-event pair around one tagged launch when nele_profile_begin(tag) armed it (bench.py's roofline figures)
bool nele_prof_armed();
bool nele_prof_match(const char* tag);
void nele_prof_mark(hipStream_t s);
#define NELE_PROF(tag, stream, launch)                      \
    do {                                                    \
        const bool prof_ = nele_prof_match(tag);            \
        if (prof_) nele_prof_mark(stream);                  \
        launch;                                             \
        if (prof_) nele_prof_mark(stream);                  \
    } while (0)

// ---- wave64 reductions (fixed-order butterflies: deterministic) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// wave64 sum with DPP row operations + readlane (fixed order; ~4x shorter dependency chain than the ds_bpermute butterflies)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)u, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double lane_value(double v, int l) {
    const long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)u, l), hi = __builtin_amdgcn_readlane((int)(u >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_move<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);    // row_half_mirror
    v += dpp_move<0x140>(v);    // row_mirror: every lane of a 16-lane row holds the row sum
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
// sum over each 16-lane row (every lane of the row receives it)
__device__ __forceinline__ double row16_sum_dpp(double v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return v;
}

// value held by lane (l ^ 32): one v_permlane32_swap per 32-bit half (gfx950) instead of two ds_bpermute through the LDS crossbar
__device__ __forceinline__ double lane_xor32(double v) {
    const long long u = __double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)u, (unsigned)u, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
    const bool up = (threadIdx.x & 32) != 0;
    const unsigned l = up ? lo[0] : lo[1], h = up ? hi[0] : hi[1];
    return __longlong_as_double(((long long)h << 32) | (long long)l);
}

// value held by lane (l ^ 16): v_permlane16_swap (gfx950), no LDS crossbar
__device__ __forceinline__ double lane_xor16(double v) {
    const long long u = __double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
    const bool up = (threadIdx.x & 16) != 0;
    const unsigned l = up ? lo[0] : lo[1], h = up ? hi[0] : hi[1];
    return __longlong_as_double(((long long)h << 32) | (long long)l);
}

// Block-wide sum of doubles through LDS scratch (>= blockDim/64 doubles); result to all threads.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}
__device__ __forceinline__ double block_max(double v, double* scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double t = scratch[0];
    for (int i = 1; i < nw; ++i) t = fmax(t, scratch[i]);
    return t;
}

// x ** p of the band features (audio_util.py:434, 451: `compute_band_E(...) ** p_power`, a float32 array to a Python float: numpy's float32
// loop, i.e. powf(x, float32(p)) - itself only accurate to an ulp).  This build returns the float32 NEAREST to x ^ float32(p).
// General p: the float64 pow (267 instructions - a quarter of the STFT kernel's issue slots at two calls per frame pair).  p = float32(1/6),
// what every caller passes (train_nele.py:39 p_power): x = m6 2^(6q), m6 in [0.5, 32); m6^(1/6) from the float32 hardware log2 / exp2
// (relative error ~2e-7) and ONE Newton step on y^6 = m6 in float64 (error 2.5 e^2 = 1e-13; the quotient by a float32 reciprocal: the
// correction term is itself 1e-6); times x^(float32(1/6) - 1/6) = 1 + ln(x) 4.97e-9.  3e-14 relative before the rounding to float32: the
// result differs from the float64 pow's in 6 of 10^7 values (a tie broken the other way, 1 ulp) - the reference's own powf differs from
// both in every fifth.  ~35 instructions.  Also the G-step glue's mask ** p (train_nele.py:137, gen.hip).
__device__ __forceinline__ float nele_pow_f32(float x, float p) {
    if (p != (float)(1.0 / 6.0)) return (float)pow((double)x, (double)p);          // (uniform: p is a kernel argument)
    const bool ok = x > 0.f && x < __builtin_inff();
    const double d = ok ? (double)x : 1.0;                                           // float32 subnormals are normal doubles
    const int e = __builtin_amdgcn_frexp_exp(d);                                     // d = m 2^e, m in [0.5, 1)
    const double m = __builtin_amdgcn_frexp_mant(d);
    const int q = (int)(((unsigned)(e + 1536) * 43691u) >> 18) - 256;               // floor(e / 6) for |e| <= 1100
    const int r = e - 6 * q;                                                         // 0 .. 5
    const double m6 = ldexp(m, r);                                                   // [0.5, 32): 24 significant bits, exact as a float
    const float l2 = __builtin_amdgcn_logf((float)m6);                               // v_log_f32
    const double y0 = (double)__builtin_amdgcn_exp2f(l2 * (float)(1.0 / 6.0));       // v_exp_f32
    const double y2 = y0 * y0, z = y2 * y2 * y2;
    const double rc = (double)__builtin_amdgcn_rcpf((float)(6.0 * z));
    double y1 = fma(-y0, (z - m6) * rc, y0);                                         // Newton: y - (y^6 - m6) / (6 y^5)
    const double eps_ln2 = ((double)(float)(1.0 / 6.0) - 1.0 / 6.0) * 0.6931471805599453;
    y1 = fma(y1, ((double)(6 * q) + (double)l2) * eps_ln2, y1);                      // x ^ (float32(1/6) - 1/6)
    const float v = (float)ldexp(y1, q);
    return ok ? v : (x == 0.f ? 0.f : (x > 0.f ? x : __builtin_nanf("")));           // 0 -> 0, inf -> inf, negative / NaN -> NaN
}

// x ** inv_p of the G-step glue (train_nele.py:133 / inference.py:100 `torch.pow(clean_in, inv_p)`, inv_p = 6): for the integer 6 three
// float64 products, rounded once - the float32 nearest to x^6 (torch's powf is within an ulp of it); energy_norm_fwd_kernel spent three
// quarters of its time in powf(x, 6.0f).  Any other exponent: powf.
__device__ __forceinline__ float nele_powi_f32(float x, float inv_p) {
    if (inv_p != 6.0f) return powf(x, inv_p);                                        // (uniform: a kernel argument)
    const double d = (double)x, d2 = d * d;
    return (float)(d2 * d2 * d2);
}
