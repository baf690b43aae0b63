// 512-point complex FFT in float64, in place in LDS, 256 threads (one radix-2 butterfly per thread
// per stage).  Two real 512-point transforms are packed into one complex transform by the callers
// (z = a + i*b), so one call serves two STFT / iSTFT frames.
#pragma once
#include "common.h"

struct Fft512Lds {
    double2 x[512];   // data (input must be stored in bit-reversed order: see fft512_brev)
    double2 tw[256];  // exp(-2*pi*i*k/512), k = 0..255
};

__device__ __forceinline__ int fft512_brev(int i) { return (int)(__brev((unsigned)i) >> 23); }

__device__ __forceinline__ void fft512_init_twiddles(Fft512Lds& s) {
    const int k = threadIdx.x;
    if (k < 256) {
        double sn, cs;
        sincospi((double)k / 256.0, &sn, &cs);
        s.tw[k] = make_double2(cs, -sn);
    }
}

// One radix-2 butterfly, a' = a + w b, b' = a - w b, as eight fused multiply-adds (round 6; ten separate multiplies / adds before - this
// header is compiled without FMA contraction in features.hip): a'.x = fma(-w.y, b.y, fma(w.x, b.x, a.x)) and so on.  With a trivial twiddle
// (1 or -+i, which the table holds exactly) the fused form IS the plain sum / difference, so code that skips those multiplications
// (fftw_triple0) gives the same bits.
template <bool INVERSE>
__device__ __forceinline__ void fft512_bfly(double2& a, double2& b, const double2 w) {
    const double wy = INVERSE ? -w.y : w.y;
    const double2 a0 = a, b0 = b;
    a = make_double2(fma(-wy, b0.y, fma(w.x, b0.x, a0.x)), fma(wy, b0.x, fma(w.x, b0.y, a0.y)));
    b = make_double2(fma(wy, b0.y, fma(-w.x, b0.x, a0.x)), fma(-wy, b0.x, fma(-w.x, b0.y, a0.y)));
}

// Forward (INVERSE=false): X[k] = sum x[n] exp(-2 pi i k n / 512).
// Inverse (INVERSE=true): unnormalised, x[n] = sum X[k] exp(+2 pi i k n / 512).
// Caller stores input element n at s.x[fft512_brev(n)], then __syncthreads(); output is in
// natural order and visible after the trailing barrier.
template <bool INVERSE>
__device__ __forceinline__ void fft512_run(Fft512Lds& s) {
    const int j = threadIdx.x;  // 0..255 take part
#pragma unroll
    for (int st = 0; st < 9; ++st) {
        const int half = 1 << st;
        if (j < 256) {
            const int pos = j & (half - 1);
            const int i0 = ((j >> st) << (st + 1)) + pos;
            const int i1 = i0 + half;
            double2 a = s.x[i0], b = s.x[i1];
            fft512_bfly<INVERSE>(a, b, s.tw[pos << (8 - st)]);
            s.x[i0] = a;
            s.x[i1] = b;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same transform, ONE WAVE per 512-point FFT with the data in registers (round 3, third session).  fft512_run() pays a workgroup
// barrier and five LDS accesses per butterfly stage; here a lane holds 8 complex values and runs three stages at a time in registers,
// so nine stages cost two LDS exchanges and no workgroup barrier (the wave's LDS accesses execute in order).  Every butterfly is the
// SAME operation on the SAME operands with the SAME twiddle (tw[pos << (8 - st)]) as in fft512_run() (fft512_bfly; trivial twiddles skipped,
// which changes no bit): results are bit-identical in a translation unit compiled without FMA contraction (features.hip).
//   in : v[r] = element (8 * lane + r) of the bit-reversed-order array, i.e. input sample  n = fftw_n(lane, r)
//   out: v[m] = X[64 * m + lane]
// xw: the wave's exchange buffer (FFTW_SLOTS double2; position i lives in slot i + (i >> 3): conflict-free 16-byte accesses in all
// three access patterns); on return it holds X in natural order at fftw_slot(k) and is ordered for reading by any lane of the wave.
#define FFTW_SLOTS 576
__device__ __forceinline__ int fftw_slot(int i) { return i + (i >> 3); }
__device__ __forceinline__ int fftw_n(int lane, int r) {
    return (int)(__brev((unsigned)lane) >> 26) + 64 * (((r & 1) << 2) | (r & 2) | ((r >> 2) & 1));
}
// orders the LDS accesses of ONE wave: DS instructions of a wave execute in issue order, so the compiler only has to keep them in place
__device__ __forceinline__ void fftw_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool INVERSE>
__device__ __forceinline__ void fftw_bfly(double2& a, double2& b, double2 w) { fft512_bfly<INVERSE>(a, b, w); }
// w = 1
__device__ __forceinline__ void fftw_bfly1(double2& a, double2& b) {
    const double2 a0 = a, b0 = b;
    a = make_double2(a0.x + b0.x, a0.y + b0.y);
    b = make_double2(a0.x - b0.x, a0.y - b0.y);
}
// w = -i (forward) / +i (inverse): w b = (b.y, -b.x) / (-b.y, b.x)
template <bool INVERSE>
__device__ __forceinline__ void fftw_bflyI(double2& a, double2& b) {
    const double2 a0 = a, b0 = b;
    if (INVERSE) {
        a = make_double2(a0.x - b0.y, a0.y + b0.x);
        b = make_double2(a0.x + b0.y, a0.y - b0.x);
    } else {
        a = make_double2(a0.x + b0.y, a0.y - b0.x);
        b = make_double2(a0.x - b0.y, a0.y + b0.x);
    }
}
// stages 0-2 of the transform (twiddle position 0): the twiddles are 1, -+i and exp(-+i pi / 4), exp(-+3 i pi / 4) - 10 of the 12
// butterflies multiply by 1 or -+i
template <bool INVERSE>
__device__ __forceinline__ void fftw_triple0(double2 (&v)[8], const double2* __restrict__ tw) {
#pragma unroll
    for (int m = 0; m < 8; m += 2) fftw_bfly1(v[m], v[m + 1]);
#pragma unroll
    for (int m = 0; m < 8; m += 4) {
        fftw_bfly1(v[m], v[m + 2]);
        fftw_bflyI<INVERSE>(v[m + 1], v[m + 3]);
    }
    fftw_bfly1(v[0], v[4]);
    fftw_bfly<INVERSE>(v[1], v[5], tw[64]);
    fftw_bflyI<INVERSE>(v[2], v[6]);
    fftw_bfly<INVERSE>(v[3], v[7], tw[192]);
}

// three consecutive stages on the 8 register values; stage s of the triple pairs (m, m + 2^s); pos0/1/2 = twiddle position of element 0
template <bool INVERSE, int ST>
__device__ __forceinline__ void fftw_triple(double2 (&v)[8], const double2* __restrict__ tw, int p) {
    // stage ST: pos = p
    {
        const double2 w = tw[p << (8 - ST)];
#pragma unroll
        for (int m = 0; m < 8; m += 2) fftw_bfly<INVERSE>(v[m], v[m + 1], w);
    }
    // stage ST + 1: pos = p + 2^ST * (m & 1)
    {
        const double2 w0 = tw[p << (7 - ST)], w1 = tw[(p + (1 << ST)) << (7 - ST)];
#pragma unroll
        for (int m = 0; m < 8; m += 4) {
            fftw_bfly<INVERSE>(v[m], v[m + 2], w0);
            fftw_bfly<INVERSE>(v[m + 1], v[m + 3], w1);
        }
    }
    // stage ST + 2: pos = p + 2^ST * m, m < 4
#pragma unroll
    for (int m = 0; m < 4; ++m) fftw_bfly<INVERSE>(v[m], v[m + 4], tw[(p + (m << ST)) << (6 - ST)]);
}

template <bool INVERSE>
__device__ __forceinline__ void fft512_wave(double2 (&v)[8], double2* __restrict__ xw, const double2* __restrict__ tw, int lane) {
    fftw_triple0<INVERSE>(v, tw);                            // stages 0-2 inside blocks of 8 positions
#pragma unroll
    for (int r = 0; r < 8; ++r) xw[9 * lane + r] = v[r];     // slot(8 lane + r)
    fftw_wave_sync();
    const int B = lane >> 3, p = lane & 7;                   // stages 3-5: positions 64 B + 8 m + p
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = xw[72 * B + 9 * m + p];
    fftw_triple<INVERSE, 3>(v, tw, p);
#pragma unroll
    for (int m = 0; m < 8; ++m) xw[72 * B + 9 * m + p] = v[m];
    fftw_wave_sync();
    const int s0 = lane + (lane >> 3);                       // stages 6-8: positions 64 m + lane
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = xw[72 * m + s0];
    fftw_triple<INVERSE, 6>(v, tw, lane);
}

// X (as left in v by fft512_wave) to the exchange buffer in natural order, readable by every lane of the wave afterwards
__device__ __forceinline__ void fft512_wave_store(const double2 (&v)[8], double2* __restrict__ xw, int lane) {
    const int s0 = lane + (lane >> 3);
#pragma unroll
    for (int m = 0; m < 8; ++m) xw[72 * m + s0] = v[m];
    fftw_wave_sync();
}
