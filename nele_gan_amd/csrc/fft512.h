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
            double2 w = s.tw[pos << (8 - st)];
            if (INVERSE) w.y = -w.y;
            const double2 a = s.x[i0], b = s.x[i1];
            const double tr = w.x * b.x - w.y * b.y;
            const double ti = w.x * b.y + w.y * b.x;
            s.x[i0] = make_double2(a.x + tr, a.y + ti);
            s.x[i1] = make_double2(a.x - tr, a.y - ti);
        }
        __syncthreads();
    }
}
