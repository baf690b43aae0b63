#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// mode 0: tagged 16B slots sc1 ; mode 1: 8-byte agent atomics ; mode 2: tagged, plain "sc1" load but store via atomic 8B
template <int MODE>
__global__ void pp(u32x4* slots, unsigned long long* flags, int rounds, int nwg, int stride) {
    const int w = blockIdx.x / stride;      // participating WGs are blockIdx = k*stride
    if (blockIdx.x % stride != 0 || w >= nwg) return;
    if (threadIdx.x != 0) return;
    for (int r = 1; r <= rounds; ++r) {
        // every WG publishes r, then waits for all others' r  (all-to-all barrier of nwg WGs)
        if (MODE == 0) {
            u32x4 q; q.x = r; q.y = r; q.z = r; q.w = r;
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&slots[w]), "v"(q) : "memory");
            for (int o = 0; o < nwg; ++o) {
                u32x4 g;
                do { asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(&slots[o]) : "memory"); } while (g.y < (unsigned)r);
            }
        } else if (MODE == 1) {
            __hip_atomic_store(&flags[w * 16], (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int o = 0; o < nwg; ++o)
                while (__hip_atomic_load(&flags[o * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)r) {}
        } else {
            __hip_atomic_fetch_add(&flags[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)r * nwg) {}
        }
    }
}
int main() {
    u32x4* slots; unsigned long long* flags;
    hipMalloc(&slots, 4096 * 16); hipMalloc(&flags, 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 2000;
    for (int stride : {8, 1}) for (int nwg : {2, 8}) for (int mode = 0; mode < 3; ++mode) {
        hipMemset(slots, 0, 4096 * 16); hipMemset(flags, 0, 4096 * 8);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(pp<0>, dim3(nwg * stride), dim3(64), 0, 0, slots, flags, rounds, nwg, stride);
        if (mode == 1) hipLaunchKernelGGL(pp<1>, dim3(nwg * stride), dim3(64), 0, 0, slots, flags, rounds, nwg, stride);
        if (mode == 2) hipLaunchKernelGGL(pp<2>, dim3(nwg * stride), dim3(64), 0, 0, slots, flags, rounds, nwg, stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("stride %d (same XCD if 8) nwg %d mode %d: %.3f us / round\n", stride, nwg, mode, ms * 1000 / rounds); fflush(stdout);
    }
    return 0;
}
