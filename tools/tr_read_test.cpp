#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[16 * 16];
    for (int i = threadIdx.x; i < 256; i += 64) { int r = i / 16, c = i % 16; tile[i] = (__bf16)(float)(r + 16 * c); }
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    auto v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(&tile[(4 * g + q) * 16 + 4 * p]));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (float)v[e];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) { int g = l >> 4, i = l & 15; float exp = (4 * g + e) + 16 * i; if (h[l*4+e] != exp) { if (bad < 8) printf("lane %d e %d got %g exp %g\n", l, e, h[l*4+e], exp); ++bad; } }
    printf("mismatches %d\n", bad);
    return 0;
}
