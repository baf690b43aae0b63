// Which compute units does a stream created with hipExtStreamCreateWithCUMask use on this part?  Every workgroup records its
// (XCC, SE, CU) from the hardware-id registers while spinning long enough for the whole grid to be resident.
// hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.cpp -o /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where(unsigned* out, long long spin) {
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}
int main() {
    const int nb = 4096;
    unsigned* d; hipMalloc(&d, nb * 8);
    std::vector<unsigned> h(nb * 2);
    struct Case { const char* name; std::vector<uint32_t> mask; };
    std::vector<Case> cases = {
        {"all", {}},
        {"low32 bits", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
        {"low64 bits", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0}},
        {"low128 bits", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}},
        {"every word 0x00ffffff", {0x00ffffffu, 0x00ffffffu, 0x00ffffffu, 0x00ffffffu, 0x00ffffffu, 0x00ffffffu, 0x00ffffffu, 0x00ffffffu}},
        {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
        {"bits 0-7 only", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
    };
    for (auto& c : cases) {
        hipStream_t s;
        hipError_t e = c.mask.empty() ? hipStreamCreate(&s) : hipExtStreamCreateWithCUMask(&s, (uint32_t)c.mask.size(), c.mask.data());
        if (e != hipSuccess) { printf("%s: create failed: %s\n", c.name, hipGetErrorString(e)); continue; }
        hipMemsetAsync(d, 0xff, nb * 8, s);
        hipLaunchKernelGGL(where, dim3(nb), dim3(64), 0, s, d, 2000000LL);   // 100 MHz wall clock: 20 ms
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
        std::set<unsigned> cus; int per_xcc[16] = {0};
        std::set<unsigned> perx[16];
        for (int i = 0; i < nb; ++i) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            const unsigned id = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            cus.insert(id); perx[xcc].insert(id);
        }
        printf("%-24s distinct CUs %3zu  per XCC:", c.name, cus.size());
        for (int x = 0; x < 8; ++x) printf(" %2zu", perx[x].size());
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
