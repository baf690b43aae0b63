// standalone harness for conv_tile16_kernel on the D.conv5 forward geometry
#define T16_PROF
#include "../nele_gan_amd/csrc/dense.hip"
#include <vector>
int nele_set_error(int code, const char* fmt, ...) { printf("error %d: %s\n", code, fmt); return code; }
int main(int argc, char** argv) {
    const int B = 32, H = 52, W = 239, C = 48, KH = 9, KW = 9, N = 64;
    ConvGeom g; g.H = H; g.W = W; g.C = C; g.ih0 = 0; g.iw0 = 0; g.Hout = H - 8; g.Wout = W - 8; g.seglen = KW * C; g.segstride = W * C;
    g.Ktot = KH * KW * C; g.OH = g.Hout; g.OW = g.Wout; g.OC = N; g.oh0 = 0; g.ow0 = 0;
    const size_t na = (size_t)B * H * W * C, no = (size_t)B * g.Hout * g.Wout * N;
    float *A, *out, *bias; __bf16* wf; long long* dbg;
    hipMalloc(&A, na * 4); hipMalloc(&out, no * 4); hipMalloc(&bias, 64 * 4); hipMalloc(&dbg, 128);
    const int sps = (g.seglen + 31) / 32, NT = 4;
    hipMalloc(&wf, (size_t)KH * sps * NT * 64 * 8 * 2 + 65536);
    hipMemset(A, 0, na * 4); hipMemset(wf, 0, (size_t)KH * sps * NT * 64 * 8 * 2 + 65536); hipMemset(bias, 0, 256);
    Tile16Args t; t.A = A; t.Wfrag = wf; t.bias = bias; t.aux = nullptr; t.out = out; t.N = N; t.NT = NT; t.epi = EPI_BIAS_LRELU; t.slope = 0.3f;
    t.KH = KH; t.KW = KW; t.steps_per_seg = sps; t.g = g; t.SB = tile16_sb(g); t.dbg = dbg;
    const size_t lds = tile16_lds(g, N, KH, KW, 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tile16_kernel<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    t.ntiles = ((g.Wout + 63) / 64) * ((g.Hout + 7) / 8) * B;
    const dim3 grid((unsigned)((t.ntiles + 7) / 8 * 8));          // 1-D, XCD-aware ids (see conv_tile16_kernel)
    printf("grid %d (tiles %d), lds %zu, SB %d sps %d\n", grid.x, t.ntiles, lds, t.SB, sps);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((conv_tile16_kernel<4, 8>), grid, dim3(256), lds, 0, t);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    long long h[10]; hipMemcpy(h, dbg, 80, hipMemcpyDeviceToHost);
    printf("prologue %.1f us, whole workgroup %.1f us\n", h[8] / 100.0, h[9] / 100.0);
    printf("wall(100MHz) %lld -> %.1f us ; clock64 %lld -> %.3f GHz\n", h[6], h[6] / 100.0, h[7], h[7] / (h[6] * 10.0));
    const double fl = 2.0 * B * g.Hout * g.Wout * N * g.Ktot;
    printf("best %.3f ms  %.1f TFLOP/s (%s)\n", best, fl / best / 1e9, hipGetErrorString(hipGetLastError()));
    printf("clocks wg(1,1,0): prologue %lld | per chunk: issue %lld  mfma-loop %lld  barrier1 %lld  wstore %lld  barrier2 %lld  (18 chunks... totals /%d)\n", h[0], h[1], h[2], h[3], h[4], h[5], KH * sps / t.SB);
    return 0;
}
