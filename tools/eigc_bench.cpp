// standalone harness: cluster vs single-workgroup tridiagonalisation
#include "../nele_gan_amd/csrc/eigh.hip"
#include <vector>
#include <cmath>
int nele_set_error(int code, const char* fmt, ...) { printf("error %d: %s\n", code, fmt); return code; }
int main(int argc, char** argv) {
    const int n = argc > 2 ? atoi(argv[2]) : 420, B = argc > 1 ? atoi(argv[1]) : 32;
    std::vector<double> h((size_t)B * n * n);
    srand(1);
    for (int b = 0; b < B; ++b) {
        std::vector<double> G((size_t)n * 64);
        for (auto& v : G) v = (rand() / (double)RAND_MAX - 0.5);
        for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = (i == j) ? 0.01 * (1 + i % 7) : 0; for (int k = 0; k < 64; ++k) s += G[(size_t)i*64+k]*G[(size_t)j*64+k] * exp(-0.05*k); h[(size_t)b*n*n + (size_t)i*n + j] = h[(size_t)b*n*n + (size_t)j*n + i] = s; }
    }
    double *A, *A0; void* wsb;
    hipMalloc(&A, sizeof(double)*h.size()); hipMalloc(&A0, sizeof(double)*h.size());
    const long long wbytes = nele_eigh_workspace_bytes(B, n);
    hipMalloc(&wsb, wbytes);
    hipMemcpy(A0, h.data(), sizeof(double)*h.size(), hipMemcpyHostToDevice);
    EighWs ws; eigh_layout(B, n, &ws, (char*)wsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = (argc > 3 ? atoi(argv[3]) : 1); mode < (argc > 3 ? atoi(argv[3]) + 1 : 2); ++mode) {   // 1 = 8 workgroups per matrix, 4 = 4
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
            hipMemcpy(A, A0, sizeof(double)*h.size(), hipMemcpyDeviceToDevice);
            hipMemset(ws.xch, 0, sizeof(uint4) * (size_t)B * 4 * EG_MAXN);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 5) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(eigh_tridiag_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * ET_M * ET_M)); const int ss = n - ET_M - 2; hipLaunchKernelGGL(eigh_tridiag_cluster4_kernel, dim3(32 * ((B + 7) / 8)), dim3(512), 0, 0, A, n, 0, B, ws, ss); hipLaunchKernelGGL(eigh_tridiag_tail_kernel, dim3(B), dim3(512), sizeof(double) * ET_M * ET_M, 0, A, n, ws, ss + 1); }
            else if (mode == 4) { for (int b0 = 0; b0 < B; b0 += 64) { int Bc = B - b0 < 64 ? B - b0 : 64; hipLaunchKernelGGL(eigh_tridiag_cluster4_kernel, dim3(32 * ((Bc + 7) / 8)), dim3(512), 0, 0, A, n, b0, Bc, ws, -2); } }
            else for (int b0 = 0; b0 < B; b0 += 32) { int Bc = B - b0 < 32 ? B - b0 : 32; hipLaunchKernelGGL(eigh_tridiag_cluster_kernel, dim3(64 * ((Bc + 7) / 8)), dim3(512), 0, 0, A, n, b0, Bc, ws); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it > 0 && ms < best) best = ms;
        }
        double t[9]; hipMemcpy(t, ws.lamp, 72, hipMemcpyDeviceToHost);
        printf("house split: scalar math %.0f, owner stores %.0f, lds writes %.0f, sync %.0f\n", t[6]/n, t[7]/n, t[8]/n, t[2]/n);
        printf("clocks/step: pv-reduce %.0f ss-reduce %.0f house+publish %.0f pass %.0f acc-reduce+store %.0f poll %.0f\n", t[0]/n, t[1]/n, t[2]/n, t[3]/n, t[4]/n, t[5]/n);
        if (mode == 5) printf("tail clocks/step: pv %.0f ss %.0f house+publish %.0f pass %.0f barrier %.0f gather %.0f\n", t[0]/ET_M, t[1]/ET_M, t[2]/ET_M, t[3]/ET_M, t[4]/ET_M, t[5]/ET_M);
        printf("mode %d: B=%d n=%d best %.3f ms (%s)\n", mode, B, n, best, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
