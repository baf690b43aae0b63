// libnele_hip.so: version / error reporting for the C ABI declared in include/nele_hip.h.
#include "common.h"
#include <cstring>

static thread_local char g_err[512] = "";

int nele_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int nele_version(void) { return 100; }  // 0.1.0

extern "C" const char* nele_last_error_string(void) { return g_err; }

extern "C" int nele_device_info(int* cu_count, int* wave_size, char* arch, int arch_len) {
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess)
        return nele_set_error(NELE_ERR_HIP, "nele_device_info: no HIP device");
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return NELE_OK;
}
