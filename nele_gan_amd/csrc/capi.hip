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

// ------------------------------------------------------------------------------------------ per-kernel timing hook
// bench.py's roofline figure needs the duration of ONE kernel that is launched from inside a multi-kernel entry point, measured with
// HIP events on the stream the kernel runs on.  nele_profile_begin(tag) arms the hook; every launch site wrapped in NELE_PROF(tag, ...)
// whose tag matches records a start / stop event pair around its launch; nele_profile_collect synchronises those events and returns
// the elapsed milliseconds.  Not thread-safe by design (one Python thread drives the library); costs one string compare when idle.
#include <vector>
static char g_prof_tag[64] = "";
static std::vector<hipEvent_t> g_prof_ev;

bool nele_prof_match(const char* tag) { return g_prof_tag[0] && strcmp(g_prof_tag, tag) == 0; }
void nele_prof_mark(hipStream_t s) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    g_prof_ev.push_back(e);
}

extern "C" int nele_profile_begin(const char* tag) {
    for (hipEvent_t e : g_prof_ev) (void)hipEventDestroy(e);
    g_prof_ev.clear();
    g_prof_tag[0] = 0;
    if (tag) { strncpy(g_prof_tag, tag, sizeof(g_prof_tag) - 1); g_prof_tag[sizeof(g_prof_tag) - 1] = 0; }
    return NELE_OK;
}

extern "C" int nele_profile_collect(float* ms_out, int max_n) {
    const int n = (int)(g_prof_ev.size() / 2);
    int k = 0;
    for (; k < n && k < max_n; ++k) {
        if (hipEventSynchronize(g_prof_ev[2 * k + 1]) != hipSuccess) break;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof_ev[2 * k], g_prof_ev[2 * k + 1]) != hipSuccess) break;
        if (ms_out) ms_out[k] = ms;
    }
    (void)nele_profile_begin(nullptr);
    return k;
}
