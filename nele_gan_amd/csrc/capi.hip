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

#ifdef NELE_AB
#include <cstdlib>
int nele_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && e[0]) ? atoi(e) : dflt;
}
#endif
extern "C" int nele_build_has_ab_switches(void) {
#ifdef NELE_AB
    return 1;
#else
    return 0;
#endif
}

// ------------------------------------------------------------------------------------------ stream / hardware-queue probe
// One wave that does nothing for `ticks` of the 100 MHz wall clock.  The host side (GanTrainer._pipeline_queues) parks it on one stream
// and times a trivial kernel on another: streams that the runtime mapped onto the same hardware queue run strictly one after the other.
__global__ void nele_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int nele_stream_spin(double microseconds, void* stream) {
    if (!(microseconds >= 0.0) || microseconds > 1e5) return nele_set_error(NELE_ERR_INVALID_ARG, "nele_stream_spin: 0 .. 100 000 us");
    hipLaunchKernelGGL(nele_spin_kernel, dim3(1), dim3(64), 0, as_stream(stream), (long long)(microseconds * 100.0));
    return hipGetLastError() == hipSuccess ? NELE_OK : nele_set_error(NELE_ERR_HIP, "nele_stream_spin: launch failed");
}

// Measurement / test aid: `workgroups` workgroups of 1024 threads that each claim `lds_bytes` of LDS and idle for `microseconds` - a stand-in
// for kernels that stay resident on part of the chip while the path runs (the channels of a collective library, another process): with
// lds_bytes = 160 KB one workgroup takes a whole CU out of service for LDS-using kernels.
__global__ __launch_bounds__(1024) void nele_occupy_kernel(long long ticks) {
    extern __shared__ int occ_lds[];
    if (threadIdx.x == 0) occ_lds[0] = 1;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int nele_stream_occupy(int workgroups, int lds_bytes, double microseconds, void* stream) {
    if (workgroups < 1 || workgroups > 4096 || lds_bytes < 4 || lds_bytes > 160 * 1024 || !(microseconds >= 0.0) || microseconds > 1e6)
        return nele_set_error(NELE_ERR_INVALID_ARG, "nele_stream_occupy: 1 .. 4096 workgroups, 4 .. 163 840 bytes of LDS, 0 .. 1 000 000 us");
    static unsigned long long attr = 0;
    if (nele_first_use_on_device(&attr))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nele_occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(nele_occupy_kernel, dim3(workgroups), dim3(1024), (size_t)lds_bytes, as_stream(stream), (long long)(microseconds * 100.0));
    return hipGetLastError() == hipSuccess ? NELE_OK : nele_set_error(NELE_ERR_HIP, "nele_stream_occupy: launch failed");
}

// ---- host-side wav decoding (no GPU work): the reference reads its wav files through libsndfile (librosa.load / sf.read, dataloader.py:34-37);
// the host mirror's loader threads call this through ctypes, which releases the interpreter lock for the whole call - file read, RIFF walk
// and the int16 -> float32 conversion run in parallel on the loader threads instead of taking turns in the interpreter.
// Mono PCM_16 only (what the reference writes and reads); anything else returns NELE_ERR_UNSUPPORTED and the caller uses its general reader.
// out [cap] float32 host memory (e.g. a row of a pinned staging buffer): samples / 32768, zeros behind them; *n_out = samples written.
#include <cstdio>
#include <cstdlib>
extern "C" int nele_wav_decode_pcm16(const char* path, float* out_host, long long cap, long long* n_out, int* sample_rate_out) {
    if (!path || !out_host || cap < 0 || !n_out) return nele_set_error(NELE_ERR_INVALID_ARG, "nele_wav_decode_pcm16: bad arguments");
    FILE* f = fopen(path, "rb");
    if (!f) return nele_set_error(NELE_ERR_INVALID_ARG, "nele_wav_decode_pcm16: cannot open %s", path);
    unsigned char hd[12];
    int st = NELE_ERR_UNSUPPORTED;
    long long n = 0;
    int fmt_ok = 0, sr = 0;
    if (fread(hd, 1, 12, f) == 12 && !memcmp(hd, "RIFF", 4) && !memcmp(hd + 8, "WAVE", 4)) {
        unsigned char ch[8];
        while (fread(ch, 1, 8, f) == 8) {
            const unsigned size = ch[4] | (ch[5] << 8) | (ch[6] << 16) | ((unsigned)ch[7] << 24);
            if (!memcmp(ch, "fmt ", 4) && size >= 16) {
                unsigned char fm[16];
                if (fread(fm, 1, 16, f) != 16) break;
                const int tag = fm[0] | (fm[1] << 8), nch = fm[2] | (fm[3] << 8), bits = fm[14] | (fm[15] << 8);
                sr = fm[4] | (fm[5] << 8) | (fm[6] << 16) | (fm[7] << 24);
                fmt_ok = (tag == 1 && nch == 1 && bits == 16);
                if (fseek(f, (long)(size - 16 + (size & 1)), SEEK_CUR)) break;
            } else if (!memcmp(ch, "data", 4)) {
                if (!fmt_ok) break;
                long long want = (long long)size / 2;
                if (want > cap) want = cap;
                // int16 samples are read into the tail of the output row and converted front to back (4 bytes written per 2 read)
                short* raw = reinterpret_cast<short*>(out_host) + want;      // second half of the first `want` floats
                const long long got = (long long)fread(raw, 2, (size_t)want, f);
                for (long long i = 0; i < got; ++i) out_host[i] = (float)raw[i] * (1.0f / 32768.0f);
                n = got;
                st = NELE_OK;
                break;
            } else if (fseek(f, (long)(size + (size & 1)), SEEK_CUR)) break;
        }
    }
    fclose(f);
    if (st != NELE_OK) return nele_set_error(NELE_ERR_UNSUPPORTED, "nele_wav_decode_pcm16: %s is not a mono PCM_16 RIFF file", path);
    for (long long i = n; i < cap; ++i) out_host[i] = 0.f;
    *n_out = n;
    if (sample_rate_out) *sample_rate_out = sr;
    return NELE_OK;
}

// ... and the writer (train_nele.py:198,313, inference.py:115: sf.write(path, wav, 16000, 'PCM_16') -> libsndfile): float32 HOST samples ->
// mono PCM_16 RIFF file.  quantised = 0: libsndfile's rule, lrintf(x * 32767) saturated (round half to even); quantised != 0: the samples
// already went through the device-side PCM_16 emulation (values k / 32768, nele_wav_post): k is recovered exactly, nothing is rounded twice.
extern "C" int nele_wav_write_pcm16(const char* path, const float* wav_host, long long n, int sample_rate, int quantised) {
    if (!path || (!wav_host && n > 0) || n < 0 || n > 0x7fffff00LL / 2 || sample_rate <= 0)
        return nele_set_error(NELE_ERR_INVALID_ARG, "nele_wav_write_pcm16: bad arguments");
    const unsigned bytes = (unsigned)(2 * n);
    unsigned char* buf = (unsigned char*)malloc(44 + (size_t)bytes);
    if (!buf) return nele_set_error(NELE_ERR_HIP, "nele_wav_write_pcm16: out of host memory");
    auto u32 = [&](int off, unsigned v) { buf[off] = v & 255; buf[off + 1] = (v >> 8) & 255; buf[off + 2] = (v >> 16) & 255; buf[off + 3] = (v >> 24) & 255; };
    auto u16 = [&](int off, unsigned v) { buf[off] = v & 255; buf[off + 1] = (v >> 8) & 255; };
    memcpy(buf, "RIFF", 4); u32(4, 36 + bytes); memcpy(buf + 8, "WAVEfmt ", 8); u32(16, 16); u16(20, 1); u16(22, 1);
    u32(24, (unsigned)sample_rate); u32(28, (unsigned)sample_rate * 2); u16(32, 2); u16(34, 16); memcpy(buf + 36, "data", 4); u32(40, bytes);
    short* out = reinterpret_cast<short*>(buf + 44);
    const float scale = quantised ? 32768.0f : 32767.0f;
    for (long long i = 0; i < n; ++i) {
        float q = nearbyintf(wav_host[i] * scale);             // round half to even (the default rounding mode), as lrintf / numpy.rint
        q = q < -32768.0f ? -32768.0f : (q > 32767.0f ? 32767.0f : q);
        out[i] = (short)q;
    }
    FILE* f = fopen(path, "wb");
    int st = NELE_OK;
    if (!f || fwrite(buf, 1, 44 + (size_t)bytes, f) != 44 + (size_t)bytes) st = nele_set_error(NELE_ERR_INVALID_ARG, "nele_wav_write_pcm16: cannot write %s", path);
    if (f) fclose(f);
    free(buf);
    return st;
}

bool nele_first_use_on_device(unsigned long long* mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;       // unknown device: set the attribute again (idempotent)
    const unsigned long long bit = 1ull << dev;
    if (*mask & bit) return false;
    *mask |= bit;
    return true;
}

// ------------------------------------------------------------------------------------------ per-kernel timing hook
// bench.py's roofline figure needs the duration of ONE kernel that is launched from inside a multi-kernel entry point, measured with
// HIP events on the stream the kernel runs on.  nele_profile_begin(tag) arms the hook; every launch site wrapped in NELE_PROF(tag, ...)
// whose tag matches records a start / stop event pair around its launch; nele_profile_collect synchronises those events and returns
// the elapsed milliseconds.  Not thread-safe by design (one Python thread drives the library); costs one string compare when idle.
#include <string>
#include <vector>
static std::vector<std::string> g_prof_tags;             // armed tags (nele_profile_begin("a,b,c") arms several)
static std::vector<std::vector<hipEvent_t>> g_prof_ev;   // per tag: start / stop pairs

static int prof_index(const char* tag) {
    for (size_t k = 0; k < g_prof_tags.size(); ++k)
        if (g_prof_tags[k] == tag) return (int)k;
    return -1;
}
static int g_prof_cur = -1;                               // tag of the launch site between its two marks
bool nele_prof_armed() { return !g_prof_tags.empty(); }
bool nele_prof_match(const char* tag) {
    if (g_prof_tags.empty()) return false;
    g_prof_cur = prof_index(tag);
    return g_prof_cur >= 0;
}
void nele_prof_mark(hipStream_t s) {
    hipEvent_t e;
    if (g_prof_cur < 0 || hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    g_prof_ev[g_prof_cur].push_back(e);
}

extern "C" int nele_profile_begin(const char* tags) {
    for (auto& v : g_prof_ev)
        for (hipEvent_t e : v) (void)hipEventDestroy(e);
    g_prof_ev.clear();
    g_prof_tags.clear();
    g_prof_cur = -1;
    if (tags) {
        std::string all(tags), cur;
        for (size_t k = 0; k <= all.size(); ++k) {
            if (k == all.size() || all[k] == ',') { if (!cur.empty()) g_prof_tags.push_back(cur); cur.clear(); }
            else cur.push_back(all[k]);
        }
        g_prof_ev.resize(g_prof_tags.size());
    }
    return NELE_OK;
}

// durations (ms) of the launches tagged `tag` since nele_profile_begin (synchronises on them); the hook stays armed
extern "C" int nele_profile_collect_tag(const char* tag, float* ms_out, int max_n) {
    const int ti = tag ? prof_index(tag) : (g_prof_tags.empty() ? -1 : 0);
    if (ti < 0) return 0;
    const std::vector<hipEvent_t>& ev = g_prof_ev[ti];
    const int n = (int)(ev.size() / 2);
    int k = 0;
    for (; k < n && k < max_n; ++k) {
        if (hipEventSynchronize(ev[2 * k + 1]) != hipSuccess) break;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]) != hipSuccess) break;
        if (ms_out) ms_out[k] = ms;
    }
    return k;
}

// the first armed tag's durations, then disarm (the round-1 form of the hook)
extern "C" int nele_profile_collect(float* ms_out, int max_n) {
    const int k = nele_profile_collect_tag(nullptr, ms_out, max_n);
    (void)nele_profile_begin(nullptr);
    return k;
}
