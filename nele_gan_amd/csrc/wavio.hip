// Batched PCM_16 file hand-off (the wav files either side of the path).  The reference reads every utterance through
// libsndfile into float32 on the host (dataloader.py:34-37, inference.py:99-101) and writes float32 back through it (inference.py:115);
// here the host only moves BYTES: a batch of mono PCM_16 files is read by the library's own threads straight into the rows of a pinned
// int16 staging buffer (one foreign call per batch - no interpreter lock, no per-file host-language work), uploaded as int16 (half the
// PCIe bytes of float32) and converted on the device (s / 32768, exact); enhanced batches come back as int16 and are written the same way.
#include "common.h"
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct WavHead { long long samples; int sr; };

// walks the RIFF chunks of an open file up to the start of the sample data; false = not mono PCM_16
bool wav_seek_data(FILE* f, WavHead* h) {
    unsigned char hd[12];
    if (fread(hd, 1, 12, f) != 12 || memcmp(hd, "RIFF", 4) || memcmp(hd + 8, "WAVE", 4)) return false;
    unsigned char ch[8];
    bool fmt_ok = false;
    while (fread(ch, 1, 8, f) == 8) {
        const unsigned size = ch[4] | (ch[5] << 8) | (ch[6] << 16) | ((unsigned)ch[7] << 24);
        if (!memcmp(ch, "fmt ", 4) && size >= 16) {
            unsigned char fm[16];
            if (fread(fm, 1, 16, f) != 16) return false;
            const int tag = fm[0] | (fm[1] << 8), nch = fm[2] | (fm[3] << 8), bits = fm[14] | (fm[15] << 8);
            h->sr = fm[4] | (fm[5] << 8) | (fm[6] << 16) | (fm[7] << 24);
            fmt_ok = (tag == 1 && nch == 1 && bits == 16);
            if (fseek(f, (long)(size - 16 + (size & 1)), SEEK_CUR)) return false;
        } else if (!memcmp(ch, "data", 4)) {
            h->samples = (long long)size / 2;
            return fmt_ok;
        } else if (fseek(f, (long)(size + (size & 1)), SEEK_CUR)) return false;
    }
    return false;
}

template <class F>
void for_each_file(int n, int threads, F&& body) {
    if (threads > n) threads = n;
    if (threads <= 1) {
        for (int i = 0; i < n; ++i) body(i);
        return;
    }
    std::atomic<int> next(0);
    auto loop = [&]() {
        for (;;) {
            const int i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) return;
            body(i);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(threads - 1);
    for (int t = 1; t < threads; ++t) {
        try { pool.emplace_back(loop); } catch (...) { break; }      // no more threads to be had: the ones that started share the files
    }
    loop();
    for (auto& t : pool) t.join();
}

}  // namespace

extern "C" int nele_wav_read_pcm16_batch(const char* const* paths, int n, short* out_host, long long row_stride, long long cap, int* n_out,
                                         int* sample_rate_out, int threads) {
    if (!paths || n < 0 || (n > 0 && (!out_host || !n_out)) || cap < 0 || row_stride < cap || threads < 1 || threads > 256)
        return nele_set_error(NELE_ERR_INVALID_ARG, "nele_wav_read_pcm16_batch: bad arguments");
    for_each_file(n, threads, [&](int i) {
        short* row = out_host + (size_t)i * row_stride;
        long long got = -2;                                  // -2: cannot open, -1: not mono PCM_16
        int sr = 0;
        FILE* f = paths[i] ? fopen(paths[i], "rb") : nullptr;
        if (f) {
            WavHead h{0, 0};
            if (wav_seek_data(f, &h)) {
                const long long want = h.samples < cap ? h.samples : cap;
                got = (long long)fread(row, 2, (size_t)want, f);
                sr = h.sr;
            } else got = -1;
            fclose(f);
        }
        const long long z0 = got > 0 ? got : 0;
        memset(row + z0, 0, sizeof(short) * (size_t)(cap - z0));
        n_out[i] = (int)got;
        if (sample_rate_out) sample_rate_out[i] = sr;
    });
    return NELE_OK;
}

// Sample counts of a batch of wav files from their headers (one foreign call instead of n stat() calls in the host language: the padded
// length of a batch must be known before its staging rows are taken).  n_out[i] = samples of a mono PCM_16 file, -1 = another wav flavour
// (the caller's general reader takes that file), -2 = cannot open.  No GPU work.
extern "C" int nele_wav_probe_pcm16_batch(const char* const* paths, int n, int* n_out, int threads) {
    if (!paths || n < 0 || (n > 0 && !n_out) || threads < 1 || threads > 256) return nele_set_error(NELE_ERR_INVALID_ARG, "nele_wav_probe_pcm16_batch: bad arguments");
    for_each_file(n, threads, [&](int i) {
        long long got = -2;
        FILE* f = paths[i] ? fopen(paths[i], "rb") : nullptr;
        if (f) {
            WavHead h{0, 0};
            got = wav_seek_data(f, &h) ? h.samples : -1;
            fclose(f);
        }
        n_out[i] = got > 0x7fffffffLL ? 0x7fffffff : (int)got;
    });
    return NELE_OK;
}

extern "C" int nele_wav_write_pcm16_batch(const char* const* paths, int n, const short* in_host, long long row_stride, const int* n_samples,
                                          int sample_rate, int threads) {
    if (!paths || n < 0 || (n > 0 && (!in_host || !n_samples)) || row_stride < 0 || sample_rate <= 0 || threads < 1 || threads > 256)
        return nele_set_error(NELE_ERR_INVALID_ARG, "nele_wav_write_pcm16_batch: bad arguments");
    for (int i = 0; i < n; ++i)
        if (!paths[i] || n_samples[i] < 0 || n_samples[i] > row_stride) return nele_set_error(NELE_ERR_INVALID_ARG, "nele_wav_write_pcm16_batch: bad entry %d", i);
    std::atomic<int> failed(-1);
    for_each_file(n, threads, [&](int i) {
        const unsigned bytes = 2u * (unsigned)n_samples[i];
        unsigned char hd[44];
        auto u32 = [&](int off, unsigned v) { hd[off] = v & 255; hd[off + 1] = (v >> 8) & 255; hd[off + 2] = (v >> 16) & 255; hd[off + 3] = (v >> 24) & 255; };
        auto u16 = [&](int off, unsigned v) { hd[off] = v & 255; hd[off + 1] = (v >> 8) & 255; };
        memcpy(hd, "RIFF", 4); u32(4, 36 + bytes); memcpy(hd + 8, "WAVEfmt ", 8); u32(16, 16); u16(20, 1); u16(22, 1);
        u32(24, (unsigned)sample_rate); u32(28, (unsigned)sample_rate * 2); u16(32, 2); u16(34, 16); memcpy(hd + 36, "data", 4); u32(40, bytes);
        FILE* f = fopen(paths[i], "wb");
        bool ok = f != nullptr;
        if (ok) ok = fwrite(hd, 1, 44, f) == 44 && fwrite(in_host + (size_t)i * row_stride, 1, bytes, f) == bytes;
        if (f) ok = (fclose(f) == 0) && ok;
        if (!ok) { int exp = -1; failed.compare_exchange_strong(exp, i); }
    });
    const int bad = failed.load();
    if (bad >= 0) return nele_set_error(NELE_ERR_INVALID_ARG, "nele_wav_write_pcm16_batch: cannot write %s", paths[bad]);
    return NELE_OK;
}

// ------------------------------------------------------------------------------------------ device side of the hand-off
// int16 rows -> float32 rows: out[b][i] = i < lengths[b] ? in[b][i] / 32768 : 0 for i < L (what sf.read / librosa.load return for PCM_16,
// zeros behind an utterance cut to its partner's length: dataloader.py:38-40 truncates clean and noise to the shorter of the two)
__global__ __launch_bounds__(256) void pcm16_to_float_kernel(const short* __restrict__ in, long long in_stride, const int* __restrict__ lengths,
                                                             long long L, float* __restrict__ out, long long out_stride, int vec) {
    const int b = blockIdx.y;
    const long long len = lengths ? (long long)lengths[b] : L;
    const short* src = in + (size_t)b * in_stride;
    float* dst = out + (size_t)b * out_stride;
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= L) return;
    if (vec && i0 + 8 <= L) {
        const int4 q = *reinterpret_cast<const int4*>(src + i0);
        const int w[4] = {q.x, q.y, q.z, q.w};
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = (float)(short)(w[k] & 0xffff) * (1.0f / 32768.0f);
            v[2 * k + 1] = (float)(short)(w[k] >> 16) * (1.0f / 32768.0f);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + k >= len) v[k] = 0.f;
        *reinterpret_cast<float4*>(dst + i0) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + i0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        for (long long i = i0; i < i0 + 8 && i < L; ++i) dst[i] = i < len ? (float)src[i] * (1.0f / 32768.0f) : 0.f;
    }
}

extern "C" int nele_pcm16_to_float(const short* in, long long in_stride, const int* lengths, int B, long long L, float* out, long long out_stride,
                                   void* stream) {
    NELE_CHECK_ARG(in && out && B > 0 && L > 0 && in_stride >= L && out_stride >= L, "nele_pcm16_to_float: bad arguments");
    const int vec = (in_stride % 8 == 0) && (out_stride % 4 == 0) && ((size_t)in % 16 == 0) && ((size_t)out % 16 == 0);
    hipLaunchKernelGGL(pcm16_to_float_kernel, dim3((unsigned)((L + 2047) / 2048), B), dim3(256), 0, as_stream(stream), in, in_stride, lengths, L, out,
                       out_stride, vec);
    NELE_CHECK_LAUNCH("nele_pcm16_to_float");
    return NELE_OK;
}

// float32 rows -> int16 rows, the sample values sf.write(..., 'PCM_16') stores: quantised != 0: the input went through the device-side
// PCM_16 emulation (values k / 32768: nele_wav_post, nele_wav_quant) and k is recovered exactly; 0: libsndfile's rule lrintf(x * 32767) saturated
__global__ __launch_bounds__(256) void float_to_pcm16_kernel(const float* __restrict__ in, long long in_stride, long long L, short* __restrict__ out,
                                                             long long out_stride, float scale) {
    const int b = blockIdx.y;
    const float* src = in + (size_t)b * in_stride;
    short* dst = out + (size_t)b * out_stride;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < L; i += (long long)gridDim.x * 256) {
        float q = rintf(src[i] * scale);
        q = fminf(fmaxf(q, -32768.f), 32767.f);
        dst[i] = (short)q;
    }
}

extern "C" int nele_float_to_pcm16(const float* in, long long in_stride, int B, long long L, short* out, long long out_stride, int quantised,
                                   void* stream) {
    NELE_CHECK_ARG(in && out && B > 0 && L > 0 && in_stride >= L && out_stride >= L, "nele_float_to_pcm16: bad arguments");
    const long long gx = (L + 255) / 256;
    hipLaunchKernelGGL(float_to_pcm16_kernel, dim3((unsigned)(gx < 512 ? gx : 512), B), dim3(256), 0, as_stream(stream), in, in_stride, L, out, out_stride,
                       quantised ? 32768.0f : 32767.0f);
    NELE_CHECK_LAUNCH("nele_float_to_pcm16");
    return NELE_OK;
}
