// Plan builders: the generator's and the discriminators' forward / backward passes as job tables made INSIDE the library
// (nele_gen_fwd / nele_gen_bwd / nele_disc_fwd / nele_disc_bwd are callable from any host language).
//
// Until round 5 a plan could only be produced by the Python mirror's recorder hooking its own per-layer loop (nele_gan_amd/model.py):
// which kernels, layouts, workspaces and stream hand-overs make up a pass of model.py:83-98 / 118-132 and their autograd was host-language
// knowledge.  nele_gen_plan_build / nele_disc_plan_build state it here: given the batch shape, the operand precision and the model's flat
// parameter / gradient buffers they lay every activation, weight layout and temporary out in ONE caller-provided workspace and return the
// forward and backward job tables over this library's own per-layer entry points - the same kernels in the same order on the same
// streams as the mirror's loop (tests/test_netplan_gpu.py: bit-identical to the module path in both precisions).
//
// Parameter order of the flat buffers = torch's nn.Module.parameters() order of the reference modules (model.py:43-82, 101-116):
//   G: for each of the 6 Conv1d blocks {conv.weight [Cout][Cin][K], conv.bias [Cout], cLN.gain0 [Cout], cLN.bias0 [Cout]}, fc1.weight, fc1.bias, fc2.weight, fc2.bias
//   D: for each of the 5 Conv2d layers and then fc1, fc2, fc3 {bias, weight_orig} (torch.nn.utils.spectral_norm re-registers the weight behind the bias)
// (nele_gen_param_layout / nele_disc_param_layout return the offsets; tests/test_model_cpu.py compares them with named_parameters()).
#include "common.h"
#include "conv_common.h"
#include "../../include/nele_hip.h"
#include <cstring>
#include <initializer_list>
#include <memory>
#include <vector>

int nele_plan_adopt(void* plan, std::vector<std::unique_ptr<char[]>>&& blobs, std::vector<hipEvent_t>&& events);   // plan.hip

namespace {

constexpr double SLOPE = 0.3;                       // nn.LeakyReLU(0.3), model.py:79,112

struct Geom { int a[15]; int KH, KW, Hout, Wout, Ktot; };
Geom mk_geom(int H, int W, int C, int Hout, int Wout, int KH, int KW, int OH, int OW, int OC, int ih0 = 0, int iw0 = 0, int oh0 = 0, int ow0 = 0) {
    Geom g;
    const int v[15] = {H, W, C, ih0, iw0, Hout, Wout, KW * C, W * C, KH * KW * C, OH, OW, OC, oh0, ow0};
    memcpy(g.a, v, sizeof(v));
    g.KH = KH; g.KW = KW; g.Hout = Hout; g.Wout = Wout; g.Ktot = KH * KW * C;
    return g;
}

struct Buf { void* p = nullptr; long long elems = 0; bool b16 = false; };

// bump allocator over the caller's workspace (base == nullptr: sizes only)
struct Ws {
    char* base; size_t off = 0;
    explicit Ws(void* b) : base(reinterpret_cast<char*>(b)) {}
    Buf take(long long elems, int elsize, bool b16 = false) {
        Buf r; r.elems = elems; r.b16 = b16;
        r.p = base ? base + off : nullptr;
        off += ((size_t)elems * elsize + 255) & ~(size_t)255;
        return r;
    }
    Buf f32(long long n) { return take(n, 4); }
    Buf f64(long long n) { return take(n, 8); }
    Buf bf16(long long n) { return take(n, 2, true); }
    Buf bytes(long long n) { return take(n, 1); }
};

struct Arg { int slot; long long i; double f; };
Arg P(const void* p) { return {-1, (long long)(uintptr_t)p, 0.0}; }
Arg P(const Buf& b) { return P(b.p); }
Arg I(long long v) { return {-1, v, 0.0}; }
Arg F(double v) { return {-1, 0, v}; }
Arg S(int slot, long long off = 0) { return {slot, off, 0.0}; }

struct Builder {
    std::vector<nele_plan_job> jobs;
    std::vector<std::unique_ptr<char[]>> blobs;
    std::vector<hipEvent_t> events;
    int nstreams = 1;
    int err = NELE_OK;

    void add(const char* name, int stream, std::initializer_list<Arg> args) {
        if (err) return;
        const int op = nele_plan_op_id(name);
        if (op < 0 || nele_plan_op_nargs(op) != (int)args.size() + 1 || (int)args.size() + 1 > NELE_PLAN_MAXARGS) {
            err = nele_set_error(NELE_ERR_INVALID_ARG, "plan builder: %s with %d arguments is not a plan operation", name, (int)args.size() + 1);
            return;
        }
        nele_plan_job j;
        memset(&j, 0, sizeof(j));
        j.op = op; j.nargs = (int)args.size() + 1; j.stream = stream;
        int i = 0;
        for (const Arg& a : args) { j.slot[i] = a.slot; j.ival[i] = a.i; j.fval[i] = a.f; ++i; }
        j.slot[i] = -1;
        if (stream + 1 > nstreams) nstreams = stream + 1;
        jobs.push_back(j);
    }
    // a host array that must live as long as the plan
    template <class T> const T* host(const T* src, size_t n) {
        std::unique_ptr<char[]> b(new char[sizeof(T) * (n ? n : 1)]);
        memcpy(b.get(), src, sizeof(T) * n);
        const T* r = reinterpret_cast<const T*>(b.get());
        blobs.push_back(std::move(b));
        return r;
    }
    const int* geom(const Geom& g) { return host(g.a, 15); }
    void hand_over(int src, int dst) {               // stream dst waits for everything enqueued on stream src so far
        if (err) return;
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { err = nele_set_error(NELE_ERR_HIP, "plan builder: hipEventCreate failed"); return; }
        events.push_back(e);
        add("nele_event_record", src, {P((void*)e)});
        add("nele_stream_wait_event", dst, {P((void*)e)});
    }
    int finish(int nslots, void** out) {
        if (err) { for (hipEvent_t e : events) (void)hipEventDestroy(e); events.clear(); return err; }
        void* h = nullptr;
        int st = nele_plan_create(jobs.data(), (int)jobs.size(), nslots, nstreams, &h);
        if (st) { for (hipEvent_t e : events) (void)hipEventDestroy(e); events.clear(); return st; }
        st = nele_plan_adopt(h, std::move(blobs), std::move(events));
        if (st) { (void)nele_plan_destroy(h); return st; }
        *out = h;
        return NELE_OK;
    }
};

long long M_of(int B, const Geom& g) { return (long long)B * g.Hout * g.Wout; }

// ops.conv_wgrad
void conv_wgrad(Builder& b, int st, const Buf& A, const Buf& dOut, const Buf& ws, int B, int N, const Geom& g, int Cvalid, const void* dW, const void* db,
                int accumulate, bool bf16) {
    const char* fn = bf16 ? "nele_conv_wgrad_bf16" : "nele_conv_wgrad";
    if (dOut.b16) fn = A.b16 ? "nele_conv_wgrad_bf16_a16d16" : "nele_conv_wgrad_bf16_d16";
    b.add(fn, st, {P(A), P(dOut), P(ws), I(ws.elems), I(M_of(B, g)), I(N), P(b.geom(g)), I(g.KH), I(g.KW), I(Cvalid), P(dW), P(db), I(accumulate)});
}

// ================================================================== generator (model.py:43-98)
struct GL { int cin, cout, k; };
const GL G_LAYERS[6] = {{128, 256, 5}, {256, 256, 7}, {256, 256, 7}, {256, 256, 7}, {256, 256, 7}, {256, 64, 5}};

struct GParams { long long w[6], b[6], gain[6], beta[6], fcw[2], fcb[2], total; };
GParams g_params() {
    GParams p; long long o = 0;
    for (int l = 0; l < 6; ++l) {
        const GL& L = G_LAYERS[l];
        p.w[l] = o; o += (long long)L.cout * L.cin * L.k;
        p.b[l] = o; o += L.cout;
        p.gain[l] = o; o += L.cout;
        p.beta[l] = o; o += L.cout;
    }
    for (int q = 0; q < 2; ++q) { p.fcw[q] = o; o += 64 * 64; p.fcb[q] = o; o += 64; }
    p.total = o;
    return p;
}

struct GWs {
    Buf wf[8], wb[8], wf16f[8], wf16b[8], wgl0[6], wgl1[6];
    Buf a5, h1, inp16[6], carry, Y[6], mean[6], rstd[6], cln_scratch, inp[6], dY[6], dY16[6], dA[6], da5, do2, dpre1, gpart, bpart, ws;
    Geom gf[6], gb[6], gw[6], gfc, gwfc;
    int nchunks = 0;
    bool fused_ok = false, fused = false;
    size_t bytes = 0;
};

void g_layout(int B, int T, bool bf16, bool need_bwd, void* base, GWs& w) {
    Ws a(base);
    const int dims[8][4] = {{256, 5 * 128, 128, 5 * 256}, {256, 7 * 256, 256, 7 * 256}, {256, 7 * 256, 256, 7 * 256}, {256, 7 * 256, 256, 7 * 256},
                            {256, 7 * 256, 256, 7 * 256}, {64, 5 * 256, 256, 5 * 64}, {64, 64, 64, 64}, {64, 64, 64, 64}};
    for (int q = 0; q < 8; ++q) { w.wf[q] = a.f32((long long)dims[q][0] * dims[q][1]); w.wb[q] = a.f32((long long)dims[q][2] * dims[q][3]); }
    w.fused_ok = true;
    for (int l = 0; l < 6; ++l) w.fused_ok = w.fused_ok && nele_glayer16_supported(G_LAYERS[l].cin, G_LAYERS[l].cout, G_LAYERS[l].k);
    w.fused = bf16 && w.fused_ok && (!need_bwd || T >= 32);
    if (bf16) {
        for (int q = 0; q < 8; ++q) {
            w.wf16f[q] = a.bf16(nele_weight_frag16_elems(dims[q][0], dims[q][1], 1));
            w.wf16b[q] = a.bf16(nele_weight_frag16_elems(dims[q][2], dims[q][3], 1));
        }
        if (w.fused_ok)
            for (int l = 0; l < 6; ++l) {
                w.wgl0[l] = a.bf16(nele_glayer16_wfrag_elems(G_LAYERS[l].cin, G_LAYERS[l].cout, G_LAYERS[l].k));
                if (l > 0) w.wgl1[l] = a.bf16(nele_glayer16_wfrag_elems(G_LAYERS[l].cout, G_LAYERS[l].cin, G_LAYERS[l].k));
            }
    }
    const long long BT = (long long)B * T;
    w.a5 = a.f32(BT * 64); w.h1 = a.f32(BT * 64);
    w.gfc = mk_geom(1, T, 64, 1, T, 1, 1, 1, T, 64);
    w.nchunks = nele_cln_chunks(T);
    const bool stats = !w.fused || need_bwd;
    if (w.fused) {
        for (int l = 0; l < 6; ++l) w.inp16[l] = a.bf16((long long)B * (T + G_LAYERS[l].k - 1) * G_LAYERS[l].cin);
        long long cb = nele_glayer16_carry_bytes(B, T);
        w.carry = a.bytes(cb < 32 ? 32 : cb);
    }
    if (stats) {
        for (int l = 0; l < 6; ++l) { w.Y[l] = a.f32(BT * G_LAYERS[l].cout); w.mean[l] = a.f32(BT); w.rstd[l] = a.f32(BT); }
        w.cln_scratch = a.f64(BT * 2);
    }
    if (!w.fused)
        for (int l = 0; l < 6; ++l) {
            const GL& L = G_LAYERS[l];
            w.inp[l] = a.f32((long long)B * (T + L.k - 1) * L.cin);
            w.gf[l] = mk_geom(1, T + L.k - 1, L.cin, 1, T, 1, L.k, 1, T, L.cout);
        }
    if (need_bwd) {
        for (int l = 0; l < 6; ++l) {
            const GL& L = G_LAYERS[l];
            if (w.fused) w.dY16[l] = a.bf16((long long)B * (T + L.k - 1) * L.cout);
            else w.dY[l] = a.f32((long long)B * (T + L.k - 1) * L.cout);
            w.dA[l] = a.f32(BT * L.cin);
            w.gb[l] = mk_geom(1, T + L.k - 1, L.cout, 1, T, 1, L.k, 1, T, L.cin);
            w.gw[l] = mk_geom(1, T + L.k - 1, L.cin, 1, T, 1, L.k, 1, T + L.k - 1, L.cout);
        }
        w.da5 = a.f32(BT * 64); w.do2 = a.f32(BT * 64); w.dpre1 = a.f32(BT * 64);
        w.gpart = a.f32((long long)B * w.nchunks * 256); w.bpart = a.f32((long long)B * w.nchunks * 256);
        w.gwfc = mk_geom(1, T, 64, 1, T, 1, 1, 1, T, 64);
        long long nws = nele_conv_wgrad_workspace_floats((int)M_of(B, w.gwfc), 64, w.gwfc.Ktot, nullptr);
        for (int l = 0; l < 6; ++l) {
            const long long v = nele_conv_wgrad_workspace_floats((int)M_of(B, w.gw[l]), G_LAYERS[l].cout, w.gw[l].Ktot, nullptr);
            if (v > nws) nws = v;
        }
        w.ws = a.f32(nws);
    }
    w.bytes = a.off;
}

// Generator_Conv1D_cLN._gemm
void g_gemm(Builder& b, const GWs& w, bool bf16, const Buf& A, int q, bool back, const void* bias, const void* aux, Arg out, int B, int N, int epi, const Geom& g) {
    const long long M = M_of(B, g);
    if (bf16 && nele_conv_span_bf16_supported((int)M, N, g.a, g.KH, g.KW)) {
        b.add("nele_conv_span_bf16", 0, {P(A), P(back ? w.wf16b[q] : w.wf16f[q]), P(bias), P(aux), out, I(M), I(N), I(epi), F(SLOPE), P(b.geom(g)), I(g.KH), I(g.KW), I(A.elems)});
        return;
    }
    b.add((bf16 && N > 48) ? "nele_conv_gemm_bf16" : "nele_conv_gemm", 0, {P(A), P(back ? w.wb[q] : w.wf[q]), P(bias), P(aux), out, I(M), I(N), I(epi), F(SLOPE), P(b.geom(g))});
}

void g_prep_weights(Builder& b, const GWs& w, const float* par, const GParams& pp, bool bf16) {
    const void* pj[32]; int dj[40];
    for (int l = 0; l < 6; ++l) {
        pj[4 * l] = par + pp.w[l]; pj[4 * l + 1] = nullptr; pj[4 * l + 2] = w.wf[l].p; pj[4 * l + 3] = w.wb[l].p;
        const int d[5] = {G_LAYERS[l].cout, G_LAYERS[l].cin, G_LAYERS[l].cin, 1, G_LAYERS[l].k};
        memcpy(dj + 5 * l, d, sizeof(d));
    }
    for (int q = 6; q < 8; ++q) {
        pj[4 * q] = par + pp.fcw[q - 6]; pj[4 * q + 1] = nullptr; pj[4 * q + 2] = w.wf[q].p; pj[4 * q + 3] = w.wb[q].p;
        const int d[5] = {64, 64, 64, 1, 1};
        memcpy(dj + 5 * q, d, sizeof(d));
    }
    b.add("nele_weight_prep_batch", 0, {P(b.host(pj, 32)), P(b.host(dj, 40)), I(8)});
    if (!bf16) return;
    const int dims[8][4] = {{256, 5 * 128, 128, 5 * 256}, {256, 7 * 256, 256, 7 * 256}, {256, 7 * 256, 256, 7 * 256}, {256, 7 * 256, 256, 7 * 256},
                            {256, 7 * 256, 256, 7 * 256}, {64, 5 * 256, 256, 5 * 64}, {64, 64, 64, 64}, {64, 64, 64, 64}};
    const void* fj[32]; int ej[64];
    for (int q = 0; q < 8; ++q) {
        fj[2 * q] = w.wf[q].p; fj[2 * q + 1] = w.wf16f[q].p;
        const int d[4] = {dims[q][0], dims[q][1], dims[q][1], 1};
        memcpy(ej + 4 * q, d, sizeof(d));
    }
    for (int q = 0; q < 8; ++q) {
        fj[16 + 2 * q] = w.wb[q].p; fj[16 + 2 * q + 1] = w.wf16b[q].p;
        const int d[4] = {dims[q][2], dims[q][3], dims[q][3], 1};
        memcpy(ej + 32 + 4 * q, d, sizeof(d));
    }
    b.add("nele_weight_prep_frag16_batch", 0, {P(b.host(fj, 32)), P(b.host(ej, 64)), I(16)});
    if (w.fused_ok) {
        const void* gj[22]; int hj[33]; int n = 0;
        for (int l = 0; l < 6; ++l) { gj[2 * n] = w.wf[l].p; gj[2 * n + 1] = w.wgl0[l].p; hj[3 * n] = G_LAYERS[l].cout; hj[3 * n + 1] = G_LAYERS[l].cin; hj[3 * n + 2] = G_LAYERS[l].k; ++n; }
        for (int l = 1; l < 6; ++l) { gj[2 * n] = w.wb[l].p; gj[2 * n + 1] = w.wgl1[l].p; hj[3 * n] = G_LAYERS[l].cin; hj[3 * n + 1] = G_LAYERS[l].cout; hj[3 * n + 2] = G_LAYERS[l].k; ++n; }
        b.add("nele_glayer16_weight_prep_batch", 0, {P(b.host(gj, 22)), P(b.host(hj, 33)), I(n)});
    }
}

// slots: 0 x, 1 y, 2 mask, 3 token
void g_forward(Builder& b, const GWs& w, const float* par, const GParams& pp, int B, int T, bool bf16, bool need_bwd, bool prep) {
    if (prep) g_prep_weights(b, w, par, pp, bf16);
    if (w.fused) {
        b.add("nele_g_pack16", 0, {S(0), S(1), P(w.inp16[0]), I(B), I(T), I(G_LAYERS[0].k - 1)});
        for (int l = 0; l < 6; ++l) {
            const GL& L = G_LAYERS[l];
            const bool last = l == 5;
            const int padn = last ? 0 : G_LAYERS[l + 1].k - 1;
            b.add("nele_glayer16_fwd", 0, {P(w.inp16[l]), P(w.wgl0[l]), P(par + pp.b[l]), P(par + pp.gain[l]), P(par + pp.beta[l]),
                                          P(need_bwd ? w.Y[l].p : nullptr), P(need_bwd ? w.mean[l].p : nullptr), P(need_bwd ? w.rstd[l].p : nullptr),
                                          P(last ? nullptr : w.inp16[l + 1].p), P(last ? w.a5.p : nullptr), P(w.carry), S(3, l), I(B), I(T), I(L.cin), I(L.cout),
                                          I(L.k), I(padn), F(SLOPE)});
        }
    } else {
        b.add("nele_g_pack", 0, {S(0), S(1), P(w.inp[0]), I(B), I(T), I(G_LAYERS[0].k - 1)});
        for (int l = 0; l < 6; ++l) {
            const GL& L = G_LAYERS[l];
            g_gemm(b, w, bf16, w.inp[l], l, false, par + pp.b[l], nullptr, P(w.Y[l]), B, L.cout, EPI_BIAS, w.gf[l]);
            const bool last = l == 5;
            b.add("nele_cln_fwd", 0, {P(w.Y[l]), P(par + pp.gain[l]), P(par + pp.beta[l]), P(last ? w.a5 : w.inp[l + 1]), P(w.mean[l]), P(w.rstd[l]), P(w.cln_scratch),
                                     I(B), I(T), I(L.cout), I(last ? 0 : G_LAYERS[l + 1].k - 1), F(SLOPE)});
        }
    }
    g_gemm(b, w, bf16, w.a5, 6, false, par + pp.fcb[0], nullptr, P(w.h1), B, 64, EPI_BIAS_LRELU, w.gfc);
    g_gemm(b, w, bf16, w.h1, 7, false, par + pp.fcb[1], nullptr, S(2), B, 64, EPI_BIAS_EXPTANH, w.gfc);
}

// slots: 0 dmask, 1 mask.  stream 1 (overlap): the weight gradients beside the data-gradient chain
void g_backward(Builder& b, const GWs& w, const float* par, float* grad, const GParams& pp, int B, int T, bool bf16, bool overlap) {
    const int wst = overlap ? 1 : 0;
    auto wgrad = [&](const Buf& A, const Buf& dOut, int N, const Geom& g, int Cvalid, float* dW, float* db, bool b16) {
        if (overlap) b.hand_over(0, 1);
        conv_wgrad(b, wst, A, dOut, w.ws, B, N, g, Cvalid, dW, db, 1, b16);
    };
    const long long n = (long long)B * T * 64;
    b.add("nele_exptanh_bwd", 0, {S(0), S(1), P(w.do2), I(n)});
    wgrad(w.h1, w.do2, 64, w.gwfc, 64, grad + pp.fcw[1], grad + pp.fcb[1], false);
    g_gemm(b, w, bf16, w.do2, 7, true, nullptr, w.h1.p, P(w.dpre1), B, 64, EPI_MASK_LRELU_GRAD, w.gfc);
    wgrad(w.a5, w.dpre1, 64, w.gwfc, 64, grad + pp.fcw[0], grad + pp.fcb[0], false);
    g_gemm(b, w, bf16, w.dpre1, 6, true, nullptr, nullptr, P(w.da5), B, 64, EPI_NONE, w.gfc);
    Buf dact = w.da5;
    for (int l = 5; l >= 0; --l) {
        const GL& L = G_LAYERS[l];
        b.add("nele_cln_bwd", 0, {P(dact), P(w.Y[l]), P(par + pp.gain[l]), P(par + pp.beta[l]), P(w.mean[l]), P(w.rstd[l]), P(w.fused ? nullptr : w.dY[l].p),
                                 P(w.fused ? w.dY16[l].p : nullptr), P(w.gpart), P(w.bpart), P(w.cln_scratch), I(B), I(T), I(L.cout), I(L.k - 1), F(SLOPE)});
        b.add("nele_colsum2", 0, {P(w.gpart), P(grad + pp.gain[l]), P(w.bpart), P(grad + pp.beta[l]), I((long long)B * w.nchunks), I(L.cout), I(1)});
        if (w.fused) wgrad(w.inp16[l], w.dY16[l], L.cout, w.gw[l], L.cin, grad + pp.w[l], grad + pp.b[l], true);
        else wgrad(w.inp[l], w.dY[l], L.cout, w.gw[l], L.cin, grad + pp.w[l], grad + pp.b[l], bf16);
        if (l > 0) {
            if (w.fused) b.add("nele_glayer16_conv", 0, {P(w.dY16[l]), P(w.wgl1[l]), P(w.dA[l]), I(B), I(T), I(L.cout), I(L.cin), I(L.k)});
            else g_gemm(b, w, bf16, w.dY[l], l, true, nullptr, nullptr, P(w.dA[l]), B, L.cin, EPI_NONE, w.gb[l]);
            dact = w.dA[l];
        }
    }
    if (overlap) b.hand_over(1, 0);
}

// ================================================================== discriminators (model.py:101-166)
const int D_CONVS[5][2] = {{8, 1}, {16, 3}, {32, 5}, {48, 7}, {64, 9}};   // (Cout, k); Cin of layer 0 is 3 (D) or 2 (D_Qua), padded to 4

struct DParams { long long bias[8], w[8]; int N[8], K[8]; long long total; };
DParams d_params(int cin, int nout) {
    DParams p; long long o = 0;
    int ci = cin;
    for (int l = 0; l < 5; ++l) {
        const int co = D_CONVS[l][0], k = D_CONVS[l][1];
        p.bias[l] = o; o += co;
        p.w[l] = o; o += (long long)co * ci * k * k;
        p.N[l] = co; p.K[l] = ci * k * k;
        ci = co;
    }
    const int fc[3][2] = {{64, 64}, {16, 64}, {nout, 16}};
    for (int q = 0; q < 3; ++q) {
        p.bias[5 + q] = o; o += fc[q][0];
        p.w[5 + q] = o; o += (long long)fc[q][0] * fc[q][1];
        p.N[5 + q] = fc[q][0]; p.K[5 + q] = fc[q][1];
    }
    p.total = o;
    return p;
}

struct DWs {
    Buf sigma, wf[5], wb[5], wff[5], wbf[5], wff16[5], wbf16[5], wf16c[5], wb16c[5];
    Buf act[5], gbuf[5], gbuf16, ddin, gap_part, pooled, h1, h2, dz1, dz2, dz3, dpooled, ws, tmpw, scratch64, ws2, tmpw2, scratch64b;
    Geom gf[5], gb[5], gw[5];
    int dims[6][3], pad[5], cins[5];
    bool c16 = false, grad16_ok = false, span_f[5], span_b[5], span16_f[5], span16_b[5];
    int P = 0, gap_parts = 0;
    size_t bytes = 0;
};

int d_layout(int B, int T, int cin0, bool bf16, void* base, DWs& w) {
    Ws a(base);
    w.sigma = a.f32(8);
    {
        int cpad = 4;
        for (int l = 0; l < 5; ++l) {
            const int co = D_CONVS[l][0], k = D_CONVS[l][1];
            w.wf[l] = a.f32((long long)co * k * k * cpad);
            w.wb[l] = a.f32((long long)cpad * k * k * co);
            if ((k * k * cpad) % 8 == 0) w.wff[l] = a.f32((long long)(k * k * cpad / 8) * ((co + 15) / 16) * 128);
            if ((k * k * co) % 8 == 0) w.wbf[l] = a.f32((long long)(k * k * co / 8) * ((cpad + 15) / 16) * 128);
            w.wff16[l] = a.bf16(nele_weight_frag16_elems(co, k * cpad, k));
            w.wbf16[l] = a.bf16(nele_weight_frag16_elems(cpad, k * co, k));
            if (k > 1) { w.wf16c[l] = a.bf16(nele_conv16_wfrag_elems(co, k * cpad, k)); w.wb16c[l] = a.bf16(nele_conv16_wfrag_elems(cpad, k * co, k)); }
            cpad = co;
        }
    }
    int H = 64, W = T, C = 4;
    w.dims[0][0] = H; w.dims[0][1] = W; w.dims[0][2] = C;
    for (int l = 0; l < 5; ++l) {
        const int co = D_CONVS[l][0], k = D_CONVS[l][1];
        const int Ho = H - k + 1, Wo = W - k + 1;
        if (Ho < 1 || Wo < 1) return nele_set_error(NELE_ERR_INVALID_ARG, "discriminator plan: T=%d is too short (needs T >= 21, model.py:105-109)", T);
        w.gf[l] = mk_geom(H, W, C, Ho, Wo, k, k, Ho, Wo, co);
        w.pad[l] = k - 1;
        w.dims[l + 1][0] = Ho; w.dims[l + 1][1] = Wo; w.dims[l + 1][2] = co;
        H = Ho; W = Wo; C = co;
    }
    w.ddin = a.f32((long long)B * 64 * T * 4);
    for (int l = 0; l < 5; ++l) {
        const int co = D_CONVS[l][0], k = D_CONVS[l][1];
        const int Hi = w.dims[l][0], Wi = w.dims[l][1], Ci = w.dims[l][2], Ho = w.dims[l + 1][0], Wo = w.dims[l + 1][1], p = k - 1;
        int OH, OW, OC, o0;
        if (l == 0) { OH = Hi; OW = Wi; OC = 4; o0 = 0; }
        else { const int pp = w.pad[l - 1]; OH = Hi + 2 * pp; OW = Wi + 2 * pp; OC = Ci; o0 = pp; }
        w.gb[l] = mk_geom(Ho + 2 * p, Wo + 2 * p, co, Hi, Wi, k, k, OH, OW, OC, 0, 0, o0, o0);
        w.gw[l] = mk_geom(Hi, Wi, Ci, Ho, Wo, k, k, Ho + 2 * p, Wo + 2 * p, co, 0, 0, p, p);
        w.cins[l] = l == 0 ? 4 : D_CONVS[l - 1][0];
    }
    (void)cin0;
    w.c16 = bf16;
    for (int l = 1; l < 5 && w.c16; ++l)
        w.c16 = nele_conv16_supported((int)M_of(B, w.gf[l]), D_CONVS[l][0], w.gf[l].a, w.gf[l].KH, w.gf[l].KW) &&
                nele_conv16_supported((int)M_of(B, w.gb[l]), w.cins[l], w.gb[l].a, w.gb[l].KH, w.gb[l].KW) &&
                nele_conv_wgrad_bf16_d16_supported((int)M_of(B, w.gw[l]), D_CONVS[l][0], w.gw[l].a, w.gw[l].KH, w.gw[l].KW);
    for (int l = 0; l < 5; ++l) {
        const int co = D_CONVS[l][0], p = w.pad[l];
        const long long na = (long long)B * w.dims[l + 1][0] * w.dims[l + 1][1] * co;
        const long long ng = (long long)B * (w.dims[l + 1][0] + 2 * p) * (w.dims[l + 1][1] + 2 * p) * co;
        w.act[l] = w.c16 ? a.bf16(na) : a.f32(na);
        w.gbuf[l] = (w.c16 && l >= 1) ? a.bf16(ng) : a.f32(ng);
        w.span_f[l] = nele_conv_span_supported((int)M_of(B, w.gf[l]), co, w.gf[l].a, w.gf[l].KH, w.gf[l].KW) != 0;
        w.span_b[l] = nele_conv_span_supported((int)M_of(B, w.gb[l]), w.cins[l], w.gb[l].a, w.gb[l].KH, w.gb[l].KW) != 0;
        w.span16_f[l] = nele_conv_span_bf16_supported((int)M_of(B, w.gf[l]), co, w.gf[l].a, w.gf[l].KH, w.gf[l].KW) != 0;
        w.span16_b[l] = nele_conv_span_bf16_supported((int)M_of(B, w.gb[l]), w.cins[l], w.gb[l].a, w.gb[l].KH, w.gb[l].KW) != 0;
    }
    w.grad16_ok = !w.c16 && nele_conv_span_bf16_a16_supported((int)M_of(B, w.gb[4]), w.cins[4], w.gb[4].a, w.gb[4].KH, w.gb[4].KW) &&
                  nele_conv_wgrad_bf16_d16_supported((int)M_of(B, w.gw[4]), D_CONVS[4][0], w.gw[4].a, w.gw[4].KH, w.gw[4].KW);
    if (bf16 && w.grad16_ok) w.gbuf16 = a.bf16(w.gbuf[4].elems);
    w.P = w.dims[5][0] * w.dims[5][1];
    w.gap_parts = w.c16 ? nele_conv16_gap_parts(D_CONVS[4][0], w.gf[4].a, w.gf[4].KH, w.gf[4].KW) : 0;
    if (w.c16) w.gap_part = a.f64((long long)B * w.gap_parts * 64);
    w.pooled = a.f32((long long)B * 64); w.h1 = a.f32((long long)B * 64); w.h2 = a.f32((long long)B * 16);
    w.dz1 = a.f32((long long)B * 64); w.dz2 = a.f32((long long)B * 16); w.dz3 = a.f32((long long)B * 4); w.dpooled = a.f32((long long)B * 64);
    long long nws = 0;
    for (int l = 0; l < 5; ++l) {
        const long long v = nele_conv_wgrad_workspace_floats((int)M_of(B, w.gw[l]), D_CONVS[l][0], w.gw[l].Ktot, nullptr);
        if (v > nws) nws = v;
    }
    const long long ns64 = (long long)B * 32 * 64 > 128 ? (long long)B * 32 * 64 : 128;
    w.ws = a.f32(nws); w.tmpw = a.f32(64 * 48 * 81 + 64); w.scratch64 = a.f64(ns64);
    w.ws2 = a.f32(nws); w.tmpw2 = a.f32(64 * 48 * 81 + 64); w.scratch64b = a.f64(ns64);
    w.bytes = a.off;
    return NELE_OK;
}

const void* const* d_mlp_ptrs(Builder& b, const DWs& w, const float* par, const DParams& pp) {
    const void* arr[9];
    for (int q = 0; q < 3; ++q) {
        arr[3 * q] = par + pp.w[5 + q];
        arr[3 * q + 1] = par + pp.bias[5 + q];
        arr[3 * q + 2] = reinterpret_cast<const float*>(w.sigma.p) + 5 + q;
    }
    return b.host(arr, 9);
}

// _DiscriminatorBase._prepare_inline: spectral norm (power iteration in training mode) + sigma + every weight layout of this shape / mode
void d_prepare(Builder& b, const DWs& w, const float* par, const DParams& pp, const void* const* uv, int cin, bool bf16, bool train) {
    const void* sp[24]; int sd[16];
    for (int l = 0; l < 8; ++l) { sp[3 * l] = par + pp.w[l]; sp[3 * l + 1] = uv[2 * l]; sp[3 * l + 2] = uv[2 * l + 1]; sd[2 * l] = pp.N[l]; sd[2 * l + 1] = pp.K[l]; }
    b.add("nele_spectral_norm", 0, {P(b.host(sp, 24)), P(b.host(sd, 16)), I(8), P(w.sigma), I(train ? 1 : 0)});
    const float* sig = reinterpret_cast<const float*>(w.sigma.p);
    if (bf16) {
        const void* pj[20]; int dj[25]; const void* fj[20]; int ej[40]; const void* cj[16]; int gj[32];
        int nf = 0, nc = 0, ci = cin, cpad = 4;
        for (int l = 0; l < 5; ++l) {
            const int co = D_CONVS[l][0], k = D_CONVS[l][1];
            pj[4 * l] = par + pp.w[l]; pj[4 * l + 1] = sig + l; pj[4 * l + 2] = w.wf[l].p; pj[4 * l + 3] = w.wb[l].p;
            const int d[5] = {co, ci, cpad, k, k};
            memcpy(dj + 5 * l, d, sizeof(d));
            if (w.c16) {
                if (l >= 1) {
                    cj[2 * nc] = w.wf[l].p; cj[2 * nc + 1] = w.wf16c[l].p;
                    const int g0[4] = {co, k * k * cpad, k * cpad, k};
                    memcpy(gj + 4 * nc, g0, sizeof(g0)); ++nc;
                    cj[2 * nc] = w.wb[l].p; cj[2 * nc + 1] = w.wb16c[l].p;
                    const int g1[4] = {cpad, k * k * co, k * co, k};
                    memcpy(gj + 4 * nc, g1, sizeof(g1)); ++nc;
                }
            } else {
                if (w.span16_f[l]) { fj[2 * nf] = w.wf[l].p; fj[2 * nf + 1] = w.wff16[l].p; const int e0[4] = {co, k * k * cpad, k * cpad, k}; memcpy(ej + 4 * nf, e0, sizeof(e0)); ++nf; }
                if (w.span16_b[l]) { fj[2 * nf] = w.wb[l].p; fj[2 * nf + 1] = w.wbf16[l].p; const int e1[4] = {cpad, k * k * co, k * co, k}; memcpy(ej + 4 * nf, e1, sizeof(e1)); ++nf; }
            }
            ci = cpad = co;
        }
        b.add("nele_weight_prep_batch", 0, {P(b.host(pj, 20)), P(b.host(dj, 25)), I(5)});
        if (nf) b.add("nele_weight_prep_frag16_batch", 0, {P(b.host(fj, 20)), P(b.host(ej, 40)), I(nf)});
        if (nc) { b.add("nele_conv16_weight_prep_batch", 0, {P(b.host(cj, 16)), P(b.host(gj, 32)), I(nc)}); return; }
        cpad = 4;
        for (int l = 0; l < 5; ++l) {                // layers the bf16 kernels decline still need the float32 fragment layout
            const int co = D_CONVS[l][0], k = D_CONVS[l][1];
            if (!w.span16_f[l] && w.span_f[l]) b.add("nele_weight_prep_frag", 0, {P(w.wf[l]), I(co), I(k * k * cpad), P(w.wff[l])});
            if (!w.span16_b[l] && w.span_b[l]) b.add("nele_weight_prep_frag", 0, {P(w.wb[l]), I(cpad), I(k * k * co), P(w.wbf[l])});
            cpad = co;
        }
    } else {
        int ci = cin, cpad = 4;
        for (int l = 0; l < 5; ++l) {
            const int co = D_CONVS[l][0], k = D_CONVS[l][1];
            b.add("nele_weight_prep", 0, {P(par + pp.w[l]), P(sig + l), I(co), I(ci), I(cpad), I(k), I(k), P(w.wf[l]), P(w.wb[l])});
            if (w.span_f[l]) b.add("nele_weight_prep_frag", 0, {P(w.wf[l]), I(co), I(k * k * cpad), P(w.wff[l])});
            if (w.span_b[l]) b.add("nele_weight_prep_frag", 0, {P(w.wb[l]), I(cpad), I(k * k * co), P(w.wbf[l])});
            ci = cpad = co;
        }
    }
}

// slots: 0 din, 1 wvalid (may be NULL), 2 score
void d_forward(Builder& b, const DWs& w, const float* par, const DParams& pp, int B, int T, int nout, bool bf16) {
    Buf din; din.p = nullptr; din.elems = (long long)B * 64 * T * 4;
    const void* const* mlp = d_mlp_ptrs(b, w, par, pp);
    for (int l = 0; l < 5; ++l) {
        const int co = D_CONVS[l][0];
        const Geom& g = w.gf[l];
        const long long M = M_of(B, g);
        const Arg A = l == 0 ? S(0) : P(w.act[l - 1]);
        const long long a_elems = l == 0 ? din.elems : w.act[l - 1].elems;
        const void* bias = par + pp.bias[l];
        if (w.c16) {
            if (l == 0) b.add("nele_conv16_pointwise_fwd", 0, {S(0), P(w.wf[0]), P(bias), P(w.act[0]), I(din.elems / 4), I(8), F(SLOPE)});
            else if (l == 4)
                b.add("nele_conv16_gap", 0, {A, P(w.wf16c[l]), P(bias), P(w.act[l]), I(M), I(co), F(SLOPE), P(b.geom(g)), I(g.KH), I(g.KW), S(1), P(w.gap_part)});
            else
                b.add("nele_conv16", 0, {A, P(w.wf16c[l]), P(bias), P((void*)nullptr), P(w.act[l]), I(1), I(M), I(co), I(EPI_BIAS_LRELU), F(SLOPE), P(b.geom(g)), I(g.KH), I(g.KW)});
        } else if (bf16 && w.span16_f[l]) {
            b.add("nele_conv_span_bf16", 0, {A, P(w.wff16[l]), P(bias), P((void*)nullptr), P(w.act[l]), I(M), I(co), I(EPI_BIAS_LRELU), F(SLOPE), P(b.geom(g)), I(g.KH), I(g.KW), I(a_elems)});
        } else if (w.span_f[l]) {
            b.add("nele_conv_span", 0, {A, P(w.wff[l]), P(bias), P((void*)nullptr), P(w.act[l]), I(M), I(co), I(EPI_BIAS_LRELU), F(SLOPE), P(b.geom(g)), I(g.KH), I(g.KW), I(a_elems)});
        } else {
            b.add("nele_conv_gemm", 0, {A, P(w.wf[l]), P(bias), P((void*)nullptr), P(w.act[l]), I(M), I(co), I(EPI_BIAS_LRELU), F(SLOPE), P(b.geom(g))});
        }
    }
    if (w.c16)
        b.add("nele_gap_mlp_fwd_parts", 0, {P(w.gap_part), I(w.gap_parts), I(B), I(w.P), I(w.dims[5][1]), S(1), P(mlp), I(nout), F(SLOPE), P(w.pooled), P(w.h1), P(w.h2), S(2)});
    else
        b.add("nele_gap_mlp_fwd_var", 0, {P(w.act[4]), I(B), I(w.P), I(w.dims[5][1]), S(1), P(mlp), I(nout), F(SLOPE), P(w.pooled), P(w.h1), P(w.h2), S(2), P(w.scratch64)});
}

// slots: 0 dscore, 1 score, 2 wvalid (may be NULL), 3 din.  streams 1, 2 (overlap): the layers' weight gradients beside the data-gradient chain
void d_backward(Builder& b, const DWs& w, const float* par, float* grad, const DParams& pp, const void* const* uv, int B, int T, int cin, int nout, bool bf16,
                bool need_din, bool wgrad, bool overlap) {
    const void* const* mlp = d_mlp_ptrs(b, w, par, pp);
    const int Ho = w.dims[5][0], Wo = w.dims[5][1], p5 = w.pad[4];
    const bool g16 = bf16 && (w.grad16_ok || w.c16);
    const Buf& glast = (g16 && !w.c16) ? w.gbuf16 : w.gbuf[4];
    b.add(g16 ? (w.c16 ? "nele_gap_mlp_bwd_var16a" : "nele_gap_mlp_bwd_var16") : "nele_gap_mlp_bwd_var", 0,
          {S(0), S(1), P(w.h1), P(w.h2), P(w.act[4]), P(mlp), I(nout), F(SLOPE), I(B), I(Ho), I(Wo), S(2), I(Ho + 2 * p5), I(Wo + 2 * p5), I(p5), I(p5), P(w.dz3), P(w.dz2),
           P(w.dz1), P(w.dpooled), P(glast)});
    if (overlap) { b.hand_over(0, 1); b.hand_over(0, 2); }
    const float* sig = reinterpret_cast<const float*>(w.sigma.p);
    if (wgrad) {
        const int st = overlap ? 2 : 0;
        const Buf& tw = overlap ? w.tmpw2 : w.tmpw;
        const Buf& sc = overlap ? w.scratch64b : w.scratch64;
        const Buf* dz[3] = {&w.dz3, &w.dz2, &w.dz1};
        const Buf* xin[3] = {&w.h2, &w.h1, &w.pooled};
        for (int i = 0; i < 3; ++i) {
            const int li = 7 - i, N = pp.N[li], K = pp.K[li];
            float* tmpb = reinterpret_cast<float*>(tw.p) + (long long)N * K;
            b.add("nele_mlp_wgrad", st, {P(*dz[i]), P(*xin[i]), I(B), I(N), I(K), P(tw), P(tmpb)});
            b.add("nele_sn_grad", st, {P(tw), P(par + pp.w[li]), P(uv[2 * li]), P(uv[2 * li + 1]), P(sig + li), I(N), I(K), P(grad + pp.w[li]), I(1), P(sc)});
            b.add("nele_vec_add", st, {P(grad + pp.bias[li]), P(tmpb), I(N)});
        }
    }
    Buf din; din.elems = (long long)B * 64 * T * 4;
    for (int l = 4; l >= 0; --l) {
        const int co = D_CONVS[l][0], k = D_CONVS[l][1];
        const int Ci = w.dims[l][2];
        const int cin_valid = l == 0 ? cin : Ci;
        if (wgrad) {
            const int N = co, K = cin_valid * k * k;
            const int q = l == 0 ? 1 : (l & 1);
            const Buf& tw = q == 0 ? w.tmpw : w.tmpw2;
            const Buf& wsb = q == 0 ? w.ws : w.ws2;
            const Buf& sc = q == 0 ? w.scratch64 : w.scratch64b;
            float* tmpb = reinterpret_cast<float*>(tw.p) + (long long)N * K;
            const int st = overlap ? 1 + q : 0;
            if (overlap) b.hand_over(0, st);
            const Buf& dOut = l == 4 ? glast : w.gbuf[l];
            const bool b16 = bf16 && l > 0;
            const char* fn = b16 ? "nele_conv_wgrad_bf16" : "nele_conv_wgrad";
            const bool a16 = l > 0 && w.act[l - 1].b16;
            if (dOut.b16) fn = a16 ? "nele_conv_wgrad_bf16_a16d16" : "nele_conv_wgrad_bf16_d16";
            const Geom& g = w.gw[l];
            b.add(fn, st, {l == 0 ? S(3) : P(w.act[l - 1]), P(dOut), P(wsb), I(wsb.elems), I(M_of(B, g)), I(co), P(b.geom(g)), I(g.KH), I(g.KW), I(cin_valid), P(tw), P(tmpb), I(0)});
            b.add("nele_sn_grad", st, {P(tw), P(par + pp.w[l]), P(uv[2 * l]), P(uv[2 * l + 1]), P(sig + l), I(N), I(K), P(grad + pp.w[l]), I(1), P(sc)});
            b.add("nele_vec_add", st, {P(grad + pp.bias[l]), P(tmpb), I(N)});
        }
        const Geom& g = w.gb[l];
        const long long M = M_of(B, g);
        if (l > 0 && w.c16) {
            b.add("nele_conv16", 0, {P(w.gbuf[l]), P(w.wb16c[l]), P((void*)nullptr), P(w.act[l - 1]), P(w.gbuf[l - 1]), I(w.gbuf[l - 1].b16 ? 1 : 0), I(M), I(Ci), I(EPI_MASK_LRELU_GRAD),
                                    F(SLOPE), P(b.geom(g)), I(g.KH), I(g.KW)});
        } else if (l > 0) {
            if (bf16 && w.span16_b[l]) {
                const Buf& src = l == 4 ? glast : w.gbuf[l];
                b.add(src.b16 ? "nele_conv_span_bf16_a16" : "nele_conv_span_bf16", 0, {P(src), P(w.wbf16[l]), P((void*)nullptr), P(w.act[l - 1]), P(w.gbuf[l - 1]), I(M), I(Ci),
                                                                                    I(EPI_MASK_LRELU_GRAD), F(SLOPE), P(b.geom(g)), I(g.KH), I(g.KW), I(src.elems)});
            } else if (w.span_b[l]) {
                b.add("nele_conv_span", 0, {P(w.gbuf[l]), P(w.wbf[l]), P((void*)nullptr), P(w.act[l - 1]), P(w.gbuf[l - 1]), I(M), I(Ci), I(EPI_MASK_LRELU_GRAD), F(SLOPE), P(b.geom(g)),
                                           I(g.KH), I(g.KW), I(w.gbuf[l].elems)});
            } else {
                b.add("nele_conv_gemm", 0, {P(w.gbuf[l]), P(w.wb[l]), P((void*)nullptr), P(w.act[l - 1]), P(w.gbuf[l - 1]), I(M), I(Ci), I(EPI_MASK_LRELU_GRAD), F(SLOPE), P(b.geom(g))});
            }
        } else if (need_din) {
            b.add("nele_conv_gemm", 0, {P(w.gbuf[0]), P(w.wb[0]), P((void*)nullptr), P((void*)nullptr), P(w.ddin), I(M), I(4), I(EPI_NONE), F(SLOPE), P(b.geom(g))});
        }
    }
    if (overlap) { b.hand_over(1, 0); b.hand_over(2, 0); }
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" long long nele_gen_param_count(void) { return g_params().total; }
extern "C" long long nele_disc_param_count(int cin, int nout) { return (cin >= 1 && cin <= 4 && nout >= 1 && nout <= 4) ? d_params(cin, nout).total : -1; }

// offsets (in floats) of the parameters inside the flat buffers, in nn.Module.parameters() order: G 28 entries, D 16
extern "C" int nele_gen_param_layout(long long* offsets, int n) {
    NELE_CHECK_ARG(offsets && n >= 28, "nele_gen_param_layout: 28 offsets");
    const GParams p = g_params();
    int k = 0;
    for (int l = 0; l < 6; ++l) { offsets[k++] = p.w[l]; offsets[k++] = p.b[l]; offsets[k++] = p.gain[l]; offsets[k++] = p.beta[l]; }
    for (int q = 0; q < 2; ++q) { offsets[k++] = p.fcw[q]; offsets[k++] = p.fcb[q]; }
    return k;
}
extern "C" int nele_disc_param_layout(int cin, int nout, long long* offsets, int n) {
    NELE_CHECK_ARG(offsets && n >= 16 && cin >= 1 && cin <= 4 && nout >= 1 && nout <= 4, "nele_disc_param_layout: 16 offsets, cin 1..4, nout 1..4");
    const DParams p = d_params(cin, nout);
    int k = 0;
    for (int l = 0; l < 8; ++l) { offsets[k++] = p.bias[l]; offsets[k++] = p.w[l]; }
    return k;
}

extern "C" long long nele_gen_workspace_bytes(int B, int T, int bf16, int need_bwd) {
    if (B <= 0 || T <= 0) return -1;
    GWs w;
    g_layout(B, T, bf16 != 0, need_bwd != 0, nullptr, w);
    return (long long)w.bytes;
}

extern "C" int nele_gen_plan_build(int B, int T, int bf16, int need_bwd, int overlap_wgrad, const float* params_flat, float* grads_flat, void* workspace,
                                   long long workspace_bytes, void* stream, void** fwd_out, void** bwd_out) {
    NELE_CHECK_ARG(B > 0 && T > 0 && params_flat && workspace && fwd_out && (!need_bwd || (grads_flat && bwd_out)), "nele_gen_plan_build: bad arguments");
    GWs w;
    g_layout(B, T, bf16 != 0, need_bwd != 0, workspace, w);
    if (workspace_bytes < (long long)w.bytes) return nele_set_error(NELE_ERR_WORKSPACE, "nele_gen_plan_build: workspace too small (%lld < %lld)", workspace_bytes, (long long)w.bytes);
    // zero rows in front of / behind the time axis, zero carry slots: written once, kept by the kernels
    if (hipMemsetAsync(workspace, 0, w.bytes, as_stream(stream)) != hipSuccess) return nele_set_error(NELE_ERR_HIP, "nele_gen_plan_build: hipMemsetAsync failed");
    const GParams pp = g_params();
    const long long io = (long long)B * T * 64 * 4;
    {
        Builder b;
        g_forward(b, w, params_flat, pp, B, T, bf16 != 0, need_bwd != 0, true);
        void* h = nullptr;
        const int st = b.finish(4, &h);
        if (st) return st;
        nele_plan_declare_slot(h, 0, io, 0); nele_plan_declare_slot(h, 1, io, 0); nele_plan_declare_slot(h, 2, io, 0);
        *fwd_out = h;
    }
    if (need_bwd) {
        Builder b;
        g_backward(b, w, params_flat, grads_flat, pp, B, T, bf16 != 0, overlap_wgrad != 0);
        void* h = nullptr;
        const int st = b.finish(2, &h);
        if (st) { (void)nele_plan_destroy(*fwd_out); *fwd_out = nullptr; return st; }
        nele_plan_declare_slot(h, 0, io, 0); nele_plan_declare_slot(h, 1, io, 0);
        *bwd_out = h;
    } else if (bwd_out) *bwd_out = nullptr;
    return NELE_OK;
}

extern "C" long long nele_disc_workspace_bytes(int B, int T, int cin, int bf16) {
    if (B <= 0 || T < 21 || cin < 1 || cin > 4) return -1;
    DWs w;
    if (d_layout(B, T, cin, bf16 != 0, nullptr, w)) return -1;
    return (long long)w.bytes;
}

// device pointer of the input gradient [B][64][T][4] the backward plan writes (need_din), inside `workspace`
extern "C" float* nele_disc_workspace_ddin(void* workspace, int B, int T, int cin, int bf16) {
    DWs w;
    if (!workspace || B <= 0 || T < 21 || d_layout(B, T, cin, bf16 != 0, workspace, w)) return nullptr;
    return reinterpret_cast<float*>(w.ddin.p);
}

extern "C" int nele_disc_plan_build(int B, int T, int cin, int nout, int bf16, int train, int need_din, int weight_grads, int overlap_wgrad, const float* params_flat,
                                    float* grads_flat, const void* const* sn_uv_host, void* workspace, long long workspace_bytes, void* stream, void** fwd_out,
                                    void** bwd_out) {
    NELE_CHECK_ARG(B > 0 && T >= 21 && cin >= 1 && cin <= 4 && nout >= 1 && nout <= 4 && params_flat && sn_uv_host && workspace && fwd_out && (!bwd_out || grads_flat),
                   "nele_disc_plan_build: bad arguments (T >= 21, cin 1..4, nout 1..4)");
    DWs w;
    { const int st = d_layout(B, T, cin, bf16 != 0, workspace, w); if (st) return st; }
    if (workspace_bytes < (long long)w.bytes) return nele_set_error(NELE_ERR_WORKSPACE, "nele_disc_plan_build: workspace too small (%lld < %lld)", workspace_bytes, (long long)w.bytes);
    if (hipMemsetAsync(workspace, 0, w.bytes, as_stream(stream)) != hipSuccess) return nele_set_error(NELE_ERR_HIP, "nele_disc_plan_build: hipMemsetAsync failed");
    const DParams pp = d_params(cin, nout);
    const long long din_bytes = (long long)B * 64 * T * 4 * 4, sc_bytes = (long long)B * nout * 4;
    {
        Builder b;
        d_prepare(b, w, params_flat, pp, sn_uv_host, cin, bf16 != 0, train != 0);
        d_forward(b, w, params_flat, pp, B, T, nout, bf16 != 0);
        void* h = nullptr;
        const int st = b.finish(3, &h);
        if (st) return st;
        nele_plan_declare_slot(h, 0, din_bytes, 0); nele_plan_declare_slot(h, 1, (long long)B * 4, 1); nele_plan_declare_slot(h, 2, sc_bytes, 0);
        *fwd_out = h;
    }
    if (bwd_out) {
        Builder b;
        d_backward(b, w, params_flat, grads_flat, pp, sn_uv_host, B, T, cin, nout, bf16 != 0, need_din != 0, weight_grads != 0, overlap_wgrad != 0);
        void* h = nullptr;
        const int st = b.finish(4, &h);
        if (st) { (void)nele_plan_destroy(*fwd_out); *fwd_out = nullptr; return st; }
        nele_plan_declare_slot(h, 0, sc_bytes, 0); nele_plan_declare_slot(h, 1, sc_bytes, 0); nele_plan_declare_slot(h, 2, (long long)B * 4, 1);
        nele_plan_declare_slot(h, 3, din_bytes, 0);
        *bwd_out = h;
    }
    return NELE_OK;
}
