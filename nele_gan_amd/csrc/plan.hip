// Job tables: the composite entry points of the dense part (nele_gen_fwd / nele_gen_bwd / nele_disc_fwd / nele_disc_bwd).
//
// A forward or backward pass of the generator / a discriminator is a fixed sequence of this library's own per-layer entry points over a
// few streams (model.py:83-98, 118-132 and their autograd): ~25 .. 60 launches whose arguments do not change from step to step except a
// handful of pointers (the batch's inputs and outputs) and counters.  Driven from the host language one call at a time, a B = 32
// GAN_epoch step spent 3.6 of its 5.6 ms in ~300 foreign-function calls.  A PLAN is that sequence recorded once per shape -
// op code + arguments, each argument either a constant or "slot k + offset" of a small array supplied per call - and nele_plan_run
// enqueues it with ONE call: same kernels, same order, same streams and hand-over events, bit-identical results.  The host mirror
// (nele_gan_amd/_lib.py: PlanRecorder) builds plans by recording its own per-layer loop the first time a shape is seen, so the layer
// logic exists once.  Not a HIP graph: the per-call pointers stay free (torch allocates outputs per step), nothing is captured, and the
// launches go to whatever streams the caller passes.
#include "common.h"
#include "../../include/nele_hip.h"
#include <cstring>
#include <memory>
#include <new>
#include <vector>

struct PlanJobI {
    int op, nargs, stream;
    int slot[NELE_PLAN_MAXARGS];
    long long ival[NELE_PLAN_MAXARGS];
    double fval[NELE_PLAN_MAXARGS];
};
struct NelePlan {
    unsigned magic;
    int nslots, nstreams;
    std::vector<PlanJobI> jobs;
    long long slot_bytes[64];                       // bytes a call must provide behind slot k (0: a scalar slot / not declared)
    unsigned char slot_nullable[64];                // the slot may be NULL (an optional argument such as wvalid)
    std::vector<std::unique_ptr<char[]>> blobs;     // host arrays the jobs point to (plans built inside the library, netplan.hip)
    std::vector<hipEvent_t> events;                 // hand-over events owned by the plan
    ~NelePlan() { for (hipEvent_t e : events) (void)hipEventDestroy(e); }
};
#define PLAN_MAGIC 0x4e504c4eu

struct PlanOpInfo { const char* name; int nargs; };
static const PlanOpInfo plan_ops[] = {
#define PLAN_OP_TABLE
#define PLAN_OP(idx, name, nargs) {#name, nargs},
#include "plan_ops.inc"
#undef PLAN_OP
#undef PLAN_OP_TABLE
};
static const int plan_nops = (int)(sizeof(plan_ops) / sizeof(plan_ops[0]));

extern "C" int nele_plan_op_id(const char* name) {
    if (!name) return -1;
    for (int k = 0; k < plan_nops; ++k)
        if (!strcmp(plan_ops[k].name, name)) return k;
    return -1;
}
extern "C" int nele_plan_op_nargs(int op) { return (op >= 0 && op < plan_nops) ? plan_ops[op].nargs : -1; }

static int plan_dispatch(int op, const long long* v, const double* d, void* S) {
#define V(i) v[i]
#define D(i) d[i]
    switch (op) {
#define PLAN_OP_CASES
#include "plan_ops.inc"
#undef PLAN_OP_CASES
        default: break;
    }
#undef V
#undef D
    return nele_set_error(NELE_ERR_INVALID_ARG, "nele_plan_run: unknown operation %d", op);
}

extern "C" int nele_plan_create(const nele_plan_job* jobs, int njobs, int nslots, int nstreams, void** plan_out) {
    NELE_CHECK_ARG(jobs && njobs > 0 && nslots >= 0 && nslots <= 64 && nstreams >= 1 && nstreams <= 16 && plan_out, "nele_plan_create: bad arguments");
    for (int j = 0; j < njobs; ++j) {
        const nele_plan_job& q = jobs[j];
        NELE_CHECK_ARG(q.op >= 0 && q.op < plan_nops, "nele_plan_create: job %d: unknown operation %d", j, q.op);
        NELE_CHECK_ARG(q.nargs == plan_ops[q.op].nargs, "nele_plan_create: job %d (%s): %d arguments, the entry point takes %d", j, plan_ops[q.op].name, q.nargs,
                       plan_ops[q.op].nargs);
        NELE_CHECK_ARG(q.stream >= 0 && q.stream < nstreams, "nele_plan_create: job %d: stream %d of %d", j, q.stream, nstreams);
        for (int i = 0; i + 1 < q.nargs; ++i)
            NELE_CHECK_ARG(q.slot[i] >= -1 && q.slot[i] < nslots, "nele_plan_create: job %d argument %d: slot %d of %d", j, i, q.slot[i], nslots);
    }
    NelePlan* p = new (std::nothrow) NelePlan;
    if (!p) return nele_set_error(NELE_ERR_HIP, "nele_plan_create: out of host memory");
    p->magic = PLAN_MAGIC; p->nslots = nslots; p->nstreams = nstreams;
    memset(p->slot_bytes, 0, sizeof(p->slot_bytes)); memset(p->slot_nullable, 0, sizeof(p->slot_nullable));
    p->jobs.resize(njobs);
    for (int j = 0; j < njobs; ++j) {
        PlanJobI& o = p->jobs[j];
        o.op = jobs[j].op; o.nargs = jobs[j].nargs; o.stream = jobs[j].stream;
        memcpy(o.slot, jobs[j].slot, sizeof(o.slot)); memcpy(o.ival, jobs[j].ival, sizeof(o.ival)); memcpy(o.fval, jobs[j].fval, sizeof(o.fval));
    }
    *plan_out = p;
    return NELE_OK;
}

// Declares what a call must provide behind slot k: `bytes` of device memory (0 = a scalar slot), `nullable` != 0: the pointer may be NULL.
// nele_plan_run_sized and the composite entry points check their arguments against it: a recorded job table holds raw offsets, and a
// short or missing buffer would otherwise be a wild device access.
extern "C" int nele_plan_declare_slot(void* plan, int slot, long long bytes, int nullable) {
    NelePlan* p = reinterpret_cast<NelePlan*>(plan);
    NELE_CHECK_ARG(p && p->magic == PLAN_MAGIC && slot >= 0 && slot < p->nslots && bytes >= 0, "nele_plan_declare_slot: bad arguments");
    p->slot_bytes[slot] = bytes; p->slot_nullable[slot] = nullable ? 1 : 0;
    return NELE_OK;
}
extern "C" long long nele_plan_slot_bytes(void* plan, int slot) {
    NelePlan* p = reinterpret_cast<NelePlan*>(plan);
    if (!p || p->magic != PLAN_MAGIC || slot < 0 || slot >= p->nslots) return -1;
    return p->slot_bytes[slot];
}

// netplan.hip: the library's own plan builders hand over the host arrays and events their jobs refer to
int nele_plan_adopt(void* plan, std::vector<std::unique_ptr<char[]>>&& blobs, std::vector<hipEvent_t>&& events) {
    NelePlan* p = reinterpret_cast<NelePlan*>(plan);
    if (!p || p->magic != PLAN_MAGIC) return nele_set_error(NELE_ERR_INVALID_ARG, "nele_plan_adopt: not a plan");
    p->blobs = std::move(blobs); p->events = std::move(events);
    return NELE_OK;
}

extern "C" int nele_plan_destroy(void* plan) {
    NelePlan* p = reinterpret_cast<NelePlan*>(plan);
    if (!p || p->magic != PLAN_MAGIC) return nele_set_error(NELE_ERR_INVALID_ARG, "nele_plan_destroy: not a plan");
    p->magic = 0;
    delete p;
    return NELE_OK;
}

extern "C" int nele_plan_run(void* plan, void* const* streams_host, int nstreams, const long long* slots_host, int nslots) {
    NelePlan* p = reinterpret_cast<NelePlan*>(plan);
    NELE_CHECK_ARG(p && p->magic == PLAN_MAGIC, "nele_plan_run: not a plan");
    NELE_CHECK_ARG(streams_host && nstreams >= p->nstreams && nslots >= p->nslots && (slots_host || p->nslots == 0), "nele_plan_run: %d streams / %d slots given, the plan needs %d / %d",
                   nstreams, nslots, p->nstreams, p->nslots);
    for (int k = 0; k < p->nslots; ++k)              // a declared pointer slot must not be NULL unless it was declared optional
        if (p->slot_bytes[k] > 0 && !p->slot_nullable[k] && slots_host[k] == 0)
            return nele_set_error(NELE_ERR_INVALID_ARG, "nele_plan_run: slot %d is NULL (the plan reads / writes %lld bytes there)", k, p->slot_bytes[k]);
    long long v[NELE_PLAN_MAXARGS];
    for (const PlanJobI& q : p->jobs) {
        const int na = q.nargs - 1;                        // the last argument is the stream
        for (int i = 0; i < na; ++i) v[i] = q.slot[i] < 0 ? q.ival[i] : slots_host[q.slot[i]] + q.ival[i];
        const int st = plan_dispatch(q.op, v, q.fval, streams_host[q.stream]);
        if (st != NELE_OK) return st;                      // (the entry point has set the error string)
    }
    return NELE_OK;
}

// The same with the sizes of the caller's buffers: sizes_host[k] = bytes behind slots_host[k] (ignored for scalar slots).
extern "C" int nele_plan_run_sized(void* plan, void* const* streams_host, int nstreams, const long long* slots_host, const long long* sizes_host, int nslots) {
    NelePlan* p = reinterpret_cast<NelePlan*>(plan);
    NELE_CHECK_ARG(p && p->magic == PLAN_MAGIC && sizes_host && slots_host && nslots >= p->nslots, "nele_plan_run_sized: bad arguments");
    for (int k = 0; k < p->nslots; ++k)
        if (p->slot_bytes[k] > 0 && slots_host[k] != 0 && sizes_host[k] < p->slot_bytes[k])
            return nele_set_error(NELE_ERR_INVALID_ARG, "nele_plan_run_sized: slot %d holds %lld bytes, the plan touches %lld", k, sizes_host[k], p->slot_bytes[k]);
    return nele_plan_run(plan, streams_host, nstreams, slots_host, nslots);
}

// ---- the composite entry points: a plan + the per-call pointers in its first slots
extern "C" int nele_gen_fwd(void* plan, const float* x, const float* y, float* mask, unsigned token, void* const* streams_host, int nstreams) {
    const long long s[4] = {(long long)(uintptr_t)x, (long long)(uintptr_t)y, (long long)(uintptr_t)mask, (long long)token};
    return nele_plan_run(plan, streams_host, nstreams, s, 4);
}
extern "C" int nele_gen_bwd(void* plan, const float* dmask, const float* mask, void* const* streams_host, int nstreams) {
    const long long s[2] = {(long long)(uintptr_t)dmask, (long long)(uintptr_t)mask};
    return nele_plan_run(plan, streams_host, nstreams, s, 2);
}
extern "C" int nele_disc_fwd(void* plan, const float* din, const int* wvalid, float* score, void* const* streams_host, int nstreams) {
    const long long s[3] = {(long long)(uintptr_t)din, (long long)(uintptr_t)wvalid, (long long)(uintptr_t)score};
    return nele_plan_run(plan, streams_host, nstreams, s, 3);
}
extern "C" int nele_disc_bwd(void* plan, const float* dscore, const float* score, const int* wvalid, const float* din, void* const* streams_host, int nstreams) {
    const long long s[4] = {(long long)(uintptr_t)dscore, (long long)(uintptr_t)score, (long long)(uintptr_t)wvalid, (long long)(uintptr_t)din};
    return nele_plan_run(plan, streams_host, nstreams, s, 4);
}

// ---- events and the one arithmetic op the host mirror used torch for inside a pass
extern "C" int nele_event_create(void** event_out) {
    NELE_CHECK_ARG(event_out, "nele_event_create: null");
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nele_set_error(NELE_ERR_HIP, "nele_event_create: hipEventCreate failed");
    *event_out = e;
    return NELE_OK;
}
extern "C" int nele_event_destroy(void* event) {
    if (event && hipEventDestroy(reinterpret_cast<hipEvent_t>(event)) != hipSuccess) return nele_set_error(NELE_ERR_HIP, "nele_event_destroy failed");
    return NELE_OK;
}
extern "C" int nele_event_record(void* event, void* stream) {
    NELE_CHECK_ARG(event, "nele_event_record: null event");
    if (hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)) != hipSuccess) return nele_set_error(NELE_ERR_HIP, "nele_event_record failed");
    return NELE_OK;
}
extern "C" int nele_stream_wait_event(void* event, void* stream) {
    NELE_CHECK_ARG(event, "nele_stream_wait_event: null event");
    if (hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0) != hipSuccess) return nele_set_error(NELE_ERR_HIP, "nele_stream_wait_event failed");
    return NELE_OK;
}

__global__ void vec_add_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] += src[i];
}
extern "C" int nele_vec_add(float* dst, const float* src, long long n, void* stream) {
    NELE_CHECK_ARG(dst && src && n > 0, "nele_vec_add: bad arguments");
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(vec_add_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, as_stream(stream), dst, src, n);
    NELE_CHECK_LAUNCH("nele_vec_add");
    return NELE_OK;
}
