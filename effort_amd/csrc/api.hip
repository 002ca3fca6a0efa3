// C ABI of libeffort_hip.so (include/effort_hip.h): contexts, weight handles, launch orchestration.
// Host-side equivalent of class BucketMul / BucketMulQ4 (bucketMul.swift:18-90, bucketMulQ4.swift:18-87)
// and of the slice of class Gpu they use (helpers/gpu.swift:109-196).
#include <dlfcn.h>
#include <rccl/rccl.h>          // types only: the library is opened when the first communicator is asked for (rccl(), below)
#include <rocblas/rocblas.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/effort_hip.h"
#include "../../include/effort_hip_debug.h"
#include "effort_internal.h"

using namespace effort;

static constexpr size_t kStampBytes = (size_t)(kTraceOff + kTraceItems * 12) * 8;   // phase stamps + per-item trace records

// What ONE multiply launch in flight owns: the reference's singleton scratch (BucketMul.shared: cutoff, dispatch counter,
// partial tiles) plus the launch machinery's (arrival tickets, per-slice counts, item queues).  A context has one such lane,
// or -- effort_set_overlap -- up to kMaxLanes of them, each with an internal stream: independent launches then overlap.
struct Lane {
    hipStream_t own = nullptr;        // internal stream (overlap mode); with one lane the launches go to the context's stream
    hipEvent_t done = nullptr;        // "everything enqueued on the lane so far": recorded LAZILY, when somebody is about to wait for the lane (lane_mark)
    bool dirty = false;               // launches were enqueued on `own` since `done` was last recorded
    unsigned long long capId = 0;     // the hipGraph capture the lane's pending launches were enqueued in (0: none -- real work on the queue)
    float* d_cutoff = nullptr;        // BucketMul.cutoff
    uint32_t* d_count = nullptr;      // dispatch.size
    float* d_slabs = nullptr;         // partial tiles (replaces tmpMulVec)
    uint32_t* d_counters = nullptr;   // per-tile arrival tickets (zero between calls)
    uint32_t* d_sliceCounts = nullptr;
    uint32_t* d_queue = nullptr;      // item queues of persistent launches
    // where each call of the lane's last (group) launch keeps its per-slice counts; slices == 0: dispatch.size is d_count
    uint32_t lastCalls = 1, lastSliceOff[effort::kMaxGroup] = {0}, lastSlices[effort::kMaxGroup] = {0};
    // address ranges the launches enqueued since the last join read / write (hazard check of the next launch)
    struct Range { uintptr_t lo, hi; };
    std::vector<Range> reads, writes;
    bool pending = false;
};

struct effort_ctx {
    static constexpr int kMaxLanes = 4;
    int device = 0;
    hipStream_t stream = nullptr;
    int numCU = 256;
    Lane lane[kMaxLanes];
    int nLanes = 1, lastLane = 0, nextLane = 0;
    int busyStreak = 0;               // overlap mode: consecutive launches of a dependent chain that found the context's stream "busy" (see the lane choice)
    bool migrated = false;            // ... and the chain was moved to another lane for it (once between joins)
    hipEvent_t forkEv = nullptr;      // overlap mode: "everything enqueued on the context's stream so far"
    size_t slabBytes = 0;
    uint32_t* d_blockScratch = nullptr;
    uint16_t* d_vhalf = nullptr;      // v.asFloat16() for the dense baseline
    size_t vhalfElems = 0;
    float* d_cos = nullptr;
    uint16_t* d_convVals = nullptr;   // converter scratch (transposed matrix)
    size_t convElems = 0;
    int* d_status = nullptr;
    int persistent = -1;              // workgroups per CU of group launches: -1 heuristic, 0 plain grid, R > 0 persistent
    static constexpr uint32_t kMaxTiles = 1024, kMaxSlices = 4096;
    static constexpr size_t kMaxRanges = 1024;   // address ranges a lane records between joins (do_group)
    unsigned long long* d_tstamp = nullptr;   // device-clock stamps of the multiply kernel (timing mode)
    double wallClockKHz = 100000.0;
    rocblas_handle blas = nullptr;
    ncclComm_t comm = nullptr;        // effort_comm_create: this rank's RCCL communicator (one process per GPU)
    int commRank = 0, commWorld = 1;
    bool denseRocblas = false;        // effort_set_dense_backend: basicMul through rocBLAS instead of dense_gemv_kernel
    // tuning overrides (0 = heuristic)
    int tuneW = 0, tuneE = 0, tuneS = 0;
    bool splitCutoff = false;     // run findCutoff32 as its own 1-workgroup kernel instead of inside every workgroup
    mutable bool thinEffort = false;   // the launch being cut streams next to nothing (mean effort under 8 %): the rules that RAISE the slice count of a group stand down
                                       // (do_group sets it before the geometry is chosen; at 2 % effort 3 / 6 / 8 calls of 4096x11008 at 13 / 13 / 10 slices are 9 / 14 / 10 % SLOWER
                                       // than at 8 -- more slabs and heads, nothing to stream -- where at 10 % they are 6 / 5 / 0 % faster and at 100 % 28 / 28 / 6 %)
    bool rowReuse = false;        // effort_set_row_reuse: the bucket-row stream with the ordinary cache policy instead of nt (GroupKArgs::split bit 3)
    // optional per-kernel timing
    bool timing = false;          // HIP events around each kernel
    bool clock = false;           // device wall-clock stamps inside the multiply kernel
    bool trace = false;           // ... plus one record per work item (effort_debug_trace)
    static constexpr int kMaxSamples = 4096;
    hipEvent_t* ev = nullptr;         // 4 events per sample
    int nSamples = 0;
    char err[256] = {0};
};

struct effort_w {
    effort_ctx* ctx = nullptr;
    Format fmt = kFp16;
    const uint16_t* buckets = nullptr;    // what the multiply reads: the caller's buffer, or `aligned`
    const uint16_t* bucketsSrc = nullptr; // the caller's buffer (borrowed)
    uint16_t* aligned = nullptr;          // own copy with rows padded to whole 128-byte lines (effort_weights_align_rows)
    uint32_t rowPitch = 0;                // bytes between rows of `buckets`
    uint32_t srcPitch = 0;                // ... of `bucketsSrc` (2*cols as the reference lays them out, or what effort_weights_fp16_pitched was given)
    const void* stats = nullptr;
    const uint16_t* probes = nullptr;
    uint32_t inDim = 0, outDim = 0, rowsPerIn = 0, numExperts = 1, cols = 0;
    bool view = false;                // a column shard (effort_weights_column_shard): buckets, row means and the outlier index point INTO the full handle's
    effort_w* parent = nullptr;       // ... that handle; it stays alive (its buffers, that is) until its last view is freed
    uint32_t viewTrim = 0;            // ... bytes `buckets` lies behind the full handle's: the multiply's buffer descriptor ends at the END of the full allocation
    int views = 0;                    // live views of this (full) handle
    bool dead = false;                // effort_weights_free was called while views were alive: freed with the last of them
    float* rankBound = nullptr;       // [numExperts] fixed-point bound of the multiply (see launch_rank_bound)
    uint16_t* means16 = nullptr;      // FP16: the row means alone (stats lane .w), one u16 per bucket row (launch_compact_means)
    // Q4 outliers
    uint64_t nOutliers = 0;
    uint32_t* olBlockPtr = nullptr;   // entry bounds per block of 64 outputs (a column shard's: a slice of the full handle's)
    uint32_t* olEntry = nullptr;      // 4 bytes per outlier, jagged-diagonal order inside a block (dispatch.hip)
    uint32_t* olMeta = nullptr;       // per (block, rank): entry count << 8 | output in block
};

static int fail(effort_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
    if (c) {
        if (e != hipSuccess) snprintf(c->err, sizeof(c->err), "%s: %s", what, hipGetErrorString(e));
        else snprintf(c->err, sizeof(c->err), "%s", what);
    }
    return code;
}
#define HIP_TRY(ctx, call)                                                   \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return fail((ctx), EFFORT_ERR_HIP, #call, e_); \
    } while (0)

// RCCL is bound at RUN time, when the first communicator is asked for: a single-GPU host needs no librccl to load this library,
// and a process that has one mapped already (PyTorch-ROCm brings its own copy) gets THAT one -- dlopen by soname returns the copy
// already in the process -- so two RCCLs never meet in one address space; a plain C / Swift host gets the system ROCm's.
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
static const RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) return a;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.lib, "ncclAllGather"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.GetErrorString;
        return a;
    }();
    return api;
}

extern "C" const char* effort_version(void) { return "effort-hip 0.1 (gfx950)"; }
static char g_createErr[256] = "null context";       // effort_last_error(NULL): why the last effort_create failed, if one did
extern "C" const char* effort_last_error(effort_ctx* c) { return c ? c->err : g_createErr; }

static bool lane_alloc(effort_ctx* c, Lane& L) {
    bool ok = hipMalloc(&L.d_cutoff, 512) == hipSuccess && hipMalloc(&L.d_count, 16) == hipSuccess &&
              hipMalloc(&L.d_slabs, c->slabBytes) == hipSuccess && hipMalloc(&L.d_counters, effort_ctx::kMaxTiles * 4) == hipSuccess &&
              hipMalloc(&L.d_sliceCounts, effort_ctx::kMaxSlices * 4) == hipSuccess && hipMalloc(&L.d_queue, kQueueWords * 4) == hipSuccess;
    if (!ok) return false;
    hipMemset(L.d_counters, 0, effort_ctx::kMaxTiles * 4);
    hipMemset(L.d_sliceCounts, 0, effort_ctx::kMaxSlices * 4);
    hipMemset(L.d_queue, 0, kQueueWords * 4);
    hipMemset(L.d_cutoff, 0, 512);
    hipMemset(L.d_count, 0, 16);
    return true;
}
// The lanes' streams and events are POOLED per process and never destroyed: a stream or event that took part in a hipGraph
// capture and is destroyed while other captured graphs are alive crashes a later hipGraphLaunch inside the runtime (ROCm 7.2:
// segfault in hipGraphLaunch after a multi-lane context was destroyed; tools/lab/lane_crash.py reproduces it).  A context returns
// them to the pool; the next one takes them from there.
// The pools are keyed by DEVICE: a stream or event belongs to the device that was current when it was created, and a context
// of another device must never be handed one (its launches would go to the wrong GPU).  The caller has made `device` current.
static std::mutex g_poolMutex;
static std::map<int, std::vector<hipStream_t>> g_streamPool;
static std::map<int, std::vector<hipEvent_t>> g_eventPool;
static hipStream_t pool_stream(int device) {
    std::lock_guard<std::mutex> lk(g_poolMutex);
    auto& pool = g_streamPool[device];
    if (!pool.empty()) { hipStream_t s = pool.back(); pool.pop_back(); return s; }
    hipStream_t s = nullptr;
    return hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess ? s : nullptr;
}
static hipEvent_t pool_event(int device) {
    std::lock_guard<std::mutex> lk(g_poolMutex);
    auto& pool = g_eventPool[device];
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess ? e : nullptr;
}
static void pool_put(int device, hipStream_t s) { if (s) { std::lock_guard<std::mutex> lk(g_poolMutex); g_streamPool[device].push_back(s); } }
static void pool_put(int device, hipEvent_t e) { if (e) { std::lock_guard<std::mutex> lk(g_poolMutex); g_eventPool[device].push_back(e); } }

static void lane_free(int device, Lane& L) {
    pool_put(device, L.own);
    pool_put(device, L.done);
    hipFree(L.d_cutoff); hipFree(L.d_count); hipFree(L.d_slabs); hipFree(L.d_counters); hipFree(L.d_sliceCounts); hipFree(L.d_queue);
    L = Lane();
}

extern "C" effort_ctx* effort_create(int device, void* stream) {
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    effort_ctx* c = new (std::nothrow) effort_ctx();
    if (!c) return nullptr;
    c->device = device;
    c->stream = reinterpret_cast<hipStream_t>(stream);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->numCU = prop.multiProcessorCount;
    c->slabBytes = (size_t)64 << 20;
    {   // function attributes belong to a device: set them when the device's first context is created
        static std::mutex m; static std::map<int, bool> prepared;
        std::lock_guard<std::mutex> lk(m);
        if (!prepared[device]) {
            const hipError_t pe = bucket_mul_prepare_device();
            if (pe != hipSuccess) { snprintf(g_createErr, sizeof(g_createErr), "effort_create: kernel attributes: %s", hipGetErrorString(pe)); delete c; return nullptr; }
            prepared[device] = true;
        }
    }
    bool ok = lane_alloc(c, c->lane[0]) && hipMalloc(&c->d_blockScratch, 4096 * 4) == hipSuccess &&
              hipMalloc(&c->d_cos, 16) == hipSuccess && hipMalloc(&c->d_status, 16) == hipSuccess &&
              hipMalloc(&c->d_tstamp, kStampBytes) == hipSuccess;
    if (!ok) { snprintf(g_createErr, sizeof(g_createErr), "effort_create: out of device memory for the context scratch"); effort_destroy(c); return nullptr; }
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) c->wallClockKHz = khz;
    hipMemset(c->d_tstamp, 0, kStampBytes);
    { unsigned long long init[2] = {~0ull, 0ull}; hipMemcpy(c->d_tstamp, init, 16, hipMemcpyHostToDevice); }
    hipMemset(c->d_status, 0, 16);
    return c;
}

// ---- lanes: independent launches of one context in flight together (effort_set_overlap) -------------------------
// The stream a launch of lane i goes to.
static hipStream_t lane_stream(effort_ctx* c, int i) { return c->nLanes > 1 ? c->lane[i].own : c->stream; }
// The lane's `done` event, brought up to date.  An event per launch made a chain of dependent lone calls -- which the hazard
// analysis keeps on ONE lane, in order -- pay an event packet between every two kernels (the reference's own timing loop,
// benchmarks/benchmark.swift:245-257: 30.1 us per call with lanes against 22.8 without, after the fork was already skipped on an
// idle stream); recorded only where something waits for the lane -- a join, or a launch on another lane that depends on it -- the
// chain is back-to-back kernels on the lane's stream and nothing else.
static hipError_t lane_mark(Lane& L) {
    if (!L.dirty) return hipSuccess;
    const hipError_t e = hipEventRecord(L.done, L.own);
    if (e == hipSuccess) L.dirty = false;
    return e;
}
// The capture the context's stream is in (0: none).
static unsigned long long capture_of(hipStream_t st) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(st, &cap, &id) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return cap == hipStreamCaptureStatusActive ? (id ? id : ~0ull) : 0ull;
}
// Lanes and hipGraph captures.  A lane's pending launches are either real work on its queue or nodes of ONE capture, the one its fork
// edge came from (Lane::capId).  Two mixtures cannot be ordered and are refused / cleaned up here instead of surfacing as a HIP error in the
// middle of a launch: (a) real work pending when the context's stream is capturing -- the capture would have to wait on an event recorded
// outside it, which stream capture forbids: the caller joins BEFORE beginning the capture (returns false, c->err says so); (b) nodes of a
// capture that has ended (or of another one): the graph's own edges order them; the bookkeeping is dropped.
static bool lanes_match_capture(effort_ctx* c, unsigned long long now) {
    for (int i = 0; i < c->nLanes; i++) {
        Lane& L = c->lane[i];
        if (!L.pending || L.capId == now) continue;
        if (L.capId == 0) {
            snprintf(c->err, sizeof(c->err), "lanes hold launches enqueued BEFORE this hipGraph capture began: call effort_join (or effort_sync) before beginning the capture");
            return false;
        }
        L.pending = false; L.dirty = false; L.capId = 0; L.reads.clear(); L.writes.clear();
    }
    return true;
}
// The context's stream waits for every lane's launches; from here on the context's stream order covers them.
static int join_lanes(effort_ctx* c) {
    if (c->nLanes <= 1) return EFFORT_OK;
    bool any = false;
    for (int i = 0; i < c->nLanes; i++) any = any || c->lane[i].pending;
    if (any && !lanes_match_capture(c, capture_of(c->stream))) return EFFORT_ERR_ARG;
    for (int i = 0; i < c->nLanes; i++) {
        Lane& L = c->lane[i];
        if (!L.pending) continue;
        hipError_t e = lane_mark(L);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, L.done, 0);
        if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "join: %s", hipGetErrorString(e)); return EFFORT_ERR_HIP; }
        L.pending = false; L.capId = 0; L.reads.clear(); L.writes.clear();
    }
    c->busyStreak = 0; c->migrated = false;
    return EFFORT_OK;
}
static bool overlaps(const std::vector<Lane::Range>& a, const std::vector<Lane::Range>& b) {
    for (const auto& x : a) for (const auto& y : b) if (x.lo < y.hi && y.lo < x.hi) return true;
    return false;
}

extern "C" int effort_set_overlap(effort_ctx* c, int lanes) {
    if (!c || lanes < 1 || lanes > effort_ctx::kMaxLanes) return EFFORT_ERR_ARG;
    hipSetDevice(c->device);
    int rc = join_lanes(c);
    if (rc != EFFORT_OK) return rc;
    if (lanes > 1) {
        if (!c->forkEv && !(c->forkEv = pool_event(c->device))) return fail(c, EFFORT_ERR_HIP, "set_overlap: event");
        for (int i = 0; i < lanes; i++) {
            Lane& L = c->lane[i];
            if (!L.d_slabs && !lane_alloc(c, L)) return fail(c, EFFORT_ERR_HIP, "set_overlap: out of device memory for a lane's scratch");
            if (!L.own && !(L.own = pool_stream(c->device))) return fail(c, EFFORT_ERR_HIP, "set_overlap: stream");
            if (!L.done && !(L.done = pool_event(c->device))) return fail(c, EFFORT_ERR_HIP, "set_overlap: event");
        }
    }
    c->nLanes = lanes; c->lastLane = 0; c->nextLane = 0;
    return EFFORT_OK;
}
extern "C" int effort_join(effort_ctx* c) {
    if (!c) return EFFORT_ERR_ARG;
    return join_lanes(c);
}

extern "C" void effort_destroy(effort_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    for (int i = 0; i < effort_ctx::kMaxLanes; i++) if (c->lane[i].own) hipStreamSynchronize(c->lane[i].own);
    hipStreamSynchronize(c->stream);
    if (c->comm) rccl().CommDestroy(c->comm);
    if (c->blas) rocblas_destroy_handle(c->blas);
    if (c->ev) { for (int i = 0; i < effort_ctx::kMaxSamples * 4; i++) if (c->ev[i]) hipEventDestroy(c->ev[i]); delete[] c->ev; }
    for (int i = 0; i < effort_ctx::kMaxLanes; i++) lane_free(c->device, c->lane[i]);
    pool_put(c->device, c->forkEv);
    hipFree(c->d_blockScratch); hipFree(c->d_vhalf); hipFree(c->d_cos); hipFree(c->d_convVals); hipFree(c->d_status); hipFree(c->d_tstamp);
    delete c;
}

extern "C" int effort_set_stream(effort_ctx* c, void* stream) {
    if (!c) return EFFORT_ERR_ARG;
    if (reinterpret_cast<hipStream_t>(stream) != c->stream && join_lanes(c) != EFFORT_OK) return EFFORT_ERR_HIP;   // (the old stream carries the join)
    c->stream = reinterpret_cast<hipStream_t>(stream);
    if (c->blas) rocblas_set_stream(c->blas, c->stream);
    return EFFORT_OK;
}

extern "C" int effort_sync(effort_ctx* c) {
    if (!c) return EFFORT_ERR_ARG;
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EFFORT_OK;
}

// ---- weights ------------------------------------------------------------------------------------
static int check_shape(uint32_t inDim, uint32_t outDim) {
    // bucketMul.swift:73 (outDim % 16), :52 tmpMulVec 16384 wide; probes need 4096 inputs (:36).  The reference also asserts
    // (outDim/16) % 4 == 0 (:76: its kernel reads four columns per thread); this kernel masks a ragged last tile, so a
    // handle -- in particular a column shard of a multi-GPU split, 11008/16/8 = 86 columns -- may have any column count.
    // (an EVEN count: the row pieces are dword / dwordx2 loads, so rows must stay 4-byte aligned -- outDim % 32 == 0)
    if (inDim < (uint32_t)kProbes || outDim == 0 || outDim % 32 || outDim > 16384) return EFFORT_ERR_SHAPE;
    return EFFORT_OK;
}

// The multiply accumulates in fixed point; its scale needs a bound on the weights: per expert, the sum over ranks
// of the largest |w| of that rank (Q4: of the largest row mean).  One pass over the buckets at registration.
static int register_bound(effort_ctx* c, effort_w* w) {
    // (Re)computed IN PLACE: launches copy these pointers by value, and a captured hipGraph bakes them in, so a refresh must
    // never move them.
    hipSetDevice(c->device);
    const size_t rows = (size_t)w->numExperts * w->rowsPerIn * w->inDim;
    float* scratch = nullptr;
    bool ok = (w->rankBound || hipMalloc(&w->rankBound, (size_t)w->numExperts * 4) == hipSuccess) && hipMalloc(&scratch, rows * 4) == hipSuccess;
    if (ok) ok = launch_rank_bound(w->fmt, w->bucketsSrc, w->srcPitch / 2u, w->stats, w->numExperts, w->rowsPerIn, w->inDim, w->cols, scratch, w->rankBound, c->stream) == hipSuccess;
    if (ok && w->fmt == kFp16) {
        ok = w->means16 || hipMalloc(&w->means16, rows * 2) == hipSuccess;
        if (ok) ok = launch_compact_means(w->stats, w->means16, (uint32_t)rows, c->stream) == hipSuccess;
    }
    if (ok) ok = hipStreamSynchronize(c->stream) == hipSuccess;
    hipFree(scratch);
    return ok ? EFFORT_OK : fail(c, EFFORT_ERR_HIP, "weight registration: rank bound");
}

extern "C" int effort_aligned_row_pitch(int outDim) {
    if (outDim <= 0 || outDim % 32 || outDim > 16384) return EFFORT_ERR_SHAPE;      // (what registration accepts: check_shape)
    return (outDim / 16 * 2 + 127) / 128 * 128;
}

extern "C" effort_w* effort_weights_fp16_pitched(effort_ctx* c, const void* buckets, int rowPitchBytes, const void* stats, const void* probes,
                                                 int inDim, int outDim, int percentLoad, int numExperts) {
    if (!c || !buckets || !stats || !probes) { fail(c, EFFORT_ERR_ARG, "effort_weights_fp16: null argument"); return nullptr; }
    const int pitch = rowPitchBytes ? rowPitchBytes : (outDim > 0 ? outDim / 16 * 2 : 0);
    if (inDim <= 0 || outDim <= 0 || percentLoad < 1 || percentLoad > 16 || numExperts < 1 || check_shape(inDim, outDim) != EFFORT_OK ||
        inDim > 65535 || pitch < outDim / 16 * 2 || pitch % 4 || (size_t)numExperts * percentLoad * inDim * pitch > 0xFFFFFFFFull) {
        fail(c, EFFORT_ERR_SHAPE, "effort_weights_fp16: unsupported shape or row pitch (a multiple of 4 bytes >= 2*cols; buckets < 4 GiB)"); return nullptr; }
    effort_w* w = new (std::nothrow) effort_w();
    if (!w) return nullptr;
    w->ctx = c; w->fmt = kFp16;
    w->buckets = static_cast<const uint16_t*>(buckets); w->stats = stats; w->probes = static_cast<const uint16_t*>(probes);
    w->inDim = inDim; w->outDim = outDim; w->rowsPerIn = percentLoad; w->numExperts = numExperts; w->cols = outDim / 16;
    w->bucketsSrc = w->buckets; w->rowPitch = w->srcPitch = (uint32_t)pitch;
    if (register_bound(c, w) != EFFORT_OK) { effort_weights_free(w); return nullptr; }
    return w;
}
extern "C" effort_w* effort_weights_fp16(effort_ctx* c, const void* buckets, const void* stats, const void* probes,
                                         int inDim, int outDim, int percentLoad, int numExperts) {
    return effort_weights_fp16_pitched(c, buckets, 0, stats, probes, inDim, outDim, percentLoad, numExperts);
}

extern "C" effort_w* effort_weights_q4(effort_ctx* c, const void* buckets, const void* stats, const void* probes,
                                       const void* outliers, int64_t nOutliers, int inDim, int outDim, int numExperts) {
    if (!c || !buckets || !stats || !probes) { fail(c, EFFORT_ERR_ARG, "effort_weights_q4: null argument"); return nullptr; }
    if (inDim <= 0 || outDim <= 0 || numExperts < 1 || check_shape(inDim, outDim) != EFFORT_OK || outDim % 32 || inDim > 65535 ||
        nOutliers < 0 || (size_t)numExperts * 8 * inDim * (outDim / 32) * 2 > 0xFFFFFFFFull) {
        fail(c, EFFORT_ERR_SHAPE, "effort_weights_q4: unsupported shape (or buckets >= 4 GiB)"); return nullptr; }
    effort_w* w = new (std::nothrow) effort_w();
    if (!w) return nullptr;
    w->ctx = c; w->fmt = kQ4;
    w->buckets = static_cast<const uint16_t*>(buckets); w->stats = stats; w->probes = static_cast<const uint16_t*>(probes);
    w->inDim = inDim; w->outDim = outDim; w->rowsPerIn = 8; w->numExperts = numExperts; w->cols = outDim / 32;
    w->bucketsSrc = w->buckets; w->rowPitch = w->srcPitch = w->cols * 2u;
    if (register_bound(c, w) != EFFORT_OK) { effort_weights_free(w); return nullptr; }
    if (outliers && nOutliers > 0) {
        if (outDim > 65536) { fail(c, EFFORT_ERR_SHAPE, "effort_weights_q4: outliers need outDim <= 65536"); effort_weights_free(w); return nullptr; }
        hipSetDevice(c->device);
        {   // every entry must name an element of THIS matrix (the index is built with unchecked scatters) and carry an f16 value
            int bad[2] = {0, 0};
            bool okv = hipMemsetAsync(c->d_status + 2, 0, 8, c->stream) == hipSuccess &&
                       launch_validate_outliers(static_cast<const float*>(outliers), (uint64_t)nOutliers, (uint32_t)inDim, (uint32_t)outDim, c->d_status + 2, c->stream) == hipSuccess &&
                       hipMemcpyAsync(bad, c->d_status + 2, 8, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
            if (!okv || bad[0] || bad[1]) {
                fail(c, okv ? EFFORT_ERR_ARG : EFFORT_ERR_HIP, !okv ? "effort_weights_q4: outlier validation" :
                     bad[0] ? "effort_weights_q4: outlier entries outside the matrix (inIdx >= inDim or outIdx >= outDim)"
                            : "effort_weights_q4: outlier values that are not f16 numbers (the table comes from an f16 matrix, q4_draft.py:58-67)");
                effort_weights_free(w); return nullptr;
            }
        }
        uint32_t *tmp = nullptr, *rowPtr = nullptr;
        w->nOutliers = (uint64_t)nOutliers;
        const uint32_t olBlocks = ((uint32_t)outDim + 63u) / 64u;
        bool ok = hipMalloc(&rowPtr, ((size_t)outDim + 2) * 4) == hipSuccess && hipMalloc(&w->olBlockPtr, ((size_t)olBlocks + 1) * 4) == hipSuccess &&
                  hipMalloc(&w->olMeta, (size_t)olBlocks * 64 * 4) == hipSuccess &&
                  hipMalloc(&w->olEntry, ((size_t)nOutliers + 64) * 4) == hipSuccess && hipMalloc(&tmp, 4 * (size_t)nOutliers * 4) == hipSuccess;
        if (ok) ok = hipMemsetAsync(w->olEntry + (size_t)nOutliers, 0, 64 * 4, c->stream) == hipSuccess;     // padding: a step's load reads 64 entries from its cursor
        if (ok) ok = launch_build_outlier_index(static_cast<const float*>(outliers), w->nOutliers, (uint32_t)inDim, (uint32_t)outDim, rowPtr, w->olBlockPtr,
                                                w->olEntry, w->olMeta, tmp, c->stream) == hipSuccess;
        if (ok) ok = hipStreamSynchronize(c->stream) == hipSuccess;
        hipFree(tmp);
        uint32_t longest = 0;
        if (ok) ok = hipMemcpy(&longest, rowPtr + (size_t)outDim + 1, 4, hipMemcpyDeviceToHost) == hipSuccess;
        hipFree(rowPtr);
        if (!ok) { fail(c, EFFORT_ERR_HIP, "effort_weights_q4: outlier index"); effort_weights_free(w); return nullptr; }
        if (longest >= (1u << 24)) { fail(c, EFFORT_ERR_SHAPE, "effort_weights_q4: more than 2^24 outliers on one output"); effort_weights_free(w); return nullptr; }
    }
    return w;
}

// The fixed-point scale of the multiply comes from rankBound, a snapshot of the weights taken at registration: weights
// rewritten in place afterwards need effort_weights_refresh (the reference's loader.swift buffers are mutable).
static int copy_aligned(effort_w* w) {
    const size_t rows = (size_t)w->numExperts * w->rowsPerIn * w->inDim;
    HIP_TRY(w->ctx, hipMemcpy2DAsync(w->aligned, w->rowPitch, w->bucketsSrc, w->srcPitch, (size_t)w->cols * 2, rows, hipMemcpyDeviceToDevice, w->ctx->stream));
    HIP_TRY(w->ctx, hipStreamSynchronize(w->ctx->stream));
    return EFFORT_OK;
}
extern "C" int effort_weights_refresh(effort_w* w) {
    if (!w || !w->ctx || w->dead) return EFFORT_ERR_ARG;
    if (w->view) return fail(w->ctx, EFFORT_ERR_ARG, "effort_weights_refresh: a column shard takes its bound from the full handle -- refresh that one and shard again");
    hipSetDevice(w->ctx->device);
    int rc = join_lanes(w->ctx);                 // multiplies still in flight on the lanes read what is recomputed here
    if (rc == EFFORT_OK) rc = register_bound(w->ctx, w);          // in place: graphs captured earlier keep valid pointers
    if (rc == EFFORT_OK && w->aligned) rc = copy_aligned(w);
    return rc;
}
// The converter's rows are 2*cols bytes apart (1376 for 11008 outputs): a 512-byte piece of a row then straddles 128-byte
// lines, the three column tiles of a row fetch 14 lines where 11 would do, and the stream runs at 5.4-5.6 TB/s instead of
// the 6.1-6.9 TB/s it reaches on line-aligned rows (measured: 4096x11264 / 12288 / 8192).  This call gives the handle its
// OWN copy of the buckets with every row starting on a 128-byte line (pitch = 2*cols rounded up to 128); the multiply
// reads the copy, and the caller's buffer is no longer read (it may be freed; effort_weights_refresh re-reads it, so
// keep it if the weights are going to change).  No-op when the pitch is line-aligned already.  Results are bit-identical.
extern "C" int effort_weights_align_rows(effort_w* w) {
    if (!w || !w->ctx || w->dead) return EFFORT_ERR_ARG;
    if (w->aligned || w->rowPitch % 128u == 0u) return EFFORT_OK;        // (already on whole lines: as converted with effort_convert_fp16_pitched, or 2*cols % 128 == 0)
    hipSetDevice(w->ctx->device);
    const uint32_t pitch = (w->cols * 2u + 127u) / 128u * 128u;
    const size_t rows = (size_t)w->numExperts * w->rowsPerIn * w->inDim;
    if (rows * pitch > 0xFFFFFFFFull) return fail(w->ctx, EFFORT_ERR_SHAPE, "effort_weights_align_rows: the padded buckets would reach 4 GiB");
    if (hipMalloc(&w->aligned, rows * pitch) != hipSuccess) return fail(w->ctx, EFFORT_ERR_HIP, "effort_weights_align_rows: out of device memory");
    HIP_TRY(w->ctx, hipMemsetAsync(w->aligned, 0, rows * pitch, w->ctx->stream));
    w->rowPitch = pitch;
    const int rc = copy_aligned(w);
    if (rc != EFFORT_OK) { hipFree(w->aligned); w->aligned = nullptr; w->rowPitch = w->srcPitch; return rc; }
    w->buckets = w->aligned;
    return EFFORT_OK;
}
extern "C" int effort_weights_row_pitch(const effort_w* w) { return w ? (int)w->rowPitch : EFFORT_ERR_ARG; }
extern "C" int effort_weights_get_bound(effort_w* w, float* host_out) {
    if (!w || !w->ctx || !host_out || !w->rankBound) return EFFORT_ERR_ARG;
    HIP_TRY(w->ctx, hipMemcpy(host_out, w->rankBound, (size_t)w->numExperts * 4, hipMemcpyDeviceToHost));
    return EFFORT_OK;
}
extern "C" int effort_weights_set_bound(effort_w* w, const float* host_in) {
    if (!w || !w->ctx || !host_in || !w->rankBound || w->dead) return EFFORT_ERR_ARG;
    for (uint32_t e = 0; e < w->numExperts; e++) if (!(host_in[e] >= 0.0f)) return EFFORT_ERR_ARG;
    { const int jrc = join_lanes(w->ctx); if (jrc != EFFORT_OK) return jrc; }
    HIP_TRY(w->ctx, hipStreamSynchronize(w->ctx->stream));
    HIP_TRY(w->ctx, hipMemcpy(w->rankBound, host_in, (size_t)w->numExperts * 4, hipMemcpyHostToDevice));
    return EFFORT_OK;
}

// A full handle's `views` / `dead` are touched from whichever threads shard and free (one context per rank-thread is a pattern the
// multi-GPU path invites): one process-wide mutex orders those few words.
static std::mutex g_viewMutex;
extern "C" void effort_weights_free(effort_w* w) {
    if (!w) return;
    if (w->view) {                               // a column shard owns its bound only; the last view of a freed parent takes the parent along
        effort_w* const par = w->parent;
        hipFree(w->rankBound);
        delete w;
        bool last = false;
        if (par) { std::lock_guard<std::mutex> lk(g_viewMutex); last = --par->views == 0 && par->dead; if (last) par->dead = false; }
        if (last) effort_weights_free(par);
        return;
    }
    {   // views still read these buffers (the header says: free the shards first -- but do not dangle if not)
        std::lock_guard<std::mutex> lk(g_viewMutex);
        if (w->views > 0) { w->dead = true; return; }
    }
    hipFree(w->olBlockPtr); hipFree(w->olEntry); hipFree(w->olMeta);
    hipFree(w->aligned); hipFree(w->rankBound); hipFree(w->means16);
    delete w;
}

// ---- multi-GPU: one process per GPU, RCCL over xGMI (the reference is single-device, helpers/gpu.swift:36-38) --------------
// Column shard of a registered bundle for rank `rank` of `world`: bucket columns [rank*C/world, (rank+1)*C/world) of every bucket
// row -- outputs [rank*outDim/world, ...) -- as a VIEW of the full handle's buffers (rows stay rowPitch bytes apart: no copy),
// stats and probes shared (they are row-global, convert.metal:105-119: every rank computes the same cutoff and selects the same
// rows), the full matrix's fixed-point bound (every rank rounds on the same grid), and for Q4 the slice of the outlier index that
// falls on these outputs (the index is grouped by blocks of outputs: a view too).
extern "C" effort_w* effort_weights_column_shard(const effort_w* full, int rank, int world) {
    if (!full || !full->ctx) return nullptr;
    effort_ctx* c = full->ctx;
    const uint32_t unit = full->fmt == kFp16 ? 16u : 32u;
    if (world < 1 || rank < 0 || rank >= world || full->cols % (uint32_t)world || (full->cols / (uint32_t)world) % 2u) {
        fail(c, EFFORT_ERR_SHAPE, "effort_weights_column_shard: the bucket columns must split evenly into an even number per rank"); return nullptr; }
    const uint32_t per = full->cols / (uint32_t)world, outDim = per * unit;
    if (check_shape(full->inDim, outDim) != EFFORT_OK) { fail(c, EFFORT_ERR_SHAPE, "effort_weights_column_shard: shard shape"); return nullptr; }
    if (full->nOutliers && outDim % 64u) {
        fail(c, EFFORT_ERR_SHAPE, "effort_weights_column_shard: the shard must hold whole blocks of the outlier index"); return nullptr; }
    effort_w* w = new (std::nothrow) effort_w();
    if (!w) return nullptr;
    {   // (`dead` and `views` under the mutex: a concurrent effort_weights_free(full) either sees this view or is seen here)
        std::lock_guard<std::mutex> lk(g_viewMutex);
        if (full->view || full->dead) { fail(c, EFFORT_ERR_ARG, "effort_weights_column_shard: shard a full, live handle"); delete w; return nullptr; }
        const_cast<effort_w*>(full)->views++;
    }
    *w = *full;
    w->view = true; w->parent = const_cast<effort_w*>(full); w->views = 0; w->dead = false;
    w->aligned = nullptr; w->rankBound = nullptr;
    w->buckets = full->buckets + (size_t)rank * per;            // what the multiply reads (the full handle's own line-aligned copy, if it made one)
    w->viewTrim = (uint32_t)((size_t)rank * per * 2u);          // < rowPitch <= 2048: the descriptor stops where the full buffer does
    w->bucketsSrc = w->buckets; w->srcPitch = full->rowPitch;
    w->cols = per; w->outDim = outDim;
    // the row means are row-global: the view reads the full handle's compact copy (means16 stays the parent's); the fixed-point bound is
    // the FULL matrix's -- every rank rounds on the same grid -- in a word of the view's own (effort_weights_set_bound may change it)
    hipSetDevice(c->device);
    if (hipMalloc(&w->rankBound, (size_t)w->numExperts * 4) != hipSuccess ||
        hipMemcpy(w->rankBound, full->rankBound, (size_t)w->numExperts * 4, hipMemcpyDeviceToDevice) != hipSuccess) {
        fail(c, EFFORT_ERR_HIP, "effort_weights_column_shard: bound"); hipFree(w->rankBound);
        effort_w* const par = w->parent;
        delete w;
        bool last;
        { std::lock_guard<std::mutex> lk(g_viewMutex); last = --par->views == 0 && par->dead; if (last) par->dead = false; }
        if (last) effort_weights_free(par);
        return nullptr; }
    if (full->nOutliers) {
        w->olBlockPtr = full->olBlockPtr + (size_t)rank * outDim / 64u;                       // block bounds index the SHARED entry array
        w->olMeta = full->olMeta + (size_t)rank * outDim;                                     // (64 meta words per block of 64 outputs)
    }
    return w;
}
extern "C" int effort_comm_unique_id(void* id_out) {
    if (!id_out) return EFFORT_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) == EFFORT_COMM_ID_BYTES, "effort_hip.h: EFFORT_COMM_ID_BYTES");
    if (!rccl().ok) return EFFORT_ERR_COMM;
    return rccl().GetUniqueId(static_cast<ncclUniqueId*>(id_out)) == ncclSuccess ? EFFORT_OK : EFFORT_ERR_COMM;
}
extern "C" int effort_comm_create(effort_ctx* c, int rank, int world, const void* id) {
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return fail(c, EFFORT_ERR_ARG, "comm_create: bad argument");
    if (c->comm) return fail(c, EFFORT_ERR_ARG, "comm_create: the context has a communicator already");
    hipSetDevice(c->device);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    if (!rccl().ok) return fail(c, EFFORT_ERR_COMM, "comm_create: librccl.so could not be opened (dlopen) -- multi-GPU needs RCCL");
    const ncclResult_t r = rccl().CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) { c->comm = nullptr; snprintf(c->err, sizeof(c->err), "ncclCommInitRank: %s", rccl().GetErrorString(r)); return EFFORT_ERR_COMM; }
    c->commRank = rank; c->commWorld = world;
    return EFFORT_OK;
}
extern "C" int effort_comm_destroy(effort_ctx* c) {
    if (!c) return EFFORT_ERR_ARG;
    if (c->comm) { hipSetDevice(c->device); effort_sync(c); rccl().CommDestroy(c->comm); c->comm = nullptr; c->commRank = 0; c->commWorld = 1; }
    return EFFORT_OK;
}
extern "C" int effort_comm_rank(effort_ctx* c) { return c ? c->commRank : EFFORT_ERR_ARG; }
extern "C" int effort_comm_world(effort_ctx* c) { return c ? c->commWorld : EFFORT_ERR_ARG; }
// All-gather of the ranks' output slices on the context's stream, after the multiplies that wrote them (the lanes are joined):
// recv = [world][count] f32.  send may be recv + rank*count (in place).
extern "C" int effort_allgather_outputs(effort_ctx* c, const float* send, float* recv, int count) {
    if (!c || !send || !recv || count < 1) return fail(c, EFFORT_ERR_ARG, "allgather_outputs: bad argument");
    if (!c->comm) return fail(c, EFFORT_ERR_ARG, "allgather_outputs: no communicator (effort_comm_create)");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    const ncclResult_t r = rccl().AllGather(send, recv, (size_t)count, ncclFloat, c->comm, c->stream);
    if (r != ncclSuccess) { snprintf(c->err, sizeof(c->err), "ncclAllGather: %s", rccl().GetErrorString(r)); return EFFORT_ERR_COMM; }
    return EFFORT_OK;
}

// ---- launch geometry ----------------------------------------------------------------------------
static bool supported(int W, int E) {
    // must match EFFORT_GEOMS in bucket_mul.hip
    return (W == 16 && (E == 1 || E == 2 || E == 4)) || (W == 8 && (E == 1 || E == 2 || E == 4)) ||
           (W == 4 && (E == 1 || E == 2 || E == 4)) || (W == 2 && E == 4);
}

// Row slices per call when the launch carries `groupSize` calls and a lane owns E columns.  Measured on MI355X
// (tools/lab/tune.py, 4096x4096 .. 14336x4096, 10-100 % effort): a workgroup's life is mostly fixed-latency steps (staging,
// cutoff, selection, hand-off), so FEWER, fatter items win even when they leave CUs idle -- about 3/4 of an item per CU
// for small groups, with slices between 128 and 512 input rows; from 8 calls on, the fattest slices (512 rows).
// Q4 groups of about 10 to 16 calls on a context WITHOUT lanes (one launch on the chip at a time): the launch as ONE round of workgroups -- just
// under two items per CU -- of 64-column tiles and tall slices.  A Q4 item's stream is bound by its CU's LDS atomic pipe, a CU's two workgroups
// run their heads, streams and tails in step (profiles/r06_q4_timelines.txt), and what a launch then costs is one head + the CU's share of the
// atomics + one tail: the unbalanced 384 items of the general rule (E = 2, 8 slices: half the CUs two items, half one) 65.4 us per 16-call
// launch, 480 items (E = 1, 5 slices) 58.3; 12 calls: 57.1 -> 50.2 at 6 slices (profiles/r06_q4_one_round_sweep.txt).  With several launches in
// flight the other launches fill the idle CUs anyway and the tall items only delay them (four in flight: 45.9 against 40.6 us): lanes keep the
// general rule.  Returns the slices per call, or 0 when the rule does not apply (the slices would be taller than two workgroups' LDS allows).
static uint32_t q4_one_round(const effort_ctx* c, uint32_t inDim, uint32_t groupTiles1) {
    if (c->nLanes > 1 || c->tuneS || !groupTiles1) return 0;
    const uint32_t sMin = (inDim + 831u) / 832u;                  // <= 832 rows per slice: 6656 candidate slots, ~78 KB of LDS, two workgroups per CU
    const uint32_t S = (uint32_t)c->numCU * 15u / 8u / groupTiles1;      // 480 items on 256 CUs
    return S >= sMin && S >= 2u ? S : 0u;
}
// Q4 groups of 3 .. 9 calls on a context without lanes: ONE item per CU -- the tallest slices (>= the slices a workgroup's LDS allows) whose items, counted on the
// padded ranges the XCDs are dealt from (a call's range is a multiple of 8; item % 8 is the XCD), still number at most the CUs.  A launch of one round lasts as long
// as its tallest item, and the first item past one per CU shares its CU for the whole launch: 3 calls of 4096x11008 at 8 slices (144 items of 512 rows) 30.2 us,
// 12 slices (216 of 342) 26.2, 14 slices (252 items, 264 padded) 30.9; 6 calls at 8 slices (288 items) 40.1, at 6 (240 padded) 35.5; 8 calls at E = 2 x 8 (192 fat items)
// 43.7, at E = 1 x 5 (256 padded) 39.5; 8 calls of 4096x4096 at E = 2 x 8 (64 items!) 42.5, at E = 1 x 16 (256) 24.8; 5 x (14336 -> 4096) at 32 slices (320 items) 56.5,
// at 24 (240) 37.9 (round 6, third session, profiles/r06_one_round_groups.txt).  Returns the slices per call, or 0 when such a round does not exist.
static uint32_t q4_one_per_cu(const effort_ctx* c, uint32_t inDim, uint32_t tiles1, int n, uint32_t groupTiles1) {
    if (c->nLanes > 1 || c->tuneS || n < 3 || n > 9) return 0;
    const uint32_t sMin = (inDim + 831u) / 832u, hi = ((inDim + 127u) / 128u + 7u) / 8u * 8u;
    const bool same = !groupTiles1 || groupTiles1 == (uint32_t)n * tiles1;          // (a mixed group: every call's padding bounded by 7)
    auto fits = [&](uint32_t s) {
        return same ? (uint32_t)n * ((tiles1 * s + 7u) / 8u * 8u) <= (uint32_t)c->numCU : groupTiles1 * s + 7u * (uint32_t)n <= (uint32_t)c->numCU;
    };
    uint32_t S = sMin > 2u ? sMin : 2u;
    if (!fits(S)) return 0;
    const uint32_t top = c->thinEffort ? ((inDim + 511u) / 512u + 7u) / 8u * 8u : hi;      // (next to nothing to stream: no more slices than the small groups' minimum)
    while (S < top && S < hi && fits(S + 1u)) S++;
    return S;
}
static uint32_t pick_slices(const effort_ctx* c, const effort_w* w, int groupSize, int E, uint32_t groupTiles = 0, bool fill = true) {
    const uint32_t tiles = (w->cols + 64 * E - 1) / (64 * E);
    const uint32_t lo = ((w->inDim + 511) / 512 + 7) / 8 * 8, hi = ((w->inDim + 127) / 128 + 7) / 8 * 8;
    if (w->fmt != kFp16 && (E == 1 || groupSize >= 8) && fill) {       // (8 / 9 calls that do not fit one E = 1 item per CU run at E = 2: one of THOSE per CU then -- 9 x (14336 ->
        const uint32_t S = q4_one_per_cu(c, w->inDim, tiles, groupSize, groupTiles);     //  4096): 32 slices = 288 items 78.4 us, 24 = 216 items 56.1; 9 x (4096 -> 14336): 8 slices 64.4, 6: 54.8)
        if (S) return S;
    }
    if (w->fmt != kFp16 && E == 1 && groupSize >= 8 && groupTiles) {       // (pick_elems chose E = 1 for this Q4 group: q4_one_round)
        const uint32_t S = q4_one_round(c, w->inDim, groupTiles);
        if (S) return S;
    }
    // Groups of >= 8 calls: the fewest slices (fat items: less fixed work per byte) -- unless the launch then leaves CUs WITHOUT an item: 8 calls on 4096x4096
    // matrices are 8 x 2 tiles x 8 slices = 128 items on 256 CUs.  FP16 groups then take the small groups' rule below (about 3/4 of an item per CU; it
    // never goes under `lo`): 8 x 4096x4096 31.8 -> 24.8 us per launch at 16 slices (32 slices: 28.7; E = 1 x 16: 27.6; E = 4 x 32: 29.7 -- round 6, third
    // session, profiles/r06_small_matrix_groups.txt).  `fill` = false: the count pick_elems prices its choice of E with (unchanged: E is chosen as before).
    if (groupSize >= 8 && (w->fmt != kFp16 || !fill || c->nLanes > 1 || c->thinEffort)) return lo;     // (with launches in flight on lanes the other launches fill the idle CUs: fat items stay -- 8 x 4096x4096, four in flight: 14.7 us per launch at 8 slices, 15.7 at 16)
    // (64-column tiles -- narrow matrices, see pick_elems -- are worked best at one item per CU: measured, 14336 -> 4096 lone, 64 slices
    //  26.9 us against 29.2 at 48)
    // (Q4 small groups are worked at E = 1 whatever the shape and want the 3/4 too -- round 6, a pair of 4096x11008 calls: 16 slices = 192 items 23.2 us
    //  against 25.4 at the 24 the full-CU target gave; lone and four per launch land on 192 items either way)
    const uint32_t target = (E == 1 && w->fmt == kFp16) ? (uint32_t)c->numCU : (uint32_t)c->numCU * 3u / 4u;
    // the launch's items come from ALL its calls: with the column tiles of the whole group known (Wq | Wk | Wv: 4 + 1 + 1 at
    // E = 1) every call takes target / tiles slices; without, the calls are taken as equals
    const uint32_t allTiles = groupTiles ? groupTiles : (uint32_t)groupSize * tiles;
    uint32_t S = (target / allTiles + 4u) / 8u * 8u;       // nearest multiple of 8
    if (S < lo) S = lo;
    if (S > hi) S = hi;
    // A slice's candidate slots are laid out per rank in blocks of 2^ceil(log2(rows)): a row count that is not a power of two wastes the
    // rest of the block in staging, selection and LDS (24 slices of a 4096-row input: 171 rows in blocks of 256, a third of the slots
    // empty).  With a power-of-two input the count snaps to the next power of two when the bounds allow.  Round 6 re-sweep of the decode
    // loop's launches and the lone FFN shapes (tools/lab/geosweep.py, profiles/r06_geosweep.txt): every launch was within 0.3 % of its best
    // geometry except the lone 4096 -> 14336 call -- the shape of the reference's own timing loop (benchmarks/benchmark.swift:245-257) --
    // where the rule above gave 24 slices: 20.0 -> 18.6 us at 25 % effort, 25.4 -> 23.4 at 50 %.
    if ((w->inDim & (w->inDim - 1u)) == 0u && (S & (S - 1u)) != 0u) {
        uint32_t up = 8u;
        while (up < S) up <<= 1;
        S = up <= hi ? up : up >> 1;
    }
    // Groups of >= 3 FP16 calls that fit ONE round of CUs: as many slices as keep the launch at one item per CU, any count, instead of the nearest power of two
    // under 3/4 of the CUs.  A launch of one round lasts as long as its tallest item: 3 calls of 4096x11008 at 8 slices are 144 items of 512 rows, at 13 slices
    // 234 of 316 (30.7 -> 25.3 us per launch).  "One item per CU" is counted the way the items are DEALT: a call's item range is padded to a multiple of 8 and item
    // % 8 is the XCD, so each call puts ceil(items / 8) on XCD 0 -- 7 calls of 33 items are 231 items but 35 on XCD 0's 32 CUs, and the launch takes 69 us
    // where 30 items per call take 45 (round 6, third session, profiles/r06_one_round_groups.txt).  Lone calls and pairs -- the decode loop's launches,
    // re-swept in round 6 -- keep the rule above.
    if (w->fmt == kFp16 && fill && groupSize >= 3 && c->nLanes <= 1) {          // (four launches in flight: 3 calls of 4096x11008 13.4 us per launch at 8 slices, 15.0 at 13)
        const bool same = allTiles == (uint32_t)groupSize * tiles;          // (a mixed group: every call's padding bounded by 7)
        auto fits = [&](uint32_t s) {
            return same ? (uint32_t)groupSize * ((tiles * s + 7u) / 8u * 8u) <= (uint32_t)c->numCU : allTiles * s + 7u * (uint32_t)groupSize <= (uint32_t)c->numCU;
        };
        if (fits(S)) { if (!c->thinEffort) while (S < hi && fits(S + 1u)) S++; }
        else {      // the rule above went OVER one item per CU (the power-of-two snap: 9 x (8192 -> 4096) 24 -> 32 slices = 288 items, 62.9 us against 46.1 at 24; the `hi`
                    // bound: 9 x (4096 -> 1024) at 32 slices 20.6 us, at 24 18.9): the most slices that fit, if any do
            uint32_t s2 = S;
            while (s2 > lo && !fits(s2)) s2--;
            if (fits(s2)) S = s2;
        }
    }
    return S;
}

// Columns per lane for the whole launch (one kernel variant serves all its calls).  FP16: 2; 1 when a small group would
// otherwise leave most of the chip without an item (small matrices); 4 for groups of >= 8 calls (fewer, fatter items: less
// fixed work per byte) unless that leaves the launch with between one and three items per CU -- half the chip would then
// run two workgroups per CU in lockstep with the other half's one -- or with less than half an item per CU (measured,
// 4096x11008: 8 calls 7.9 vs 8.4 us/call, 16 calls 7.2 vs 6.6, 32 calls 5.7 vs 6.3).  Q4 (a word = 4 sub-buckets): 1, or 2 from 8 calls on.
static int pick_elems(const effort_ctx* c, Format fmt, int n, const effort_w* const* ws) {
    if (c->tuneE) return c->tuneE;
    if (fmt != kFp16) {      // measured, 4096x11008 Q4: 32 calls 5.3 vs 6.4 us/call, 8 calls 8.3 vs 8.3, 2 calls 20.8 vs 18.8
        if (n < 8) return 1;
        if (n < 10) {        // (8 / 9 calls: one E = 1 item per CU where that fits: q4_one_per_cu)
            uint32_t t1 = 0, inDim = 0, tMax = 0;
            bool same = true;
            for (int i = 0; i < n; i++) if (ws[i]) {
                const uint32_t t = (ws[i]->cols + 63u) / 64u;
                t1 += t; same = same && (tMax == 0 || t == tMax); tMax = tMax > t ? tMax : t; inDim = inDim > ws[i]->inDim ? inDim : ws[i]->inDim;
            }
            if (q4_one_per_cu(c, inDim, tMax, n, same ? 0u : t1)) return 1;
        }
        if (n >= 10) {       // (one round of narrow, tall items where that fits: q4_one_round)
            uint32_t t1 = 0, inDim = 0;
            for (int i = 0; i < n; i++) if (ws[i]) { t1 += (ws[i]->cols + 63u) / 64u; inDim = inDim > ws[i]->inDim ? inDim : ws[i]->inDim; }
            if (q4_one_round(c, inDim, t1)) return 1;
        }
        return 2;
    }
    if (c->tuneS) return 2;
    auto group_tiles = [&](int E) {
        uint32_t t = 0;
        for (int i = 0; i < n; i++) if (ws[i]) t += (ws[i]->cols + 64 * E - 1) / (64 * E);
        return t;
    };
    auto items = [&](int E) {
        uint32_t t = 0;
        const uint32_t gt = group_tiles(E);
        for (int i = 0; i < n; i++) if (ws[i]) t += (ws[i]->cols + 64 * E - 1) / (64 * E) * pick_slices(c, ws[i], n, E, gt, false);
        return t;
    };
    const uint32_t numCU = (uint32_t)c->numCU;
    // how much of its column tiles' lanes a choice keeps busy: a narrow handle -- a column shard of a multi-GPU split, 11008
    // outputs over 8 ranks = 86 columns -- fills a third of ONE 256-column tile (E = 4: 22 of 64 lanes), two thirds at E = 2
    auto fill = [&](int E) {
        double used = 0, have = 0;
        for (int i = 0; i < n; i++) if (ws[i]) { used += ws[i]->cols; have += (double)((ws[i]->cols + 64 * E - 1) / (64 * E)) * 64 * E; }
        return have > 0 ? used / have : 1.0;
    };
    const double f1 = fill(1), f2 = fill(2), f4 = fill(4), best = f1 > f2 ? (f1 > f4 ? f1 : f4) : (f2 > f4 ? f2 : f4);
    if (n >= 8) {
        const uint32_t i4 = items(4);
        if (((i4 > numCU / 2 && i4 <= numCU) || i4 >= 3u * numCU) && f4 >= 0.8 * best) return 4;
        if (f2 >= 0.8 * best) return 2;
        // 64-column tiles because 128-column ones would leave lanes idle (k sequences' Wq | Wk | Wv: 4096 + 1024 + 1024 outputs fill 3/4 of their 128-column tiles) --
        // unless the 64-column items outnumber the CUs at the fewest slices and the 128-column ones do not: a launch of one item per CU at 3/4 lane fill beats a
        // second round (6 / 7 / 8 sequences, 18 / 21 / 24 calls: 37.6 -> 32.7 / 39.3 -> 33.8 / 39.7 -> 34.0 us per launch; 10 sequences overflow either way and stay)
        if (c->nLanes <= 1 && f2 >= 0.7 * best) {
            uint32_t p1 = 0, p2 = 0;
            for (int i = 0; i < n; i++) if (ws[i]) {
                const uint32_t lo = ((ws[i]->inDim + 511u) / 512u + 7u) / 8u * 8u;
                p1 += ((ws[i]->cols + 63u) / 64u * lo + 7u) / 8u * 8u; p2 += ((ws[i]->cols + 127u) / 128u * lo + 7u) / 8u * 8u;
            }
            if (p1 > numCU && p2 <= numCU) return 2;
        }
        return 1;
    }
    const uint32_t i2 = items(2);
    if (f2 < 0.8 * best) return 1;
    // 3..7 calls of BIG matrices whose E = 2 items overflow one round of CUs even at the fewest slices (6 calls of 4096x11008: 6 x 6 tiles x 8 = 288 items, the 32
    // over the 256 CUs run as a round of their own) while E = 4 items fit: E = 4, and pick_slices then fills the round (18 tiles x 13 slices = 234 items)
    if (n >= 3 && f4 >= 0.8 * best && c->nLanes <= 1) {
        uint32_t lo2 = 0, lo4 = 0;
        for (int i = 0; i < n; i++) if (ws[i]) {
            const uint32_t lo = ((ws[i]->inDim + 511u) / 512u + 7u) / 8u * 8u;
            lo2 += (ws[i]->cols + 127u) / 128u * lo; lo4 += (ws[i]->cols + 255u) / 256u * lo;
        }
        if (lo2 > numCU && lo4 <= numCU * 15u / 16u) return 4;
    }
    // narrow matrices (<= 256 bucket columns: 4096 outputs) in small groups: 64-column tiles -- more tiles, each reduced by its
    // own last arriver (measured, us per launch at 25 %: Wq|Wk|Wv 21.2 vs 23.8, 14336 -> 4096 lone 26.6-27.5 vs 29.7)
    bool narrow = true;
    for (int i = 0; i < n; i++) narrow = narrow && (!ws[i] || ws[i]->cols <= 256u);
    // (... for launches that have the chip to themselves, and for lone calls.  GROUPS on a context with lanes -- launches in flight beside one another, the CUs
    //  never short of items -- want the fatter 128-column tiles: four in flight, us per launch E = 1 / E = 2: 14336 -> 4096 x 2 / 3 / 4 / 6 calls 14.7 / 13.0, 21.6 / 16.9,
    //  28.6 / 22.3, 44.6 / 31.7; 4096x4096 x 2 / 3 / 4 / 6: 7.7 / 6.7, 10.9 / 7.8, 10.7 / 10.2, 13.1 / 11.8; lone calls 9.7 / 10.0 and 5.4 / 5.9 -- round 6, third session,
    //  profiles/r06_one_round_groups.txt)
    if (narrow && f1 >= 0.8 * best && n >= 3 && c->nLanes <= 1 && f2 >= 0.8 * best) {
        // ... unless the 64-column tiles overflow ONE round of CUs even at the fewest slices while the 128-column tiles fit it (tall narrow matrices: 3 calls of 11008 -> 4096
        // are 3 x 4 tiles x 24 slices = 288 items, at E = 2 144 -- and pick_slices then fills the round: 35.6 -> 27.9 us per launch, 4 / 5 calls -15 / -7 %, 3 x (14336 -> 4096) -12 %)
        uint32_t p1 = 0, p2 = 0;
        for (int i = 0; i < n; i++) if (ws[i]) {
            const uint32_t lo = ((ws[i]->inDim + 511u) / 512u + 7u) / 8u * 8u;
            p1 += ((ws[i]->cols + 63u) / 64u * lo + 7u) / 8u * 8u; p2 += ((ws[i]->cols + 127u) / 128u * lo + 7u) / 8u * 8u;
        }
        if (p1 > numCU && p2 <= numCU) return 2;
    }
    if (narrow && f1 >= 0.8 * best && !(c->nLanes > 1 && n >= 3)) return 1;      // (pairs keep the lone calls' tiles: -11 % left on the table, and a pair's bits do not depend on the lanes)
    return (i2 * 10u < numCU * 3u / 4u * 6u && items(1) > i2) ? 1 : 2;
}

static int choose_geom(const effort_ctx* c, const effort_w* w, int groupSize, int E, MulGeom* g, int* Wout, int* Eout, uint32_t sliceMult = 1, uint32_t groupTiles = 0) {
    const int W = c->tuneW ? c->tuneW : 8;                 // 8 waves per workgroup
    if (!supported(W, E)) return EFFORT_ERR_ARG;
    const uint32_t nacc = w->fmt == kFp16 ? 16 : 32;
    g->inDim = w->inDim; g->outDim = w->outDim; g->cols = w->cols; g->rowsPerIn = w->rowsPerIn;
    g->expertRows = w->rowsPerIn * w->inDim; g->numExperts = w->numExperts;
    g->tiles = (w->cols + 64 * E - 1) / (64 * E);
    g->elems = (uint32_t)E;
    g->rowPitch = w->rowPitch;
    const uint32_t tileFloats = nacc * E * 64;
    const size_t ldsMax = 160 * 1024;
    uint32_t S;
    if (c->tuneS) S = c->tuneS;                            // any count: the item grid is padded to a multiple of 8 slices
    else {
        const uint32_t want = pick_slices(c, w, groupSize, E, groupTiles) * sliceMult;
        const uint32_t cap = (c->numCU * 2u) / g->tiles / 8 * 8;              // one round of workgroups
        S = cap < want ? cap : want;
    }
    if (S > w->inDim) S = w->inDim / 8 * 8;
    if (S < 1) S = 1;
    const uint32_t maxCand = bucket_mul_max_candidates(W);
    for (;;) {
        g->sliceRows = (w->inDim + S - 1) / S;
        g->slices = (w->inDim + g->sliceRows - 1) / g->sliceRows;
        g->sliceLog2 = 0; while ((1u << g->sliceLog2) < g->sliceRows) g->sliceLog2++;
        g->slots = w->fmt == kFp16 ? (g->rowsPerIn << g->sliceLog2) : g->sliceRows * 8u;
        const size_t lds = bucket_mul_lds_bytes(w->fmt, W, E, *g, true);    // (whether the launch will be a lean plain grid is known only once all its calls are: budget for the larger plan, the lean one's)
        const bool fits = lds <= ldsMax && g->slots <= maxCand && (size_t)g->slots * 4 + (size_t)g->sliceRows * 8 + 1024 <= 65536 &&   // staged regions below 64 KB
                          (w->fmt == kFp16 ? (1u << g->sliceLog2) <= 64u * (uint32_t)W : g->sliceRows <= 128u * (uint32_t)W);   // a thread stages one (Q4: two) inputs of the slice
        const size_t slab = (size_t)g->slices * g->tiles * tileFloats * 4;
        if (fits && slab <= c->slabBytes) break;
        if (!fits) { S += 1; if (S > w->inDim + 8) return EFFORT_ERR_SHAPE; }
        else return EFFORT_ERR_SHAPE;
    }
    *Wout = W; *Eout = E;
    return EFFORT_OK;
}

static int ensure_timing(effort_ctx* c) {
    if (c->ev) return EFFORT_OK;
    c->ev = new (std::nothrow) hipEvent_t[effort_ctx::kMaxSamples * 4]();       // (value-initialised: effort_destroy walks the whole array)
    if (!c->ev) return EFFORT_ERR_HIP;
    for (int i = 0; i < effort_ctx::kMaxSamples * 4; i++) HIP_TRY(c, hipEventCreate(&c->ev[i]));
    return EFFORT_OK;
}

// One launch for a group of independent calls (a lone call is a group of one).
static int do_group(effort_ctx* c, Format fmt, int n, const effort_w* const* ws, const float* const* vs,
                    const uint32_t* const* expNos, float* const* outs, const double* efforts,
                    const int* prologues = nullptr, const void* const* vAux = nullptr, const float* const* resids = nullptr) {
    if (!c || !ws || !vs || !outs || !efforts) return fail(c, EFFORT_ERR_ARG, "bucketmul: null argument");
    if (n < 1 || n > kMaxGroup) return fail(c, EFFORT_ERR_ARG, "bucketmul: group size outside 1..32");
    // The A/B switches below read the environment in LAB builds only (-DEFFORT_LAB: tools/build_variant*.sh); the shipped
    // library has no getenv on this path.
#ifdef EFFORT_LAB
    static const uint32_t ablate = getenv("EFFORT_ABLATE") ? (uint32_t)atoi(getenv("EFFORT_ABLATE")) : 0u;   // profiling only
#else
    constexpr uint32_t ablate = 0u;
#endif
    for (int i = 0; i < n; i++) {                      // every argument first: nothing is launched for a group with a bad call
        if (!ws[i] || !vs[i] || !outs[i]) return fail(c, EFFORT_ERR_ARG, "bucketmul: null argument");
        if (ws[i]->dead) return fail(c, EFFORT_ERR_ARG, "bucketmul: the handle was freed (its buffers live on only for its column shards)");
        if (ws[i]->fmt != fmt) return fail(c, EFFORT_ERR_KIND, "bucketmul: weight handle of the wrong kind");
        if (!(efforts[i] >= 0.0 && efforts[i] <= 1.0)) return fail(c, EFFORT_ERR_EFFORT, "bucketmul: effort outside [0,1]");
        const int pre = prologues ? prologues[i] : 0;
        if (pre < 0 || pre > 2 || (pre && (!vAux || !vAux[i]))) return fail(c, EFFORT_ERR_ARG, "bucketmul: bad input prologue");
        if (pre && (fmt != kFp16 || c->splitCutoff)) return fail(c, EFFORT_ERR_KIND, "bucketmul: input prologues need FP16 weights and the fused cutoff");
    }
    { double sum = 0.0; for (int i = 0; i < n; i++) sum += efforts[i]; c->thinEffort = sum < 0.08 * n; }
    const int groupE = pick_elems(c, fmt, n, ws);         // columns per lane: one choice for a group launch
    const bool tm = c->timing && c->nSamples < effort_ctx::kMaxSamples;
    hipEvent_t* ev = tm ? c->ev + 4 * c->nSamples : nullptr;
    // ---- which lane (overlap mode): the launch may run beside the launches in flight on the OTHER lanes unless it reads or
    // writes what one of them writes, or writes what one of them reads; then it waits for that lane (or simply joins it: a
    // lane's stream is in order).  Always ordered after everything enqueued on the context's stream before this call.
    // The lane is CHOSEN here and forked (made to wait) only when the first kernel of the group is about to be launched:
    // a group that fails validation or finds no launch geometry has then touched no stream.  From the fork on, every way out
    // records the lane's `done` event and marks the lane pending (LaneGuard), so a later join -- in particular the join a
    // caller owes the capturing stream before it ends a hipGraph capture -- always rejoins a lane that was forked.
    int li = 0;
    const bool laned = c->nLanes > 1 && !c->timing && !c->clock;
    std::vector<Lane::Range> rd, wr;
    int hazard[effort_ctx::kMaxLanes], nh = 0;
    unsigned long long capNow = 0;
    if (laned) {
        capNow = capture_of(c->stream);
        if (!lanes_match_capture(c, capNow)) return EFFORT_ERR_ARG;
        // bounded bookkeeping: a caller may enqueue arbitrarily many multiplies between joins (helpers/gpu.swift:109-119: one
        // eval() per token); once a lane has recorded kMaxRanges ranges the lanes are JOINED -- the context's stream waits for
        // all of them, and every later launch forks from that stream -- never forgotten
        bool full = false;
        for (int i = 0; i < c->nLanes; i++) full = full || c->lane[i].reads.size() + c->lane[i].writes.size() > effort_ctx::kMaxRanges;
        if (full) { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
        auto add = [](std::vector<Lane::Range>& v, const void* p, size_t bytes) { if (p && bytes) v.push_back({(uintptr_t)p, (uintptr_t)p + bytes}); };
        for (int i = 0; i < n; i++) {
            add(rd, vs[i], (size_t)ws[i]->inDim * 4);
            if (expNos && expNos[i]) add(rd, expNos[i], 4);
            if (vAux && vAux[i]) add(rd, vAux[i], (size_t)ws[i]->inDim * 4);
            if (resids && resids[i]) add(rd, resids[i], (size_t)ws[i]->outDim * 4);
            add(wr, outs[i], (size_t)ws[i]->outDim * 4);
        }
        for (int i = 0; i < c->nLanes; i++) {
            const Lane& L = c->lane[i];
            if (L.pending && (overlaps(rd, L.writes) || overlaps(wr, L.writes) || overlaps(wr, L.reads))) hazard[nh++] = i;
        }
        li = nh ? hazard[0] : c->nextLane;
        // A chain of dependent launches lives on ONE lane and forks from the context's stream only when that stream is busy
        // (fork_lane).  HIP multiplexes streams over a few hardware queues, and a lane that happens to share its queue with the
        // context's stream makes hipStreamQuery(context's stream) say "busy" while the LANE works: every link of the chain then
        // pays the event-record-and-wait fork, +8 us per call (tools/lab/lane_probe.py with EXTRA_CONTEXTS=1: one lane of four;
        // round 5's bench record showed it at three of five efforts).  Four busy answers in a row on a pure chain: the chain MOVES
        // to an idle lane -- one cross-lane edge, the ordinary hazard wait below -- where the stream's answer is its own again.
        // Once between joins: if the busy answers were true (the caller really enqueues between the calls) nothing is lost but that edge.
        if (nh == 1 && c->busyStreak >= 4 && !c->migrated) {
            for (int k = 1; k < c->nLanes; k++) {
                const int cand = (hazard[0] + k) % c->nLanes;
                if (!c->lane[cand].pending) { li = cand; c->migrated = true; c->busyStreak = 0; break; }
            }
        }
    } else if (c->nLanes > 1) {
        { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }        // timing modes: one launch at a time, on lane 0
    }
    Lane& L = c->lane[li];
    const hipStream_t st = laned ? L.own : c->stream;
    struct LaneGuard {                                 // (see above)
        Lane& L; hipStream_t st; bool forked = false;
        ~LaneGuard() { if (forked) { L.dirty = true; L.pending = true; } }     // (the event itself: lane_mark, when somebody waits for the lane)
    } guard{L, st};
    auto fork_lane = [&]() -> int {                    // before the group's first launch
        if (!laned || guard.forked) return EFFORT_OK;
        // "after everything enqueued on the context's stream before this call": an event recorded there, the lane waits -- UNLESS the
        // stream is idle: then everything enqueued on it has completed and there is nothing to wait for.  That is the state of a
        // caller who enqueues multiply after multiply and evaluates once (helpers/gpu.swift:109-119; the reference's own timing
        // loop, benchmarks/benchmark.swift:245-257): all its work sits on the lanes, and the two HIP calls plus the cross-stream
        // edge per call made that loop 45 % SLOWER with lanes than without (round 4: 32.5 against 22.5 us per call).  A capturing
        // stream cannot be queried (and "idle" means nothing inside a capture): there the fork is a graph edge, as before.
        bool wait = true;
        if (capNow == 0) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) wait = false;
            else {
                (void)hipGetLastError();                    // ("not ready" must not linger as the thread's last error: the launchers read it after their kernels)
                if (q != hipErrorNotReady) return fail(c, EFFORT_ERR_HIP, "hipStreamQuery", q);
            }
        }
        if (wait) {
            HIP_TRY(c, hipEventRecord(c->forkEv, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(L.own, c->forkEv, 0));
        }
        c->busyStreak = (wait && nh == 1 && hazard[0] == li) ? c->busyStreak + 1 : 0;     // (a dependent chain's link that had to fork)
        guard.forked = true;
        L.capId = capNow;
        // (a range the lane holds already is not recorded twice: a loop that multiplies into the same vectors over and over --
        //  the reference's timing loop -- would otherwise grow the lists to their cap, and every call scan thousands of ranges)
        auto add_new = [](std::vector<Lane::Range>& have, const std::vector<Lane::Range>& more) {
            for (const auto& x : more) {
                bool seen = false;
                for (const auto& y : have) if (y.lo == x.lo && y.hi == x.hi) { seen = true; break; }
                if (!seen) have.push_back(x);
            }
        };
        // a lane this launch waits for: from here on this lane's order covers everything that lane has enqueued, so its ranges move
        // here and it is no longer pending (a join need not wait for it, a later launch that touches those ranges follows THIS lane)
        for (int k = 0; k < nh; k++) if (hazard[k] != li) {
            Lane& H = c->lane[hazard[k]];
            HIP_TRY(c, lane_mark(H));
            HIP_TRY(c, hipStreamWaitEvent(L.own, H.done, 0));
            add_new(L.reads, H.reads); add_new(L.writes, H.writes);
            H.reads.clear(); H.writes.clear(); H.pending = false;
        }
        if (!nh) c->nextLane = (c->nextLane + 1) % c->nLanes;
        add_new(L.reads, rd);
        add_new(L.writes, wr);
        return EFFORT_OK;
    };
    c->lastLane = li;
    if (tm) HIP_TRY(c, hipEventRecord(ev[0], st));
    GroupKArgs ga;
    int W = 0, E = 0;
    uint32_t nGeoms = 0, wg = 0, realItems = 0, first = 0;
    size_t slabOff = 0; uint32_t tileOff = 0, sliceOff = 0;
    auto begin = [&](uint32_t firstCall) {
        memset(&ga, 0, sizeof(ga));
        ga.groupDone = L.d_counters + effort_ctx::kMaxTiles - 1;
        ga.slabs = L.d_slabs; ga.counters = L.d_counters; ga.sliceCounts = L.d_sliceCounts; ga.cutoff = L.d_cutoff + firstCall;
        ga.tstamp = c->clock ? c->d_tstamp : nullptr;
        ga.ablate = ablate; ga.split = (c->splitCutoff ? 1u : 0u) | (c->rowReuse ? 8u : 0u); ga.trace = (c->clock && c->trace) ? 1u : 0u;
        ga.numCU = (uint32_t)c->numCU; ga.queue = L.d_queue;
        nGeoms = 0; wg = 0; realItems = 0; first = firstCall;
    };
    auto flush = [&]() -> int {                        // one kernel launch for the calls gathered so far
        // grid: persistent workgroups once the items outnumber what the chip holds at R per CU
        const uint32_t R = c->persistent < 0 ? 2u : (uint32_t)c->persistent;
        // FP16 on a context WITHOUT lanes -- one launch on the chip at a time -- stays a PLAIN grid: the lean kernel (half the
        // instructions, every workgroup evaluates its cutoff at once instead of awaiting a job), and the dispatcher hands the third round of workgroups
        // to whichever CU frees a slot.  Round 6, 4096x11008, us per launch persistent -> plain: 12 calls 80.0 -> 74.9, 16: 89.9 -> 87.4, 20: 111.9 -> 103.2,
        // 16 at 50 % effort 157.6 -> 145.8, 16 x (4096 -> 14336) 108.9 -> 101.8, 24 / 32 calls level (122.2 / 122.4, 153.0 / 152.2), 32 at 10 % +2 %;
        // with four launches in flight the persistent grid wins (16 calls 66.9 against 67.6, 32: 127.2 against 130.1): lanes keep it from 2 per CU on
        // (profiles/r06_plain_vs_persistent.txt).
        // (1024 items -- 32 calls of 4096 -> 14336 -- 205.4 -> 197.7; 1536 -- 32 calls at E = 2 -- plain 0.98 x the heuristic's launch where persistent was 1.04 x:
        //  plain up to six items per CU, as far as was measured)
        const uint32_t perCU = (c->persistent < 0 && fmt == kFp16 && !laned) ? 6u : R;
        ga.persistent = (R && realItems > ga.numCU * perCU) ? R : 0u;      // (the items that exist, not the padded item range)
        // persistent launches evaluate every call's cutoff ONCE, in a job of its own at the head of the item queues, instead
        // of once per workgroup and call (measured: 6.8 of the ~90 us of an item at 32 calls per launch)
        bool plain = true;
        for (uint32_t i = 0; i < ga.count; i++) plain = plain && !ga.call[i].pre;
#ifdef EFFORT_LAB
        static const bool noJobs = getenv("EFFORT_NO_CUTJOBS") != nullptr;
#else
        constexpr bool noJobs = false;
#endif
        ga.cutJobs = (ga.persistent && !c->splitCutoff && plain && !noJobs) ? (ga.count + 7u) / 8u * 8u : 0u;
        // FP16: the multiply stages the compact row means where every slice starts on an even row (an LDS-direct load lands two):
        // a quarter of the lines of the 8-byte stats entries, half the loads.  Persistent launches, and the plain grids the lean
        // instantiation serves (8 waves, no stamps): with the path a template parameter it costs them no code (as a run-time
        // switch inside one kernel it cost lone calls 3 %); measured on plain grids: decode 298 -> 300 tokens/s.
#ifdef EFFORT_LAB
        static const bool noCompact = getenv("EFFORT_NO_COMPACT_MEANS") != nullptr;
#else
        constexpr bool noCompact = false;
#endif
        const bool leanGrid = ga.persistent == 0u && W == 8 && !c->clock && !ablate;      // (launch_mul_t's condition for the lean instantiations)
        bool compact = fmt == kFp16 && !noCompact && (leanGrid || (ga.persistent != 0u && plain));    // lean: with prologues / residuals too
        for (uint32_t i = 0; compact && !leanGrid && i < ga.count; i++) compact = !ga.call[i].resid;
        for (uint32_t i = 0; compact && i < ga.count; i++) {
            const MulGeom& g = ga.geom[ga.call[i].geom];
            compact = ws[first + i]->means16 != nullptr && g.inDim % 2u == 0u && g.sliceRows % 2u == 0u;
        }
        if (compact) {
            for (uint32_t i = 0; i < ga.count; i++) ga.call[i].stats = ws[first + i]->means16;
            ga.split |= 4u;
        }
        // Persistent Q4 launches of a context WITHOUT lanes -- one launch on the chip at a time -- start the second workgroup of every CU ~10 us late,
        // which takes a CU's pair out of step (bucket_mul_kernel; profiles/r06_ab_q4_stagger.txt: 32 calls per launch 109.5 -> 99.8 us).  With lanes the
        // launches overlap, the CUs are busy anyway and the wait is a loss (2-5 %): off.  FP16 launches are bound by the CU's pull from memory, not
        // by an LDS pipe, and their pairs fall out of step by themselves (the older workgroup wins the arbitration two to one): measured, no gain
        // (profiles/r06_ab_fp16_stagger.txt): off.
        ga.staggerSleeps = (ga.persistent && fmt == kQ4 && !laned) ? 11u : 0u;
        const int frc = fork_lane();
        if (frc != EFFORT_OK) return frc;
        if (c->splitCutoff && !(ablate & 1u)) HIP_TRY(c, launch_find_cutoff_group(ga, st));
        HIP_TRY(c, launch_bucket_mul(fmt, W, E, ga, st));
        return EFFORT_OK;
    };
    uint32_t groupTiles = 0;
    for (int i = 0; i < n; i++) groupTiles += (ws[i]->cols + 64 * groupE - 1) / (64 * groupE);
    // Thin slices for the LAST TWO calls of an FP16 launch whose last round of workgroups would be nearly empty (round 6, third session;
    // profiles/r06_tail_slices.txt).  A plain grid's workgroups are handed out in call order, two per CU at a time: 11 calls of 48 items are 528 items --
    // one round of 512 and 16 stragglers that start when the others END, a whole item's duration for 3 % of the work.  With the last two calls cut into twice the
    // slices the tail of the launch is made of half-height items: the first of them finish while the round is still running and hand their slots on.  us per
    // launch, 4096x11008 at 25 %, without / with the rule: 11 calls 72.8 -> 66.0 (-9.4 %), 12: 74.5 -> 69.3 (-7 %), 22: 114.2 -> 108.4 (-5 %), 23: 115.6 -> 110.9
    // (-4 %), 12 at 10 / 50 / 100 % effort -5.7 / -9.1 / -11.5 %, 11 at 100 % -13.7 %, 19 x (4096 -> 14336) -4.8 %, 11 x (4096 -> 14336) -6 %.  The gain shrinks as
    // the last round fills -- 13 calls (112 of 512 slots) -2.5 %, 24 calls or 18 x (14336 -> 4096) (128) 0 / +3 % -- so the rule ends at 7/32 of a round.  Applied
    // to EVERY mid-size launch (its first form) it was level or worse from a quarter-full last round on: 14 / 15 calls +2 / +3 %, 16 at 50 / 100 % effort +2.5 /
    // +4.5 %, 16 calls of a 4096x4096 matrix (256 items: not even one round) +12 %.  More than
    // two thin calls, or four times the slices: never better (a thin item pays the same head and hand-off for half the rows); E = 4 launches (32 calls: 149.0 ->
    // 150.8), persistent grids (151 -> 157) and launches in flight on lanes (a launch's tail runs under the next one's head; round 2: 124.3 -> 125.8): worse, off.
    // HOW MANY calls: enough that the thin calls' items cover the stragglers (the items past the last whole round), at least two, at most four -- 24 Q4 calls of 24
    // items are 64 over a round: two thin calls (48 items) 93.7 -> 92.6 us, four (96) 86.3.  Q4 (E = 2 launches past one round of two workgroups per CU are PERSISTENT
    // grids on a context without lanes: the queue hands the items out in call order just the same): 22 / 24 calls of 4096x11008 93.0 -> 84.9 / 93.7 -> 86.3 us, 17 / 18 x
    // (4096 -> 14336) 90.0 -> 83.1 / 91.0 -> 83.8, 24 calls at 50 % effort 139.6 -> 128.7; 26 calls (112 over) level.
    int thinFrom = n;                                     // calls [thinFrom, n) take twice the slices
    if (groupE == 2 && n >= 8 && !laned && !c->tuneS && c->persistent < 0 && !c->thinEffort) {
        uint32_t base = 0, it1[kMaxGroup];
        bool ok = true;
        for (int i = 0; ok && i < n; i++) {
            MulGeom g1; int Wi, Ei;
            memset(&g1, 0, sizeof(g1));
            ok = choose_geom(c, ws[i], n, groupE, &g1, &Wi, &Ei, 1, groupTiles) == EFFORT_OK;
            it1[i] = (g1.tiles * g1.slices + 7u) / 8u * 8u;
            base += it1[i];
        }
        const uint32_t round = 2u * (uint32_t)c->numCU, over = ok ? base % round : 0u;
        // (FP16 stays a plain grid up to six items per CU; Q4 past one round is a persistent grid whatever its size)
        if (ok && base > round && over != 0u && over * 32u <= round * 7u && (fmt != kFp16 || base <= 6u * (uint32_t)c->numCU)) {
            int tc = 0;
            uint32_t thin = 0;
            while (tc < n && tc < 4 && (tc < 2 || thin < over)) thin += it1[n - 1 - tc++];
            // the thin geometries: they must exist (more slices than the call has) and fit the launch descriptor's kMaxGeoms shapes beside the others
            MulGeom seen[kMaxGeoms + 1];
            uint32_t nSeen = 0, extra = 0;
            auto note = [&](const MulGeom& g) {
                uint32_t k = 0;
                while (k < nSeen && memcmp(&seen[k], &g, sizeof(g)) != 0) k++;
                if (k == nSeen) { if (nSeen == kMaxGeoms) ok = false; else seen[nSeen++] = g; }
            };
            ok = thin >= over;
            for (int i = 0; ok && i < n; i++) {
                MulGeom g; int Wi, Ei;
                memset(&g, 0, sizeof(g));
                const bool t = i >= n - tc;
                ok = choose_geom(c, ws[i], n, groupE, &g, &Wi, &Ei, t ? 2u : 1u, groupTiles) == EFFORT_OK;
                if (ok && t) { const uint32_t it2 = (g.tiles * g.slices + 7u) / 8u * 8u; ok = it2 > it1[i]; extra += it2 - it1[i]; }
                if (ok) note(g);
            }
            if (ok && (fmt != kFp16 || base + extra <= 6u * (uint32_t)c->numCU)) thinFrom = n - tc;
        }
    }
    begin(0);
    for (int i = 0; i < n; i++) {
        const effort_w* w = ws[i];
        MulGeom g;
        memset(&g, 0, sizeof(g));
        int Wi, Ei;
        // the calls at the END of a mid-size group are cut into thinner slices (thinFrom, above): their items are the last ones handed out, and
        // the launch ends when the last item does
        uint32_t mult = i >= thinFrom ? 2u : 1u;
#ifdef EFFORT_LAB
        if (n >= 8 && !c->tuneS && getenv("EFFORT_TAIL_CALLS")) {       // profiling knobs (read at every call: tools/qbench.py --tails sweeps them in one process): the rule above replaced by "the last tc calls at tailMult x the slices"
            const int tc = atoi(getenv("EFFORT_TAIL_CALLS"));
            const int tailMult = getenv("EFFORT_TAIL_MULT") ? atoi(getenv("EFFORT_TAIL_MULT")) : 2;
            mult = 1;
            if (i >= n - tc) mult = (uint32_t)tailMult;
            if (tailMult >= 4 && i >= n - tc && i < n - tc / 2) mult = (uint32_t)tailMult / 2;      // two steps: ... x2 x2 x4 x4
        }
#endif
        int rc = choose_geom(c, w, n, groupE, &g, &Wi, &Ei, mult, groupTiles);
        if (rc != EFFORT_OK) return fail(c, rc, "bucketmul: no launch geometry for this shape/tuning");
        if (i == 0) { W = Wi; E = Ei; }
        else if (Wi != W || Ei != E) return fail(c, EFFORT_ERR_SHAPE, "bucketmul: the calls of a group must agree on the kernel variant");
        uint32_t gi = 0;
        while (gi < nGeoms && memcmp(&ga.geom[gi], &g, sizeof(g)) != 0) gi++;
        if (gi == nGeoms && nGeoms == kMaxGeoms) {     // a launch carries kMaxGeoms distinct shapes: this call opens the next one
            rc = flush();
            if (rc != EFFORT_OK) return rc;
            begin((uint32_t)i);
            gi = 0;
        }
        if (gi == nGeoms) ga.geom[nGeoms++] = g;
        CallDesc& a = ga.call[(uint32_t)i - first];
        const size_t slab = (size_t)g.slices * g.tiles * ((fmt == kFp16 ? 16u : 32u) * (uint32_t)Ei * 64u) * 4;
        if (tileOff + g.tiles + 1 > effort_ctx::kMaxTiles || sliceOff + g.slices > effort_ctx::kMaxSlices || slabOff + slab > c->slabBytes)
            return fail(c, EFFORT_ERR_SHAPE, "bucketmul: group exceeds the context scratch");
        a.buckets = w->buckets; a.stats = w->stats; a.rankBound = w->rankBound; a.probes = w->probes; a.v = vs[i];
        a.expNo = expNos ? expNos[i] : nullptr; a.out = outs[i];
        a.ol = OutlierIndex{fmt == kQ4 ? w->olBlockPtr : nullptr, w->olEntry, w->olMeta};
        a.q = (uint16_t)(int)((double)(kProbes - 1) * (1.0 - efforts[i]));            // bucketMul.swift:39
        a.bucketsTrim = (uint16_t)w->viewTrim;
        const int pre = prologues ? prologues[i] : 0;
        a.pre = (uint16_t)pre; a.vAux = pre ? vAux[i] : nullptr; a.resid = resids ? resids[i] : nullptr;
        a.slabOff = (uint32_t)(slabOff / 256); a.tileOff = (uint16_t)tileOff; a.sliceOff = (uint16_t)sliceOff; a.geom = (uint16_t)gi;
        wg += (g.tiles * g.slices + 7u) / 8u * 8u;            // the call's item range: a multiple of 8 (item % 8 = the XCD; locate_item)
        realItems += g.tiles * g.slices;
        if (wg / 8u > 0xFFFFu) return fail(c, EFFORT_ERR_SHAPE, "bucketmul: group exceeds the launch descriptor's item range");
        ga.wgEnd8[(uint32_t)i - first] = (uint16_t)(wg / 8u); ga.totalItems = wg;
        ga.count = (uint32_t)i - first + 1u;
        ga.totalTiles += g.tiles;
        L.lastSliceOff[i] = sliceOff; L.lastSlices[i] = g.slices;                   // dispatch.size = sum of the per-slice counts
        slabOff += (slab + 255) / 256 * 256; tileOff += g.tiles; sliceOff += g.slices;
    }
    L.lastCalls = (uint32_t)n;
    int rc = flush();
    if (rc != EFFORT_OK) return rc;
    if (tm) { HIP_TRY(c, hipEventRecord(ev[1], st)); c->nSamples++; }
    return EFFORT_OK;                                  // (LaneGuard records the lane's `done` event)
}

extern "C" int effort_bucketmul(effort_ctx* c, const effort_w* w, const float* v, const uint32_t* expNo, float* out, double effort) {
    return do_group(c, kFp16, 1, &w, &v, &expNo, &out, &effort);
}
extern "C" int effort_bucketmul_q4(effort_ctx* c, const effort_w* w, const float* v, const uint32_t* expNo, float* out, double effort) {
    return do_group(c, kQ4, 1, &w, &v, &expNo, &out, &effort);
}
extern "C" int effort_bucketmul_group(effort_ctx* c, int n, const effort_w* const* ws, const float* const* vs,
                                      const uint32_t* const* expNos, float* const* outs, const double* efforts) {
    return do_group(c, kFp16, n, ws, vs, expNos, outs, efforts);
}
extern "C" int effort_bucketmul_group_fused(effort_ctx* c, int n, const effort_w* const* ws, const float* const* vs,
                                            const uint32_t* const* expNos, float* const* outs, const double* efforts,
                                            const int* prologues, const void* const* vAux, const float* const* resids) {
    return do_group(c, kFp16, n, ws, vs, expNos, outs, efforts, prologues, vAux, resids);
}
extern "C" int effort_bucketmul_q4_group(effort_ctx* c, int n, const effort_w* const* ws, const float* const* vs,
                                         const uint32_t* const* expNos, float* const* outs, const double* efforts) {
    return do_group(c, kQ4, n, ws, vs, expNos, outs, efforts);
}

extern "C" int effort_calc_dispatch(effort_ctx* c, const effort_w* w, const float* v, const uint32_t* expNo, double effort,
                                    float* dispatch, uint32_t* count) {
    if (!c || !w || !v || !dispatch || w->dead) return fail(c, EFFORT_ERR_ARG, "calc_dispatch: null argument (or a freed handle)");
    if (!(effort >= 0.0 && effort <= 1.0)) return fail(c, EFFORT_ERR_EFFORT, "calc_dispatch: effort outside [0,1]");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    Lane& L = c->lane[0];
    c->lastLane = 0;
    MulGeom g; int W, E;
    int rc = choose_geom(c, w, 1, pick_elems(c, w->fmt, 1, &w), &g, &W, &E);
    if (rc != EFFORT_OK) return fail(c, rc, "calc_dispatch: geometry");
    const uint32_t q = (uint32_t)(int)((double)(kProbes - 1) * (1.0 - effort));
    HIP_TRY(c, launch_find_cutoff(v, w->probes, expNo, q, L.d_cutoff, L.d_count, nullptr, c->stream));
    HIP_TRY(c, launch_calc_dispatch(w->fmt, w->stats, v, expNo, L.d_cutoff, g, dispatch, count, L.d_count, c->d_blockScratch, c->stream));
    L.lastCalls = 1; L.lastSlices[0] = 0;        // dispatch.size is the scalar written by the scan kernel
    return EFFORT_OK;
}

extern "C" int effort_group_dispatch_count(effort_ctx* c, int idx, uint32_t* host_out) {
    if (!c || !host_out || idx < 0 || (uint32_t)idx >= c->lane[c->lastLane].lastCalls) return EFFORT_ERR_ARG;
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    Lane& L = c->lane[c->lastLane];
    if (L.lastSlices[idx] == 0) {
        HIP_TRY(c, hipMemcpyAsync(host_out, L.d_count, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return EFFORT_OK;
    }
    static thread_local uint32_t h[effort_ctx::kMaxSlices];
    HIP_TRY(c, hipMemcpyAsync(h, L.d_sliceCounts + L.lastSliceOff[idx], (size_t)L.lastSlices[idx] * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint32_t n = 0;
    for (uint32_t i = 0; i < L.lastSlices[idx]; i++) n += h[i];
    *host_out = n;
    return EFFORT_OK;
}
extern "C" int effort_debug_slice_counts(effort_ctx* c, int idx, uint32_t* host, int maxSlices) {
    if (!c || !host || idx < 0 || (uint32_t)idx >= c->lane[c->lastLane].lastCalls || maxSlices < 1) return EFFORT_ERR_ARG;
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    Lane& L = c->lane[c->lastLane];
    const uint32_t n = L.lastSlices[idx] < (uint32_t)maxSlices ? L.lastSlices[idx] : (uint32_t)maxSlices;
    if (n) HIP_TRY(c, hipMemcpyAsync(host, L.d_sliceCounts + L.lastSliceOff[idx], (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return (int)n;
}
extern "C" int effort_group_cutoff(effort_ctx* c, int idx, float* host_out) {
    if (!c || !host_out || idx < 0 || (uint32_t)idx >= c->lane[c->lastLane].lastCalls) return EFFORT_ERR_ARG;
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    Lane& L = c->lane[c->lastLane];
    HIP_TRY(c, hipMemcpyAsync(host_out, L.d_cutoff + idx, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EFFORT_OK;
}
extern "C" int effort_last_dispatch_count(effort_ctx* c, uint32_t* host_out) { return effort_group_dispatch_count(c, 0, host_out); }
extern "C" int effort_last_cutoff(effort_ctx* c, float* host_out) { return effort_group_cutoff(c, 0, host_out); }

// ---- dense baseline ------------------------------------------------------------------------------
extern "C" int effort_set_dense_backend(effort_ctx* c, int rocblas) {
    if (!c) return EFFORT_ERR_ARG;
    c->denseRocblas = rocblas != 0;
    return EFFORT_OK;
}
extern "C" int effort_dense_gemv(effort_ctx* c, const void* W, const float* v, float* out, int inDim, int outDim) {
    if (!c || !W || !v || !out || inDim <= 0 || outDim <= 0) return fail(c, EFFORT_ERR_ARG, "dense_gemv: bad argument");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    if (inDim % 16) return fail(c, EFFORT_ERR_SHAPE, "dense_gemv: inDim % 16 != 0 (helpers/mps.swift:18)");
    if (!c->denseRocblas && dense_gemv_supported((uint32_t)inDim, (uint32_t)outDim)) {
        HIP_TRY(c, launch_dense_gemv(static_cast<const uint16_t*>(W), v, out, (uint32_t)inDim, (uint32_t)outDim, c->stream));
        return EFFORT_OK;
    }
    if (!c->blas) {
        if (rocblas_create_handle(&c->blas) != rocblas_status_success) return fail(c, EFFORT_ERR_BLAS, "rocblas_create_handle");
        rocblas_set_stream(c->blas, c->stream);
        rocblas_set_pointer_mode(c->blas, rocblas_pointer_mode_host);
    }
    if ((size_t)inDim > c->vhalfElems) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        hipFree(c->d_vhalf); c->d_vhalf = nullptr; c->vhalfElems = 0;
        HIP_TRY(c, hipMalloc(&c->d_vhalf, (size_t)inDim * 2));
        c->vhalfElems = inDim;
    }
    HIP_TRY(c, launch_f32_to_f16(v, c->d_vhalf, inDim, c->stream));                  // v.asFloat16(), mps.swift:19
    // W is [outDim][inDim] row-major == column-major inDim x outDim with lda = inDim: y = A^T x
    const float alpha = 1.0f, beta = 0.0f;
    rocblas_status s = rocblas_hssgemv_strided_batched(c->blas, rocblas_operation_transpose, inDim, outDim, &alpha,
                                                       static_cast<const rocblas_half*>(W), inDim, 0,
                                                       reinterpret_cast<const rocblas_half*>(c->d_vhalf), 1, 0, &beta, out, 1, 0, 1);
    if (s != rocblas_status_success) return fail(c, EFFORT_ERR_BLAS, "rocblas_hssgemv_strided_batched");
    return EFFORT_OK;
}

// ---- converter -----------------------------------------------------------------------------------
extern "C" int effort_convert_fp16_pitched(effort_ctx* c, const void* W, int outDim, int inDim, void* buckets, int rowPitchBytes, void* stats, void* probes) {
    if (!c || !W || !buckets || !stats || !probes) return fail(c, EFFORT_ERR_ARG, "convert_fp16: null argument");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    // convert.swift:210-215,239 + the 16384-wide limit of the multiply (bucketMul.swift:52)
    if (outDim <= 0 || inDim < kProbes || !(outDim >= kProbes || kProbes % outDim == 0) || outDim > 16384 || inDim > 32000 ||
        outDim % 16 || (outDim / 16) % 4)
        return fail(c, EFFORT_ERR_CONVERT, "convert_fp16: bucketize preconditions violated");
    const int pitch = rowPitchBytes ? rowPitchBytes : outDim / 16 * 2;
    if (pitch < outDim / 16 * 2 || pitch % 8) return fail(c, EFFORT_ERR_CONVERT, "convert_fp16: row pitch must be a multiple of 8 bytes >= 2*cols");   // (the stats pass reads 8 bytes at a time)
    const size_t elems = (size_t)outDim * inDim;
    if (elems > c->convElems) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        hipFree(c->d_convVals); c->d_convVals = nullptr; c->convElems = 0;
        HIP_TRY(c, hipMalloc(&c->d_convVals, elems * 2));
        c->convElems = elems;
    }
    HIP_TRY(c, launch_convert_fp16(static_cast<const uint16_t*>(W), outDim, inDim, static_cast<uint16_t*>(buckets), (uint32_t)pitch / 2u,
                                   static_cast<uint16_t*>(stats), static_cast<uint16_t*>(probes), c->d_convVals, c->d_status, c->stream));
    return EFFORT_OK;
}
extern "C" int effort_convert_fp16(effort_ctx* c, const void* W, int outDim, int inDim, void* buckets, void* stats, void* probes) {
    return effort_convert_fp16_pitched(c, W, outDim, inDim, buckets, 0, stats, probes);
}

// q4_draft.convert (q4_draft.py:70-322) on the GPU: core2 = W.T, f16 [inDim][outDim].
extern "C" int64_t effort_q4_outlier_count(int inDim, int outDim, double perc) {
    if (inDim <= 0 || outDim <= 0 || !(perc >= 0.0 && perc <= 1.0)) return EFFORT_ERR_ARG;
    return (int64_t)((double)((int64_t)inDim * outDim) * perc);           // int(len(flat) * perc), q4_draft.py:76
}
extern "C" int effort_convert_q4(effort_ctx* c, const void* core2, int inDim, int outDim, double perc, void* buckets, void* stats, void* probes,
                                 void* outliers) {
    if (!c || !core2 || !buckets || !stats || !probes) return fail(c, EFFORT_ERR_ARG, "convert_q4: null argument");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    if (inDim <= 0 || outDim <= 0 || outDim % 32 || (int64_t)inDim * outDim >= (1ll << 32) || !(perc >= 0.0 && perc <= 1.0))
        return fail(c, EFFORT_ERR_CONVERT, "convert_q4: outDim % 32 != 0 (q4_draft.py:299), or a matrix of 2^32 elements or more");
    const int64_t cnt = effort_q4_outlier_count(inDim, outDim, perc);
    if (cnt >= (1ll << 31)) return fail(c, EFFORT_ERR_CONVERT, "convert_q4: 2^31 outliers or more (the candidate sort counts in int)");
    if (cnt > 0 && !outliers) return fail(c, EFFORT_ERR_ARG, "convert_q4: outliers buffer missing");
    hipSetDevice(c->device);
    HIP_TRY(c, launch_convert_q4(static_cast<const uint16_t*>(core2), (uint32_t)inDim, (uint32_t)outDim, (uint32_t)cnt, static_cast<uint16_t*>(buckets),
                                 static_cast<float*>(stats), static_cast<uint16_t*>(probes), static_cast<float*>(outliers), c->numCU, c->stream));
    return EFFORT_OK;
}

extern "C" int effort_cosine(effort_ctx* c, const float* a, const float* b, int n, float* host_out) {
    if (!c || !a || !b || !host_out || n <= 0) return EFFORT_ERR_ARG;
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    HIP_TRY(c, launch_cosine(a, b, n, c->d_cos, c->stream));
    HIP_TRY(c, hipMemcpyAsync(host_out, c->d_cos, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EFFORT_OK;
}

// ---- decode-loop glue (runNetwork.swift:68-316; kernels in decode.hip) ---------------------------------------
extern "C" int effort_add_rmsnorm_mul(effort_ctx* c, float* h, const float* delta, const void* w, float* out, int n) {
    if (!c || !h || !w || !out || n <= 0) return fail(c, EFFORT_ERR_ARG, "add_rmsnorm_mul: bad argument");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    HIP_TRY(c, launch_add_rmsnorm_mul(h, delta, static_cast<const uint16_t*>(w), out, (uint32_t)n, c->stream));
    return EFFORT_OK;
}
extern "C" int effort_rope_kv(effort_ctx* c, const float* xq, const float* xk, const float* xv, float* qOut, float* kCache,
                              float* vCache, const uint32_t* pos, int numHeads, int numHeadsKV, int headDim, int maxTokens, float ropeBase) {
    if (!c || !xq || !xk || !xv || !qOut || !kCache || !vCache || !pos) return fail(c, EFFORT_ERR_ARG, "rope_kv: null argument");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    if (numHeads <= 0 || numHeadsKV <= 0 || numHeads % numHeadsKV || headDim < 2 || headDim > 1024 || headDim % 2 || !(ropeBase > 1.0f) || maxTokens <= 0)
        return fail(c, EFFORT_ERR_SHAPE, "rope_kv: bad head geometry");
    HIP_TRY(c, launch_rope_kv(xq, xk, xv, qOut, kCache, vCache, pos, numHeads, numHeadsKV, headDim, ropeBase, (uint32_t)maxTokens, c->d_status + 1, c->stream));
    return EFFORT_OK;
}
extern "C" int effort_attention(effort_ctx* c, const float* q, const float* kCache, const float* vCache, const uint32_t* pos,
                                float* out, int numHeads, int headDim, int maxTokens) {
    if (!c || !q || !kCache || !vCache || !pos || !out) return fail(c, EFFORT_ERR_ARG, "attention: null argument");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    if (numHeads <= 0 || maxTokens <= 0 || maxTokens > 8192 || (headDim != 64 && headDim != 128 && headDim != 256))
        return fail(c, EFFORT_ERR_SHAPE, "attention: headDim 64/128/256, maxTokens <= 8192");
    HIP_TRY(c, launch_attention(q, kCache, vCache, pos, out, numHeads, headDim, maxTokens, c->stream));
    return EFFORT_OK;
}
extern "C" int effort_rope_attention(effort_ctx* c, const float* xq, const float* xk, const float* xv, float* kCache, float* vCache,
                                     const uint32_t* pos, float* out, int numHeads, int numHeadsKV, int headDim, int maxTokens, float ropeBase) {
    if (!c || !xq || !xk || !xv || !kCache || !vCache || !pos || !out) return fail(c, EFFORT_ERR_ARG, "rope_attention: null argument");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    if (numHeads <= 0 || numHeadsKV <= 0 || numHeads % numHeadsKV || maxTokens <= 0 || maxTokens > 8192 || !(ropeBase > 1.0f) ||
        (headDim != 64 && headDim != 128 && headDim != 256))
        return fail(c, EFFORT_ERR_SHAPE, "rope_attention: headDim 64/128/256, maxTokens <= 8192");
    HIP_TRY(c, launch_rope_attention(xq, xk, xv, kCache, vCache, pos, out, numHeads, numHeadsKV, headDim, maxTokens, ropeBase, c->d_status + 1, c->stream));
    return EFFORT_OK;
}
extern "C" int effort_silu_mul(effort_ctx* c, const float* x1, const float* x3, float* out, int n) {
    if (!c || !x1 || !x3 || !out || n <= 0) return fail(c, EFFORT_ERR_ARG, "silu_mul: bad argument");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    HIP_TRY(c, launch_silu_mul(x1, x3, out, (uint32_t)n, c->stream));
    return EFFORT_OK;
}
extern "C" int effort_fetch_row(effort_ctx* c, const void* emb, const uint32_t* id, float* out, int n) {
    if (!c || !emb || !id || !out || n <= 0) return fail(c, EFFORT_ERR_ARG, "fetch_row: bad argument");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    HIP_TRY(c, launch_fetch_row(static_cast<const uint16_t*>(emb), id, out, (uint32_t)n, c->stream));
    return EFFORT_OK;
}
extern "C" int effort_top2_softmax(effort_ctx* c, const float* gate, int n, uint32_t* idx2, float* val2) {
    if (!c || !gate || !idx2 || !val2 || n < 1) return fail(c, EFFORT_ERR_ARG, "top2_softmax: bad argument");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    HIP_TRY(c, launch_top2_softmax(gate, (uint32_t)n, idx2, val2, c->stream));
    return EFFORT_OK;
}
extern "C" int effort_mix2(effort_ctx* c, const float* f0, const float* f1, const float* val2, float* out, int n) {
    if (!c || !f0 || !f1 || !val2 || !out || n <= 0) return fail(c, EFFORT_ERR_ARG, "mix2: bad argument");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    HIP_TRY(c, launch_mix2(f0, f1, val2, out, (uint32_t)n, c->stream));
    return EFFORT_OK;
}
extern "C" int effort_argmax(effort_ctx* c, const float* logits, int n, uint32_t* idOut, uint32_t* pos, uint32_t* history, int historyLen) {
    if (!c || !logits || !idOut || !pos || n <= 0 || (history && historyLen <= 0)) return fail(c, EFFORT_ERR_ARG, "argmax: bad argument");
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    HIP_TRY(c, launch_argmax(logits, (uint32_t)n, idOut, pos, history, (uint32_t)(history ? historyLen : 0), c->d_status + 1, c->stream));
    return EFFORT_OK;
}
// Device-side conditions the decode glue could not report through a return code (the position lives in device memory):
// bit 0 = a step ran past the key/value cache or the history buffer (nothing was written there), bit 1 = argmax over NaN
// logits (token 0 returned).  Reads and clears the word.
extern "C" int effort_decode_status(effort_ctx* c, int* host_out) {
    if (!c || !host_out) return EFFORT_ERR_ARG;
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    HIP_TRY(c, hipMemcpyAsync(host_out, c->d_status + 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_status + 1, 0, 4, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EFFORT_OK;
}
// Rows of the last effort_convert_fp16 calls whose bucket 0 overflowed in the literal preBucketize path (zero padding
// tying with real zeros, convert.metal:40-61: the reference drops those elements silently).  Reads and clears the count.
extern "C" int effort_convert_status(effort_ctx* c, int* host_out) {
    if (!c || !host_out) return EFFORT_ERR_ARG;
    { const int jrc = join_lanes(c); if (jrc != EFFORT_OK) return jrc; }
    HIP_TRY(c, hipMemcpyAsync(host_out, c->d_status, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_status, 0, 4, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EFFORT_OK;
}

// ---- tuning / timing -----------------------------------------------------------------------------
extern "C" int effort_set_tuning(effort_ctx* c, int W, int E, int S) {
    if (!c) return EFFORT_ERR_ARG;
    if ((W || E) && !supported(W ? W : 16, E ? E : 1)) return fail(c, EFFORT_ERR_ARG, "set_tuning: unsupported (waves, elems)");
    if (S < 0) return EFFORT_ERR_ARG;
    c->tuneW = W; c->tuneE = E; c->tuneS = S;
    return EFFORT_OK;
}

extern "C" int effort_debug_occupancy(effort_ctx* c, int q4, int W, int E, int ldsBytes) {
    if (!c || !supported(W, E)) return EFFORT_ERR_ARG;
    return bucket_mul_occupancy(q4 ? kQ4 : kFp16, W, E, (size_t)ldsBytes);
}

extern "C" int effort_set_persistent(effort_ctx* c, int wgPerCU) {
    if (!c || wgPerCU < -1 || wgPerCU > 8) return EFFORT_ERR_ARG;
    c->persistent = wgPerCU;
    return EFFORT_OK;
}

extern "C" int effort_debug_hook_lane(effort_ctx* c, int lane) {
    if (!c || lane < 0 || lane >= c->nLanes) return EFFORT_ERR_ARG;
    c->lastLane = lane;
    return EFFORT_OK;
}

extern "C" int effort_set_row_reuse(effort_ctx* c, int reuse) {
    if (!c) return EFFORT_ERR_ARG;
    c->rowReuse = reuse != 0;
    return EFFORT_OK;
}

extern "C" int effort_set_split_cutoff(effort_ctx* c, int split) {
    if (!c) return EFFORT_ERR_ARG;
    c->splitCutoff = split != 0;
    return EFFORT_OK;
}

// The device-clock stamps and the per-item trace live in the LAB library only (libeffort_hip_lab.so, -DEFFORT_LAB; csrc/Makefile): the
// shipped kernels carry no stamp code at all.  In the shipped library mode 1 times launches with HIP events alone and modes 2 / 3 are refused.
#ifdef EFFORT_LAB
static constexpr bool kLabBuild = true;
#else
static constexpr bool kLabBuild = false;
#endif
static int lab_only(effort_ctx* c) { return fail(c, EFFORT_ERR_KIND, "device-clock stamps / traces are compiled into libeffort_hip_lab.so only (EFFORT_HIP_LIB=lab)"); }
extern "C" int effort_is_lab_build(void) { return kLabBuild ? 1 : 0; }

extern "C" int effort_enable_kernel_timing(effort_ctx* c, int enable) {
    if (!c) return EFFORT_ERR_ARG;
    if (!kLabBuild && enable >= 2) return lab_only(c);
    if (enable == 1) { int rc = ensure_timing(c); if (rc != EFFORT_OK) return rc; }
    c->timing = enable == 1;      // 1: events + device clock, 2: device clock only (graph-capture safe), 3: 2 + per-item trace
    c->clock = kLabBuild && enable != 0;
    c->trace = kLabBuild && enable == 3;
    if (c->trace) HIP_TRY(c, hipMemsetAsync(c->d_tstamp + kTraceOff, 0, (size_t)kTraceItems * 96, c->stream));
    c->nSamples = 0;
    HIP_TRY(c, hipMemsetAsync(c->d_tstamp, 0, 4096, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_tstamp, 0xFF, 8, c->stream));
    return EFFORT_OK;
}

extern "C" int effort_debug_stamps(effort_ctx* c, unsigned long long* host32) {
    if (!c || !host32) return EFFORT_ERR_ARG;
    if (!kLabBuild) return lab_only(c);
    HIP_TRY(c, hipMemcpyAsync(host32, c->d_tstamp + 8, 192, hipMemcpyDeviceToHost, c->stream));
    unsigned long long lines[32 * 8];
    HIP_TRY(c, hipMemcpyAsync(lines, c->d_tstamp + 64, sizeof(lines), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 8; i++) { host32[24 + i] = 0; for (int l = 0; l < 32; l++) host32[24 + i] += lines[l * 8 + i]; }   // all-workgroup phase sums
    return EFFORT_OK;
}

extern "C" int effort_debug_trace(effort_ctx* c, unsigned long long* host, int maxRecords) {
    if (!c || !host || maxRecords < 1 || maxRecords > kTraceItems) return EFFORT_ERR_ARG;
    if (!kLabBuild) return lab_only(c);
    HIP_TRY(c, hipMemcpyAsync(host, c->d_tstamp + kTraceOff, (size_t)maxRecords * 64, hipMemcpyDeviceToHost, c->stream));
    // (then, per item, four progress stamps of its streaming phase: wave 0 a quarter / half / three quarters through its rows)
    HIP_TRY(c, hipMemcpyAsync(host + (size_t)maxRecords * 8, c->d_tstamp + kTraceOff + (size_t)kTraceItems * 8, (size_t)maxRecords * 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return EFFORT_OK;
}

extern "C" int effort_kernel_clock(effort_ctx* c, double* mul_us_avg, int* n_launches) {
    if (!c) return EFFORT_ERR_ARG;
    if (!kLabBuild) return lab_only(c);
    unsigned long long h[4] = {0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(h, c->d_tstamp, 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (mul_us_avg) *mul_us_avg = h[3] ? (double)h[2] / (double)h[3] * 1000.0 / c->wallClockKHz : 0.0;
    if (n_launches) *n_launches = (int)h[3];
    HIP_TRY(c, hipMemsetAsync(c->d_tstamp + 2, 0, 16, c->stream));
    return EFFORT_OK;
}

extern "C" int effort_kernel_timing(effort_ctx* c, double* mul_us, double* cutoff_us, double* integrate_us, int* n) {
    if (!c) return EFFORT_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double a = 0, b = 0, d = 0;
    for (int i = 0; i < c->nSamples; i++) {
        float ms;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[4 * i + 0], c->ev[4 * i + 1])); b += ms;   // the one fused kernel
    }
    const int ns = c->nSamples ? c->nSamples : 1;
    if (cutoff_us) *cutoff_us = a * 1000.0 / ns;
    if (mul_us) *mul_us = b * 1000.0 / ns;
    if (integrate_us) *integrate_us = d * 1000.0 / ns;
    if (n) *n = c->nSamples;
    c->nSamples = 0;
    return EFFORT_OK;
}
