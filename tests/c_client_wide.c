/*
 * c_client_wide.c -- the REST of the surface a Swift host would bind, called from plain C (effort_hip.h + the HIP runtime API,
 * nothing else; see tests/c_client.c for the basic FP16 call).  Three modes:
 *
 *   c_client_wide q4    core2.f16 v.f32 inDim outDim effort out.f32
 *       bucketMulQ4(v:by:expNo:out:effort:) (bucketMulQ4.swift:11-17) on a bundle made by effort_convert_q4 (= q4_draft.convert,
 *       q4_draft.py:70-322, 2 % outliers): convert -> effort_weights_q4 -> effort_bucketmul_q4 -> hooks.  core2 = W.T, f16
 *       [inDim][outDim].  Also dumps the converted bundle next to out.f32 (.buckets / .stats / .probes / .outliers) so that the
 *       checker can compare the layout with the oracle's byte for byte.
 *   c_client_wide group Wq.f16 Wk.f16 Wv.f16 v.f32 inDim outQ outK outV effort out.f32
 *       the decode loop's Wq|Wk|Wv on ONE input vector as ONE launch (runNetwork.swift:132-134): effort_bucketmul_group on three
 *       handles sharing v; out.f32 = the three outputs one after the other; prints every call's dispatch size and cutoff bits.
 *   c_client_wide shard W.f16 v.f32 inDim outDim effort world out.f32
 *       north_star's "thin C-ABI shim ... RCCL all-gather": effort_comm_unique_id -> effort_comm_create(rank 0 of a world of ONE:
 *       this box has one GPU) -> effort_weights_column_shard(r, world) for every r -> one multiply per shard into its slice of
 *       the send buffer (what rank r would run) -> effort_allgather_outputs.  The RCCL the library opens here is the system
 *       ROCm's (no PyTorch in this process).
 *
 * Every mode prints "dispatch <n> cutoff <hex bits>" per call; tests/test_c_client.py (-m gpu) checks the dumps against the oracle.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "effort_hip.h"

static void* read_file(const char* path, size_t bytes) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "c_client_wide: cannot open %s\n", path); exit(2); }
    void* p = malloc(bytes);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "c_client_wide: %s is not %zu bytes\n", path, bytes); exit(2); }
    fclose(f);
    return p;
}
static int write_file(const char* path, const char* suffix, const void* p, size_t bytes) {
    char name[4096];
    snprintf(name, sizeof(name), "%s%s", path, suffix);
    FILE* f = fopen(name, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "c_client_wide: cannot write %s\n", name); return 2; }
    fclose(f);
    return 0;
}
static uint32_t bits_of(float x) { uint32_t b; memcpy(&b, &x, 4); return b; }

#define HIP_OK(call)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) { fprintf(stderr, "c_client_wide: %s: %s\n", #call, hipGetErrorString(e_)); return 3; } \
    } while (0)
#define EFFORT_OK_OR_DIE(call)                                                                 \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != EFFORT_OK) { fprintf(stderr, "c_client_wide: %s = %d (%s)\n", #call, rc_, effort_last_error(ctx)); return 4; } \
    } while (0)

/* device buffer holding a host file's contents */
static void* upload(const char* path, size_t bytes) {
    void* h = read_file(path, bytes);
    void* d = NULL;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "c_client_wide: upload of %s failed\n", path); exit(3); }
    free(h);
    return d;
}
static int download(const char* path, const char* suffix, const void* d, size_t bytes) {
    void* h = malloc(bytes);
    if (!h || hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "c_client_wide: download failed\n"); return 3; }
    const int rc = write_file(path, suffix, h, bytes);
    free(h);
    return rc;
}

static int mode_q4(effort_ctx* ctx, char** a) {
    const int inDim = atoi(a[2]), outDim = atoi(a[3]);
    const double effort = atof(a[4]);
    const char* outPath = a[5];
    const size_t rows = (size_t)inDim * 8, cols = (size_t)outDim / 32;
    void* dCore = upload(a[0], (size_t)inDim * outDim * 2);
    void* dV = upload(a[1], (size_t)inDim * 4);
    const int64_t nOl = effort_q4_outlier_count(inDim, outDim, 0.02);                 /* int(len(flat) * perc), q4_draft.py:76 */
    if (nOl < 0) return 4;
    const int nProbes = inDim < outDim ? inDim : outDim;
    void *dB, *dS, *dP, *dO, *dOut;
    HIP_OK(hipMalloc(&dB, rows * cols * 2));
    HIP_OK(hipMalloc(&dS, rows * 2 * 4));
    HIP_OK(hipMalloc(&dP, (size_t)nProbes * 2));
    HIP_OK(hipMalloc(&dO, (size_t)(nOl ? nOl : 1) * 16));
    HIP_OK(hipMalloc(&dOut, (size_t)outDim * 4));
    HIP_OK(hipMemset(dOut, 0xFF, (size_t)outDim * 4));                               /* NaNs: the call itself zeroes out (expertMul.swift:27) */
    EFFORT_OK_OR_DIE(effort_convert_q4(ctx, dCore, inDim, outDim, 0.02, dB, dS, dP, dO));
    effort_w* w = effort_weights_q4(ctx, dB, dS, dP, dO, nOl, inDim, outDim, 1);      /* loader.swift:70,98,124 */
    if (!w) { fprintf(stderr, "c_client_wide: effort_weights_q4: %s\n", effort_last_error(ctx)); return 4; }
    /* an FP16 call on a Q4 bundle must be refused, not run (expertMul.swift:24-38 routes by the bundle's kind) */
    if (effort_bucketmul(ctx, w, (const float*)dV, NULL, (float*)dOut, effort) != EFFORT_ERR_KIND) { fprintf(stderr, "c_client_wide: kind check\n"); return 5; }
    EFFORT_OK_OR_DIE(effort_bucketmul_q4(ctx, w, (const float*)dV, NULL, (float*)dOut, effort));
    EFFORT_OK_OR_DIE(effort_sync(ctx));
    uint32_t count = 0; float cutoff = 0.0f;
    EFFORT_OK_OR_DIE(effort_last_dispatch_count(ctx, &count));
    EFFORT_OK_OR_DIE(effort_last_cutoff(ctx, &cutoff));
    printf("dispatch %u cutoff %08x outliers %lld\n", count, bits_of(cutoff), (long long)nOl);
    int rc = download(outPath, "", dOut, (size_t)outDim * 4);
    if (!rc) rc = download(outPath, ".buckets", dB, rows * cols * 2);
    if (!rc) rc = download(outPath, ".stats", dS, rows * 2 * 4);
    if (!rc) rc = download(outPath, ".probes", dP, (size_t)nProbes * 2);
    if (!rc && nOl) rc = download(outPath, ".outliers", dO, (size_t)nOl * 16);
    effort_weights_free(w);
    hipFree(dCore); hipFree(dV); hipFree(dB); hipFree(dS); hipFree(dP); hipFree(dO); hipFree(dOut);
    return rc;
}

/* convert an HF matrix [outDim][inDim] on the GPU and register it (bucketize() + ExpertWeights) */
static effort_w* make_fp16(effort_ctx* ctx, const char* path, int inDim, int outDim, void** keep /* [4] device buffers to free */) {
    const size_t rows = (size_t)inDim * 16, cols = (size_t)outDim / 16;
    keep[0] = upload(path, (size_t)inDim * outDim * 2);
    if (hipMalloc(&keep[1], rows * cols * 2) != hipSuccess || hipMalloc(&keep[2], rows * 8) != hipSuccess || hipMalloc(&keep[3], 4096 * 2) != hipSuccess) return NULL;
    if (effort_convert_fp16(ctx, keep[0], outDim, inDim, keep[1], keep[2], keep[3]) != EFFORT_OK || effort_sync(ctx) != EFFORT_OK) return NULL;
    return effort_weights_fp16(ctx, keep[1], keep[2], keep[3], inDim, outDim, 16, 1);
}

static int mode_group(effort_ctx* ctx, char** a) {
    const int inDim = atoi(a[4]);
    const int outDims[3] = {atoi(a[5]), atoi(a[6]), atoi(a[7])};
    const double effort = atof(a[8]);
    void* keep[3][4];
    const effort_w* ws[3];
    const float* vs[3];
    float* outs[3];
    double efforts[3];
    void* dV = upload(a[3], (size_t)inDim * 4);
    size_t total = 0;
    for (int i = 0; i < 3; i++) total += (size_t)outDims[i];
    void* dOut;
    HIP_OK(hipMalloc(&dOut, total * 4));
    size_t off = 0;
    for (int i = 0; i < 3; i++) {
        ws[i] = make_fp16(ctx, a[i], inDim, outDims[i], keep[i]);
        if (!ws[i]) { fprintf(stderr, "c_client_wide: matrix %d: %s\n", i, effort_last_error(ctx)); return 4; }
        vs[i] = (const float*)dV; outs[i] = (float*)dOut + off; efforts[i] = effort;
        off += (size_t)outDims[i];
    }
    /* runNetwork.swift:132-134: three expertMul calls on one h_norm -- here ONE launch */
    EFFORT_OK_OR_DIE(effort_bucketmul_group(ctx, 3, ws, vs, NULL, outs, efforts));
    EFFORT_OK_OR_DIE(effort_sync(ctx));
    for (int i = 0; i < 3; i++) {
        uint32_t count = 0; float cutoff = 0.0f;
        EFFORT_OK_OR_DIE(effort_group_dispatch_count(ctx, i, &count));
        EFFORT_OK_OR_DIE(effort_group_cutoff(ctx, i, &cutoff));
        printf("dispatch %u cutoff %08x\n", count, bits_of(cutoff));
    }
    /* a group of 33 is refused with a code, nothing enqueued (the reference would assert) */
    if (effort_bucketmul_group(ctx, 33, ws, vs, NULL, outs, efforts) != EFFORT_ERR_ARG) { fprintf(stderr, "c_client_wide: group size check\n"); return 5; }
    const int rc = download(a[9], "", dOut, total * 4);
    for (int i = 0; i < 3; i++) { effort_weights_free((effort_w*)ws[i]); for (int k = 0; k < 4; k++) hipFree(keep[i][k]); }
    hipFree(dV); hipFree(dOut);
    return rc;
}

static int mode_shard(effort_ctx* ctx, char** a) {
    const int inDim = atoi(a[2]), outDim = atoi(a[3]);
    const double effort = atof(a[4]);
    const int world = atoi(a[5]);
    if (world < 1 || world > 16) return 1;
    void* keep[4];
    effort_w* full = make_fp16(ctx, a[0], inDim, outDim, keep);
    if (!full) { fprintf(stderr, "c_client_wide: %s\n", effort_last_error(ctx)); return 4; }
    void* dV = upload(a[1], (size_t)inDim * 4);
    /* the communicator: rank 0 makes the id, the host ships it (here: nowhere -- a world of one rank, this box's GPU) */
    unsigned char id[EFFORT_COMM_ID_BYTES];
    EFFORT_OK_OR_DIE(effort_comm_unique_id(id));
    EFFORT_OK_OR_DIE(effort_comm_create(ctx, 0, 1, id));
    if (effort_comm_rank(ctx) != 0 || effort_comm_world(ctx) != 1) return 5;
    void *dSend, *dRecv;
    HIP_OK(hipMalloc(&dSend, (size_t)outDim * 4));
    HIP_OK(hipMalloc(&dRecv, (size_t)outDim * 4));
    HIP_OK(hipMemset(dRecv, 0xFF, (size_t)outDim * 4));
    const int per = outDim / world;
    effort_w* shard[16];
    for (int r = 0; r < world; r++) {                             /* what rank r of `world` would run; here one after the other */
        shard[r] = effort_weights_column_shard(full, r, world);
        if (!shard[r]) { fprintf(stderr, "c_client_wide: column_shard(%d, %d): %s\n", r, world, effort_last_error(ctx)); return 4; }
        EFFORT_OK_OR_DIE(effort_bucketmul(ctx, shard[r], (const float*)dV, NULL, (float*)dSend + (size_t)r * per, effort));
        EFFORT_OK_OR_DIE(effort_sync(ctx));
        uint32_t count = 0; float cutoff = 0.0f;
        EFFORT_OK_OR_DIE(effort_last_dispatch_count(ctx, &count));
        EFFORT_OK_OR_DIE(effort_last_cutoff(ctx, &cutoff));
        printf("dispatch %u cutoff %08x\n", count, bits_of(cutoff));
    }
    /* ONE gather of the whole vector (a world of one: every "rank's" slice sits in this rank's send buffer) */
    EFFORT_OK_OR_DIE(effort_allgather_outputs(ctx, (const float*)dSend, (float*)dRecv, outDim));
    EFFORT_OK_OR_DIE(effort_sync(ctx));
    const int rc = download(a[6], "", dRecv, (size_t)outDim * 4);
    effort_weights_free(full);                                      /* before its views: the library defers it */
    for (int r = 0; r < world; r++) effort_weights_free(shard[r]);
    EFFORT_OK_OR_DIE(effort_comm_destroy(ctx));
    for (int k = 0; k < 4; k++) hipFree(keep[k]);
    hipFree(dV); hipFree(dSend); hipFree(dRecv);
    return rc;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: c_client_wide q4|group|shard ...\n"); return 1; }
    HIP_OK(hipSetDevice(0));
    effort_ctx* ctx = effort_create(0, NULL);
    if (!ctx) { fprintf(stderr, "c_client_wide: effort_create failed: %s\n", effort_last_error(NULL)); return 3; }
    int rc = 1;
    if (!strcmp(argv[1], "q4") && argc == 8) rc = mode_q4(ctx, argv + 2);
    else if (!strcmp(argv[1], "group") && argc == 12) rc = mode_group(ctx, argv + 2);
    else if (!strcmp(argv[1], "shard") && argc == 9) rc = mode_shard(ctx, argv + 2);
    else fprintf(stderr, "c_client_wide: bad arguments for mode %s\n", argv[1]);
    effort_destroy(ctx);
    return rc;
}
