/*
 * c_client.c -- a plain-C caller of the drop-in boundary: nothing but include/effort_hip.h and the HIP runtime API, no Python,
 * no ctypes, no torch.  It stands in for the Swift shim of INTEGRATION.md (Swift cannot be built in this image): what
 * `bucketMul(v:by:expNo:out:effort:)` (bucketMul.swift:11-15) does through `Gpu.deploy` / `gpu.eval()` (helpers/gpu.swift:
 * 109-196) it does through the C ABI -- create a context, convert an HF matrix into the bucketed layout on the GPU, register
 * the bundle, multiply, read BucketMul's dispatch size and cutoff, and hand the result back.
 *
 *   c_client W.f16 v.f32 inDim outDim effort out.f32
 *
 * W.f16: [outDim][inDim] f16 row-major (the HF matrix), v.f32: [inDim].  Writes out.f32 [outDim] and prints
 * "dispatch <n> cutoff <bits as hex>".  Compiled as C11 by `make -C effort_amd/csrc c_client` (so the header is also proven
 * to be C, not just C++), run by tests/test_c_client.py (-m gpu), which checks the dump against the oracle.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "effort_hip.h"

static void* read_file(const char* path, size_t bytes) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "c_client: cannot open %s\n", path); exit(2); }
    void* p = malloc(bytes);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "c_client: %s is not %zu bytes\n", path, bytes); exit(2); }
    fclose(f);
    return p;
}

#define HIP_OK(call)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) { fprintf(stderr, "c_client: %s: %s\n", #call, hipGetErrorString(e_)); return 3; } \
    } while (0)
#define EFFORT_OK_OR_DIE(call)                                                                 \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != EFFORT_OK) { fprintf(stderr, "c_client: %s = %d (%s)\n", #call, rc_, effort_last_error(ctx)); return 4; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 7) { fprintf(stderr, "usage: c_client W.f16 v.f32 inDim outDim effort out.f32\n"); return 1; }
    const int inDim = atoi(argv[3]), outDim = atoi(argv[4]);
    const double effort = atof(argv[5]);
    const size_t rows = (size_t)inDim * 16, cols = (size_t)outDim / 16;
    void* W = read_file(argv[1], (size_t)inDim * outDim * 2);
    void* v = read_file(argv[2], (size_t)inDim * 4);

    HIP_OK(hipSetDevice(0));
    effort_ctx* ctx = effort_create(0, NULL);                  /* Gpu() + BucketMul.shared */
    if (!ctx) { fprintf(stderr, "c_client: effort_create failed: %s\n", effort_last_error(NULL)); return 3; }

    void *dW, *dBuckets, *dStats, *dProbes, *dV, *dOut;
    HIP_OK(hipMalloc(&dW, (size_t)inDim * outDim * 2));
    HIP_OK(hipMalloc(&dBuckets, rows * cols * 2));
    HIP_OK(hipMalloc(&dStats, rows * 4 * 2));
    HIP_OK(hipMalloc(&dProbes, 4096 * 2));
    HIP_OK(hipMalloc(&dV, (size_t)inDim * 4));
    HIP_OK(hipMalloc(&dOut, (size_t)outDim * 4));
    HIP_OK(hipMemcpy(dW, W, (size_t)inDim * outDim * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dV, v, (size_t)inDim * 4, hipMemcpyHostToDevice));

    /* bucketize() -- convert.swift:209-260 */
    EFFORT_OK_OR_DIE(effort_convert_fp16(ctx, dW, outDim, inDim, dBuckets, dStats, dProbes));
    EFFORT_OK_OR_DIE(effort_sync(ctx));
    int dropped = -1;
    EFFORT_OK_OR_DIE(effort_convert_status(ctx, &dropped));
    /* ExpertWeights -- loader.swift:46-167 */
    effort_w* w = effort_weights_fp16(ctx, dBuckets, dStats, dProbes, inDim, outDim, 16, 1);
    if (!w) { fprintf(stderr, "c_client: effort_weights_fp16: %s\n", effort_last_error(ctx)); return 4; }
    /* bucketMul(v:by:expNo:out:effort:) + gpu.eval() -- bucketMul.swift:11-15, helpers/gpu.swift:109-119 */
    EFFORT_OK_OR_DIE(effort_bucketmul(ctx, w, (const float*)dV, NULL, (float*)dOut, effort));
    EFFORT_OK_OR_DIE(effort_sync(ctx));
    uint32_t count = 0;
    float cutoff = 0.0f;
    EFFORT_OK_OR_DIE(effort_last_dispatch_count(ctx, &count));   /* dispatch.size, bucketMul.swift:46-47 */
    EFFORT_OK_OR_DIE(effort_last_cutoff(ctx, &cutoff));          /* BucketMul.cutoff, bucketMul.swift:22 */

    float* out = (float*)malloc((size_t)outDim * 4);
    HIP_OK(hipMemcpy(out, dOut, (size_t)outDim * 4, hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[6], "wb");
    if (!f || fwrite(out, 4, (size_t)outDim, f) != (size_t)outDim) { fprintf(stderr, "c_client: cannot write %s\n", argv[6]); return 2; }
    fclose(f);
    uint32_t bits;
    memcpy(&bits, &cutoff, 4);
    printf("dispatch %u cutoff %08x dropped %d version %s\n", count, bits, dropped, effort_version());

    effort_weights_free(w);
    effort_destroy(ctx);
    hipFree(dW); hipFree(dBuckets); hipFree(dStats); hipFree(dProbes); hipFree(dV); hipFree(dOut);
    free(W); free(v); free(out);
    return 0;
}
