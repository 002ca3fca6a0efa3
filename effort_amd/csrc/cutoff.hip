// Standalone findCutoff32 launch -- replaces kernel findCutoff32 (bucketMul.metal:141-247) where the cutoff is
// needed on its own (effort_calc_dispatch).  The algorithm lives in cutoff_device.h and is shared with the
// fused multiply kernel.
#include "cutoff_device.h"

namespace effort {

__global__ __launch_bounds__(1024) void find_cutoff_kernel(const float* __restrict__ v, const uint16_t* __restrict__ probes,
                                                           const uint32_t* __restrict__ expNo, uint32_t q,
                                                           float* __restrict__ cutoff, uint32_t* __restrict__ dispatchCount,
                                                           unsigned long long* __restrict__ tstamp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t e = expNo ? expNo[0] : 0u;
    const uint16_t* pr = probes + (size_t)e * kProbes;
    float vj[4]; uint16_t prj[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { vj[i] = v[threadIdx.x + 1024 * i]; prj[i] = pr[threadIdx.x + 1024 * i]; }
    const float c = block_find_cutoff<1024>(vj, prj, q, smem, reinterpret_cast<uint32_t*>(smem + kCutoffLdsBytes), []() {},
                                            tstamp ? tstamp + 8 : nullptr);
    if (threadIdx.x == 0) {
        cutoff[0] = c;
        dispatchCount[0] = 0;      // dispatch.size.zero(), bucketMul.swift:38
    }
}

// One workgroup per call of a group (split mode of grouped launches): the multiply workgroups then read the
// cutoff instead of each re-deriving it.
__global__ __launch_bounds__(1024) void find_cutoff_group_kernel(const GroupKArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const CallDesc& a = ga.call[blockIdx.x];
    const uint32_t e = a.expNo ? a.expNo[0] : 0u;
    const uint16_t* pr = a.probes + (size_t)e * kProbes;
    float vj[4]; uint16_t prj[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { vj[i] = a.v[threadIdx.x + 1024 * i]; prj[i] = pr[threadIdx.x + 1024 * i]; }
    const float c = block_find_cutoff<1024>(vj, prj, a.q, smem, reinterpret_cast<uint32_t*>(smem + kCutoffLdsBytes), []() {}, nullptr);
    if (threadIdx.x == 0) ga.cutoff[blockIdx.x] = c;
}

static hipError_t cutoff_prepare() {       // (both kernels ask for kCutoffLdsBytes + the table of dynamic LDS)
    hipError_t e = allow_full_lds(reinterpret_cast<const void*>(&find_cutoff_group_kernel));
    return e != hipSuccess ? e : allow_full_lds(reinterpret_cast<const void*>(&find_cutoff_kernel));
}

hipError_t launch_find_cutoff_group(const GroupKArgs& ga, hipStream_t st) {
    hipError_t e = cutoff_prepare();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(find_cutoff_group_kernel, dim3(ga.count), dim3(1024), kCutoffLdsBytes + cutoff_table_bytes(1024), st, ga);
    return hipGetLastError();
}

hipError_t launch_find_cutoff(const float* v, const uint16_t* probes, const uint32_t* expNo, uint32_t q,
                              float* cutoff, uint32_t* dispatchCount, unsigned long long* tstamp, hipStream_t st) {
    hipError_t e = cutoff_prepare();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(find_cutoff_kernel, dim3(1), dim3(1024), kCutoffLdsBytes + cutoff_table_bytes(1024), st, v, probes, expNo, q, cutoff, dispatchCount, tstamp);
    return hipGetLastError();
}

}  // namespace effort
