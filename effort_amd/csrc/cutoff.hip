// findCutoff32 for gfx950 -- replaces kernel findCutoff32 (bucketMul.metal:141-247).
//
// Effort -> scalar cutoff.  The reference runs one 1024-thread threadgroup (32 simdgroups of 32):
// 4096 probe products are rounded to bfloat, then a bisection on a threshold counts how many exceed it
// until the count hits 4096-q (or the bounds/counters converge, or 100 rounds pass).  Here: ONE
// 256-thread workgroup (4 wave64), 16 values per lane held in VGPRs, per-round count = 16 wave
// ballots + popcounts (scalar) and one LDS exchange between the 4 waves.  Every lane carries the
// bisection state redundantly so a round costs a single barrier.  The arithmetic is bit-for-bit
// the reference's: (1e5*v)*bf16(probe) in f32, bf16 rounding, f32 midpoint, the same exit tests in
// the same order, and the value written is the NEXT midpoint (bucketMul.metal:222,230-232).
#include "effort_internal.h"

namespace effort {

__global__ __launch_bounds__(256) void find_cutoff_kernel(const float* __restrict__ v,
                                                          const uint16_t* __restrict__ probes,
                                                          const uint32_t* __restrict__ expNo, uint32_t q,
                                                          float* __restrict__ cutoff,
                                                          uint32_t* __restrict__ dispatchCount,
                                                          unsigned long long* __restrict__ tstamp) {
    __shared__ float s_min[4], s_max[4];
    __shared__ uint32_t s_cnt[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t e = expNo ? expNo[0] : 0u;
    const uint16_t* pr = probes + (size_t)e * kProbes;

    float val[16];
    float mx = -999.0f, mn = 999.0f;                            // bucketMul.metal:155-156
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int j = tid + 256 * k;
        const float t = kCutoffScale * v[j];                    // :160, evaluated left to right
        const float u = t * bf16_round(half_bits_to_float(pr[j]));
        val[k] = bf16_round(fabsf(u));
        mx = fmaxf(mx, val[k]);
        mn = fminf(mn, val[k]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off));
        mn = fminf(mn, __shfl_xor(mn, off));
    }
    if (lane == 0) { s_min[wave] = mn; s_max[wave] = mx; }
    __syncthreads();
    // The reference clamps each simdgroup's min at 999 and stores it as bfloat (-> 1000) before the
    // cross-simdgroup min (:169-190).  All values are bf16, so the net effect is min(globalMin, 1000);
    // the max side starts at -999 and the values are >= 0, so it is the plain global max.
    float minBound = fminf(fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3])), 1000.0f);
    float maxBound = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    float newBound = (minBound + maxBound) / 2;                 // :195

    const uint32_t effort = 4096u - q;                          // :154
    int loops = 0, minCount = 4096, maxCount = 0;               // :175-176,198
    for (;;) {
        loops += 1;
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) c += (uint32_t)__popcll(__ballot(val[k] > newBound));
        if (lane == 0) s_cnt[loops & 1][wave] = c;
        __syncthreads();
        const uint32_t countAbove = s_cnt[loops & 1][0] + s_cnt[loops & 1][1] + s_cnt[loops & 1][2] + s_cnt[loops & 1][3];
        if (countAbove < effort) { maxBound = newBound; maxCount = (int)countAbove; }   // :214-220
        else { minBound = newBound; minCount = (int)countAbove; }
        newBound = (maxBound + minBound) / 2;                                            // :222
        int d = maxCount - minCount; if (d < 0) d = -d;
        if (countAbove == effort || (maxBound - minBound < 0.00001f) || d < 3) break;    // :227-229
        if (loops > 100) break;                                                          // :236
    }
    if (tid == 0) {
        cutoff[0] = newBound;
        dispatchCount[0] = 0;      // dispatch.size.zero(), bucketMul.swift:38
        if (tstamp) { tstamp[0] = ~0ull; tstamp[1] = 0ull; }
    }
}

hipError_t launch_find_cutoff(const float* v, const uint16_t* probes, const uint32_t* expNo, uint32_t q,
                              float* cutoff, uint32_t* dispatchCount, unsigned long long* tstamp, hipStream_t st) {
    hipLaunchKernelGGL(find_cutoff_kernel, dim3(1), dim3(256), 0, st, v, probes, expNo, q, cutoff, dispatchCount, tstamp);
    return hipGetLastError();
}

}  // namespace effort
