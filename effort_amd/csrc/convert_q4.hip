// Q4 weight layout converter on the GPU: q4_draft.convert (q4_draft.py:70-322), one matrix per call.
// The reference converts a matrix with pure-Python loops over every weight (minutes per matrix); this is the same
// transformation in a handful of kernels, bit-identical in buckets, stats and probes and identical -- including the order
// of the table -- in the outlier set (checked against the reference-generated fixtures, tests/golden/q4_*.npz):
//   * the top perc (2 %) weights by |w| become outliers f32 [n][4] = (value, inIdx, outIdx, 0) and are zeroed (:71-102).
//     Ties in |w| are ordered by flat index (the reference's unstable argsort leaves them unspecified): a histogram of the
//     15-bit magnitudes finds the threshold magnitude, the candidates at or above it are collected and radix-sorted
//     (rocPRIM through hipCUB: a plain library sort) by (magnitude descending, flat index ascending);
//   * per input row, every 8 consecutive outputs are sorted by |w| descending, stably (:117-134); rank row
//     inRow*8 + rank collects one element per bucket (:147-168);
//   * bucket.stats = mean |rank row| as numpy computes np.mean of a float16 array -- float32 PAIRWISE accumulation in
//     numpy's order (eight partial sums per block of up to 128, recursive halving above), float32 division, float16
//     result -- stored as f32 in both lanes (:179-194,244-245);
//   * 4-bit codes (w < 0 ? 8 : 0) + position, four per 16-bit word, first bucket in the top nibble (:264-318);
//   * probes = the diagonal after outlier removal (:240).
#include <hipcub/hipcub.hpp>

#include "effort_internal.h"

namespace effort {

constexpr int kMagBins = 32768;

// histogram of the magnitudes (|w| as its 15-bit f16 pattern: monotone for halfs), privatised in LDS
__global__ __launch_bounds__(1024) void q4_hist_kernel(const uint16_t* __restrict__ core, size_t n, uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t sh[];
    for (int i = threadIdx.x; i < kMagBins; i += 1024) sh[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024) atomicAdd(&sh[core[i] & 0x7FFFu], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < kMagBins; i += 1024) if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// res[0] = m*: the largest magnitude with count(mag >= m*) >= cnt; res[1] = count(mag > m*); res[2] = count(mag >= m*)
__global__ __launch_bounds__(1024) void q4_threshold_kernel(const uint32_t* __restrict__ hist, uint32_t cnt, uint32_t* __restrict__ res) {
    __shared__ uint32_t part[1024];
    // thread t owns bins [32*t, 32*t + 32), scanned from the top
    uint32_t own = 0;
    for (int b = 0; b < 32; b++) own += hist[threadIdx.x * 32 + b];
    part[threadIdx.x] = own;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int t = 1023; t >= 0; t--) { const uint32_t c = part[t]; part[t] = run; run += c; } }   // part[t] = count in bins above thread t's
    __syncthreads();
    uint32_t above = part[threadIdx.x];
    for (int b = 31; b >= 0; b--) {
        const uint32_t m = threadIdx.x * 32 + b, c = hist[m];
        if (above < cnt && above + c >= cnt) { res[0] = m; res[1] = above; res[2] = above + c; }
        above += c;
    }
}

// candidates: every element with magnitude >= m*, as key (32767 - mag) << 32 | flat index (ascending key = the table's order; the sort
// that follows fixes the order, so where a candidate lands here is free).  A thread takes eight elements (one 16-byte load), the block
// counts its candidates in LDS and reserves their places with ONE global atomic per 2048 elements.  (Round 5's version asked the global
// counter once per wave and 64 elements -- half a million atomics on one address for a 4096 x 11008 matrix: 5.8 ms, two thirds of a
// conversion's GPU time; profiles/r05_q4_rocprofv3_kernel_stats_16_per_launch.csv.)
__global__ __launch_bounds__(256) void q4_collect_kernel(const uint16_t* __restrict__ core, size_t n, const uint32_t* __restrict__ res,
                                                         unsigned long long* __restrict__ keys, uint32_t* __restrict__ counter) {
    __shared__ uint32_t s_cnt, s_base;
    const uint32_t mstar = res[0];
    constexpr size_t kChunk = 256 * 8;
    for (size_t c0 = (size_t)blockIdx.x * kChunk; c0 < n; c0 += (size_t)gridDim.x * kChunk) {       // (n is a multiple of 32: outDim % 32 == 0)
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const size_t i = c0 + (size_t)threadIdx.x * 8;
        uint32_t mag[8], t = 0;
        if (i < n) {
            const uint4 w = *reinterpret_cast<const uint4*>(core + i);
            const uint32_t d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 8; k++) { mag[k] = (d[k >> 1] >> (16 * (k & 1))) & 0x7FFFu; t += mag[k] >= mstar ? 1u : 0u; }
        }
        uint32_t my = t ? atomicAdd(&s_cnt, t) : 0u;
        __syncthreads();
        if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(counter, s_cnt) : 0u;
        __syncthreads();
        if (t) {
            my += s_base;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (mag[k] >= mstar) keys[my++] = ((unsigned long long)(32767u - mag[k]) << 32) | (unsigned long long)(i + k);
        }
    }
}

__global__ void q4_emit_outliers_kernel(const unsigned long long* __restrict__ keys, uint32_t cnt, uint16_t* __restrict__ core, uint32_t outDim,
                                        float4* __restrict__ table) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cnt) return;
    const uint32_t idx = (uint32_t)keys[i];
    table[i] = make_float4(half_bits_to_float(core[idx]), (float)(idx / outDim), (float)(idx % outDim), 0.0f);     // rearrangeOutliers, :58-67
    core[idx] = 0;
}

// one thread per (input row, 16-bit word): 4 buckets of 8 outputs -> a word of each of the row's 8 rank rows, and the sorted
// magnitudes (f16 bits) for the stats
__global__ __launch_bounds__(256) void q4_bucket_kernel(const uint16_t* __restrict__ core, uint32_t inDim, uint32_t outDim,
                                                        uint16_t* __restrict__ words, uint16_t* __restrict__ absRows) {
    const uint32_t wpr = outDim / 32, nb = outDim / 8;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)inDim * wpr) return;
    const uint32_t row = (uint32_t)(t / wpr), q = (uint32_t)(t % wpr);
    const uint4* src = reinterpret_cast<const uint4*>(core + (size_t)row * outDim + (size_t)q * 32);
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int gq = 0; gq < 4; gq++) {
        const uint4 v = src[gq];
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
        uint32_t h[8];
#pragma unroll
        for (int i = 0; i < 8; i++) h[i] = (d[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t mi = h[i] & 0x7FFFu;
            uint32_t rank = 0;                              // stable descending order: larger magnitudes first, earlier positions first on ties
#pragma unroll
            for (int j = 0; j < 8; j++) { const uint32_t mj = h[j] & 0x7FFFu; rank += (mj > mi || (mj == mi && j < i)) ? 1u : 0u; }
            const uint32_t nib = ((h[i] & 0x8000u) && mi ? 8u : 0u) + (uint32_t)i;     // (w < 0): -0.0 is not negative
#pragma unroll
            for (int r = 0; r < 8; r++) if (rank == (uint32_t)r) { w[r] |= nib << (12 - 4 * gq); absRows[((size_t)row * 8 + r) * nb + (size_t)q * 4 + gq] = (uint16_t)mi; }
        }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) words[((size_t)row * 8 + r) * wpr + q] = (uint16_t)w[r];
}

// numpy's pairwise summation (loops_utils.h.src, PW_BLOCKSIZE 128) of n non-negative halfs in float32, numpy's order
__device__ inline float np_pairwise_leaf(const uint16_t* a, int n) {
    if (n < 8) { float res = 0.0f; for (int i = 0; i < n; i++) res = res + half_bits_to_float(a[i]); return res; }
    float r[8];
    for (int k = 0; k < 8; k++) r[k] = half_bits_to_float(a[k]);
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] = r[k] + half_bits_to_float(a[i + k]);
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + half_bits_to_float(a[i]);
    return res;
}
__device__ inline float np_pairwise(const uint16_t* a, int n) {
    struct Frame { int off, n, st; float left; };
    Frame stk[24];
    int sp = 0;
    stk[0] = {0, n, 0, 0.0f};
    float ret = 0.0f;
    bool returning = false;
    while (sp >= 0) {
        Frame& f = stk[sp];
        if (!returning) {
            if (f.n <= 128) { ret = np_pairwise_leaf(a + f.off, f.n); returning = true; sp--; continue; }
            int n2 = f.n / 2; n2 -= n2 % 8;
            f.st = 1;
            stk[sp + 1] = {f.off, n2, 0, 0.0f}; sp++;
        } else if (f.st == 1) {
            int n2 = f.n / 2; n2 -= n2 % 8;
            f.left = ret; f.st = 2; returning = false;
            stk[sp + 1] = {f.off + n2, f.n - n2, 0, 0.0f}; sp++;
        } else { ret = f.left + ret; sp--; }
    }
    return ret;
}
// bucket.stats[row] = (avg, avg), avg = f16(sum / nb) (:179-194,244-245)
__global__ void q4_stats_kernel(const uint16_t* __restrict__ absRows, uint32_t rows, uint32_t nb, float2* __restrict__ stats) {
    const uint32_t row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    const float s = np_pairwise(absRows + (size_t)row * nb, (int)nb);
    const float avg = __half2float(__float2half_rn(s / (float)nb));
    stats[row] = make_float2(avg, avg);
}
__global__ void q4_probes_kernel(const uint16_t* __restrict__ core, uint32_t outDim, uint32_t n, uint16_t* __restrict__ probes) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) probes[j] = core[(size_t)j * outDim + j];
}

// work: 2 * inDim * outDim halfs (a copy of core2 that loses its outliers | the sorted magnitudes); keys: 2 * (cnt + ties) u64
hipError_t launch_convert_q4(const uint16_t* core2, uint32_t inDim, uint32_t outDim, uint32_t cnt, uint16_t* buckets, float* stats, uint16_t* probes,
                             float* outliers, int numCU, hipStream_t st) {
    const size_t n = (size_t)inDim * outDim;
    uint16_t* work = nullptr;
    uint32_t* small = nullptr;                    // [32768] histogram | [3] threshold result | [1] counter
    unsigned long long* keys = nullptr;
    void* cubTmp = nullptr;
    hipError_t e = hipMalloc(&work, n * 2 * 2);
    if (e == hipSuccess) e = hipMalloc(&small, (kMagBins + 8) * 4);
    auto done = [&](hipError_t r) { hipFree(work); hipFree(small); hipFree(keys); hipFree(cubTmp); return r; };
    if (e != hipSuccess) return done(e);
    uint16_t* core = work, *absRows = work + n;
    e = hipMemcpyAsync(core, core2, n * 2, hipMemcpyDeviceToDevice, st); if (e != hipSuccess) return done(e);
    if (cnt) {
        e = hipMemsetAsync(small, 0, (kMagBins + 8) * 4, st); if (e != hipSuccess) return done(e);
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&q4_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMagBins * 4); if (e != hipSuccess) return done(e);
        hipLaunchKernelGGL(q4_hist_kernel, dim3(numCU), dim3(1024), kMagBins * 4, st, core, n, small);
        hipLaunchKernelGGL(q4_threshold_kernel, dim3(1), dim3(1024), 0, st, small, cnt, small + kMagBins);
        uint32_t res[3] = {0, 0, 0};
        e = hipMemcpyAsync(res, small + kMagBins, 12, hipMemcpyDeviceToHost, st); if (e != hipSuccess) return done(e);
        e = hipStreamSynchronize(st); if (e != hipSuccess) return done(e);
        const uint32_t cand = res[2];             // everything at or above the threshold magnitude (>= cnt)
        e = hipMalloc(&keys, (size_t)cand * 8 * 2); if (e != hipSuccess) return done(e);
        hipLaunchKernelGGL(q4_collect_kernel, dim3(numCU * 8), dim3(256), 0, st, core, n, small + kMagBins, keys, small + kMagBins + 4);
        size_t tmpBytes = 0;
        e = hipcub::DeviceRadixSort::SortKeys(nullptr, tmpBytes, keys, keys + cand, (int)cand, 0, 64, st); if (e != hipSuccess) return done(e);
        e = hipMalloc(&cubTmp, tmpBytes ? tmpBytes : 16); if (e != hipSuccess) return done(e);
        e = hipcub::DeviceRadixSort::SortKeys(cubTmp, tmpBytes, keys, keys + cand, (int)cand, 0, 64, st); if (e != hipSuccess) return done(e);
        hipLaunchKernelGGL(q4_emit_outliers_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, keys + cand, cnt, core, outDim, reinterpret_cast<float4*>(outliers));
    }
    const size_t words = (size_t)inDim * (outDim / 32);
    hipLaunchKernelGGL(q4_bucket_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, st, core, inDim, outDim, buckets, absRows);
    hipLaunchKernelGGL(q4_stats_kernel, dim3((inDim * 8 + 63) / 64), dim3(64), 0, st, absRows, inDim * 8, outDim / 8, reinterpret_cast<float2*>(stats));
    const uint32_t np = inDim < outDim ? inDim : outDim;
    hipLaunchKernelGGL(q4_probes_kernel, dim3((np + 255) / 256), dim3(256), 0, st, core, outDim, np, probes);
    e = hipGetLastError(); if (e != hipSuccess) return done(e);
    e = hipStreamSynchronize(st);
    return done(e);
}

// Stable radix sort of (key, value) pairs by key -- the Q4 outlier index's by-output order (dispatch.hip) -- kept in THIS
// translation unit: it is the one that carries rocPRIM (whose own getenv import tests/test_abi.py allows here only).
hipError_t launch_sort_pairs_u32(const uint32_t* keysIn, uint32_t* keysOut, const uint32_t* valsIn, uint32_t* valsOut, uint64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (n >= (1ull << 31)) return hipErrorInvalidValue;
    size_t tmpBytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, keysIn, keysOut, valsIn, valsOut, (int)n, 0, 32, st);
    if (e != hipSuccess) return e;
    void* cubTmp = nullptr;
    e = hipMalloc(&cubTmp, tmpBytes ? tmpBytes : 16);
    if (e != hipSuccess) return e;
    e = hipcub::DeviceRadixSort::SortPairs(cubTmp, tmpBytes, keysIn, keysOut, valsIn, valsOut, (int)n, 0, 32, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);           // (the scratch is freed below)
    hipFree(cubTmp);
    return e;
}

}  // namespace effort
