// Materialised dispatch list (BucketMul.calcDispatch, bucketMul.swift:34-47) and small helper kernels.
//
// The fused multiply kernel never writes a global dispatch list; this file produces one, in the
// reference's float2 format {value, float(row*cols)} (bucketMul.metal:74, bucketMulQ4.metal:52), for
// the parity tests and for callers that inspect BucketMul.shared.dispatch.  Order: ascending bucket
// row (the reference's atomic append order is unspecified).  Three small launches: per-block counts
// (ballot + popcount), one-block exclusive scan, ordered write.
#include "effort_internal.h"

namespace effort {

constexpr int kDispBlock = 1024;

template <int FMT>
__device__ __forceinline__ bool keep_row(const void* stats, const float* v, uint32_t e, uint32_t r, const MulGeom& g,
                                         float cutoff, float* value) {
    const size_t i = (size_t)e * g.expertRows + r;               // bucket row incl. expert offset (:58-59)
    float mean, x;
    if (FMT == kFp16) {
        mean = half_bits_to_float(reinterpret_cast<const uint16_t*>(stats)[i * 4 + 3]);
        x = v[r % g.inDim];                                       // v[i % rowsCount], expertSize % inDim == 0
        *value = x;
    } else {
        mean = reinterpret_cast<const float*>(stats)[i * 2 + 1];
        x = v[r / 8u];                                            // v[i / 8] (expert 0)
        *value = x * mean;
    }
    return cutoff < (kCutoffScale * mean) * fabsf(x);
}

template <int FMT>
__global__ __launch_bounds__(kDispBlock) void disp_count_kernel(const void* stats, const float* v, const uint32_t* expNo,
                                                                const float* cutoff, const MulGeom g, uint32_t* blockCounts) {
    __shared__ uint32_t s_w[16];
    const uint32_t r = blockIdx.x * kDispBlock + threadIdx.x;
    const uint32_t e = expNo ? expNo[0] : 0u;
    float val;
    const bool keep = r < g.expertRows && keep_row<FMT>(stats, v, e, r, g, cutoff[0], &val);
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int i = 0; i < 16; i++) t += s_w[i]; blockCounts[blockIdx.x] = t; }
}

__global__ __launch_bounds__(256) void disp_scan_kernel(uint32_t* blockCounts, uint32_t nBlocks, uint32_t* count, uint32_t* ctxCount) {
    // nBlocks <= 229376*2/1024 = 448: one thread walks it (a few hundred adds)
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < nBlocks; i++) { const uint32_t c = blockCounts[i]; blockCounts[i] = run; run += c; }
        if (count) count[0] = run;
        ctxCount[0] = run;
    }
}

template <int FMT>
__global__ __launch_bounds__(kDispBlock) void disp_write_kernel(const void* stats, const float* v, const uint32_t* expNo,
                                                                const float* cutoff, const MulGeom g,
                                                                const uint32_t* blockOffsets, float2* dispatch) {
    __shared__ uint32_t s_w[16];
    const uint32_t r = blockIdx.x * kDispBlock + threadIdx.x;
    const uint32_t e = expNo ? expNo[0] : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float val = 0.0f;
    const bool keep = r < g.expertRows && keep_row<FMT>(stats, v, e, r, g, cutoff[0], &val);
    const unsigned long long m = __ballot(keep);
    const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t woff = 0;
    for (int i = 0; i < wave; i++) woff += s_w[i];
    if (keep) {
        const uint32_t i = e * g.expertRows + r;
        dispatch[blockOffsets[blockIdx.x] + woff + pre] = make_float2(val, (float)(uint32_t)(i * g.cols));
    }
}

hipError_t launch_calc_dispatch(Format fmt, const void* stats, const float* v, const uint32_t* expNo,
                                const float* cutoff, const MulGeom& g, float* dispatch, uint32_t* count,
                                uint32_t* ctxCount, uint32_t* blockScratch, hipStream_t st) {
    const uint32_t nBlocks = (g.expertRows + kDispBlock - 1) / kDispBlock;
    if (fmt == kFp16) hipLaunchKernelGGL(disp_count_kernel<kFp16>, dim3(nBlocks), dim3(kDispBlock), 0, st, stats, v, expNo, cutoff, g, blockScratch);
    else hipLaunchKernelGGL(disp_count_kernel<kQ4>, dim3(nBlocks), dim3(kDispBlock), 0, st, stats, v, expNo, cutoff, g, blockScratch);
    hipLaunchKernelGGL(disp_scan_kernel, dim3(1), dim3(256), 0, st, blockScratch, nBlocks, count, ctxCount);
    if (fmt == kFp16) hipLaunchKernelGGL(disp_write_kernel<kFp16>, dim3(nBlocks), dim3(kDispBlock), 0, st, stats, v, expNo, cutoff, g, blockScratch, reinterpret_cast<float2*>(dispatch));
    else hipLaunchKernelGGL(disp_write_kernel<kQ4>, dim3(nBlocks), dim3(kDispBlock), 0, st, stats, v, expNo, cutoff, g, blockScratch, reinterpret_cast<float2*>(dispatch));
    return hipGetLastError();
}

// ---- v.asFloat16() (helpers/mps.swift:19) -----------------------------------------------------
__global__ void f32_to_f16_kernel(const float* in, uint16_t* out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = __half_as_ushort(__float2half_rn(in[i]));
}
// max |w| of every bucket row (position bits included, exactly the value the multiply uses): one wave per row.
// Run once when a weight handle is registered (first half of launch_rank_bound).
__global__ __launch_bounds__(256) void row_max_kernel(const uint16_t* __restrict__ buckets, uint32_t pitchCols, uint32_t rows, uint32_t cols,
                                                      float* __restrict__ rowMax) {
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const uint16_t* p = buckets + (size_t)row * pitchCols;
    float m = 0.0f;
    for (uint32_t c = lane; c < cols; c += 64) m = fmaxf(m, fabsf(half_bits_to_float(p[c])));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) rowMax[row] = m;
}

// rankBound[e] = sum over ranks r of  max over input rows j of  rowValue(e, r, j), where rowValue is the row's max |w|
// (FP16, from row_max_kernel) or its |mean| (Q4, the magnitude every weight of the row decodes to).
__global__ __launch_bounds__(1024) void rank_bound_kernel(const float* __restrict__ rowMax, const float* __restrict__ statsQ4,
                                                          uint32_t rowsPerIn, uint32_t inDim, float* __restrict__ rankBound) {
    __shared__ float red[16];
    const uint32_t e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)e * rowsPerIn * inDim;
    float total = 0.0f;
    for (uint32_t r = 0; r < rowsPerIn; r++) {
        float m = 0.0f;
        for (uint32_t j = tid; j < inDim; j += 1024)
            m = fmaxf(m, statsQ4 ? fabsf(statsQ4[(base + (size_t)j * rowsPerIn + r) * 2 + 1]) : rowMax[base + (size_t)r * inDim + j]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        float mm = 0.0f;
        for (int w2 = 0; w2 < 16; w2++) mm = fmaxf(mm, red[w2]);
        total += mm;
        __syncthreads();
    }
    if (tid == 0) rankBound[e] = total;
}

hipError_t launch_rank_bound(Format fmt, const uint16_t* buckets, uint32_t pitchCols, const void* stats, uint32_t numExperts, uint32_t rowsPerIn,
                             uint32_t inDim, uint32_t cols, float* rowScratch, float* rankBound, hipStream_t st) {
    const uint32_t rows = numExperts * rowsPerIn * inDim;
    if (fmt == kFp16) hipLaunchKernelGGL(row_max_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, buckets, pitchCols, rows, cols, rowScratch);
    hipLaunchKernelGGL(rank_bound_kernel, dim3(numExperts), dim3(1024), 0, st, rowScratch,
                       fmt == kQ4 ? static_cast<const float*>(stats) : nullptr, rowsPerIn, inDim, rankBound);
    return hipGetLastError();
}

// The FP16 multiply's keep test reads ONE half of every bucket row's stats (lane .w, the row mean): a copy of just those
// halves, made at registration, is a quarter of the bytes to stage per item (and to keep in L2 between a slice's tiles).
__global__ void compact_means_kernel(const uint16_t* stats4, uint16_t* means, uint32_t rows) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) means[r] = stats4[(size_t)r * 4u + 3u];
}
hipError_t launch_compact_means(const void* stats, uint16_t* means, uint32_t rows, hipStream_t st) {
    hipLaunchKernelGGL(compact_means_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, static_cast<const uint16_t*>(stats), means, rows);
    return hipGetLastError();
}

hipError_t launch_f32_to_f16(const float* in, uint16_t* out, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3((n + 255) / 256), dim3(256), 0, st, in, out, n);
    return hipGetLastError();
}

// ---- cosineSimilarityTo (aux.metal:293-312): dot, |a|^2, |b|^2 in f32; one block, fixed order ----
__global__ __launch_bounds__(1024) void cosine_kernel(const float* a, const float* b, uint32_t n, float* out3) {
    __shared__ float s[3][16];
    float d = 0, ma = 0, mb = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) { const float x = a[i], y = b[i]; d += x * y; ma += x * x; mb += y * y; }
    for (int off = 32; off >= 1; off >>= 1) { d += __shfl_xor(d, off); ma += __shfl_xor(ma, off); mb += __shfl_xor(mb, off); }
    if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = d; s[1][threadIdx.x >> 6] = ma; s[2][threadIdx.x >> 6] = mb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float D = 0, A = 0, Bq = 0;
        for (int i = 0; i < 16; i++) { D += s[0][i]; A += s[1][i]; Bq += s[2][i]; }
        out3[0] = D / (sqrtf(A) * sqrtf(Bq)); out3[1] = A; out3[2] = Bq;
    }
}
hipError_t launch_cosine(const float* a, const float* b, uint32_t n, float* out3, hipStream_t st) {
    hipLaunchKernelGGL(cosine_kernel, dim3(1), dim3(1024), 0, st, a, b, n, out3);
    return hipGetLastError();
}

// ---- Q4 outliers -> index (registration time) -------------------------------------------------------------------
// outliers float4 = (value, inIdx, outIdx, 0) (q4_draft.py:58-67).  The multiply's outlier phase (bucket_mul.hip, O) gives every
// OUTPUT to one lane, which sums that output's products in a register -- no atomics -- while the wave's loads stay coalesced: the
// index is a jagged-diagonal layout per block of 64 consecutive outputs.  The table is sorted by output, stably (a radix sort by
// key: an output's entries keep the TABLE's order, so the sums are added in a defined order whatever the registration's atomics
// did); within a block the 64 outputs are ranked by entry count, descending (ties: lower output first); entry k of the output of
// rank i sits at block start + sum over k' < k of #{outputs with more than k' entries} + i -- step k of the multiply reads the k-th
// entry of every output that has one with ONE contiguous load, lane i <-> rank i, and no padding exists.  Per rank a meta word
// (entry count << 8 | output within the block) tells the lane whose sum it holds.  Entries are FOUR bytes: f16 value << 16 | input.
__global__ void ol_validate_kernel(const float4* ol, uint64_t n, uint32_t inDim, uint32_t outDim, int* bad) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 o = ol[i];
    const bool idx = o.y >= 0.0f && o.y < (float)inDim && o.z >= 0.0f && o.z < (float)outDim;
    const bool f16 = __half2float(__float2half_rn(o.x)) == o.x;
    if (!idx) atomicAdd(bad, 1);
    else if (!f16) atomicAdd(bad + 1, 1);
}
hipError_t launch_validate_outliers(const float* outliers, uint64_t n, uint32_t inDim, uint32_t outDim, int* bad, hipStream_t st) {
    hipLaunchKernelGGL(ol_validate_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(outliers), n, inDim, outDim, bad);
    return hipGetLastError();
}
__global__ void ol_count_kernel(const float4* ol, uint64_t n, uint32_t* rowPtr, uint32_t* keys, uint32_t* vals) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t out = (uint32_t)ol[i].z;
    atomicAdd(&rowPtr[out + 1], 1u);
    keys[i] = out; vals[i] = (uint32_t)i;                  // sorted by output below (stable: table order inside an output)
}
__global__ __launch_bounds__(1024) void ol_scan_kernel(uint32_t* rowPtr, uint32_t outDim) {
    __shared__ uint32_t s_part[1024];
    // inclusive scan of rowPtr[1..outDim] in place (rowPtr[0] = 0): thread t owns a contiguous chunk
    const uint32_t per = (outDim + 1023) / 1024, lo = 1 + threadIdx.x * per, hi = min(outDim + 1, lo + per);
    uint32_t sum = 0, longest = 0;
    for (uint32_t i = lo; i < hi; i++) { sum += rowPtr[i]; longest = max(longest, rowPtr[i]); }
    s_part[threadIdx.x] = sum;
    atomicMax(&rowPtr[outDim + 1], longest);               // one more word: the longest segment (zeroed by the launcher)
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int i = 0; i < 1024; i++) { const uint32_t c = s_part[i]; s_part[i] = run; run += c; } rowPtr[0] = 0; }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (uint32_t i = lo; i < hi; i++) { run += rowPtr[i]; rowPtr[i] = run; }
}
// one wave per block of 64 outputs (lane = output): rank the outputs by entry count, write the meta words, then walk the steps --
// step k places the k-th entry of every output that has one at (cursor + rank), the cursor advancing by the step's population
__global__ __launch_bounds__(64) void ol_jds_kernel(const float4* ol, const uint32_t* sortedIdx, const uint32_t* rowPtr, uint32_t outDim,
                                                    uint32_t* entry, uint32_t* meta, uint32_t* blockPtr) {
    const uint32_t lane = threadIdx.x, out = blockIdx.x * 64u + lane;
    const uint32_t lo = rowPtr[min(out, outDim)], len = out < outDim ? rowPtr[out + 1] - lo : 0u;
    uint32_t rank = 0;
    for (uint32_t o = 0; o < 64u; o++) {
        const uint32_t lo2 = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)o);
        rank += (lo2 > len || (lo2 == len && o < lane)) ? 1u : 0u;
    }
    meta[(size_t)blockIdx.x * 64u + rank] = (len << 8) | lane;
    uint32_t cursor = rowPtr[min(blockIdx.x * 64u, outDim)];
    if (lane == 0) blockPtr[blockIdx.x] = cursor;
    if (lane == 0 && blockIdx.x == gridDim.x - 1u) blockPtr[gridDim.x] = rowPtr[outDim];
    for (uint32_t k = 0;; k++) {
        const unsigned long long active = __ballot(len > k);
        if (!active) break;
        if (len > k) {
            const float4 o = ol[sortedIdx[lo + k]];
            entry[cursor + rank] = ((uint32_t)__half_as_ushort(__float2half_rn(o.x)) << 16) | (uint32_t)o.y;
        }
        cursor += (uint32_t)__popcll(active);
    }
}
// rowPtr: [outDim + 1] by-output bounds, then one word: the longest segment.  tmp: [n] keys | [n] table indices (unsorted) | [n] keys |
// [n] table indices (sorted by output, stably: launch_sort_pairs_u32, convert_q4.hip -- the one translation unit that carries rocPRIM)
hipError_t launch_build_outlier_index(const float* outliers, uint64_t n, uint32_t inDim, uint32_t outDim, uint32_t* rowPtr,
                                      uint32_t* blockPtr, uint32_t* entry, uint32_t* meta, uint32_t* tmp, hipStream_t st) {
    (void)inDim;
    hipError_t e = hipMemsetAsync(rowPtr, 0, ((size_t)outDim + 2) * 4, st); if (e != hipSuccess) return e;
    const float4* ol = reinterpret_cast<const float4*>(outliers);
    const uint32_t nb = (uint32_t)((n + 255) / 256), nBlocks = (outDim + 63u) / 64u;
    uint32_t *keys = tmp, *vals = tmp + n, *keysOut = tmp + 2 * n, *valsOut = tmp + 3 * n;
    hipLaunchKernelGGL(ol_count_kernel, dim3(nb), dim3(256), 0, st, ol, n, rowPtr, keys, vals);
    hipLaunchKernelGGL(ol_scan_kernel, dim3(1), dim3(1024), 0, st, rowPtr, outDim);
    e = launch_sort_pairs_u32(keys, keysOut, vals, valsOut, n, st); if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ol_jds_kernel, dim3(nBlocks), dim3(64), 0, st, ol, valsOut, rowPtr, outDim, entry, meta, blockPtr);
    return hipGetLastError();
}

}  // namespace effort
