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
// outliers float4 = (value, inIdx, outIdx, 0) (q4_draft.py:58-67).  Pass 1 counts per output, a one-block scan makes
// rowPtr, pass 2 places each outlier in its output's segment (atomic cursor: arbitrary order inside a segment -- the
// multiply adds the products as integers, so the order does not matter), pass 3 packs the segments of every block of
// 2^(16 - bitsIn) consecutive outputs into 4-byte entries (f16 value | output in block | input), interleaved -- first
// entry of each output, then the second of each ... -- so that neighbouring lanes of the multiply's coalesced stream
// mostly hold different outputs (their LDS atomics collide 64 / blockOutputs ways instead of 64).
// entries the format does not allow: an index outside the matrix (or NaN) -- e.g. a full-matrix table handed to a column
// shard -- or a value that is not an f16 number (the table comes from an f16 matrix; the 4-byte entry keeps 16 bits of it)
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
__global__ void ol_count_kernel(const float4* ol, uint64_t n, uint32_t* rowPtr) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) atomicAdd(&rowPtr[(uint32_t)ol[i].z + 1], 1u);
}
__global__ __launch_bounds__(1024) void ol_scan_kernel(uint32_t* rowPtr, uint32_t outDim) {
    __shared__ uint32_t s_part[1024];
    // inclusive scan of rowPtr[1..outDim] in place (rowPtr[0] = 0): thread t owns a contiguous chunk
    const uint32_t per = (outDim + 1023) / 1024, lo = 1 + threadIdx.x * per, hi = min(outDim + 1, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += rowPtr[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int i = 0; i < 1024; i++) { const uint32_t c = s_part[i]; s_part[i] = run; run += c; } rowPtr[0] = 0; }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (uint32_t i = lo; i < hi; i++) { run += rowPtr[i]; rowPtr[i] = run; }
}
__global__ void ol_place_kernel(const float4* ol, uint64_t n, const uint32_t* rowPtr, uint32_t* cursor, uint32_t* inIdx, float* value) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 o = ol[i];
    const uint32_t out = (uint32_t)o.z;
    const uint32_t p = rowPtr[out] + atomicAdd(&cursor[out], 1u);
    inIdx[p] = (uint32_t)o.y; value[p] = o.x;
}
// one wave per 64 outputs (lane = output), which hold 64 / bs blocks of bs outputs: round r moves the r-th entry of every
// output that has one; inside its block an output's entry lands after those of the block's lower outputs of that round
__global__ __launch_bounds__(64) void ol_pack_kernel(const uint32_t* rowPtr, uint32_t outDim, uint32_t bitsIn, const uint32_t* inIdx,
                                                     const float* valIn, uint32_t* entry, uint32_t* blockPtr) {
    const uint32_t out = blockIdx.x * 64u + threadIdx.x, lane = threadIdx.x;
    const uint32_t bs = 1u << (16u - bitsIn);                                  // outputs per block (a power of two <= 64... 32768 for bitsIn = 1: clamped below)
    const uint32_t bsl = min(bs, 64u);                                         // lanes of this wave per block
    const uint32_t lo = rowPtr[min(out, outDim)], len = out < outDim ? rowPtr[out + 1] - lo : 0u;
    const uint32_t first = lane / bsl * bsl;                                   // first lane of this lane's block (within the wave)
    const unsigned long long blockMask = (bsl == 64u ? ~0ull : ((1ull << bsl) - 1ull)) << first;
    uint32_t base = rowPtr[min(blockIdx.x * 64u + first, outDim)];
    if (lane == first && blockIdx.x * 64u + first < outDim) blockPtr[(blockIdx.x * 64u + first) / bs] = base;      // (bs <= 16: inDim >= 4096)
    for (uint32_t r = 0;; r++) {
        const unsigned long long active = __ballot(len > r);
        if (!active) break;
        const unsigned long long mine = active & blockMask;
        if (len > r) {
            const uint32_t p = base + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
            const uint32_t h = (uint32_t)__half_as_ushort(__float2half_rn(valIn[lo + r]));
            entry[p] = (h << 16) | ((out & (bs - 1u)) << bitsIn) | inIdx[lo + r];
        }
        base += (uint32_t)__popcll(mine);
    }
}
// rowPtr[outDim + 1 + b] = bits of max over the outputs of block b (64 outputs) of sum |value|: bounds the multiply's
// fixed-point outlier sums; after the blocks, the largest number of entries of one output
__global__ void ol_bound_kernel(const uint32_t* rowPtr, uint32_t outDim, const float* value, uint32_t* boundBits) {
    const uint32_t out = blockIdx.x * 256u + threadIdx.x;
    if (out >= outDim) return;
    float sum = 0.0f;
    for (uint32_t i = rowPtr[out]; i < rowPtr[out + 1]; i++) sum += fabsf(value[i]);
    atomicMax(&boundBits[out / 64u], __float_as_uint(sum));       // non-negative floats order like their bit patterns
    atomicMax(&boundBits[(outDim + 63u) / 64u], rowPtr[out + 1] - rowPtr[out]);      // one more word: the longest segment
}
__global__ void ol_block_end_kernel(const uint32_t* rowPtr, uint32_t outDim, uint32_t nBlocks, uint32_t* blockPtr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) blockPtr[nBlocks] = rowPtr[outDim];
}
// rowPtr: [outDim + 1] by-output bounds, then [ceil(outDim/64)] bound bits, then the longest segment (kept: bound64 points
// into it).  tmp: [outDim] cursors, then [n] inputs and [n] values of the by-output order (before packing)
hipError_t launch_build_outlier_index(const float* outliers, uint64_t n, uint32_t inDim, uint32_t outDim, uint32_t* rowPtr,
                                      uint32_t* blockPtr, uint32_t* entry, uint32_t* tmp, hipStream_t st) {
    hipError_t e = hipMemsetAsync(rowPtr, 0, ((size_t)outDim + 2 + (outDim + 63) / 64) * 4, st); if (e != hipSuccess) return e;
    e = hipMemsetAsync(tmp, 0, (size_t)outDim * 4, st); if (e != hipSuccess) return e;
    const float4* ol = reinterpret_cast<const float4*>(outliers);
    const uint32_t nb = (uint32_t)((n + 255) / 256);
    const uint32_t bitsIn = ol_bits_in(inDim), bs = 1u << (16u - bitsIn), nBlocks = (outDim + bs - 1) / bs;
    uint32_t* in0 = tmp + outDim;
    float* val0 = reinterpret_cast<float*>(in0 + n);
    hipLaunchKernelGGL(ol_count_kernel, dim3(nb), dim3(256), 0, st, ol, n, rowPtr);
    hipLaunchKernelGGL(ol_scan_kernel, dim3(1), dim3(1024), 0, st, rowPtr, outDim);
    hipLaunchKernelGGL(ol_place_kernel, dim3(nb), dim3(256), 0, st, ol, n, rowPtr, tmp, in0, val0);
    hipLaunchKernelGGL(ol_bound_kernel, dim3((outDim + 255) / 256), dim3(256), 0, st, rowPtr, outDim, val0, rowPtr + outDim + 1);
    hipLaunchKernelGGL(ol_pack_kernel, dim3((outDim + 63) / 64), dim3(64), 0, st, rowPtr, outDim, bitsIn, in0, val0, entry, blockPtr);
    hipLaunchKernelGGL(ol_block_end_kernel, dim3(1), dim3(64), 0, st, rowPtr, outDim, nBlocks, blockPtr);
    return hipGetLastError();
}

}  // namespace effort
