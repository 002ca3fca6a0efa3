// GPU converter: HF weight matrix -> the reference's FP16 bucket layout.
// Replaces bucketize() (convert.swift:209-260) and its kernels getProbes, prepareValsIdxs,
// idxsBitonicSortAbs (driven inDim times from the host by Vector.sortAbs, model.swift:660-683 -- the
// reference's minutes-per-layer part), preBucketize, bucketize, makeStats (convert.metal:14-119,315-342).
//
// MI355X design: one 1024-thread workgroup per INPUT row keeps the whole row (values + u16 indices,
// zero-padded to the next power of two exactly as sortAbs pads) in LDS -- up to 16384 x 4 B = 64 KiB of
// the CU's 160 KiB -- and runs the reference's compare-exchange network there, stage by stage with one
// barrier each, so ties in |w| land exactly where the reference's network puts them.  The rank of an
// element inside its bucket (= how many members of the same 16-output bucket precede it in the sorted
// order, what preBucketize's running counters compute) is found in parallel from the inverse
// permutation; the 16 rank rows are staged in LDS and written coalesced into the rank-major matrix.
// Rows where the zero padding ties with real zeros and leaks into the first outDim sorted entries
// (non-power-of-two outDim only; the reference then overfills bucket 0) take a literal sequential
// emulation of preBucketize on one lane, so the output stays bit-identical to the reference's.
#include "effort_internal.h"

namespace effort {

__global__ void probes_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ probes, uint32_t inDim, uint32_t rep) {
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;                     // getProbes, convert.metal:14-22
    if (id >= kProbes / rep) return;
    for (uint32_t i = 0; i < rep; i++) probes[id * rep + i] = W[(size_t)id * inDim + id + i];
}

// vals[in][out] = W[out][in]   (prepareValsIdxs, convert.metal:27-41; the idxs are implicit: idx == out)
__global__ __launch_bounds__(256) void transpose_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ vals,
                                                        uint32_t outDim, uint32_t inDim) {
    __shared__ uint16_t tile[64][66];
    const uint32_t bx = blockIdx.x * 64u, by = blockIdx.y * 64u;             // bx: in, by: out
    const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
    for (uint32_t r = ty; r < 64; r += 4) {
        const uint32_t o = by + r, i = bx + tx;
        tile[r][tx] = (o < outDim && i < inDim) ? W[(size_t)o * inDim + i] : (uint16_t)0;
    }
    __syncthreads();
    for (uint32_t r = ty; r < 64; r += 4) {
        const uint32_t i = bx + r, o = by + tx;
        if (i < inDim && o < outDim) vals[(size_t)i * outDim + o] = tile[tx][r];
    }
}

__device__ __forceinline__ uint16_t half_bits_to_ushort(uint16_t h) {       // Metal half -> ushort
    const float f = half_bits_to_float(h);
    if (!(f > 0.0f)) return 0;
    if (f >= 65535.0f) return 65535;
    return (uint16_t)f;
}

constexpr uint32_t kB = 16;   // bucket size of the FP16 layout (convert.swift:232)

__global__ __launch_bounds__(1024) void sort_bucketize_kernel(const uint16_t* __restrict__ vals, uint16_t* __restrict__ buckets, uint32_t pitchCols,
                                                              uint32_t outDim, uint32_t inDim, uint32_t P, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t n = outDim, C = outDim / kB;
    uint16_t* sv = reinterpret_cast<uint16_t*>(smem);            // [P] sorted values
    uint16_t* si = sv + P;                                       // [P] their output indices
    uint16_t* aux = si + P;                                      // posOf[n]  |  bVals[C*17] (slow path)
    uint16_t* tileOut = aux + C * (kB + 1);                      // [16][C] rank rows
    uint32_t& s_cnt0 = *reinterpret_cast<uint32_t*>(tileOut + n);   // all LDS in the one dynamic region
    const uint32_t row = blockIdx.x, tid = threadIdx.x;
    const uint16_t* src = vals + (size_t)row * n;

    if (tid == 0) s_cnt0 = 0;
    for (uint32_t i = tid; i < P; i += 1024) { sv[i] = i < n ? src[i] : (uint16_t)0; si[i] = i < n ? (uint16_t)i : (uint16_t)0; }
    __syncthreads();

    // idxsBitonicSortAbs network (convert.metal:315-342): for p, for q<=p, compare-exchange at distance 2^(p-q)
    uint32_t logn = 0; while ((1u << logn) < P) logn++;
    for (uint32_t p = 0; p < logn; p++) {
        for (uint32_t q = 0; q <= p; q++) {
            const uint32_t sh = p - q, distance = 1u << sh;
            for (uint32_t m = tid; m < P / 2; m += 1024) {
                const uint32_t gid = ((m >> sh) << (sh + 1)) | (m & (distance - 1u));   // (gid & distance) == 0
                const uint32_t partner = gid | distance;
                const bool direction = ((gid >> p) & 2u) == 0u;
                const uint16_t a = sv[gid], b = sv[partner];
                const bool less = (a & 0x7FFFu) < (b & 0x7FFFu);     // |a| < |b| on halfs
                if (less == direction) {
                    sv[gid] = b; sv[partner] = a;
                    const uint16_t ia = si[gid], ib = si[partner];
                    si[gid] = ib; si[partner] = ia;
                }
            }
            __syncthreads();
        }
    }

    // Does the zero padding (value 0, index 0) appear among the first n entries?  Then index 0 shows up
    // more than once there (sortAbs copies back only the first n, model.swift:674-675).
    uint32_t c0 = 0;
    for (uint32_t i = tid; i < n; i += 1024) c0 += (si[i] == 0) ? 1u : 0u;
    for (int off = 32; off >= 1; off >>= 1) c0 += __shfl_xor(c0, off);
    if ((tid & 63u) == 0 && c0) atomicAdd(&s_cnt0, c0);
    __syncthreads();
    const bool fast = (s_cnt0 == 1);

    if (fast) {
        for (uint32_t i = tid; i < n; i += 1024) aux[si[i]] = (uint16_t)i;            // inverse permutation
        __syncthreads();
        for (uint32_t c = tid; c < n; c += 1024) {
            const uint32_t bkt = c / kB, pos = c % kB, my = aux[c];
            uint32_t rank = 0;
#pragma unroll
            for (uint32_t k = 0; k < kB; k++) rank += (aux[bkt * kB + k] < my) ? 1u : 0u;
            // preBucketize :66-70: the low 4 mantissa bits are replaced by the position in the bucket
            tileOut[rank * C + bkt] = (uint16_t)((sv[my] & 0xFFF0u) | pos);
        }
    } else {
        // literal preBucketize (convert.metal:43-78) on one lane, counters kept as halfs in slot 0
        for (uint32_t i = tid; i < C * (kB + 1); i += 1024) aux[i] = 0;
        __syncthreads();
        if (tid == 0) {
            int oob = 0;
            const uint32_t lim = C * (kB + 1);
            for (uint32_t i = 0; i < n; i++) {
                uint16_t val = sv[i]; const uint16_t idx = si[i];
                const uint16_t bkt = (uint16_t)(idx / kB), pos = (uint16_t)(idx % kB);
                val = (uint16_t)((val & 0xFFF0u) | pos);
                const uint16_t bOff = (uint16_t)(bkt * (kB + 1));
                const uint16_t counter = half_bits_to_ushort(aux[bOff]);
                const uint32_t slot = (uint32_t)bOff + 1u + counter;
                if (slot < lim) aux[slot] = val; else oob++;
                aux[bOff] = __half_as_ushort(__float2half_rn(half_bits_to_float(aux[bOff]) + 1.0f));
            }
            if (oob) atomicAdd(status, oob);
        }
        __syncthreads();
        for (uint32_t c = tid; c < n; c += 1024) {
            const uint32_t bkt = c / kB, r = c % kB;
            tileOut[r * C + bkt] = aux[bkt * (kB + 1) + 1 + r];
        }
    }
    __syncthreads();
    // bucketize (convert.metal:83-100): buckets[(rank*inDim + row)*C + bucket]
    for (uint32_t i = tid; i < n; i += 1024) {
        const uint32_t r = i / C, bkt = i - r * C;
        buckets[((size_t)r * inDim + row) * pitchCols + bkt] = tileOut[i];      // (pitchCols = C as the reference writes it, or padded: whole 128-byte lines)
    }
}

// makeStats (convert.metal:105-119): f32 sum of |row| in column order, / bCols, stored as half x4
__global__ __launch_bounds__(256) void make_stats_kernel(const uint16_t* __restrict__ buckets, uint16_t* __restrict__ stats,
                                                         uint32_t rows, uint32_t C, uint32_t pitchCols) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= rows) return;
    const uint2* p = reinterpret_cast<const uint2*>(buckets + (size_t)r * pitchCols);   // C % 4 == 0, pitchCols % 4 == 0
    float sum = 0.0f;
    for (uint32_t i = 0; i < C / 4; i++) {
        const uint2 w = p[i];
        sum += fabsf(half_bits_to_float((uint16_t)(w.x & 0xFFFFu)));
        sum += fabsf(half_bits_to_float((uint16_t)(w.x >> 16)));
        sum += fabsf(half_bits_to_float((uint16_t)(w.y & 0xFFFFu)));
        sum += fabsf(half_bits_to_float((uint16_t)(w.y >> 16)));
    }
    const uint16_t m = __half_as_ushort(__float2half_rn(sum / (float)C));
    reinterpret_cast<uint2*>(stats)[r] = make_uint2((uint32_t)m | ((uint32_t)m << 16), (uint32_t)m | ((uint32_t)m << 16));
}

hipError_t launch_convert_fp16(const uint16_t* W, uint32_t outDim, uint32_t inDim, uint16_t* buckets, uint32_t pitchCols,
                               uint16_t* stats, uint16_t* probes, uint16_t* scratchVals, int* status, hipStream_t st) {
    const uint32_t rep = outDim >= (uint32_t)kProbes ? 1u : (uint32_t)kProbes / outDim;
    hipLaunchKernelGGL(probes_kernel, dim3((kProbes / rep + 255) / 256), dim3(256), 0, st, W, probes, inDim, rep);
    hipLaunchKernelGGL(transpose_kernel, dim3((inDim + 63) / 64, (outDim + 63) / 64), dim3(256), 0, st, W, scratchVals, outDim, inDim);
    // sortAbs padding: exact powers of two are sorted in place, everything else in 2^(floor(log2 n)+1)
    uint32_t P = 1; while (P < outDim) P <<= 1;
    const uint32_t C = outDim / kB;
    const uint32_t lds = P * 4 + C * (kB + 1) * 2 + outDim * 2 + 16;
    if (lds > 160u * 1024u - 64u) return hipErrorInvalidValue;
    {
        hipError_t e = allow_full_lds(reinterpret_cast<const void*>(&sort_bucketize_kernel));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(sort_bucketize_kernel, dim3(inDim), dim3(1024), lds, st, scratchVals, buckets, pitchCols, outDim, inDim, P, status);
    hipLaunchKernelGGL(make_stats_kernel, dim3((inDim * kB + 255) / 256), dim3(256), 0, st, buckets, stats, inDim * kB, C, pitchCols);
    return hipGetLastError();
}

}  // namespace effort
