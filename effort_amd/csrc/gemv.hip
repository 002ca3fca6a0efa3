// Dense baseline: basicMul (helpers/mps.swift:14-47) = out = W * f16(v), W f16 [outDim][inDim] row-major, f32 accumulate.
// The reference hands this to MPSMatrixVectorMultiplication; rocBLAS' hssgemv moves the 90 MB of a 4096x11008 matrix at
// 1.9 TB/s on MI355X, so the baseline (and the decode loop's LM head) gets a kernel of its own here: HBM-bound streaming,
// 16 bytes per lane per load, eight loads in flight per lane, v kept as f16 in LDS, v_fma_mix_f32.
#include "effort_internal.h"

namespace effort {

constexpr int kGemvWaves = 4;          // waves per workgroup
// Cache policy of the weight stream (AUX, an immediate of the load): a matrix of more than kGemvKeepBytes is streamed NON-TEMPORALLY
// (every element is read once and nothing of it will still be cached when it is read again; 4096 x 11008, 32 matrices rotated: 17.55 ->
// 17.0 us per call, the decode loop's dense line 291 -> 310 tokens/s, round 6 -- the baseline gets what the bucket-row stream got); a
// smaller one keeps the ordinary policy: the reference's dense benchmark line multiplies ONE 4096 x 4096 matrix three times in a row
// (benchmarks/benchmark.swift:237-241), and its second and third pass come from the Infinity Cache (360 projected tokens/s against 264 nt).
constexpr size_t kGemvKeepBytes = 64u << 20;

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// ROWS = rows a wave works on at once (they share the LDS reads of v): 2, or 1 when the matrix has too few rows to fill the chip
template <int kGemvRows, int AUX>       // eight 16-byte loads in flight per lane: 8 / ROWS 1 KB chunks of each row
__global__ __launch_bounds__(64 * kGemvWaves) void dense_gemv_kernel(const uint16_t* __restrict__ W, const float* __restrict__ v,
                                                                     float* __restrict__ out, uint32_t inDim, uint32_t outDim) {
    extern __shared__ __attribute__((aligned(16))) uint16_t vh[];              // f16(v), padded with zeros to a multiple of 512
    const uint32_t padded = (inDim + 511u) / 512u * 512u;
    auto h16 = [](float x) -> uint32_t { return __half_as_ushort(__float2half_rn(x)); };       // v.asFloat16(), mps.swift:19
    if (((size_t)v & 15u) == 0) {                        // four 16-byte loads in flight per thread (inDim % 16 == 0)
        const float4* v4 = reinterpret_cast<const float4*>(v);
        const uint32_t n4 = inDim / 4u;
        for (uint32_t i0 = 0; i0 < n4; i0 += 64 * kGemvWaves * 4) {
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) t[u] = v4[min(i0 + u * 64 * kGemvWaves + threadIdx.x, n4 - 1u)];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * 64 * kGemvWaves + threadIdx.x;
                if (i < n4) *reinterpret_cast<uint2*>(vh + 4u * i) = make_uint2(h16(t[u].x) | (h16(t[u].y) << 16), h16(t[u].z) | (h16(t[u].w) << 16));
            }
        }
        for (uint32_t i = inDim + threadIdx.x; i < padded; i += 64 * kGemvWaves) vh[i] = 0;
    } else {
        for (uint32_t i = threadIdx.x; i < padded; i += 64 * kGemvWaves) vh[i] = (uint16_t)h16(i < inDim ? v[i] : 0.0f);
    }
    __syncthreads();
    constexpr int kGemvUnroll = 8 / kGemvRows;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t r0 = (blockIdx.x * kGemvWaves + wave) * kGemvRows;
    if (r0 >= outDim) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(W), 0, (int)min((size_t)0xFFFFFFFFu, (size_t)outDim * inDim * 2u), 0x00020000);
    float acc[kGemvRows];
    uint32_t rowOff[kGemvRows];
#pragma unroll
    for (int r = 0; r < kGemvRows; r++) { acc[r] = 0.0f; rowOff[r] = __builtin_amdgcn_readfirstlane(min(r0 + r, outDim - 1u) * inDim * 2u); }   // a row past the end re-reads the last one
    // a chunk = 64 lanes x 8 halves = 512 elements; elements past inDim (inDim % 512 != 0) read as 0 through the descriptor only
    // at the very end of W, so the tail chunk is masked by hand
    for (uint32_t c0 = 0; c0 < padded; c0 += 512u * kGemvUnroll) {
        uint32_t w[kGemvRows][kGemvUnroll][4];
#pragma unroll
        for (int u = 0; u < kGemvUnroll; u++) {
            const uint32_t e = min(c0 + u * 512u + lane * 8u, inDim - 8u);      // clamped, branch-free (inDim % 16 == 0)
#pragma unroll
            for (int r = 0; r < kGemvRows; r++) {
                const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, e * 2u, rowOff[r], AUX);
                w[r][u][0] = t[0]; w[r][u][1] = t[1]; w[r][u][2] = t[2]; w[r][u][3] = t[3];
            }
        }
#pragma unroll
        for (int u = 0; u < kGemvUnroll; u++) {
            const uint32_t e = c0 + u * 512u + lane * 8u;
            const bool live = e < inDim;                                        // a clamped lane holds a copy of another lane's data
            const uint4 xq = *reinterpret_cast<const uint4*>(vh + min(e, padded - 8u));
            const uint32_t x[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
            for (int r = 0; r < kGemvRows; r++) {
                float a = 0.0f;
#pragma unroll
                for (int h = 0; h < 4; h++) {                                   // exact f16 products, f32 sums
                    const half2v wh = __builtin_bit_cast(half2v, w[r][u][h]), xh = __builtin_bit_cast(half2v, x[h]);
                    a = fmaf((float)wh[0], (float)xh[0], a);
                    a = fmaf((float)wh[1], (float)xh[1], a);
                }
                acc[r] += live ? a : 0.0f;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kGemvRows; r++) {
        float a = acc[r];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off);
        if (lane == 0 && r0 + r < outDim) out[r0 + r] = a;
    }
}

// inDim % 16 == 0, inDim * 2 padded to 1 KB must fit the LDS, W below 4 GiB; otherwise the caller falls back to rocBLAS
bool dense_gemv_supported(uint32_t inDim, uint32_t outDim) {
    return inDim % 16 == 0 && inDim >= 16 && inDim <= 65536 && (size_t)outDim * inDim * 2 <= 0xFFFFFFFFull;
}

template <int ROWS, int AUX>
static hipError_t launch_dense_gemv_t(const uint16_t* W, const float* v, float* out, uint32_t inDim, uint32_t outDim, hipStream_t st) {
    const uint32_t lds = (inDim + 511u) / 512u * 512u * 2u;
    if (lds > 48u * 1024u) {
        hipError_t e = allow_full_lds(reinterpret_cast<const void*>(&dense_gemv_kernel<ROWS, AUX>));
        if (e != hipSuccess) return e;
    }
    const uint32_t rowsPerWg = kGemvWaves * ROWS;
    hipLaunchKernelGGL((dense_gemv_kernel<ROWS, AUX>), dim3((outDim + rowsPerWg - 1) / rowsPerWg), dim3(64 * kGemvWaves), lds, st, W, v, out, inDim, outDim);
    return hipGetLastError();
}

hipError_t launch_dense_gemv(const uint16_t* W, const float* v, float* out, uint32_t inDim, uint32_t outDim, hipStream_t st) {
    const bool nt = (size_t)inDim * outDim * 2u > kGemvKeepBytes;
    if (outDim <= 8192u) return nt ? launch_dense_gemv_t<1, 2>(W, v, out, inDim, outDim, st) : launch_dense_gemv_t<1, 0>(W, v, out, inDim, outDim, st);
    return nt ? launch_dense_gemv_t<2, 2>(W, v, out, inDim, outDim, st) : launch_dense_gemv_t<2, 0>(W, v, out, inDim, outDim, st);
}

}  // namespace effort
