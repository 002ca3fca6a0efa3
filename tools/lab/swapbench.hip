// swapbench -- can a tile's reducer CONSUME the partial slabs with atomic swaps (value out, sentinel in: self-cleaning, no ticket,
// no store acknowledgement on the writers' path) at the speed it reads them today (tools only)?
// 192 writer workgroups store an 8 KB slab each, write-through; then 6 reducer workgroups of 512 threads take 31 slabs x 8 KB each:
//   mode 0  16-byte loads past L1 (sc1), 16 in flight per thread                  [today's last arriver]
//   mode 1  64-bit atomic swaps with return (device scope), 32 in flight per thread
//   mode 2  32-bit atomic swaps with return, 64 in flight
// Prints the reducer kernel's duration.      hipcc --offload-arch=gfx950 -O3 -o build/swapbench tools/lab/swapbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kSc1 = 16, kTiles = 6, kSlices = 32, kTileF = 2048;       // floats per slab (E = 2)

__global__ __launch_bounds__(512) void writer(float* slabs) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, kTiles * kSlices * kTileF * 4, 0x00020000);
    const uint32_t base = blockIdx.x * kTileF * 4u;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 v; v[0] = v[1] = v[2] = v[3] = __float_as_uint(1.0f + blockIdx.x);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + threadIdx.x * 16u, 0, kSc1);
}

template <int MODE>
__global__ __launch_bounds__(512) void reducer(float* slabs, float* out) {
    const uint32_t t = blockIdx.x, tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, kTiles * kSlices * kTileF * 4, 0x00020000);
    float s[4] = {0, 0, 0, 0};
    if constexpr (MODE == 0) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        for (int sl0 = 0; sl0 < kSlices - 1; sl0 += 16) {
            u4 r[16];
#pragma unroll
            for (int i = 0; i < 16; i++) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16u, (uint32_t)((min(sl0 + i, kSlices - 2) * kTiles + t) * kTileF * 4), kSc1);
#pragma unroll
            for (int i = 0; i < 16; i++) if (sl0 + i < kSlices - 1) for (int h = 0; h < 4; h++) s[h] += __uint_as_float(r[i][h]);
        }
    } else if constexpr (MODE == 1) {
        unsigned long long* p = reinterpret_cast<unsigned long long*>(slabs);
        for (int sl0 = 0; sl0 < kSlices - 1; sl0 += 16) {
            unsigned long long r[32];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const size_t o = ((size_t)(min(sl0 + i, kSlices - 2) * kTiles + t) * kTileF * 4 + tid * 16u) / 8;
                r[2 * i] = __hip_atomic_exchange(&p[o], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r[2 * i + 1] = __hip_atomic_exchange(&p[o + 1], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int i = 0; i < 32; i++) if (sl0 + i / 2 < kSlices - 1) { s[(i & 1) * 2] += __uint_as_float((uint32_t)r[i]); s[(i & 1) * 2 + 1] += __uint_as_float((uint32_t)(r[i] >> 32)); }
        }
    } else {
        uint32_t* p = reinterpret_cast<uint32_t*>(slabs);
        for (int sl0 = 0; sl0 < kSlices - 1; sl0 += 8) {
            uint32_t r[32];
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const size_t o = ((size_t)(min(sl0 + i, kSlices - 2) * kTiles + t) * kTileF * 4 + tid * 16u) / 4 + h;
                    r[4 * i + h] = __hip_atomic_exchange(&p[o], ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
            for (int i = 0; i < 32; i++) if (sl0 + i / 4 < kSlices - 1) s[i & 3] += __uint_as_float(r[i]);
        }
    }
    for (int h = 0; h < 4; h++) out[(t * 512 + tid) * 4 + h] = s[h];
}

int main() {
    float *d_s, *d_o;
    const size_t bytes = (size_t)kTiles * kSlices * kTileF * 4;
    CK(hipMalloc(&d_s, bytes)); CK(hipMalloc(&d_o, kTiles * 512 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f, sum = 0; int n = 0;
        for (int rep = 0; rep < 12; rep++) {
            hipLaunchKernelGGL(writer, dim3(kTiles * kSlices), dim3(512), 0, 0, d_s);
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(reducer<0>, dim3(kTiles), dim3(512), 0, 0, d_s, d_o);
            else if (mode == 1) hipLaunchKernelGGL(reducer<1>, dim3(kTiles), dim3(512), 0, 0, d_s, d_o);
            else hipLaunchKernelGGL(reducer<2>, dim3(kTiles), dim3(512), 0, 0, d_s, d_o);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2) { best = ms < best ? ms : best; sum += ms; n++; }
        }
        float h[4]; CK(hipMemcpy(h, d_o, 16, hipMemcpyDeviceToHost));
        printf("mode %d  %-44s  min %.2f  mean %.2f us   (check %.1f)\n", mode,
               mode == 0 ? "16-byte sc1 loads" : mode == 1 ? "64-bit atomic swaps (value out, sentinel in)" : "32-bit atomic swaps", best * 1e3, sum / n * 1e3, h[0]);
    }
    return 0;
}
