// Standalone gfx950 microbenchmarks that informed the design of the bucket_mul kernel:
//   * device facts (CUs, clocks, wall-clock rate)
//   * HBM streaming read rate with 16-B and 2-B per-lane loads
//   * scatter-accumulate primitives: ds_add_f32, ds_add_u32, LDS read-add-write, register select chain
// Build: hipcc --offload-arch=gfx950 -O3 tools/lab/microbench.hip -o tools/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_stream16(const uint4* __restrict__ p, size_t n, unsigned* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n; i += stride) { uint4 a = p[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

// each wave reads whole 128-B pieces (2 B per lane), 16 in flight, pieces strided like bucket rows
__global__ void k_stream2(const unsigned short* __restrict__ p, size_t nPieces, size_t pitchElems, unsigned* out) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    unsigned acc = 0;
    for (size_t r = wave * 16; r + 16 <= nPieces; r += nWaves * 16) {
        unsigned short v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = p[(r + u) * pitchElems + lane];
#pragma unroll
        for (int u = 0; u < 16; u++) acc += v[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_scatter(unsigned* out, int iters, unsigned long long* clk) {
    __shared__ float acc[16 * 17 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* my = acc + wave * 17 * 64 + lane;
    for (int i = 0; i < 17; i++) my[i * 64] = 0.f;
    unsigned x = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = 0.f;
    __syncthreads();
    unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = x * 1664525u + 1013904223u;
            const unsigned pos = (x >> 20) & 15u;
            const float val = __uint_as_float((x & 0x007FFFFFu) | 0x3F800000u) - 1.5f;
            if (MODE == 0) __hip_atomic_fetch_add(my + pos * 64, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else if (MODE == 1) __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(my + pos * 64), __float_as_uint(val) >> 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else if (MODE == 2) { volatile float* q = my + pos * 64; const float o = *q; *q = o + val; }
            else if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 16; i++) r[i] += (pos == (unsigned)i) ? val : 0.f;
            }
        }
    }
    unsigned long long t1 = wall_clock64();
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += my[i * 64] + r[i];
    if (s == 1234.5f) out[0] = 1;
    if (tid == 0) { atomicMin(&clk[0], t0); atomicMax(&clk[1], t1); }
}

// Residency probe: every workgroup spins for a fixed wall-clock time; total time / spin time = rounds, hence how
// many workgroups the chip really keeps resident for a given (threads, LDS, VGPR) footprint.
template <int VG>
__global__ void k_spin(unsigned* out, unsigned long long ticks, unsigned long long* clk) {
    extern __shared__ char sm[];
    float r[VG];
#pragma unroll
    for (int i = 0; i < VG; i++) r[i] = threadIdx.x * 0.5f + i;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < VG; i++) r[i] = r[i] * 1.0001f + 0.5f;
        __builtin_amdgcn_s_sleep(8);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VG; i++) s += r[i];
    if (s == 1234.5f) { out[0] = 1; sm[threadIdx.x] = 1; }
    if (threadIdx.x == 0) { atomicMin(&clk[0], t0); atomicMax(&clk[1], (unsigned long long)wall_clock64()); }
}

template <int VG>
static void run_spin(int threads, int lds, int blocks, double spinUs, double wallKHz, unsigned* dOut, unsigned long long* dClk) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spin<VG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&k_spin<VG>), threads, lds));
    for (int rep = 0; rep < 2; rep++) {
        unsigned long long h[2] = {~0ull, 0};
        CK(hipMemcpy(dClk, h, 16, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_spin<VG>, dim3(blocks), dim3(threads), lds, 0, dOut, (unsigned long long)(spinUs * wallKHz / 1000.0), dClk);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, dClk, 16, hipMemcpyDeviceToHost));
        const double us = (double)(h[1] - h[0]) * 1000.0 / wallKHz;
        if (rep) printf("spin vgpr~%3d threads %4d lds %6d blocks %5d spin %5.1f us: total %7.1f us  => %.2f rounds, ~%.0f resident (runtime occupancy %d/CU)\n",
                        VG, threads, lds, blocks, spinUs, us, us / spinUs, blocks / (us / spinUs), occ);
    }
}

template <int MODE>
static void run_scatter(const char* name, int nCU, double wallKHz, unsigned* dOut, unsigned long long* dClk) {
    const int iters = 2000, blocks = nCU * 2;
    unsigned long long h[2] = {~0ull, 0};
    CK(hipMemcpy(dClk, h, 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_scatter<MODE>, dim3(blocks), dim3(1024), 0, 0, dOut, iters, dClk);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, dClk, 16, hipMemcpyDeviceToHost));
    const double us = (double)(h[1] - h[0]) * 1000.0 / wallKHz;
    const double elems = (double)blocks * 1024 * iters * 8;
    printf("scatter %-28s %8.1f us  %7.2f Gelem/s  %6.3f elem/ns/CU\n", name, us, elems / us / 1e3, elems / us / 1e3 / nCU);
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    int wall = 0, clk = 0;
    hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("device %s arch %s CUs %d clock %d kHz wallclock %d kHz mem %.1f GB L2 %d B maxLDS/block %zu\n", p.name, p.gcnArchName,
           p.multiProcessorCount, clk, wall, p.totalGlobalMem / 1e9, p.l2CacheSize, p.sharedMemPerBlock);
    const int nCU = p.multiProcessorCount;
    const double wallKHz = wall > 0 ? wall : 100000.0;

    const size_t bytes = (size_t)3 << 30;
    void* buf; unsigned* dOut; unsigned long long* dClk;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&dOut, 64)); CK(hipMalloc(&dClk, 64));
    CK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {nCU * 4, nCU * 8, nCU * 16}) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_stream16, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, dOut);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("stream 16B/lane grid %5d x256: %7.1f GB/s\n", grid, bytes / ms / 1e6);
        }
    }
    for (size_t pitch : {(size_t)64, (size_t)688, (size_t)11008}) {
        const size_t nPieces = bytes / 2 / pitch - 16;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_stream2, dim3(nCU * 8), dim3(256), 0, 0, (const unsigned short*)buf, nPieces, pitch, dOut);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("stream 2B/lane 128-B pieces, pitch %6zu elems: %7.1f GB/s (useful bytes)\n", pitch, nPieces * 128.0 / ms / 1e6);
        }
    }
    if (getenv("MB_SPIN")) {
        for (int threads : {256, 512, 1024})
            for (int lds : {8192, 36000, 70000}) {
                run_spin<16>(threads, lds, 1536, 20.0, wallKHz, dOut, dClk);
                run_spin<16>(threads, lds, 6144, 20.0, wallKHz, dOut, dClk);
            }
        run_spin<16>(512, 20000, 1536, 5.0, wallKHz, dOut, dClk);
        run_spin<16>(512, 20000, 6144, 5.0, wallKHz, dOut, dClk);
        run_spin<64>(512, 20000, 1536, 20.0, wallKHz, dOut, dClk);
        return 0;
    }
    run_scatter<0>("ds_add_f32 (private slots)", nCU, wallKHz, dOut, dClk);
    run_scatter<1>("ds_add_u32 (private slots)", nCU, wallKHz, dOut, dClk);
    run_scatter<2>("ds_read+add+ds_write", nCU, wallKHz, dOut, dClk);
    run_scatter<3>("16-way register select", nCU, wallKHz, dOut, dClk);
    return 0;
}
