// rampbench -- how long the dispatcher takes to START a grid (tools only): every workgroup stamps the device wall clock first thing;
// the spread first -> last start, by workgroup size, dynamic LDS and register footprint.
//   hipcc --offload-arch=gfx950 -O3 -o build/rampbench tools/lab/rampbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int REGS>
__global__ void probe(unsigned long long* t, float* sink, int spin) {
    extern __shared__ char smem[];
    unsigned long long t0 = wall_clock64();
    float r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; i++) r[i] = (float)(threadIdx.x + i);
    for (int k = 0; k < spin; k++) {
#pragma unroll
        for (int i = 0; i < REGS; i++) r[i] = r[i] * 1.0001f + r[(i + 1) % REGS];
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < REGS; i++) s += r[i];
    if (threadIdx.x == 0) { t[blockIdx.x] = t0; smem[0] = 1; }
    if (s == 12345.678f) sink[0] = s;
}

template <int REGS>
static void run(int grid, int threads, int lds, const char* tag) {
    unsigned long long* d_t; float* d_s;
    CK(hipMalloc(&d_t, grid * 8)); CK(hipMalloc(&d_s, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<REGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<unsigned long long> h(grid);
    double best = 1e9, sum = 0; int n = 0;
    for (int rep = 0; rep < 12; rep++) {
        hipLaunchKernelGGL(probe<REGS>, dim3(grid), dim3(threads), lds, 0, d_t, d_s, 200);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d_t, grid * 8, hipMemcpyDeviceToHost));
        const auto mm = std::minmax_element(h.begin(), h.end());
        const double us = (double)(*mm.second - *mm.first) / 100.0;
        if (rep >= 2) { best = std::min(best, us); sum += us; n++; }
    }
    printf("%-10s grid %4d  threads %4d  lds %6d B  regs ~%3d : first -> last workgroup start  min %.2f  mean %.2f us\n", tag, grid, threads, lds, REGS, best, sum / n);
    CK(hipFree(d_t)); CK(hipFree(d_s));
}

int main() {
    for (int grid : {192, 256, 512}) {
        run<8>(grid, 256, 0, "small");
        run<8>(grid, 512, 0, "512t");
        run<8>(grid, 512, 40 * 1024, "512t+lds");
        run<8>(grid, 512, 70 * 1024, "512t+LDS");
        run<96>(grid, 512, 40 * 1024, "512t+regs");
        run<96>(grid, 1024, 40 * 1024, "1024t+regs");
        run<96>(grid, 256, 40 * 1024, "256t+regs");
    }
    return 0;
}
