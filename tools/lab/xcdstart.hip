// xcdstart -- WHICH workgroups of a grid start late (tools only).  tools/lab/rampbench.hip found that any grid of 192-512 workgroups takes
// ~1.3 us from its first to its last workgroup's first instruction; the per-item trace of a lone multiply (tools/timeline.py) shows
// that this is not a gradual ramp: seven XCDs start within 0.2 us and ONE starts ~0.9 us later, and its items are the launch's last.
// This probe stamps every workgroup's start with its XCC_ID, for isolated launches and for back-to-back launches of one hipGraph
// (each preceded by a kernel shaped like the previous multiply), and prints per XCD: the mean offset of its first and of its last
// workgroup start from the launch's first start, and how often it was the last XCD to start.
//   hipcc --offload-arch=gfx950 -O3 -o build/xcdstart tools/lab/xcdstart.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void probe(unsigned long long* t, unsigned* xcc, float* sink, int spin) {
    extern __shared__ char smem[];
    const unsigned long long t0 = wall_clock64();
    float r = (float)threadIdx.x;
    for (int k = 0; k < spin; k++) r = r * 1.0001f + 0.5f;
    if (threadIdx.x == 0) { t[blockIdx.x] = t0; xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu; smem[0] = 1; }
    if (r == 12345.678f) sink[0] = r;
}

static void report(const char* tag, int grid, int launches, const std::vector<unsigned long long>& T, const std::vector<unsigned>& X) {
    double first[8] = {0}, last[8] = {0}; int lastCount[8] = {0}, n = 0;
    for (int l = 0; l < launches; l++) {
        const unsigned long long* t = T.data() + (size_t)l * grid; const unsigned* x = X.data() + (size_t)l * grid;
        unsigned long long t0 = ~0ull; for (int b = 0; b < grid; b++) t0 = t[b] < t0 ? t[b] : t0;
        unsigned long long f[8], la[8]; for (int i = 0; i < 8; i++) { f[i] = ~0ull; la[i] = 0; }
        for (int b = 0; b < grid; b++) { const unsigned k = x[b] & 7u; if (t[b] < f[k]) f[k] = t[b]; if (t[b] > la[k]) la[k] = t[b]; }
        int worst = 0;
        for (int i = 0; i < 8; i++) { if (f[i] == ~0ull) continue; first[i] += (double)(f[i] - t0) / 100.0; last[i] += (double)(la[i] - t0) / 100.0; if (f[i] > f[worst] || f[worst] == ~0ull) worst = i; }
        lastCount[worst]++; n++;
    }
    printf("%s grid %d, %d launches: per XCD mean first-start / last-start offset (us) [times it was the last XCD to start]\n  ", tag, grid, n);
    for (int i = 0; i < 8; i++) printf("xcc%d %.2f/%.2f [%d]  ", i, first[i] / n, last[i] / n, lastCount[i]);
    printf("\n");
}

int main() {
    const int threads = 512, lds = 40 * 1024, L = 64;
    for (int grid : {128, 192, 256}) {
        unsigned long long* d_t; unsigned* d_x; float* d_s;
        CK(hipMalloc(&d_t, (size_t)L * grid * 8)); CK(hipMalloc(&d_x, (size_t)L * grid * 4)); CK(hipMalloc(&d_s, 4));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        std::vector<unsigned long long> T((size_t)L * grid); std::vector<unsigned> X((size_t)L * grid);
        // (a) isolated launches
        for (int l = 0; l < L; l++) {
            hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), lds, 0, d_t + (size_t)l * grid, d_x + (size_t)l * grid, d_s, 2000);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(T.data(), d_t, T.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(X.data(), d_x, X.size() * 4, hipMemcpyDeviceToHost));
        report("isolated      ", grid, L, T, X);
        // (b) back-to-back kernel nodes of one hipGraph (each ~8 us long), replayed
        hipStream_t st; CK(hipStreamCreate(&st));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int l = 0; l < L; l++) hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), lds, st, d_t + (size_t)l * grid, d_x + (size_t)l * grid, d_s, 4000);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; r++) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(T.data(), d_t, T.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(X.data(), d_x, X.size() * 4, hipMemcpyDeviceToHost));
        report("graph, chained", grid, L, T, X);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
        CK(hipFree(d_t)); CK(hipFree(d_x)); CK(hipFree(d_s));
    }
    return 0;
}
