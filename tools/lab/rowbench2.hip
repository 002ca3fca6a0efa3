// rowbench2 -- rowbench's mode 0 (the kernel's E = 4 items: the 512-byte piece of every kept row of a 512-input slice, 768 items on
// persistent workgroups) read by a SOFTWARE-PIPELINED reader whose depth is a parameter: how much of a single launch's gap to the
// 8 TB/s roofline is the pattern's, and how much is the reader's memory-level parallelism?  (rowbench loads a batch's list entries,
// then its 16 rows, then waits: list latency + row latency exposed per batch.)
//   R   = rows per batch, NB = batches in flight per wave (loads in flight per lane = R * NB), WGS = workgroups (512-thread)
// Also prints when the workgroups END (device clock, relative to the first start): the launch's tail.
//   hipcc --offload-arch=gfx950 -O3 -o build/rowbench2 tools/lab/rowbench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr uint32_t kIn = 4096, kRanks = 16, kRowsPerMat = kIn * kRanks, kPitch = 1408, kRowBytes = 1376, kMats = 32;
struct Item { uint32_t first, count, off, bytes; };
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

template <int R, int NB, int AUX>
__global__ __launch_bounds__(512, 4) void read_kernel(const char* __restrict__ base, const uint32_t* __restrict__ list, const Item* __restrict__ items,
                                                       uint32_t nItems, uint32_t* __restrict__ queue, uint32_t* __restrict__ sink, unsigned long long* __restrict__ stamps) {
    __shared__ uint32_t s_item;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    if (threadIdx.x == 0) stamps[blockIdx.x * 2] = wall_clock64();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, -1, 0x00020000);
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(queue, 1u);
        __syncthreads();
        const uint32_t it = s_item;
        __syncthreads();
        if (it >= nItems) break;
        const Item I = items[it];
        const uint32_t o = I.off + min((uint32_t)lane * 8u, I.bytes - 8u);
        // batches b = wave, wave + 8, ...: batch b = rows [b * R, b * R + R).  NB batches in flight: x[q] holds batch k + q.
        u2 x[NB][R];
        uint32_t rowsN[R];                                   // the NEXT batch's row offsets (list entries asked for one batch ahead)
        const uint32_t nBat = (I.count + R - 1) / R;
        auto ask = [&](uint32_t b) {
#pragma unroll
            for (int u = 0; u < R; u++) rowsN[u] = list[I.first + min(b * R + (uint32_t)u, I.count - 1u)] * kPitch;
        };
        auto issue = [&](int q) {
#pragma unroll
            for (int u = 0; u < R; u++) x[q][u] = __builtin_amdgcn_raw_buffer_load_b64(rs, o, __builtin_amdgcn_readfirstlane(rowsN[u]), AUX);
        };
        uint32_t b = wave;
        ask(min(b, nBat - 1));
#pragma unroll
        for (int q = 0; q < NB - 1; q++) { issue(q); ask(min(b + 8u * (q + 1), nBat - 1)); }
        // steady state: issue batch k + NB - 1, consume batch k
        for (; b < nBat; b += 8u * NB) {
#pragma unroll
            for (int q = 0; q < NB; q++) {
                issue((q + NB - 1) % NB);
                ask(min(b + 8u * (q + NB), nBat - 1));
#pragma unroll
                for (int u = 0; u < R; u++) acc ^= x[q][u][0] ^ x[q][u][1];
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
    if (threadIdx.x == 0) stamps[blockIdx.x * 2 + 1] = wall_clock64();
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)kMats * kRowsPerMat * kPitch;
    char* d_base; CK(hipMalloc(&d_base, bytes)); CK(hipMemset(d_base, 1, bytes));
    std::mt19937 rng(1);
    std::normal_distribution<float> nd;
    std::vector<uint32_t> list; std::vector<Item> items;
    size_t want = 0;
    for (uint32_t m = 0; m < kMats; m++) {
        std::vector<float> av(kIn);
        for (auto& x : av) x = fabsf(nd(rng));
        for (uint32_t s0 = 0; s0 < kIn; s0 += 512u) {
            const uint32_t first = (uint32_t)list.size();
            auto thr = [](uint32_t rank) { return 0.06f + 0.22f * (float)rank * (1.0f + 0.035f * (float)rank); };
            for (uint32_t rank = 0; rank < kRanks; rank++)
                for (uint32_t j = s0; j < s0 + 512u; j++)
                    if (av[j] > thr(rank)) list.push_back(m * kRowsPerMat + rank * kIn + j);
            const uint32_t count = (uint32_t)list.size() - first;
            if (!count) continue;
            for (uint32_t t = 0; t < 3; t++) { const uint32_t b = t < 2 ? 512u : kRowBytes - 1024u; items.push_back({first, count, t * 512u, b}); want += (size_t)count * b; }
        }
    }
    uint32_t *d_list, *d_queue, *d_sink; Item* d_items; unsigned long long* d_st;
    CK(hipMalloc(&d_list, list.size() * 4)); CK(hipMemcpy(d_list, list.data(), list.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_items, items.size() * sizeof(Item))); CK(hipMemcpy(d_items, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_queue, 4)); CK(hipMalloc(&d_sink, 4)); CK(hipMalloc(&d_st, 1024 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, int wgs) {
        float best = 1e9f; std::vector<unsigned long long> st(2048), bestSt;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipMemset(d_queue, 0, 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 0, 0, d_base, d_list, d_items, (uint32_t)items.size(), d_queue, d_sink, d_st);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) { best = ms; CK(hipMemcpy(st.data(), d_st, wgs * 16, hipMemcpyDeviceToHost)); bestSt = st; }
        }
        unsigned long long t0 = ~0ull; for (int i = 0; i < wgs; i++) t0 = std::min(t0, bestSt[2 * i]);
        std::vector<double> ends; for (int i = 0; i < wgs; i++) ends.push_back((bestSt[2 * i + 1] - t0) / 100.0);     // 100 MHz
        std::sort(ends.begin(), ends.end());
        printf("%-28s wgs %4d  %7.1f MB  %7.1f us  %.2f TB/s   workgroup ends (us): min %.1f p10 %.1f med %.1f p90 %.1f max %.1f\n", name, wgs, want / 1e6, best * 1e3,
               want / (best * 1e-3) / 1e12, ends[0], ends[wgs / 10], ends[wgs / 2], ends[wgs * 9 / 10], ends[wgs - 1]);
    };
    run("R=8  NB=2 (16 in flight)", read_kernel<8, 2, 0>, 512);
    run("R=16 NB=2 (32)", read_kernel<16, 2, 0>, 512);
    run("R=8  NB=3 (24)", read_kernel<8, 3, 0>, 512);
    run("R=8  NB=4 (32)", read_kernel<8, 4, 0>, 512);
    run("R=16 NB=3 (48)", read_kernel<16, 3, 0>, 512);
    run("R=4  NB=2 (8)", read_kernel<4, 2, 0>, 512);
    run("R=8  NB=2 nt", read_kernel<8, 2, 2>, 512);
    run("R=8  NB=4 nt", read_kernel<8, 4, 2>, 512);
    run("R=8  NB=2, 768 wgs", read_kernel<8, 2, 0>, 768);
    run("R=8  NB=4, 768 wgs", read_kernel<8, 4, 0>, 768);
    run("R=8  NB=4, 256 wgs", read_kernel<8, 4, 0>, 256);
    run("R=16 NB=3, 256 wgs", read_kernel<16, 3, 0>, 256);
    return 0;
}
