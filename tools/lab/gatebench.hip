// gatebench -- does launching a DEPENDENT kernel early pay on this chip (tools only)?
// A decode loop is a chain  C_0 -> P_0 -> C_1 -> P_1 -> ...: C = a 192-workgroup kernel shaped like a lone bucketMul (68 KB of
// LDS, 512 threads; weight-only head: 16 KB of "means" per workgroup; then the input v: 16 KB; then a latency chain; then a ticket
// and a tail), P = a one-workgroup glue kernel that turns C's output into the next C's input.
//   serial : one stream, every kernel after the previous one (what a captured decode graph does today)
//   gated  : C_{k+1} is enqueued after P_{k-1} (two side streams), so it starts -- dispatch ramp, kernel arguments, the weight-only
//            head -- while C_k still runs, then SPINS on a flag that P_k raises when v is complete.  At most two C's are resident.
// Prints us per (C, P) pair for both.     hipcc --offload-arch=gfx950 -O3 -o build/gatebench tools/lab/gatebench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kSc1 = 16;
constexpr int kWG = 192, kThreads = 512, kV = 4096;

__global__ __launch_bounds__(512, 2) void consumer(const uint32_t* __restrict__ weights, const float* v, uint32_t* gate, float* out, uint32_t* ticket,
                                                    const uint32_t* __restrict__ chase, int chain, float* sink) {
    extern __shared__ char smem[];
    float* lds = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    // weight-only head: 32 bytes per thread
    const uint4* wp = reinterpret_cast<const uint4*>(weights) + ((size_t)blockIdx.x * kThreads + tid) * 2;
    uint4 w0 = wp[0], w1 = wp[1];
    lds[tid] = __uint_as_float(w0.x ^ w1.y);
    // the gate: v is complete
    if (tid == 0) {
        while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(v), 0, kV * 4, 0x00020000);
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < kV / kThreads; i++) acc += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, (uint32_t)(tid + i * kThreads) * 4u, 0, kSc1));
    // a chain of dependent memory round trips and barriers (cutoff, selection, first rows ...)
    uint32_t p = (uint32_t)(blockIdx.x * 64 + (tid & 63));
    for (int k = 0; k < chain; k++) {
        p = chase[p & 0xFFFFFu];
        lds[tid] += __uint_as_float(p);
        __syncthreads();
        acc += lds[(tid + 64) & (kThreads - 1)];
    }
    // tail: partial result out (write-through), ticket, the last arriver writes the output
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, (kV + kWG * kThreads) * 4, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc), ro, (uint32_t)(kV + blockIdx.x * kThreads + tid) * 4u, 0, kSc1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ uint32_t s_last;
    if (tid == 0) s_last = atomicAdd(ticket, 1u) == (uint32_t)kWG - 1u;
    __syncthreads();
    if (s_last) {
        float s = 0.0f;
        for (int i = 0; i < 8; i++) s += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ro, (uint32_t)(kV + (i * 24) * kThreads + tid) * 4u, 0, kSc1));
#pragma unroll
        for (int i = 0; i < kV / kThreads; i++) out[tid + i * kThreads] = s * 1e-9f + (float)i;
        if (tid == 0) *ticket = 0u;
    }
    if (acc == 1234.5f) sink[0] = acc;
}

// the glue kernel: one workgroup, out -> v (e.g. a normalisation), then raises the next consumer's gate and lowers its own predecessor's
__global__ __launch_bounds__(1024) void producer(const float* out, float* v, uint32_t* gateNext, uint32_t* gatePrev) {
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(v, 0, kV * 4, 0x00020000);
    __shared__ float part[16];
    float x[4], s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) { x[i] = out[tid + i * 1024]; s += x[i] * x[i]; }
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    float tot = 0.0f;
    for (int i = 0; i < 16; i++) tot += part[i];
    const float inv = 1.0f / sqrtf(tot / kV + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; i++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x[i] * inv), rv, (uint32_t)(tid + i * 1024) * 4u, 0, kSc1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(gatePrev, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(gateNext, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int N = 64, chain = argc > 1 ? atoi(argv[1]) : 14, lds = 68 * 1024, only = argc > 2 ? atoi(argv[2]) : -1;      // only: 0 serial, 1 gated
    uint32_t *d_w, *d_gate, *d_ticket, *d_chase; float *d_v, *d_out, *d_sink;
    CK(hipMalloc(&d_w, (size_t)kWG * kThreads * 32)); CK(hipMemset(d_w, 1, (size_t)kWG * kThreads * 32));
    CK(hipMalloc(&d_gate, (N + 1) * 4)); CK(hipMalloc(&d_ticket, 2 * 4)); CK(hipMemset(d_ticket, 0, 8));
    CK(hipMalloc(&d_v, kV * 4)); CK(hipMemset(d_v, 0, kV * 4));
    CK(hipMalloc(&d_out, (kV + kWG * kThreads) * 4)); CK(hipMemset(d_out, 0, (kV + kWG * kThreads) * 4));
    CK(hipMalloc(&d_sink, 4));
    std::vector<uint32_t> h(1 << 20);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)((i * 2654435761ull + 12345ull) & 0xFFFFFu);
    CK(hipMalloc(&d_chase, h.size() * 4)); CK(hipMemcpy(d_chase, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&consumer), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipStream_t m, s[2];
    CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    std::vector<hipEvent_t> evC(N), evP(N); hipEvent_t evStart;
    for (auto& e : evC) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : evP) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&evStart, hipEventDisableTiming));
    auto reset_gates = [&]() { std::vector<uint32_t> g(N + 1, 0u); g[0] = 1u; CK(hipMemcpy(d_gate, g.data(), (N + 1) * 4, hipMemcpyHostToDevice)); };
    hipGraph_t graphs[2]; hipGraphExec_t execs[2];
    for (int mode = 0; mode < 2; mode++) {
        if (only >= 0 && mode != only) continue;       // (only >= 2: no graphs at all)
        CK(hipStreamBeginCapture(m, hipStreamCaptureModeGlobal));
        if (mode == 1) CK(hipEventRecord(evStart, m));
        for (int k = 0; k < N; k++) {
            hipStream_t sk = mode == 1 ? s[k & 1] : m;
            if (mode == 1) CK(hipStreamWaitEvent(sk, k >= 2 ? evP[k - 2] : evStart, 0));
            // two tickets: consecutive consumers may be resident together
            hipLaunchKernelGGL(consumer, dim3(kWG), dim3(kThreads), lds, sk, d_w, d_v, d_gate + k, d_out, d_ticket + (k & 1), d_chase, chain, d_sink);
            if (mode == 1) { CK(hipEventRecord(evC[k], sk)); CK(hipStreamWaitEvent(m, evC[k], 0)); }
            hipLaunchKernelGGL(producer, dim3(1), dim3(1024), 0, m, d_out, d_v, d_gate + (k + 1) % N, d_gate + k);
            if (mode == 1) CK(hipEventRecord(evP[k], m));
        }
        CK(hipStreamEndCapture(m, &graphs[mode]));
        CK(hipGraphInstantiate(&execs[mode], graphs[mode], nullptr, nullptr, 0));
    }
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    if (only >= 2) {                 // eager: the same two topologies enqueued directly (three streams = three hardware queues), no graph
        const int mode = only - 2;
        auto pass = [&]() {
            if (mode == 1) CK(hipEventRecord(evStart, m));
            for (int k = 0; k < N; k++) {
                hipStream_t sk = mode == 1 ? s[k & 1] : m;
                if (mode == 1) CK(hipStreamWaitEvent(sk, k >= 2 ? evP[k - 2] : evStart, 0));
                hipLaunchKernelGGL(consumer, dim3(kWG), dim3(kThreads), lds, sk, d_w, d_v, d_gate + k, d_out, d_ticket + (k & 1), d_chase, chain, d_sink);
                if (mode == 1) { CK(hipEventRecord(evC[k], sk)); CK(hipStreamWaitEvent(m, evC[k], 0)); }
                hipLaunchKernelGGL(producer, dim3(1), dim3(1024), 0, m, d_out, d_v, d_gate + (k + 1) % N, d_gate + k);
                if (mode == 1) CK(hipEventRecord(evP[k], m));
            }
        };
        for (int round = 0; round < 3; round++) {
            reset_gates();
            pass(); CK(hipDeviceSynchronize());
            const int reps = 20;
            CK(hipEventRecord(t0, m));
            for (int r = 0; r < reps; r++) pass();
            CK(hipEventRecord(t1, m)); CK(hipEventSynchronize(t1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            printf("chain %2d  eager %-6s  %.2f us per (consumer, producer) pair\n", chain, mode ? "gated" : "serial", ms * 1e3 / (reps * N));
        }
        return 0;
    }
    for (int round = 0; round < 3; round++)
        for (int mode = 0; mode < 2; mode++) {
            if (only >= 0 && mode != only) continue;
            reset_gates();
            CK(hipGraphLaunch(execs[mode], m)); CK(hipStreamSynchronize(m));
            const int reps = 20;
            CK(hipEventRecord(t0, m));
            for (int r = 0; r < reps; r++) CK(hipGraphLaunch(execs[mode], m));
            CK(hipEventRecord(t1, m)); CK(hipEventSynchronize(t1));
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            printf("chain %2d  %-6s  %.2f us per (consumer, producer) pair\n", chain, mode ? "gated" : "serial", ms * 1e3 / (reps * N));
        }
    return 0;
}
