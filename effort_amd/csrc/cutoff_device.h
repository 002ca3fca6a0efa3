// block_find_cutoff -- findCutoff32 (bucketMul.metal:141-247) as a workgroup-level device function, shared by
// the standalone cutoff kernel and the fused multiply kernel (where every workgroup evaluates it redundantly
// instead of waiting on another kernel).
//
// The reference's loop is a bisection on an f32 threshold: every round counts how many of the 4096
// bf16-rounded probe products exceed the threshold, until the count hits 4096-q, or the bounds/counters
// converge, or 100 rounds pass.  With quantised (bf16) values the count usually steps OVER the requested
// rank; the bracket then shrinks float by float and the loop runs ~25-100 rounds (measured: the literal port
// took 30 us on MI355X).  Three observations make it cheap without changing a single result bit:
//   * the values are non-negative bf16 numbers, so  value > threshold  <=>  pattern(value) > bits(threshold)>>16
//     as integers: a count depends only on the threshold's top 16 bits;
//   * therefore the counts already obtained at the two current bounds answer every later threshold that
//     falls into one of their two bf16 cells -- which is every round once the bracket is narrower than two
//     cells, i.e. all of the long tail; only the first ~10 rounds need a real (workgroup-wide) count;
//   * once the midpoint stops moving the loop body has reached a fixed point, so the value the reference
//     would write after grinding on to round 101 is already known.
// A real count is 4096/NT integer compares per lane, wave ballots, one LDS atomic per wave and one barrier.
#pragma once
#include "effort_internal.h"

namespace effort {

constexpr uint32_t kCutoffLdsBytes = 64;      // 4 rotating count slots + min/max words

// NT threads (multiple of 64, <= 1024, dividing 4096); lds = kCutoffLdsBytes of scratch (4-byte aligned).
// v / pr are the loaded inputs of THIS thread: v[j], probes[j] for j = tid + NT*k.  Returns the cutoff in
// every lane.  Contains barriers: call from uniform control flow.
template <int NT>
__device__ __forceinline__ float block_find_cutoff(const float (&vj)[4096 / NT], const uint16_t (&prj)[4096 / NT],
                                                   uint32_t q, char* lds, unsigned long long* dbg = nullptr) {
    static_assert(4096 % NT == 0 && NT % 64 == 0 && NT <= 1024, "block_find_cutoff: bad workgroup size");
    constexpr int VPT = 4096 / NT;                      // probe products per thread
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(lds);            // [4] rotating count slots
    uint32_t* s_mm = reinterpret_cast<uint32_t*>(lds) + 4;         // [0] min pattern, [1] max pattern
    const int tid = threadIdx.x, lane = tid & 63;
    if (dbg && tid == 0) { dbg[0] = wall_clock64(); dbg[6] = clock64(); }

    if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; s_cnt[2] = 0; s_cnt[3] = 0; s_mm[0] = 0xFFFFFFFFu; s_mm[1] = 0; }
    __syncthreads();
    // the 4096 values bf16(|(1e5*v[j]) * bf16(probe[j])|), products evaluated left to right (:160), kept as
    // their 16-bit patterns
    uint32_t vp[VPT];
    uint32_t pmin = 0xFFFFu, pmax = 0u;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const float t = kCutoffScale * vj[k];
        const float u = t * bf16_round(half_bits_to_float(prj[k]));
        vp[k] = __float_as_uint(bf16_round(fabsf(u))) >> 16;
        pmin = min(pmin, vp[k]);
        pmax = max(pmax, vp[k]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        pmin = min(pmin, (uint32_t)__shfl_xor((int)pmin, off));
        pmax = max(pmax, (uint32_t)__shfl_xor((int)pmax, off));
    }
    if (lane == 0) { atomicMin(&s_mm[0], pmin); atomicMax(&s_mm[1], pmax); }
    __syncthreads();
    if (dbg && tid == 0) dbg[1] = wall_clock64();
    // The reference starts each thread's min at 999 / max at -999, clamps per simdgroup and stores the
    // simdgroup results as bfloat (999 -> 1000) before the cross-simdgroup reduction (:155-190).  All values
    // are non-negative bf16 numbers, so the net effect is minBound = min(globalMin, 1000), maxBound = globalMax.
    float minBound = fminf(__uint_as_float(s_mm[0] << 16), 1000.0f);
    float maxBound = __uint_as_float(s_mm[1] << 16);

    int nCounts = 0;
    auto count_above = [&](uint32_t p) -> uint32_t {    // #{value > threshold}; workgroup-wide, uniform call
        uint32_t wc = 0;
#pragma unroll
        for (int k = 0; k < VPT; k++) wc += (uint32_t)__popcll(__ballot(vp[k] > p));
        const int slot = nCounts & 3;
        if (tid == 0) s_cnt[(nCounts + 1) & 3] = 0;     // slot of the NEXT count: idle since three counts ago
        if (lane == 0) atomicAdd(&s_cnt[slot], wc);
        __syncthreads();
        nCounts++;
        return s_cnt[slot];
    };

    float newBound = (minBound + maxBound) / 2;          // :195
    const uint32_t effort = 4096u - q;                   // :154
    int loops = 0, minCount = 4096, maxCount = 0;        // :175-176,198
    uint32_t patLo = 0xFFFFFFFFu, patHi = 0xFFFFFFFFu, cLo = 0, cHi = 0;   // counts known at the current bounds
    for (;;) {
        loops += 1;
        const uint32_t p = __float_as_uint(newBound) >> 16;
        uint32_t countAbove;
        if (p == patHi) countAbove = cHi;
        else if (p == patLo) countAbove = cLo;
        else countAbove = count_above(p);
        if (countAbove < effort) { maxBound = newBound; maxCount = (int)countAbove; patHi = p; cHi = countAbove; }   // :214-220
        else { minBound = newBound; minCount = (int)countAbove; patLo = p; cLo = countAbove; }
        const float prev = newBound;
        newBound = (maxBound + minBound) / 2;                                               // :222
        int d = maxCount - minCount; if (d < 0) d = -d;
        if (countAbove == effort || (maxBound - minBound < 0.00001f) || d < 3) break;      // :227-229
        if (loops > 100) break;                                                            // :236
        if (newBound == prev) break;                     // fixed point of the loop body: rounds ..101 change nothing
    }
    if (dbg && tid == 0) { dbg[2] = wall_clock64(); dbg[3] = dbg[2]; dbg[4] = dbg[2]; dbg[5] = (unsigned long long)loops * 1000ull + nCounts; dbg[7] = clock64(); }
    return newBound;
}

}  // namespace effort
