// block_find_cutoff -- findCutoff32 (bucketMul.metal:141-247) as a workgroup-level device function, shared by
// the standalone cutoff kernel and the fused multiply kernel (where every workgroup evaluates it redundantly
// instead of waiting on another kernel).
//
// The reference's loop is a bisection on an f32 threshold: every round counts how many of the 4096
// bf16-rounded probe products exceed the threshold, until the count hits 4096-q, or the bounds/counters
// converge, or 100 rounds pass.  With quantised (bf16) values the count usually steps OVER the requested
// rank; the bracket then shrinks float by float and the loop runs ~25-100 rounds (a literal port: 30 us on
// MI355X, two barriers per round).  This version returns the same bits in ~5 us (of which ~3 us is the serial bisection: ~26 dependent rounds on one wave):
//   * the values are non-negative bf16 numbers, so  value > threshold  <=>  pattern(value) > bits(threshold)>>16
//     as integers: a count depends only on the bf16 cell the threshold falls into;
//   * so ONE pass builds a table over the cells between the smallest and the largest value -- an LDS histogram
//     (integer LDS atomics run at full rate) turned into suffix counts by a workgroup scan -- after which
//     count(threshold) is a single LDS lookup;
//   * the bisection itself is serial and identical in every lane, so one wave runs it (the others would only
//     compete for issue slots), with the reference's arithmetic and exit tests in the reference's order;
//   * once the two bounds sit in adjacent cells with known counts every later threshold falls into one of the
//     two cells: the long tail of the loop needs no lookups, just a few f32 operations per round;
//   * once the midpoint stops moving the loop body has reached a fixed point, so the value the reference would
//     write after grinding on to round 101 is already known.
// If the values span more cells than the table holds (only when some products are ~0: > 64 octaves of range)
// the first rounds are counted with wave ballots until the bracket fits.
#pragma once
#include "effort_internal.h"

namespace effort {

constexpr uint32_t kCutoffBinsPerThread = 8;
constexpr uint32_t kCutoffLdsBytes = 512;     // small scratch: count slots, per-wave min/max, scan totals, result
// The table's cells are read back by ONE wave, lane L taking the kSeg = cells / 64 consecutive cells [L * kSeg, (L + 1) * kSeg) with
// 16-byte loads: four pad dwords after every kSeg cells make the lanes' stride 4 (mod 64) dwords -- no bank conflicts.
__host__ __device__ constexpr uint32_t cutoff_table_bytes(int NT) { return ((uint32_t)NT * kCutoffBinsPerThread + 64u * 4u) * 4u; }
template <int NT> __device__ __forceinline__ uint32_t cutoff_cell_index(uint32_t c) {
    constexpr uint32_t kSeg = (uint32_t)NT * kCutoffBinsPerThread / 64u;
    static_assert((kSeg & (kSeg - 1u)) == 0u && kSeg >= kCutoffBinsPerThread, "cutoff table: cells per lane must be a power of two holding whole threads");
    return c + ((c / kSeg) << 2);
}
// every thread clears its kCutoffBinsPerThread cells (a caller that passes PREZERO = true does this, and a barrier, before the call)
template <int NT> __device__ __forceinline__ void cutoff_table_zero(uint32_t* tbl, int tid) {
    uint4* z4 = reinterpret_cast<uint4*>(tbl + cutoff_cell_index<NT>((uint32_t)tid * kCutoffBinsPerThread));
    z4[0] = make_uint4(0, 0, 0, 0); z4[1] = make_uint4(0, 0, 0, 0);
}
// PREZERO callers count into a FIXED window of cells before the value range is known (one barrier and one LDS round trip less on
// the chain): [2^-10, 2^-10 * 2^(cells / 128)) -- 2^22 for 4096 cells.  Nonzero values outside it: the table is rebuilt the usual way.
constexpr uint32_t kCutoffWindowBase = (127u - 10u) << 7;

// NT threads (multiple of 64, <= 1024, dividing 4096); lds = kCutoffLdsBytes of scratch; tbl =
// cutoff_table_bytes(NT) of scratch, 16-byte aligned (may alias memory the caller initialises afterwards).
// vj / prj are THIS thread's inputs: v[j], probes[j] for j = tid + NT*k.  Returns the cutoff in every lane.
// Contains barriers: call from uniform control flow.  `idle` is invoked by the waves that do not run the
// serial part, with the table still live (it may not touch tbl).
// The reference's bisection once its bounds sit in ADJACENT bfloat cells (minBound below X, maxBound at or above X = the first
// float of the upper cell): the count no longer tells the halves apart, every round just moves the bound on the midpoint's side
// (findCutoff32's loop with the count test decided by newBound >= X), until the midpoint repeats, the bounds are 1e-5 apart, or
// 100 rounds are spent.  Up to 19 dependent float rounds (0.5 us) -- whose outcome is known: non-adjacent floats have a midpoint
// strictly between them, so the bounds close in on [pred(X), X]; there the midpoint is a tie that rounds to the even mantissa --
// X, a bfloat pattern -- and repeats.  The other two exits cannot come first when ulp(pred(X)) >= 7.6e-6 (X >= 128: at
// adjacency the midpoint is computed BEFORE the 1e-5 test) and fewer than 60 rounds are spent (19 more at most).  Checked
// against the loop on 110 000 random states (bounds anywhere in their cells); smaller X and late rounds take the loop.
__device__ __forceinline__ float bisect_to_cell_edge(float newBound, float minBound, float maxBound, const float X, int& loops) {
    if (X >= 128.0f && loops <= 60) { loops += 19; return X; }
    for (;;) {
        loops += 1;
        if (newBound >= X) maxBound = newBound; else minBound = newBound;
        const float prev = newBound;
        newBound = (maxBound + minBound) / 2;
        if (maxBound - minBound < 0.00001f) break;
        if (loops > 100) break;
        if (newBound == prev) break;
    }
    return newBound;
}

#ifdef EFFORT_CUT_FINE            // lab builds: shader-clock stamps inside the cutoff (tools/lab/cutfine.py), ten of them at dbg[8..17]
#define EFFORT_FSTAMP(i) if (dbg && tid == 0) fine[i] = clock64();
#else
#define EFFORT_FSTAMP(i)
#endif
template <int NT, bool PREZERO = false, typename Idle>
__device__ __forceinline__ float block_find_cutoff(const float (&vj)[4096 / NT], const uint16_t (&prj)[4096 / NT],
                                                   uint32_t q, char* lds, uint32_t* tbl, Idle idle,
                                                   unsigned long long* dbg = nullptr) {
    static_assert(4096 % NT == 0 && NT % 64 == 0 && NT <= 1024, "block_find_cutoff: bad workgroup size");
    constexpr int VPT = 4096 / NT;                      // probe products per thread
    constexpr int NW = NT / 64;
    constexpr uint32_t BPT = kCutoffBinsPerThread, CAP = (uint32_t)NT * BPT;
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(lds);            // [4] rotating count slots (ballot passes)
    uint32_t* s_wm = reinterpret_cast<uint32_t*>(lds) + 32;        // [3][NW] per wave: min pattern, max pattern, min NONZERO pattern (16-byte aligned)
    float* s_res = reinterpret_cast<float*>(lds) + 24;             // [0] result
    int tid0 = threadIdx.x;
    asm volatile("" : "+v"(tid0));      // opaque: a caller looping over work items must not hoist tid-derived state
    const int tid = tid0, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (dbg && tid == 0) { dbg[0] = wall_clock64(); dbg[6] = clock64(); }
#ifdef EFFORT_CUT_FINE
    unsigned long long fine[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    EFFORT_FSTAMP(0)

    if (tid < 4) s_cnt[tid] = 0;
    // the 4096 values bf16(|(1e5*v[j]) * bf16(probe[j])|), products evaluated left to right (:160), kept as
    // their 16-bit patterns
    uint32_t vp[VPT];
    uint32_t pmin = 0xFFFFu, pmax = 0u, pminnz = 0xFFFFu;
    static_assert(VPT % 2 == 0, "block_find_cutoff: the values are converted to bfloat two at a time");
#pragma unroll
    for (int k = 0; k < VPT; k += 2) {
        // bfloat(probe) and bfloat(|product|), two values per conversion instruction (bf16_pack2)
        const uint32_t pr2 = bf16_pack2(half_bits_to_float(prj[k]), half_bits_to_float(prj[k + 1]));
        const float t0 = kCutoffScale * vj[k], t1 = kCutoffScale * vj[k + 1];
        const float u0 = t0 * __uint_as_float(pr2 << 16), u1 = t1 * __uint_as_float(pr2 & 0xFFFF0000u);
        const uint32_t v2 = bf16_pack2(fabsf(u0), fabsf(u1));
        vp[k] = v2 & 0xFFFFu; vp[k + 1] = v2 >> 16;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            pmin = min(pmin, vp[k + h]);
            pmax = max(pmax, vp[k + h]);
            pminnz = min(pminnz, vp[k + h] ? vp[k + h] : 0xFFFFu);
        }
    }
    // PREZERO (the table was cleared, and a barrier passed, before the call): count into the fixed window NOW -- the adds run
    // under the reductions below and are complete at the barrier that publishes the value range
    if constexpr (PREZERO) {
        // (no branch around the add: a value outside the window -- a zero, mostly -- counts into this lane's pad dwords, which nobody reads)
        const uint32_t pad = (uint32_t)lane * (CAP / 64u + 4u) + CAP / 64u;
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const uint32_t c = vp[k] - kCutoffWindowBase;
            atomicAdd(&tbl[c < CAP ? cutoff_cell_index<NT>(c) : pad], 1u);
        }
    }
    EFFORT_FSTAMP(1)
    // wave results on DPP (no LDS round trips), one slot per wave; the count table is zeroed under the same barrier (its
    // region is free on entry): ONE barrier where there were three
    pmin = wave_min_u32(pmin); pmax = wave_max_u32(pmax); pminnz = wave_min_u32(pminnz);
    if (lane == 0) { s_wm[wave] = pmin; s_wm[NW + wave] = pmax; s_wm[2 * NW + wave] = pminnz; }
    if constexpr (!PREZERO) cutoff_table_zero<NT>(tbl, tid);
    __syncthreads();
    EFFORT_FSTAMP(2)
    // (the waves' results in ONE batch of 16-byte loads: left to the scheduler they came in two, a second LDS latency on the chain)
    uint32_t mmin = 0xFFFFFFFFu, mmax = 0u, mnz = 0xFFFFFFFFu;
    if constexpr (NW % 4 == 0) {
        uint4 wm[3 * NW / 4];
#pragma unroll
        for (int i = 0; i < 3 * NW / 4; i++) wm[i] = reinterpret_cast<const uint4*>(s_wm)[i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NW / 4; i++) {
            const uint4 a = wm[i], b = wm[NW / 4 + i], c = wm[2 * NW / 4 + i];
            mmin = min(min(mmin, a.x), min(min(a.y, a.z), a.w));
            mmax = max(max(mmax, b.x), max(max(b.y, b.z), b.w));
            mnz = min(min(mnz, c.x), min(min(c.y, c.z), c.w));
        }
    } else {
#pragma unroll
        for (int w2 = 0; w2 < NW; w2++) { mmin = min(mmin, s_wm[w2]); mmax = max(mmax, s_wm[NW + w2]); mnz = min(mnz, s_wm[2 * NW + w2]); }
    }
    const uint32_t pminAll = mmin, pmaxAll = mmax;
    // Exact zeros (a zero input, or a probe zeroed as a Q4 outlier) exceed no threshold, so the count table only has
    // to start at the smallest NONZERO value: below it every count is the same.  (Without this, zeros make the value
    // range span every bf16 cell down to 0 and the bracket never fits the table: ~100 ballot rounds at effort 1.)
    const uint32_t pminNZ = min(mnz, pmaxAll);
    // The reference starts each thread's min at 999 / max at -999, clamps per simdgroup and stores the
    // simdgroup results as bfloat (999 -> 1000) before the cross-simdgroup reduction (:155-190).  All values
    // are non-negative bf16 numbers, so the net effect is minBound = min(globalMin, 1000), maxBound = globalMax.
    float minBound = fminf(__uint_as_float(pminAll << 16), 1000.0f);
    float maxBound = __uint_as_float(pmaxAll << 16);

    float newBound = (minBound + maxBound) / 2;          // :195
    const uint32_t effort = 4096u - q;                   // :154
    int loops = 0, minCount = 4096, maxCount = 0;        // :175-176,198
    constexpr uint32_t kNoLo = 0xFFFF0000u, kNoHi = 0xFFF00000u;
    uint32_t patLo = kNoLo, patHi = kNoHi;               // bf16 cells of the bounds whose counts are known
    int nPasses = 0;
    bool done = false;

    // one round of the reference's loop body (:199-246) given the count at newBound; returns true on exit
    auto round = [&](uint32_t countAbove) -> bool {
        loops += 1;
        const uint32_t p = __float_as_uint(newBound) >> 16;
        if (countAbove < effort) { maxBound = newBound; maxCount = (int)countAbove; patHi = p; }     // :214-220
        else { minBound = newBound; minCount = (int)countAbove; patLo = p; }
        const float prev = newBound;
        newBound = (maxBound + minBound) / 2;                                               // :222
        int d = maxCount - minCount; if (d < 0) d = -d;
        if (countAbove == effort || (maxBound - minBound < 0.00001f) || d < 3) return true;  // :227-229
        if (loops > 100) return true;                                                      // :236
        return newBound == prev;                         // fixed point: rounds ..101 would change nothing
    };

    // ---- rare: the value range exceeds the table; count with ballots until the bracket fits ----------------
    auto lowCell = [&]() { return patLo != kNoLo ? max(patLo, pminNZ) : pminNZ; };
    auto topCell = [&]() { return patHi != kNoHi ? patHi : pmaxAll; };
    while (!done && patHi != patLo + 1u && topCell() - lowCell() + 1u > CAP) {
        const uint32_t p = __float_as_uint(newBound) >> 16;
        uint32_t wc = 0;
#pragma unroll
        for (int k = 0; k < VPT; k++) wc += (uint32_t)__popcll(__ballot(vp[k] > p));
        const int slot = nPasses & 3;
        if (tid == 0) s_cnt[(nPasses + 1) & 3] = 0;     // slot of the NEXT pass: idle since three passes ago
        if (lane == 0) atomicAdd(&s_cnt[slot], wc);
        __syncthreads();
        nPasses++;
        done = round(s_cnt[slot]);
    }
    if (dbg && tid == 0) dbg[1] = wall_clock64();
    EFFORT_FSTAMP(3)

    if (!done && patHi != patLo + 1u) {                  // uniform
        // ---- histogram over the cells [base, top]; ONE wave turns it into the five order statistics the rounds need ----
        uint32_t base = lowCell();
        const uint32_t top = topCell();
        const uint32_t above = (patHi != kNoHi) ? (uint32_t)maxCount : 0u;     // values beyond the top cell
        bool counted = false;                                                  // (uniform)
        if constexpr (PREZERO) {
            // every nonzero value inside the window: the counts are in place, cells below the smallest value simply hold 0
            counted = nPasses == 0 && pminNZ >= kCutoffWindowBase && pmaxAll - kCutoffWindowBase < CAP;
            if (counted) base = kCutoffWindowBase;
            else { cutoff_table_zero<NT>(tbl, tid); __syncthreads(); }
        }
        if (!counted) {
#pragma unroll
            for (int k = 0; k < VPT; k++) if (vp[k] >= base && vp[k] <= top) atomicAdd(&tbl[cutoff_cell_index<NT>(vp[k] - base)], 1u);
            __syncthreads();
        }

        if (wave == 0) {
            // count(c) = #{values with pattern > base + c} (+ above).  Lane L adds up its kSeg cells; a DPP scan over the lanes
            // gives count(last cell of lane L) -- the first level of every search below; the second level reads the 64 (128)
            // cells of the one lane the answer lies in, one (two) per lane, and scans them the same way.  (Until round 5 the whole
            // workgroup turned the histogram into a table of suffix counts -- a scan across the waves, two more barriers, four
            // LDS round trips -- which wave 0 then probed.)
            constexpr uint32_t kSeg = CAP / 64u, kSub = (kSeg + 63u) / 64u, kSegP = kSeg + 4u;
            const uint4* row = reinterpret_cast<const uint4*>(tbl + (uint32_t)lane * kSegP);
            // (the loads of a batch are ALL issued before the first sum -- one LDS latency per batch of 16, ~250 cycles here, not one
            //  per four loads as the scheduler would have it)
            // (measured with shader-clock stamps, tools/lab/cutfine.py: a wave's sixteen 16-byte reads cost ~64 cycles each whatever the
            //  number of active lanes -- masking the lanes whose cells cannot hold a value was 140 cycles SLOWER; read + sum ~1050 cycles)
            uint32_t mine = 0;
            constexpr uint32_t kBatch = kSeg / 4u < 16u ? kSeg / 4u : 16u;
#pragma unroll
            for (uint32_t i0 = 0; i0 < kSeg / 4u; i0 += kBatch) {
                uint4 xs[kBatch];
#pragma unroll
                for (uint32_t i = 0; i < kBatch; i++) xs[i] = row[i0 + i];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (uint32_t i = 0; i < kBatch; i++) mine += (xs[i].x + xs[i].y) + (xs[i].z + xs[i].w);
                __builtin_amdgcn_sched_barrier(0);
            }
            EFFORT_FSTAMP(4)
            const uint32_t pre = wave_prefix_sum_u32(mine), wtotal = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63);
            const uint32_t c1 = above + wtotal - pre;                          // count(last cell of this lane's segment)
            const uint32_t allGE = above + wtotal;                             // values with pattern >= base
            // THE BISECTION WITHOUT ITS LOOKUPS.  A count enters a round of the reference's loop in three places: the comparison
            // `countAbove < effort` that steers the bounds, and the exit tests `countAbove == effort` and |maxCount - minCount| < 3.
            // Counts are monotone in the threshold's bf16 cell, so each of them is a comparison of CELLS with a few order
            // statistics of the values: with T(k) = the first cell whose count is below k (count(p) >= k  <=>  p < T(k)) and
            // m = effort,
            //     countAbove <  m                 <=>  p >= T(m)
            //     countAbove == m                 <=>  T(m+1) <= p < T(m)
            //     |maxCount - minCount| < 3       <=>  (count(hi) >= m-1 and count(lo) <= m+1) or (count(hi) >= m-2 and count(lo) <= m)
            // (count(hi) < m <= count(lo) always; bounds never set yet count 0 / 4096, the reference's initial values).  The five
            // cells T(m-2) .. T(m+2) are found ONCE; the rounds then run with no LDS round trip inside -- it was ~330 cycles a
            // round with the lookup on the dependent chain, 1.4 us for 9 rounds and 2.7 for 29 -- as a chain of VALU instructions.
            // Same float operations in the same order, the same exits in the same round: bit-identical
            // (tests/test_cutoff_trajectory_model.py restates this on the CPU).
            // T(k) = base + #{cells with count >= k} (counts fall with the cell): whole segments from the ballot over c1, the
            // rest from the segment the boundary lies in.  The five k are neighbours, so their segments nearly always coincide:
            // a segment is read and scanned once per run of equal segments (uniform branch).
            const int m = (int)effort;
            uint32_t tk[5], cnt[kSub], sgHeld = 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t u = 0; u < kSub; u++) cnt[u] = 0u;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const int k = m - 2 + j;
                const uint32_t sgAll = (uint32_t)__popcll(__ballot((int)c1 >= k));   // (counts <= 4096: the signed compare also serves k <= 0)
                const uint32_t sg = min(sgAll, 63u);
                if (sg != sgHeld) {                                                 // uniform
                    sgHeld = sg;
                    uint32_t x[kSub], later = (uint32_t)__builtin_amdgcn_readlane((int)c1, (int)sg);
#pragma unroll
                    for (uint32_t u = 0; u < kSub; u++) x[u] = (u * 64u + (uint32_t)lane < kSeg) ? tbl[sg * kSegP + min(u * 64u + (uint32_t)lane, kSeg - 1u)] : 0u;
#pragma unroll
                    for (int u = (int)kSub - 1; u >= 0; u--) {                       // count(cell) = everything in later cells
                        const uint32_t p2 = wave_prefix_sum_u32(x[u]), t2 = (uint32_t)__builtin_amdgcn_readlane((int)p2, 63);
                        cnt[u] = later + t2 - p2;
                        later += t2;
                    }
                }
                uint32_t n = sg * kSeg;
#pragma unroll
                for (uint32_t u = 0; u < kSub; u++) n += (uint32_t)__popcll(__ballot(u * 64u + (uint32_t)lane < kSeg && (int)cnt[u] >= k));
                // every count is >= k (k <= 0, or the whole table and -- its last cell holds `above` -- everything beyond it): +inf;
                // none is (k > 4096, or more than the values at or above the table's first cell): 0
                tk[j] = (k <= 0 || sgAll >= 64u) ? 0xFFFFFFFFu : ((k > 4096 || (int)allGE < k) ? 0u : base + n);
            }
            if (dbg && tid == 0) dbg[2] = wall_clock64();
            EFFORT_FSTAMP(5)
            uint32_t tM2 = tk[0], tM1 = tk[1], tM = tk[2], tP1 = tk[3], tP2 = tk[4];
            float nb = newBound, lo = minBound, hi = maxBound;
            uint32_t pLo = patLo, pHi = patHi, nLoops = (uint32_t)loops;
            // the third test needs count(hi) only as "how many of m-1, m-2 it reaches" and count(lo) as "how many of m+1, m+2":
            //   |maxCount - minCount| < 3  <=>  catHi > catLo.   Bounds never set yet count 0 / 4096 (the reference's initial values).
            uint32_t catHi = patHi != kNoHi ? (uint32_t)(patHi < tM1) + (uint32_t)(patHi < tM2) : (uint32_t)(maxCount >= m - 1) + (uint32_t)(maxCount >= m - 2);
            uint32_t catLo = patLo != kNoLo ? (uint32_t)(patLo < tP1) + (uint32_t)(patLo < tP2) : (uint32_t)(minCount >= m + 1) + (uint32_t)(minCount >= m + 2);
            // THE ROUNDS, kBlk AT A TIME, THEIR EXIT TESTS IN PARALLEL (round 5).  What makes a round depend on the one before it is
            // the bounds alone: newBound = (hi + lo) / 2, and which bound it replaces -- `countAbove < effort`, i.e. the midpoint's
            // bits against the first float of cell T(m) (non-negative floats order like their bit patterns).  That recurrence is six
            // VALU instructions.  Everything else a round does -- the three count-driven exit tests, the 1e-5 test, the 100-round
            // cap, the fixed point, the hand-over to the closed-form tail at adjacent cells -- only decides WHERE the loop stops, and
            // rounds run past that point have no side effect.  So a block of kBlk rounds runs the bare recurrence, identical in every
            // lane, round r leaving the midpoint it tested and the bounds after it in lane r; then lane r reconstructs round r on
            // its own -- whether a round of this block set each bound (a ballot), their cells and count categories
            // follow -- and evaluates the reference's exits for it; the first lane that stops (a ballot) hands its state
            // to the wave.  Same float operations on the same operands, the same exit taken in the same round: bit-identical
            // (tests/test_cutoff_trajectory_model.py restates the block form on the CPU against the count-by-count loop).  It was a
            // chain of ~38 instructions and a branch per round (~260 cycles: 1.0 us for 9 rounds, 3.2 for 29); a block of 16 rounds
            // is ~100 + ~70 instructions.
            constexpr int kBlk = 16;
            const uint32_t tMs0 = tM >= 0x10000u ? 0xFFFFFFFFu : (tM << 16);        // (T(m) is a cell, 0, or +inf)
            bool fin = done;
            if (!fin && pHi != pLo + 1u) {
                for (;;) {
                    float a = nb, l = lo, h = hi, rec = 0.0f, hAt = hi, lAt = lo;
                    uint32_t tMs = tMs0;
                    asm volatile("" : "+v"(a), "+v"(l), "+v"(h), "+v"(tMs));          // VGPRs: left uniform, hipcc splits every round between SALU and VALU
#pragma unroll
                    for (int r = 0; r < kBlk; r++) {
                        rec = lane == r ? a : rec;                                    // the midpoint round r tests
                        const bool below = __float_as_uint(a) >= tMs;                 // countAbove < effort
                        h = below ? a : h;                                            // :214-220
                        l = below ? l : a;
                        hAt = lane == r ? h : hAt;                                    // the bounds AFTER round r, kept by lane r (two selects off
                        lAt = lane == r ? l : lAt;                                    //  the chain; they were two ds_bpermutes after the block)
                        a = (h + l) / 2;                                              // :222
                    }
                    // lane r < kBlk: round r of the block
                    const uint32_t lr = (uint32_t)lane & (uint32_t)(kBlk - 1);
                    const uint32_t p = __float_as_uint(rec) >> 16;
                    const bool wentHi = __float_as_uint(rec) >= tMs;                  // this round's midpoint became the upper bound
                    const uint32_t kmask = (1u << kBlk) - 1u;
                    const uint32_t hb = (uint32_t)__ballot(wentHi) & kmask, lb = ~hb & kmask;      // (uniform) which rounds went which way
                    const uint32_t upto = (2u << lr) - 1u;
                    const uint32_t hSeen = hb & upto, lSeen = lb & upto;
                    const float hr = hAt, lor = lAt;                                  // the bounds AFTER round r (hSeen / lSeen: whether a round of THIS block set them)
                    const uint32_t pHr = hSeen ? __float_as_uint(hr) >> 16 : pHi, pLr = lSeen ? __float_as_uint(lor) >> 16 : pLo;
                    const uint32_t cHr = hSeen ? (uint32_t)(pHr < tM1) + (uint32_t)(pHr < tM2) : catHi;
                    const uint32_t cLr = lSeen ? (uint32_t)(pLr < tP1) + (uint32_t)(pLr < tP2) : catLo;
                    const float nbr = (hr + lor) / 2;                                 // :222
                    // :227-229 (countAbove == effort | the bounds | the counts), :236, and the fixed point
                    const bool finr = ((p >= tP1) & (p < tM)) | (hr - lor < 0.00001f) | (cHr > cLr) | (nLoops + lr + 1u > 100u) | (nbr == rec);
                    const bool stopr = finr | (pHr == pLr + 1u);                      // ... or the bounds sit in adjacent cells: the closed-form tail
                    const uint32_t sm = (uint32_t)__ballot(stopr) & kmask;
                    const int e = sm ? __builtin_ctz(sm) : kBlk - 1;                  // (uniform) the round the loop stops in, or the block's last
                    nb = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(nbr), e));
                    lo = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lor), e));
                    hi = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hr), e));
                    pLo = __builtin_amdgcn_readlane(pLr, e); pHi = __builtin_amdgcn_readlane(pHr, e);
                    catLo = __builtin_amdgcn_readlane(cLr, e); catHi = __builtin_amdgcn_readlane(cHr, e);
                    nLoops += (uint32_t)e + 1u;
                    if (sm) { fin = (__builtin_amdgcn_readlane((uint32_t)finr, e) & 1u) != 0u; break; }
                }
            }
            EFFORT_FSTAMP(6)
            done = fin;
            newBound = nb; minBound = lo; maxBound = hi; patLo = pLo; patHi = pHi; loops = (int)nLoops;
            if (!done) {
                // tail: bounds in adjacent cells with known counts (below the target at and above X, not below it
                // under X); those counts differ by >= 3 and neither equals the target, or the loop had exited.
                newBound = bisect_to_cell_edge(newBound, minBound, maxBound, __uint_as_float(patHi << 16), loops);
            }
            EFFORT_FSTAMP(7)
            if (lane == 0) s_res[0] = newBound;
        } else {
            idle();
        }
        __syncthreads();
        newBound = s_res[0];
    } else if (!done) {
        // adjacent cells already (possible only after ballot passes): tail needs no table
        newBound = bisect_to_cell_edge(newBound, minBound, maxBound, __uint_as_float(patHi << 16), loops);
        idle();
    } else {
        idle();
    }
    EFFORT_FSTAMP(8)
#ifdef EFFORT_CUT_FINE
    if (dbg && tid == 0) { for (int i = 3; i < 7; i++) dbg[8 + i] = fine[i]; }     // ([16..19]: the kernel's entry stamps; [23..24]: the reduction's)
#endif
    if (dbg && tid == 0) { dbg[3] = wall_clock64(); dbg[4] = dbg[3]; dbg[5] = (unsigned long long)loops * 1000ull + nPasses; dbg[7] = clock64(); }
    return newBound;
}

}  // namespace effort
