// bucket_mul -- the hot path as ONE kernel launch per bucketMul call.
// Replaces the reference's six launches: findCutoff32, prepareDispatch, roundUp, zeroRange32, bucketMul,
// bucketIntegrate (bucketMul.metal:11-247) and, for Q4, prepareDispatchQ4, bucketMulQ4, calcOutliers
// (bucketMulQ4.metal:13-92).  On MI355X a dependent kernel boundary costs 1.5-5 us and the whole call moves
// ~23 MB (2.8 us of HBM time), so the call is a latency chain first and a bandwidth problem second: the design
// goal is one launch, one memory round trip per dependent step, and no host involvement.
//
// Work decomposition (MI355X-first, not the reference's [cols x 32] grid of 32-thread groups):
//   grid   = T column tiles x S row slices, one workgroup of W wave64 each; tiles*S <= resident capacity so
//            the grid is one round of workgroups; block id -> (tile, slice) remapped so all tiles of a slice sit
//            on one XCD (same L2: they share the stats lines and the 128-B lines that straddle two tiles).
//   slice  = a contiguous block of B input rows (all ranks of them) -> its kept bucket rows are neighbours in
//            HBM within each rank plane.
//   tile   = 64*E u16 columns; lane l of every wave owns columns l*E .. l*E+E-1 of the tile, so one wave load
//            instruction reads one contiguous 128*E-byte piece of one bucket row.
// Per workgroup:
//   A. every thread issues ALL the loads the selection needs up front -- its share of v and the probes, its
//      slice of v, the stats of the candidate rows it will test -- one memory round trip; the private
//      accumulator tiles are zeroed while those loads fly.
//   B. the cutoff is evaluated redundantly by every workgroup (cutoff_device.h: bit-exact findCutoff32), which
//      is cheaper than a kernel boundary or a cross-workgroup flag.
//   C. the 16*B (Q4: 8*B) candidate rows are tested exactly as prepareDispatch does (cutoff < (1e5*mean)*|v|)
//      and the survivors compacted, in ascending bucket-row order, into an LDS list (wave ballots + mbcnt
//      prefix; no global atomics, deterministic).
//   D. waves take list entries round-robin and stream those rows from HBM: two batches of 16 buffer loads in
//      flight per lane, the row offset in an SGPR (v_readlane of the decoded entry -> buffer soffset, no per-row
//      address VALU).  Each product is added into a PRIVATE per-wave LDS accumulator tile acc[slot][j][lane]
//      (slot = the 4 position bits of the f16 weight; Q4: sub-bucket*8 + the 3 position bits of the nibble) by a
//      plain ds_read / v_add / ds_write: the LDS float atomic (ds_add_f32) measured 0.5 elements/clk/CU on
//      MI355X against 6.8 for read-add-write and 2.1 for a 16-way register select chain (tools/microbench.hip).
//      The layout puts lane l on LDS bank l%32 whatever the slot, so the scatter is bank-conflict free; the E
//      (Q4: 4E) slots one lane touches for one row are distinct by construction, so their read-add-writes are
//      issued together; rows are processed in order, which fixes the f32 summation order.
//   E. the W private tiles are summed in wave order into one partial "slab", stored write-through; the workgroup
//      takes a ticket on its tile's arrival counter, and the LAST workgroup of each tile sums the S slabs in
//      slice order (and, for Q4, adds that output's outliers in table order) and writes out[].  The result does
//      not depend on arrival order: deterministic end to end.  (Q4 outliers: q4_outliers_kernel, launched next.)
#include "cutoff_device.h"

namespace effort {

constexpr int kBatch = 16;   // bucket rows per batch; two batches in flight per wave
constexpr int kPre = 4;      // candidate rows per thread whose stats are preloaded into registers

// Ablation builds for profiling (-DEFFORT_ABLATE_NOSCATTER=1 / -DEFFORT_ABLATE_NOLOAD=1); never shipped.
#ifndef EFFORT_ABLATE_NOSCATTER
#define EFFORT_ABLATE_NOSCATTER 0
#endif
#ifndef EFFORT_ABLATE_NOLOAD
#define EFFORT_ABLATE_NOLOAD 0
#endif

constexpr int kSc1 = 16;     // buffer aux bit: sc1 = write-through store / L1-bypassing load (cross-XCD visible)

template <int FMT> struct Fmt;
template <> struct Fmt<kFp16> { static constexpr int kAcc = 16; };
template <> struct Fmt<kQ4> { static constexpr int kAcc = 32; };

// One lane's piece of a bucket row: E u16 words.
template <int E> struct Piece;
template <> struct Piece<1> {
    uint32_t w;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { w = __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0); }
    __device__ __forceinline__ uint32_t word(int) const { return w & 0xFFFFu; }
    __device__ __forceinline__ uint32_t dword(int) const { return w; }
};
template <> struct Piece<2> {
    uint32_t w;
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { w = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0); }
    __device__ __forceinline__ uint32_t word(int j) const { return (w >> (16 * j)) & 0xFFFFu; }
    __device__ __forceinline__ uint32_t dword(int) const { return w; }
};
template <> struct Piece<4> {
    uint32_t w[2];
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
        auto t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0); w[0] = t[0]; w[1] = t[1];
    }
    __device__ __forceinline__ uint32_t word(int j) const { return (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu; }
    __device__ __forceinline__ uint32_t dword(int j) const { return w[j >> 1]; }
};
template <> struct Piece<8> {
    uint32_t w[4];
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
        auto t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0); w[0] = t[0]; w[1] = t[1]; w[2] = t[2]; w[3] = t[3];
    }
    __device__ __forceinline__ uint32_t word(int j) const { return (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu; }
    __device__ __forceinline__ uint32_t dword(int j) const { return w[j >> 1]; }
};

__host__ __device__ inline uint32_t align_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

// LDS carve (bytes): [W][tileFloats] f32 | vblk[B] f32 | misc 768 B | list[rows*B] u16 | dlist (Q4) f32
template <int FMT, int E, int W>
__host__ __device__ inline uint32_t lds_layout(uint32_t B, uint32_t rowsPerIn, uint32_t* offV, uint32_t* offC,
                                               uint32_t* offL, uint32_t* offD) {
    uint32_t o = (uint32_t)W * Fmt<FMT>::kAcc * E * 64 * 4;
    *offV = o; o += align_up(B * 4, 16);
    *offC = o; o += 768;                                   // [0..255] cutoff scratch, [256..511] wave counts [kPre][16], [512] flags
    *offL = o; o += align_up(rowsPerIn * B * 2, 16);
    *offD = o; if (FMT == kQ4) o += align_up(rowsPerIn * B * 4, 16);
    return o;
}

template <int FMT, int E, int W>
__global__ __launch_bounds__(64 * W) void bucket_mul_kernel(const MulArgs a) {
    constexpr int NACC = Fmt<FMT>::kAcc;
    constexpr int TILE_F = NACC * E * 64;
    constexpr int NT = 64 * W;
    constexpr int VPT = 4096 / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const MulGeom& g = a.g;
    // XCD-aware id -> (tile, slice): the dispatcher places block b on XCD b%8 (speed only).
    const uint32_t b = blockIdx.x, xcd = b & 7u, k = b >> 3;
    const uint32_t s = (k / g.tiles) * 8u + xcd, t = k % g.tiles;
    if (s >= g.slices) return;

    const int tid = threadIdx.x, lane = tid & 63;
    if (a.tstamp && tid == 0) { const unsigned long long now = wall_clock64(); atomicMin(&a.tstamp[0], now); atomicMax(&a.tstamp[25], now); }
    const bool stamp = a.tstamp && blockIdx.x == 0 && tid == 0;      // phase stamps of workgroup 0 (profiling aid)
    if (stamp) a.tstamp[16] = wall_clock64();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t B = g.sliceRows;
    uint32_t offV, offC, offL, offD;
    lds_layout<FMT, E, W>(B, g.rowsPerIn, &offV, &offC, &offL, &offD);
    float* acc = reinterpret_cast<float*>(smem);
    float* vblk = reinterpret_cast<float*>(smem + offV);
    uint32_t* wcnt = reinterpret_cast<uint32_t*>(smem + offC + 256);         // [kPre][16]
    uint32_t* flags = reinterpret_cast<uint32_t*>(smem + offC + 512);
    uint16_t* list = reinterpret_cast<uint16_t*>(smem + offL);
    float* dlist = reinterpret_cast<float*>(smem + offD);
    float* myacc = acc + wave * TILE_F + lane;

    const uint32_t j0 = s * B;
    const uint32_t nb = min(B, g.inDim - j0);
    const uint32_t e = a.expNo ? a.expNo[0] : 0u;

    // ---- A. everything the selection needs, in one round trip --------------------------------
    float vj[VPT]; uint16_t prj[VPT];
    const uint16_t* pr = a.probes + (size_t)e * kProbes;
    const bool fused = a.cutoffIn == nullptr;                    // uniform
#pragma unroll
    for (int i = 0; i < VPT; i++) { vj[i] = 0.0f; prj[i] = 0; }
    if (fused) {
#pragma unroll
        for (int i = 0; i < VPT; i++) { vj[i] = a.v[tid + NT * i]; prj[i] = pr[tid + NT * i]; }
    }
    // candidate slot c = r*NT + tid, in ascending bucket-row order:
    //   FP16 -> rank = c >> lg, jl = c & (2^lg - 1), 2^lg >= B  (rank-major rows, convert.metal:83-100; v index = row % inDim)
    //   Q4   -> jl = c >> 3,  rank = c & 7                      (input-major rows, bucketMulQ4.metal:46)
    const uint32_t lg = g.sliceLog2;
    const uint32_t nSlots = FMT == kFp16 ? (g.rowsPerIn << lg) : (nb << 3);
    float mean[kPre];
    uint32_t codes[kPre];
#pragma unroll
    for (int r = 0; r < kPre; r++) {
        const uint32_t c = r * NT + tid;
        mean[r] = 0.0f; codes[r] = 0xFFFFFFFFu;                 // 0xFFFFFFFF: no candidate in this slot
        if ((uint32_t)(r * NT) >= nSlots) continue;             // uniform: this round holds no slots at all
        if (FMT == kFp16) {
            const uint32_t rank = c >> lg, jl = c & ((1u << lg) - 1u);
            if (rank < g.rowsPerIn && jl < nb) {
                const size_t row = (size_t)e * g.expertRows + (size_t)rank * g.inDim + j0 + jl;
                mean[r] = half_bits_to_float(reinterpret_cast<const uint16_t*>(a.stats)[row * 4 + 3]);
                codes[r] = (rank << 12) | jl;
            }
        } else {
            const uint32_t jl = c >> 3, rank = c & 7u;
            if (jl < nb) {
                const size_t row = (size_t)e * g.expertRows + (size_t)(j0 + jl) * 8u + rank;
                mean[r] = reinterpret_cast<const float*>(a.stats)[row * 2 + 1];
                codes[r] = (jl << 3) | rank;
            }
        }
    }
    for (uint32_t jl = tid; jl < nb; jl += NT) vblk[jl] = a.v[j0 + jl];
    if (stamp) a.tstamp[17] = wall_clock64();

    // ---- B. cutoff (findCutoff32), redundantly per workgroup; its first barrier also publishes vblk.  Its
    //         lookup table borrows the tail of the accumulator region, so the tiles are zeroed around it: the
    //         waves that idle during the serial bisection zero theirs meanwhile, the rest right after. ---------
    uint32_t* tbl = reinterpret_cast<uint32_t*>(acc + W * TILE_F) - NT * kCutoffBinsPerThread;
    const bool tileUnderTable = (uint32_t)(wave + 1) * TILE_F > (uint32_t)W * TILE_F - NT * kCutoffBinsPerThread;
    auto zero_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NACC * E; i++) myacc[i * 64] = 0.0f;
    };
    float cutoff;
    if (fused) {
        cutoff = block_find_cutoff<NT>(vj, prj, a.q, smem + offC, tbl,
                                       [&]() { if (wave != 0 && !tileUnderTable) zero_tile(); },
                                       stamp ? a.tstamp + 8 : nullptr);
        if (wave == 0 || tileUnderTable) zero_tile();
        if (blockIdx.x == 0 && tid == 0) a.cutoffOut[0] = cutoff;    // BucketMul.cutoff (bucketMul.swift:22)
    } else {
        // split mode: the standalone cutoff kernel ran first on this stream (cheaper in aggregate when several
        // calls overlap: one workgroup evaluates it instead of all of them)
        cutoff = a.cutoffIn[0];
        zero_tile();
        __syncthreads();                                             // publishes vblk
    }
    if (stamp) a.tstamp[18] = wall_clock64();

    // ---- C. keep test (bucketMul.metal:69 / bucketMulQ4.metal:47) + ordered compaction -----------
    // All rounds are balloted first; the kPre*W per-wave counts meet in LDS at ONE barrier; list position of a
    // kept slot = (kept slots of earlier rounds) + (earlier waves of its round) + (earlier lanes of its wave).
    bool keep[kPre]; uint32_t pre[kPre]; float xs[kPre];
#pragma unroll
    for (int r = 0; r < kPre; r++) {
        keep[r] = false; pre[r] = 0; xs[r] = 0.0f;
        if ((uint32_t)(r * NT) < nSlots && !(a.ablate & 8u)) {  // uniform
            const bool cand = codes[r] != 0xFFFFFFFFu;
            const uint32_t jl = FMT == kFp16 ? (codes[r] & 4095u) : (codes[r] >> 3);
            xs[r] = cand ? vblk[jl] : 0.0f;
            keep[r] = cand && (cutoff < (kCutoffScale * mean[r]) * fabsf(xs[r]));
            const unsigned long long m = __ballot(keep[r]);
            pre[r] = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (lane == 0) wcnt[r * 16 + wave] = (uint32_t)__popcll(m);
        } else if (lane == 0) {
            wcnt[r * 16 + wave] = 0u;
        }
    }
    __syncthreads();
    // lanes 0..kPre*W-1 hold one (round, wave) count each, in list order; inclusive scan with shuffles
    static_assert(kPre * W <= 64, "selection scan: kPre*W must fit one wave");
    uint32_t inc = (lane < kPre * W) ? wcnt[(lane / W) * 16 + (lane % W)] : 0u;
    const uint32_t own = inc;
#pragma unroll
    for (int off = 1; off < kPre * W; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off);
        inc += (lane >= off) ? o : 0u;
    }
    const uint32_t exc = inc - own;
    uint32_t n = __shfl(inc, kPre * W - 1);
#pragma unroll
    for (int r = 0; r < kPre; r++) {
        const uint32_t base = __shfl(exc, r * W + wave);
        if (keep[r]) {
            list[base + pre[r]] = (uint16_t)codes[r];
            if (FMT == kQ4) dlist[base + pre[r]] = xs[r] * mean[r];          // entry value = v*mean (:52)
        }
    }
    __syncthreads();
    if (t == 0 && tid == 0) a.sliceCounts[s] = n;                  // dispatch.size = sum over slices (test hook)
    if (stamp) a.tstamp[19] = wall_clock64();

    // ---- D. stream the kept rows, scatter-accumulate into the private LDS tile ----------------
    const uint32_t col = t * (64u * E) + (uint32_t)lane * E;
    const bool colOK = col < g.cols;                               // a piece past the last column is skipped whole
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.buckets), 0, (int)min((size_t)0xFFFFFFFFu, (size_t)g.numExperts * g.expertRows * g.cols * 2u), 0x00020000);
    const uint32_t voff = (colOK ? col : 0u) * 2u;                 // lanes past the last column re-read column 0
    const uint32_t nU = (a.ablate & 4u) ? 0u : __builtin_amdgcn_readfirstlane(n);   // n is workgroup-uniform; keep it in an SGPR
    const uint32_t myRows = (nU > (uint32_t)wave) ? (nU - wave + W - 1) / W : 0u;   // entries wave, wave+W, ...

    // Software pipeline: the loads of batch k+1 are issued before batch k is accumulated, so a wave keeps
    // up to 2*kBatch row pieces in flight.  Loads are never predicated: a batch that runs past the wave's
    // last row re-reads that last row (an L1/L2 hit) -- a branch around a load makes hipcc drain
    // vmcnt(0) -- and the accumulate step skips the surplus with a wave-uniform test.
    auto decode = [&](uint32_t i0, uint32_t& boff, float& dv) {
        // lane u decodes list entry i0+u (clamped); the row loops read it back with v_readlane into SGPRs
        boff = 0; dv = 0.0f;
        if (lane < kBatch && myRows) {
            const uint32_t kk = wave + W * min(i0 + (uint32_t)lane, myRows - 1u);
            const uint32_t code = list[kk];
            uint32_t rowIdx;
            if (FMT == kFp16) { const uint32_t rank = code >> 12, jl = code & 4095u; rowIdx = rank * g.inDim + j0 + jl; dv = vblk[jl]; }
            else { const uint32_t jl = code >> 3, rank = code & 7u; rowIdx = (j0 + jl) * 8u + rank; dv = dlist[kk]; }
            boff = (e * g.expertRows + rowIdx) * g.cols * 2u;       // byte offset of the bucket row (< 4 GiB, checked at registration)
        }
    };
    auto issue = [&](Piece<E> (&piece)[kBatch], uint32_t boff) {
#pragma unroll
        for (int u = 0; u < kBatch; u++) piece[u].load(rsrc, voff, EFFORT_ABLATE_NOLOAD ? 0u : __builtin_amdgcn_readlane(boff, u));
    };
    // One row piece -> E (Q4: 4E) read-add-writes on this lane's private accumulator column.  The LDS byte
    // address is built with v_bfe_u32 + v_lshl_add_u32 (hipcc otherwise spends three VALU ops per element on it)
    // and the product is accumulated with one v_fma_f32 (a single rounding, where the reference's
    // `v = d.x*float(w); acc += v` rounds twice: well inside the parity tolerance and never less accurate).
    using lds_f = __attribute__((address_space(3))) float;
    const uint32_t accB = (uint32_t)(size_t)(lds_f*)myacc;
    constexpr int kShift = (E == 1 ? 8 : E == 2 ? 9 : E == 4 ? 10 : 11);      // log2(E * 64 lanes * 4 bytes)
    auto row = [&](const Piece<E>& pc, float dd) {
        if (FMT == kFp16) {
            // bucketMul.metal:100-106: v = d.x*float(w) with the position bits left in w; acc[pos] += v
            float w[E], old[E]; lds_f* p[E];
#pragma unroll
            for (int j = 0; j < E; j++) {
                const uint32_t dw = pc.dword(j);                       // the dword holding element j (bits 16*(j&1)..)
                w[j] = half_bits_to_float((uint16_t)(dw >> (16 * (j & 1))));
                uint32_t a2;
                asm("v_bfe_u32 %0, %1, %2, 4\n\tv_lshl_add_u32 %0, %0, %3, %4" : "=&v"(a2) : "v"(dw), "n"(16 * (j & 1)), "n"(kShift), "v"(accB));
                p[j] = (lds_f*)(size_t)a2 + j * 64;
            }
#pragma unroll
            for (int j = 0; j < E; j++) old[j] = *p[j];
#pragma unroll
            for (int j = 0; j < E; j++) *p[j] = __builtin_fmaf(dd, w[j], old[j]);
        } else {
            // bucketMulQ4.metal:76-81: low nibble first <-> sub-bucket 3,2,1,0; acc += (n&8) ? -d : d
            float val[4 * E], old[4 * E]; lds_f* p[4 * E];
#pragma unroll
            for (int j = 0; j < E; j++) {
                const uint32_t x = pc.word(j);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t nib = (x >> (4 * q)) & 15u;
                    val[4 * j + q] = (nib & 8u) ? -dd : dd;
                    p[4 * j + q] = (lds_f*)myacc + (((3 - q) * 8 + (nib & 7u)) * E + j) * 64;
                }
            }
#pragma unroll
            for (int i = 0; i < 4 * E; i++) old[i] = *p[i];
#pragma unroll
            for (int i = 0; i < 4 * E; i++) *p[i] = old[i] + val[i];
        }
    };
    auto accumulate = [&](const Piece<E> (&piece)[kBatch], float dv, uint32_t nv) {
        if (EFFORT_ABLATE_NOSCATTER) {       // ablation build: keep the loads live, skip the LDS scatter
#pragma unroll
            for (int u = 0; u < kBatch; u++) asm volatile("" ::"v"(piece[u].word(E - 1)), "v"(dv));
            return;
        }
        if (!colOK) return;                  // lanes past the last column sit the batch out (one exec change)
        if (nv == (uint32_t)kBatch) {        // uniform: full batch, no per-row test
#pragma unroll
            for (int u = 0; u < kBatch; u++) row(piece[u], __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dv), u)));
        } else {
#pragma unroll
            for (int u = 0; u < kBatch; u++)
                if ((uint32_t)u < nv) row(piece[u], __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dv), u)));
        }
    };

    if (myRows) {
        Piece<E> pa[kBatch], pb[kBatch];
        uint32_t boffA, boffB; float dvA, dvB;
        decode(0, boffA, dvA);
        issue(pa, boffA);
        for (uint32_t i0 = 0; i0 < myRows; i0 += 2 * kBatch) {
            decode(i0 + kBatch, boffB, dvB);
            issue(pb, boffB);
            accumulate(pa, dvA, __builtin_amdgcn_readfirstlane(min((uint32_t)kBatch, myRows - i0)));
            decode(i0 + 2 * kBatch, boffA, dvA);
            issue(pa, boffA);
            accumulate(pb, dvB, __builtin_amdgcn_readfirstlane(i0 + kBatch < myRows ? min((uint32_t)kBatch, myRows - i0 - kBatch) : 0u));
        }
    }
    __syncthreads();
    if (stamp) a.tstamp[20] = wall_clock64();
    if (a.tstamp && tid == 0) atomicMax(&a.tstamp[26], (unsigned long long)wall_clock64());

    // ---- E. W private tiles -> one slab (native [slot][j][lane] order), write-through; ticket; last arriver
    //         of the tile reduces the S slabs in slice order and writes out[] -------------------------------
    const size_t slabBytes = (size_t)g.slices * g.tiles * TILE_F * 4;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.slabs, 0, (int)slabBytes, 0x00020000);
    const uint32_t slabOff = (s * g.tiles + t) * (uint32_t)(TILE_F * 4);
    for (int o = tid * 2; o < TILE_F; o += NT * 2) {
        float s0 = acc[o], s1 = acc[o + 1];
#pragma unroll
        for (int w2 = 1; w2 < W; w2++) { s0 += acc[w2 * TILE_F + o]; s1 += acc[w2 * TILE_F + o + 1]; }
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        u2 pk; pk[0] = __float_as_uint(s0); pk[1] = __float_as_uint(s1);
        __builtin_amdgcn_raw_buffer_store_b64(pk, srs, (uint32_t)o * 4u, slabOff, kSc1);
    }
    if (a.tstamp && tid == 0) atomicMax(&a.tstamp[1], (unsigned long long)wall_clock64());
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's slab stores have left the CU
    if (stamp) { a.tstamp[21] = wall_clock64(); a.tstamp[22] = n; }
    if (a.tstamp && tid == 0) atomicMax(&a.tstamp[27], (unsigned long long)wall_clock64());
    __syncthreads();
    if (tid == 0) {
        const uint32_t ticket = __hip_atomic_fetch_add(&a.counters[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flags[0] = (ticket == g.slices - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (flags[0] == 0u) return;
    if (a.ablate & 2u) { if (tid == 0) __hip_atomic_store(&a.counters[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }

    // last arriver of tile t: every slab of the tile was stored write-through (sc1) and drained before its ticket;
    // read them past L1 (sc1), sum in slice order, un-permute, add the Q4 outliers, write out[].  Each thread owns
    // two adjacent tile slots and keeps up to kRed 8-byte loads in flight (each is a fabric round trip); the four
    // running sums per slot are combined in a fixed order.
    const bool rstamp = a.tstamp && t == 0 && tid == 0;
    if (rstamp) a.tstamp[23] = wall_clock64();
    constexpr int kRed = 32;
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    const uint32_t sliceStride = g.tiles * (uint32_t)(TILE_F * 4);
    for (int o = tid * 2; o < TILE_F; o += NT * 2) {
        const uint32_t vo = t * (uint32_t)(TILE_F * 4) + (uint32_t)o * 4u;
        float sa[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (uint32_t sl = 0; sl < g.slices; sl += kRed) {
            u2v r[kRed];
#pragma unroll
            for (int i = 0; i < kRed; i++)
                r[i] = __builtin_amdgcn_raw_buffer_load_b64(srs, vo, min(sl + i, g.slices - 1u) * sliceStride, kSc1);
#pragma unroll
            for (int i = 0; i < kRed; i++) {
                if (sl + i < g.slices) { sa[i & 3] += __uint_as_float(r[i][0]); sb[i & 3] += __uint_as_float(r[i][1]); }
            }
        }
        float sum2[2] = {(sa[0] + sa[1]) + (sa[2] + sa[3]), (sb[0] + sb[1]) + (sb[2] + sb[3])};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t oo = (uint32_t)o + h;
            const uint32_t lane2 = oo & 63u, sj = oo >> 6, j = sj % E, slot = sj / E;
            const uint32_t c2 = t * (64u * E) + lane2 * E + j;
            float sum = sum2[h];
            if (c2 < g.cols) {
                const uint32_t oi = c2 * NACC + slot;
                a.out[oi] = sum;
            }
        }
    }
    if (rstamp) a.tstamp[24] = wall_clock64();
    if (tid == 0) {
        __hip_atomic_store(&a.counters[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next call
        if (a.tstamp) {
            atomicMax(&a.tstamp[1], (unsigned long long)wall_clock64());
            const uint32_t done = __hip_atomic_fetch_add(&a.counters[g.tiles], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (done == g.tiles - 1u) {                                        // whole kernel finished: fold the stamps
                const unsigned long long t0 = __hip_atomic_load(&a.tstamp[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long t1 = __hip_atomic_load(&a.tstamp[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.tstamp[2] += t1 - t0; a.tstamp[3] += 1;
                __hip_atomic_store(&a.tstamp[0], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&a.tstamp[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&a.counters[g.tiles], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// calcOutliers (bucketMulQ4.metal:13-21): out[o] += sum over the outliers of output o of v[in]*value.  The reference
// fires one atomic per outlier in table order; here one wave owns one output (its outliers are contiguous in the
// by-output index built at registration), lanes stride its segment with coalesced loads, and a fixed xor-butterfly
// adds the 64 partial sums -- no atomics, deterministic.  Launched right after the multiply kernel on the same stream.
__global__ __launch_bounds__(256) void q4_outliers_kernel(const OutlierIndex ol, const float* __restrict__ v,
                                                          float* __restrict__ out, uint32_t outDim) {
    const uint32_t o = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (o >= outDim) return;
    const int lane = threadIdx.x & 63;
    const uint32_t lo = ol.rowPtr[o], hi = ol.rowPtr[o + 1];
    if (lo == hi) return;
    float part = 0.0f;
    for (uint32_t k = lo + lane; k < hi; k += 64) part += v[ol.inIdx[k]] * ol.value[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) out[o] += part;
}

hipError_t launch_q4_outliers(const OutlierIndex& ol, const float* v, float* out, uint32_t outDim, hipStream_t st) {
    hipLaunchKernelGGL(q4_outliers_kernel, dim3((outDim + 3) / 4), dim3(256), 0, st, ol, v, out, outDim);
    return hipGetLastError();
}

// ---- host side ------------------------------------------------------------------------------
template <int FMT, int E, int W>
static hipError_t launch_mul_t(const MulArgs& a, hipStream_t st) {
    uint32_t o1, o2, o3, o4;
    const uint32_t lds = lds_layout<FMT, E, W>(a.g.sliceRows, a.g.rowsPerIn, &o1, &o2, &o3, &o4);
    static uint32_t maxSet = 0;   // per instantiation
    if (lds > maxSet) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&bucket_mul_kernel<FMT, E, W>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
        maxSet = lds;
    }
    if ((FMT == kFp16 ? (a.g.rowsPerIn << a.g.sliceLog2) : a.g.sliceRows * 8u) > (uint32_t)kPre * 64 * W) return hipErrorInvalidValue;   // candidate slots must fit the preload
    const uint32_t grid = a.g.tiles * align_up(a.g.slices, 8);
    hipLaunchKernelGGL((bucket_mul_kernel<FMT, E, W>), dim3(grid), dim3(64 * W), lds, st, a);
    return hipGetLastError();
}

#define EFFORT_GEOMS(X) X(16, 1) X(16, 2) X(8, 1) X(8, 2) X(8, 4) X(4, 1) X(4, 2) X(4, 4) X(4, 8) X(2, 4) X(2, 8)

template <int FMT>
static hipError_t launch_mul_fmt(int W, int E, const MulArgs& a, hipStream_t st) {
#define EFFORT_CASE(w, e) if (W == w && E == e) return launch_mul_t<FMT, e, w>(a, st);
    EFFORT_GEOMS(EFFORT_CASE)
#undef EFFORT_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_bucket_mul(Format fmt, int W, int E, const MulArgs& a, hipStream_t st) {
    return fmt == kFp16 ? launch_mul_fmt<kFp16>(W, E, a, st) : launch_mul_fmt<kQ4>(W, E, a, st);
}

size_t bucket_mul_lds_bytes(Format fmt, int W, int E, uint32_t B, uint32_t rowsPerIn) {
    uint32_t o1, o2, o3, o4;
#define EFFORT_CASE(w, e)                                                                     \
    if (W == w && E == e)                                                                     \
        return fmt == kFp16 ? lds_layout<kFp16, e, w>(B, rowsPerIn, &o1, &o2, &o3, &o4)       \
                            : lds_layout<kQ4, e, w>(B, rowsPerIn, &o1, &o2, &o3, &o4);
    EFFORT_GEOMS(EFFORT_CASE)
#undef EFFORT_CASE
    return 0;
}

uint32_t bucket_mul_max_candidates(int W) { return (uint32_t)kPre * 64u * (uint32_t)W; }

}  // namespace effort
