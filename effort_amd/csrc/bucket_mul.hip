// bucket_mul -- the hot path as ONE kernel launch per bucketMul call, or per GROUP of independent calls.
// Replaces the reference's six launches: findCutoff32, prepareDispatch, roundUp, zeroRange32, bucketMul,
// bucketIntegrate (bucketMul.metal:11-247) and, for Q4, prepareDispatchQ4, bucketMulQ4, calcOutliers
// (bucketMulQ4.metal:13-92).  On MI355X a dependent kernel boundary costs 1.5-5 us and a whole call moves
// ~23 MB (2.9 us of HBM time), so a call is a latency chain first and a bandwidth problem second: one launch, one
// memory round trip per dependent step, no host involvement -- and, for throughput, several independent calls
// (the decode loop's Wq|Wk|Wv, W1|W3) sharing one launch so that their workgroups overlap on the CUs.
//
// Work decomposition (MI355X-first, not the reference's [cols x 32] grid of 32-thread groups):
//   item   = one (call, column tile, row slice); one workgroup of W wave64 works an item.  Item id -> (tile, slice)
//            is remapped so all tiles of a slice sit on one XCD (same L2: they share the stats lines and the 128-B
//            lines that straddle two tiles).
//   slice  = a contiguous block of B input rows (all ranks of them) -> its kept bucket rows are neighbours in
//            HBM within each rank plane.  B = 128 for a lone call, up to 512 when 8 calls share the launch.
//   tile   = 64*E u16 columns; lane l of every wave owns columns l*E .. l*E+E-1 of the tile, so one wave load
//            instruction reads one contiguous 128*E-byte piece of one bucket row.
// Per item:
//   A. stage in LDS, in one memory round trip, the slice of v and the row means of all candidate rows of the slice
//      (16*B, Q4 8*B); sum |v| (the fixed-point bound, D).
//   B. the cutoff is evaluated redundantly by every workgroup (cutoff_device.h: bit-exact findCutoff32), which for a
//      lone call is cheaper than a kernel boundary or a cross-workgroup flag; a persistent launch (many calls) has one
//      cutoff job per call at the head of its item queues instead, and the items wait for its flag (cutoff_job).
//   C. the candidate rows are tested exactly as prepareDispatch does (cutoff < (1e5*mean)*|v|) and the survivors
//      compacted, in ascending bucket-row order, into an LDS list (wave ballots + mbcnt, ONE barrier; no global
//      atomics, deterministic).
//   D. waves take list entries round-robin and stream those rows from HBM: two batches of 16 buffer loads in
//      flight per lane, the row offset in an SGPR (v_readlane of the decoded entry -> buffer soffset, no per-row
//      address VALU).  Each product is added into ONE LDS accumulator tile acc[slot][j][lane] shared by the
//      workgroup (slot = the 4 position bits of the f16 weight; Q4: sub-bucket*8 + the 3 position bits of the
//      nibble) with an INTEGER LDS atomic: products are converted to fixed point on a per-workgroup power-of-two
//      grid.  Measured on MI355X (tools/lab/microbench.hip, elements/clk/CU): ds_add_f32 0.5, read-add-write on
//      private per-wave tiles 6.8, 16-way register select 2.1, ds_add_u32 13.  Integer addition is associative:
//      the result does not depend on the order waves run in.  The layout puts lane l on LDS bank l%32 whatever
//      the slot, so a wave's scatter is bank-conflict free.
//   E. the tile is converted back to f32 into one partial "slab", stored write-through; the workgroup takes a ticket
//      on its tile's arrival counter, and the LAST workgroup of each tile sums the S slabs in slice order and writes
//      out[].  The result does not depend on arrival order: deterministic end to end.  (Q4: before E, phase O adds the
//      outliers of the item's share of the tile's outputs to its slab.)
#include <cstring>
#include <type_traits>

#include "cutoff_device.h"

namespace effort {

#ifndef EFFORT_KBATCH
#define EFFORT_KBATCH 16
#endif
constexpr int kBatch = EFFORT_KBATCH;   // bucket rows per batch; two batches in flight per wave
#ifndef EFFORT_KBATCH4
#define EFFORT_KBATCH4 8
#endif
// 8-byte pieces (E = 4) fly 8 per batch: the same bytes in flight per wave as 16 4-byte pieces, in half the registers (the
// freed ones hold the next item's staged loads across the streaming phase)
template <int E> __host__ __device__ constexpr int batch_rows() { return E >= 4 ? EFFORT_KBATCH4 : kBatch; }
constexpr int kRounds = 16;  // selection rounds of NT candidate slots a workgroup can run: slots per slice <= kRounds * NT

// Ablation builds for profiling (-DEFFORT_ABLATE_NOSCATTER=1 / -DEFFORT_ABLATE_NOLOAD=1); never shipped.
#ifndef EFFORT_ABLATE_NOSCATTER
#define EFFORT_ABLATE_NOSCATTER 0
#endif
#ifndef EFFORT_ABLATE_NOLOAD
#define EFFORT_ABLATE_NOLOAD 0
#endif

// Register budget of the 8-wave (and smaller) instantiations, as waves per SIMD the kernel must fit.  4 = 128 VGPRs
// (two 8-wave workgroups per CU).  Measured: forcing 6 (80 VGPRs, three per CU; hipcc spills 64 B) is 5-15 % SLOWER.
#ifndef EFFORT_MIN_WAVES_PER_EU
#define EFFORT_MIN_WAVES_PER_EU 4
#endif

// Cache policy of the bucket-row stream: NON-TEMPORAL (aux bit 1, `nt`; -DEFFORT_ROW_AUX=0 builds the temporal policy).  A kept row is
// read once per call and a model's weights are tens of times the chip's caches (Mistral-7B at 25 % effort reads 3.5 GB per token against
// 32 MB of L2 and 256 MB of Infinity Cache): marked nt, the stream does not push the row means, the partial tiles and the other
// workgroups' lines out on its way through.  Round 6, second session, A/B builds on rows that sit on whole 128-byte lines
// (profiles/r06_ab_row_stream_nt.txt): one 32-call launch 160 -> 150 us, four in flight on disjoint matrices 133 -> 127, a lone call
// 18.1 -> 17.75, 50 % effort 286 -> 264, 4096 -> 14336 lone 18.8 -> 18.3; sc0 / sc1 beside nt change nothing, sc1 alone nothing; rows
// 1376 bytes apart (not line-aligned) 169 -> 167.  (Round 2 had measured nt WORSE -- 6.8 vs 6.6 us per call at 16 per launch -- on those
// unaligned rows with that round's kernel: neighbouring tiles' pieces share the lines a row straddles.)  What nt gives up is reuse
// BETWEEN launches: four launches in flight over the SAME 32 matrices (a batch of inputs on one set of weights, in lockstep within the
// Infinity Cache's 256 MB) 117 -> 130 us per launch; and 32 calls on 4096 x 4096 matrices re-read every launch (270 MB, the Infinity
// Cache's size) 62 -> 64.  Such a caller asks for the ordinary policy at RUN TIME (effort_set_row_reuse -> GroupKArgs::split bit 3): the two policies
// are two copies of the whole streaming loop, chosen once per item (mul_item: `stream`).  The first form of the switch -- a uniform branch between two
// batches of eight loads inside ONE loop -- cost a lone call 18.1 -> 19.8 us and a 32-call launch 160 -> 165: at the join hipcc's wait-count pass
// assumed the worse of the two paths and every `s_waitcnt vmcnt(15..8)` of the accumulate became `vmcnt(7..0)` -- a batch waited for the batch
// issued AFTER it, i.e. no software pipeline at all.
#ifndef EFFORT_ROW_AUX
#define EFFORT_ROW_AUX 2
#endif
constexpr int kRowAux = EFFORT_ROW_AUX;
// (Q4: the outlier entries are read once per call too, but nt on THEIR loads measured slower -- 16 calls per launch 58.8 -> 63.7 us, 32:
//  100.4 -> 106.3 -- and stays off: -DEFFORT_OL_NT=1 builds it.)
#ifndef EFFORT_OL_NT
#define EFFORT_OL_NT 0
#endif
#if EFFORT_OL_NT
#define EFFORT_OL_LOAD(p) __builtin_nontemporal_load(p)
#else
#define EFFORT_OL_LOAD(p) (*(p))
#endif
// PERSIST (template parameter of the kernel and of mul_item) = false: the LEAN instantiation for PLAIN grids -- one item per
// workgroup, no item queues, no cutoff jobs, no staging of a next item, no stamps or ablation switches.  It is the same kernel
// with those compiled out: 4300 instead of 8000 instructions and 116 instead of 128 VGPRs.  A plain grid's workgroup runs its
// path once and every instruction on it is four cycles of the call's dependent chain (a wave issues one instruction per four
// cycles; A/B with EFFORT_PAD_TEST: 1000 executed s_nop = +1.65 us per launch, 24 KB of code nobody executes = nothing), so the
// polls, flag tests and queue handling a plain grid does not need were a microsecond of every lone call: 22.8 -> 21.5 us, the
// decode loop 267 -> 290 tokens/s (A/B on one box).  The generic instantiation serves persistent launches and the debug modes.
#define GA_PERSISTENT(ga) (PERSIST ? (ga).persistent : 0u)
#define GA_CUTJOBS(ga) (PERSIST ? (ga).cutJobs : 0u)
// Two libraries are built from this file (csrc/Makefile).  The SHIPPED one, libeffort_hip.so, is the product alone: no device-clock
// stamps, no per-item trace, no ablation switches in ANY instantiation -- the macros below are constants and the code behind them is
// gone (tests/test_abi.py disassembles the library: no s_memrealtime in bucket_mul_kernel).  libeffort_hip_lab.so (-DEFFORT_LAB) is
// the lab bench the tools load (EFFORT_HIP_LIB=lab): the generic instantiation then carries the stamps / trace / ablation paths, and
// the A/B macros (EFFORT_PAD_TEST, EFFORT_CUT_FINE, EFFORT_ABLATE_*, EFFORT_NO_TOUCH, ...) are accepted.  Round 6 A/B of the two
// builds on the throughput launches: profiles/r06_ab_product_only.txt.
#ifndef EFFORT_LAB
#define EFFORT_PRODUCT_ONLY 1
#if defined(EFFORT_PAD_TEST) || defined(EFFORT_CUT_FINE) || defined(EFFORT_NO_TOUCH) || defined(EFFORT_NO_STAMPS) || EFFORT_ABLATE_NOSCATTER || EFFORT_ABLATE_NOLOAD
#error "lab switches need -DEFFORT_LAB (they belong in libeffort_hip_lab.so / build/variants, never in the shipped library)"
#endif
#endif
#ifdef EFFORT_PRODUCT_ONLY
constexpr bool kProductOnly = true;
#else
constexpr bool kProductOnly = false;
#endif
#ifdef EFFORT_PRODUCT_ONLY
#define GA_TSTAMP(ga) ((unsigned long long*)nullptr)
#define GA_ABLATE(ga) 0u
#define GA_TRACE(ga) 0u
#elif defined(EFFORT_LEAN_STAMPS)          // lab: the LEAN instantiations write the fine stamps too (with EFFORT_CUT_FINE; tools/lab/cutfine.py) -- timing only
#define GA_TSTAMP(ga) ((ga).tstamp)
#define GA_ABLATE(ga) (PERSIST ? (ga).ablate : 0u)
#define GA_TRACE(ga) 0u
#else
#define GA_TSTAMP(ga) (PERSIST ? (ga).tstamp : nullptr)
#define GA_ABLATE(ga) (PERSIST ? (ga).ablate : 0u)
#define GA_TRACE(ga) (PERSIST ? (ga).trace : 0u)
#endif
constexpr uint32_t kMaxLdsBytes = 160u * 1024u - 1024u;     // dynamic LDS a launch may ask for: a gfx950 CU's 160 KB less the kernel's static words (rounded up generously)
constexpr int kSc1 = 16;     // buffer aux bit: sc1 = write-through store / L1-bypassing load (cross-XCD visible)

template <int FMT> struct Fmt;
// kAcc: outputs per u16 column (16 positions / 4 sub-buckets x 8 positions); kSlots: LDS accumulators per u16 column.
// Q4 keeps TWO accumulators per output, one for each sign nibble (slot = sub-bucket*16 + the whole 4-bit nibble): every
// nibble then adds the same +d and the address comes straight out of the nibble -- 3 instructions per nibble instead of
// 6 (no sign select) -- and the hand-off subtracts the planes.
// (A byte-indexed variant -- one atomic per BYTE into 512 slots per column, folded afterwards -- was built and measured slower in
//  round 4: branch `chain-launch`)
template <> struct Fmt<kFp16> { static constexpr int kAcc = 16, kSlots = 16; };
template <> struct Fmt<kQ4> { static constexpr int kAcc = 32, kSlots = 64; };

template <int FMT> struct MeanT;                       // what the staged row means are kept as in LDS
template <> struct MeanT<kFp16> { typedef uint16_t type; };    // f16 bits (stats lane .w)
template <> struct MeanT<kQ4> { typedef float type; };        // f32 (stats lane .y)

// One lane's piece of a bucket row: E u16 words.
template <int E> struct Piece;
template <> struct Piece<1> {
    uint32_t w;
    template <int AUX> __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { w = __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, AUX); }
    __device__ __forceinline__ uint32_t word(int) const { return w & 0xFFFFu; }
    __device__ __forceinline__ uint32_t dword(int) const { return w; }
};
template <> struct Piece<2> {
    uint32_t w;
    template <int AUX> __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { w = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX); }
    __device__ __forceinline__ uint32_t word(int j) const { return (w >> (16 * j)) & 0xFFFFu; }
    __device__ __forceinline__ uint32_t dword(int) const { return w; }
};
template <> struct Piece<4> {
    uint32_t w[2];
    template <int AUX> __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
        auto t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX); w[0] = t[0]; w[1] = t[1];
    }
    __device__ __forceinline__ uint32_t word(int j) const { return (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu; }
    __device__ __forceinline__ uint32_t dword(int j) const { return w[j >> 1]; }
};
__host__ __device__ inline uint32_t align_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

// Q4 outliers: an item adds the outliers of 1/slices of its tile's outputs to its slab (phase O below).
constexpr int kOlBatch = 16;                               // outlier entries in flight per lane
constexpr uint32_t kOlLdsFloats = 16384;                   // v is staged whole in LDS up to this inDim (64 KB), else gathered from memory
template <int E> __host__ __device__ inline uint32_t ol_outputs_per_item(const MulGeom& g) { return align_up((32u * E * 64u + g.slices - 1u) / g.slices, 64u); }   // whole interleave blocks
constexpr uint32_t kOlEarlyFloats = 4096;                  // ... up to this inDim in a region of its OWN (LdsPlan::offO), filled at the top of the item: the lean
                                                           // kernels then work the outliers INSIDE the streaming loop (mul_item, D); above it the copy overlays the
                                                           // staging regions once the rows are in
constexpr int kOlMerged = 8;                               // outlier steps per half of a streaming-loop trip (merged mode)
constexpr uint32_t kOlSumFloats = 1024;                    // partial sums of a thin share split among the waves: [parts][blocks * 64], parts * blocks <= waves <= 16
template <int E> __host__ __device__ inline uint32_t ol_scratch_bytes(const MulGeom& g) {
    const uint32_t per = ol_outputs_per_item<E>(g);
    return (per > kOlSumFloats ? per : kOlSumFloats) * 4u + (g.inDim <= kOlLdsFloats ? g.inDim * 4u : 0u);           // sums | v (the overlay copy: generic kernels, inDim > 4096)
}

// LDS carve (bytes), one plan for the whole launch (the largest of its geometries, so that a persistent workgroup can stage
// its NEXT item -- possibly of another geometry -- while the current one still uses its regions):
//   means[slots] one dword per candidate slot | vblk[2][B] f32, alternating between consecutive items |
//   { acc[tileFloats] i32 | list[slots] u16 }  (the cutoff's lookup table and the tile reduction's partial sums borrow this
//   region: the table is dead before the tile is zeroed and the list written, the tile is dead when it is reduced) | misc 2 KB.
// means / vblk are filled by LDS-direct buffer loads (stage_issue), whose destination (M0) is kept below 64 KB.  Q4: after
// the streaming phase means | vblk are dead and hold the outlier phase's scratch (sums | whole v); Q4 items are not
// pipelined and use vblk[0] only.
// `merge`: the plan of an instantiation whose Q4 items work their outliers inside the streaming loop (kOlMerge: the 8-wave kernels of the
// shipped library, lean and persistent) and read offO, the region of their own for the whole input vector; the lab library's generic
// kernels (stamps: no registers to spare) and the other workgroup sizes do not carry it.
template <int FMT, int E, int W>
__host__ __device__ inline LdsPlan plan_lds(const MulGeom* geoms, int nGeoms, bool merge) {
    uint32_t slots = 0, vrows = 0, ol = 0, vEarly = 0;
    for (int i = 0; i < nGeoms; i++) {
        const MulGeom& g = geoms[i];
        if (!g.slices) continue;                            // unused entry
        slots = slots > g.slots ? slots : g.slots;
        const uint32_t vr = align_up(g.sliceRows, 64);      // the loads land a whole wave (64 dwords) at a time
        vrows = vrows > vr ? vrows : vr;
        if (FMT != kFp16) {
            const uint32_t x = ol_scratch_bytes<E>(g); ol = ol > x ? ol : x;
            if (merge && W == 8 && g.inDim <= kOlEarlyFloats) vEarly = vEarly > g.inDim * 4u ? vEarly : g.inDim * 4u;
        }
    }
    LdsPlan p;
    uint32_t o = 0;
    p.offM = o; o += align_up(slots, 64) * 4;
    p.offV[0] = o; o += vrows * 4;
    p.offV[1] = FMT == kFp16 ? o : p.offV[0];
    if (FMT == kFp16) o += vrows * 4;
    if (FMT != kFp16 && o < ol) o = align_up(ol, 16);
    p.offA = o; o += (uint32_t)Fmt<FMT>::kSlots * E * 64 * 4;
    p.offL = o; o += align_up(slots * 2, 16);
    const uint32_t tbl = cutoff_table_bytes(64 * W);        // (>= the tile reduction's [G][tileFloats] partial sums)
    if (o < p.offA + tbl) o = p.offA + tbl;
    p.offC = o; o += 2048;           // [0..kCutoffLdsBytes) cutoff scratch, [1280] flags, [1344..1407] wave bounds
    p.offO = vEarly ? o : 0u; o += vEarly;       // Q4: the whole input vector, for the outlier phase (inDim <= kOlEarlyFloats)
    static_assert(kCutoffLdsBytes <= 1280, "the cutoff scratch must end below the flags of the misc region");
    p.total = o;
    return p;
}
// Where a work item sits: which call of the group, which (tile, slice) of it.  All workgroup-uniform.
struct ItemRef { uint32_t ci, t, s; };

// `item` numbers the items the way a plain grid would number its blocks (item % 8 = the XCD it should run on).
__device__ __forceinline__ bool locate_item(const GroupKArgs& ga, const uint32_t item, ItemRef& r) {
    // which call of the group this item belongs to (item ranges are multiples of 8, so item%8 is still the XCD)
    uint32_t ci = 0;
    for (uint32_t i = 0; i + 1 < ga.count; i++) if ((item >> 3) >= (uint32_t)ga.wgEnd8[i]) ci = i + 1;
    ci = __builtin_amdgcn_readfirstlane(ci);
    const MulGeom& g = ga.geom[ga.call[ci].geom];
    // XCD-aware id -> (tile, slice): all tiles of a slice run on one XCD (speed only).
    const uint32_t b = item - (ci ? (uint32_t)ga.wgEnd8[ci - 1] * 8u : 0u), xcd = b & 7u, k = b >> 3;
    // (the slice <-> XCD assignment rotates with the call: calls sharing an input vector -- Wq|Wk|Wv, or a batch on one v --
    //  have the same heavy and light slices, and an XCD that got the same slice of every call would finish 15 % late)
    r.ci = ci;
    if ((g.slices & 7u) == 0u) {               // slices dealt to the XCDs in rounds of 8: every block of the call's range is an item
        r.s = (k / g.tiles) * 8u + ((xcd + ci) & 7u); r.t = k % g.tiles;
        return true;
    }
    // A slice count that is not a multiple of 8 (round 6: Q4's one-round launches, 5-7 tall slices): dealing slices in rounds of 8 would leave
    // three blocks of every eight as padding -- a 16-call launch 768 blocks for 480 items.  The call's tiles x slices items are numbered
    // slice-major and every XCD takes a CONTIGUOUS eighth of them (its items then belong to one or two slices: the same locality), the range
    // padded to a multiple of 8 at its end only: 512 blocks for those 480 items.
    const uint32_t per = (uint32_t)ga.wgEnd8[ci] - (ci ? (uint32_t)ga.wgEnd8[ci - 1] : 0u);      // items per XCD of this call
    const uint32_t idx = ((xcd + ci) & 7u) * per + k;
    r.s = idx / g.tiles; r.t = idx - r.s * g.tiles;
    return r.s < g.slices;
}

// Phase A of an item, issued: the row means of its candidate slots and its slice of v travel from memory straight into
// LDS (buffer_load ... lds: no registers, nothing to wait for until the item needs them).  A persistent workgroup ISSUES
// these loads for its NEXT item right before it streams the current one, so the round trip (2-3 us, during which the
// workgroup would ask nothing of HBM) hides under ~50 us of streaming.
// Candidate slot c of the slice, in ascending bucket-row order:
//   FP16 -> rank = c >> lg, jl = c & (2^lg - 1), 2^lg >= B   (rank-major rows, convert.metal:83-100; v index = row % inDim)
//   Q4   -> jl = c >> 3,  rank = c & 7                        (input-major rows, bucketMulQ4.metal:46)
// Thread tid owns the slots r*NT + tid: lane l of wave w lands its dword at means[r*NT + w*64 + l] (the destination of an
// LDS-direct load is a wave-uniform base + 4*lane).  FP16: the dword holding stats lanes .z|.w (the mean the keep test
// reads is .w, the high half); Q4: stats lane .y as f32.  Slots past the slice load the slice's first row; the keep test
// knows which slots exist.  v: thread tid lands v[j0 + tid (+ NT)] at vblk[tid (+ NT)] of buffer `par`.
// One LDS-direct load: every lane's dword at byte offset `voff` of the buffer lands at LDS byte address ldsAddr + 4*lane
// (ldsAddr wave-uniform, below 64 KB).  Issued from inline asm so that hipcc does not know about it: with the builtin, every
// later wait on an ordinary load in the same loop degrades to vmcnt(0).  The price: hipcc's counted waits see these loads
// as its own youngest ones, so the first wait after an issue drains the wave's batches in flight once.  Completion is
// awaited explicitly (s_waitcnt vmcnt(0)) where the data is read.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_dma_dword(const u32x4 rsrc, const uint32_t voff, const uint32_t ldsAddr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(ldsAddr), "s"(rsrc) : "memory");
}
__device__ __forceinline__ u32x4 make_rsrc(const void* base, uint32_t bytes) {
    const uint64_t p = (uint64_t)(size_t)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)p); r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32) & 0xFFFFu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes); r[3] = 0x00020000u;
    return r;
}

template <int FMT, int W, bool COMPACT>
__device__ __forceinline__ void stage_issue(const GroupKArgs& ga, const ItemRef& r, const int tid, char* smem, const LdsPlan& lp, const uint32_t par) {
    constexpr int NT = 64 * W;
    using lds_v = __attribute__((address_space(3))) void;
    const CallDesc& a = ga.call[__builtin_amdgcn_readfirstlane(r.ci)];
    const MulGeom& g = ga.geom[a.geom];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t B = g.sliceRows, j0 = __builtin_amdgcn_readfirstlane(r.s) * B, nb = min(B, g.inDim - j0), lg = g.sliceLog2;
    const uint32_t e = a.expNo ? a.expNo[0] : 0u;
    const uint32_t rowBase = e * g.expertRows;
    const uint32_t rowsPerIn = g.rowsPerIn, inDim = g.inDim, mask = (1u << lg) - 1u;
    const uint32_t nSlots = FMT == kFp16 ? (rowsPerIn << lg) : (nb << 3);
    constexpr bool compact = COMPACT && FMT == kFp16;              // a.stats = the row means alone, u16 per bucket row (persistent launches)
    const u32x4 rs = make_rsrc(a.stats, (uint32_t)((size_t)g.numExperts * g.expertRows * (compact ? 2u : 8u)));
    const u32x4 rv = make_rsrc(a.v, inDim * 4u);
    const uint32_t ldsM = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_v*)(smem + lp.offM));
    const uint32_t ldsV = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_v*)(smem + ((par & 1u) ? lp.offV[1] : lp.offV[0])));
    if constexpr (compact) {
        // one dword = the means of candidate slots 2d and 2d+1 (rows j and j+1 of one rank: neighbours in memory; every slice
        // starts on an even row -- the host checks -- so the dword is aligned): half the loads, means[] holds u16 per slot
#pragma unroll
        for (int rr = 0; rr < kRounds / 2; rr++) {
            if (((uint32_t)(rr * NT) + wave * 64u) * 2u >= nSlots) continue;   // uniform per wave
            const uint32_t c = ((uint32_t)(rr * NT) + (uint32_t)tid) * 2u, rank = c >> lg, jl = c & mask;
            const bool ok = rank < rowsPerIn && jl < nb;
            lds_dma_dword(rs, (rowBase + (ok ? rank * inDim + j0 + jl : j0)) * 2u, ldsM + ((uint32_t)(rr * NT) + wave * 64u) * 4u);
        }
    } else
#pragma unroll
    for (int rr = 0; rr < kRounds; rr++) {
        if ((uint32_t)(rr * NT) + wave * 64u >= nSlots) continue;   // uniform per wave: none of its 64 slots exists
        const uint32_t c = rr * NT + tid;
        uint32_t voff;
        if (FMT == kFp16) {
            const uint32_t rank = c >> lg, jl = c & mask;
            const bool ok = rank < rowsPerIn && jl < nb;
            voff = (rowBase + (ok ? rank * inDim + j0 + jl : j0)) * 8u + 4u;
        } else {
            voff = ((rowBase + j0 * 8u + (c < nSlots ? c : 0u)) * 2u + 1u) * 4u;
        }
        lds_dma_dword(rs, voff, ldsM + ((uint32_t)(rr * NT) + wave * 64u) * 4u);
    }
#pragma unroll
    for (int u = 0; u < (FMT == kFp16 ? 1 : 2); u++) {
        if ((uint32_t)(u * NT) + wave * 64u >= nb) continue;    // uniform per wave (reads past inDim return 0; past the slice, a neighbour's input nobody looks at)
        lds_dma_dword(rv, (j0 + (uint32_t)(u * NT + tid)) * 4u, ldsV + ((uint32_t)(u * NT) + wave * 64u) * 4u);
    }
}

// One work item = one (call, tile, slice) of the group: stage, select, stream, hand the partial tile over.
// `staged` (per wave): this wave's share of the item's stage loads was issued while the previous item streamed (into vblk
// buffer `par`).  `prefetch` is polled by every wave near the end of its streaming loop until it returns true: there the
// caller pulls the next item from the queue (wave 0) and issues the wave's share of its stage loads (into buffer par ^ 1).
template <int FMT, int E, int W, bool FUSED, bool COMPACT, bool PERSIST, typename Prefetch>
__device__ __forceinline__ void mul_item(const GroupKArgs& ga, const uint32_t item, const ItemRef& ref, char* smem, const LdsPlan& lp, uint32_t& cachedCall,
                                         float& cachedCutoff, const uint32_t par, const bool staged, const bool firstItem, Prefetch prefetch) {
    constexpr int NACC = Fmt<FMT>::kAcc;
    constexpr int TILE_F = NACC * E * 64;                    // outputs of a tile (slab / out[] granularity)
    constexpr int TILE_L = Fmt<FMT>::kSlots * E * 64;        // LDS accumulators of a tile
    constexpr int NT = 64 * W;
    constexpr int VPT = 4096 / NT;
    constexpr int KB = batch_rows<E>();                      // bucket rows per batch; two batches in flight per wave

    // (uniform, and the compiler must know it: everything derived from these -- descriptors, buffer resources -- stays scalar)
    const uint32_t ci = __builtin_amdgcn_readfirstlane(ref.ci), s = __builtin_amdgcn_readfirstlane(ref.s), t = __builtin_amdgcn_readfirstlane(ref.t);
    const CallDesc& a = ga.call[ci];
    const MulGeom& g = ga.geom[a.geom];
    float* const a_slabs = ga.slabs + (size_t)a.slabOff * 64u;
    uint32_t* const a_counters = ga.counters + a.tileOff;
    uint32_t* const a_sliceCounts = ga.sliceCounts + a.sliceOff;
    float* const a_cutoff = ga.cutoff + ci;
    (void)item;

    int tid0 = threadIdx.x;
    asm volatile("" : "+v"(tid0));      // opaque per item: keeps the compiler from hoisting every tid-derived value out of the item loop (+50 VGPRs)
    const int tid = tid0, lane = tid & 63;
#if defined(EFFORT_NO_STAMPS) || defined(EFFORT_CUT_FINE)
    const bool wstamp = false;
#else
    const bool wstamp = GA_TSTAMP(ga) && tid == 0;
#endif                        // every workgroup: phase durations summed into tstamp[32..]
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    if (wstamp) ph[0] = wall_clock64();
#ifdef EFFORT_CUT_FINE                                                    // (lab: the cutoff's fine stamps take the item stamps' words)
    const bool stampCut = GA_TSTAMP(ga) && item == 0 && tid == 0, stamp = false;
#define EFFORT_PSTAMP(i) if (stampCut) GA_TSTAMP(ga)[26 + (i)] = clock64();
#else
#define EFFORT_PSTAMP(i)
    const bool stamp = GA_TSTAMP(ga) && item == 0 && tid == 0;            // phase stamps of item 0 (profiling aid)
    const bool stampCut = stamp;
#endif
    if (stamp) GA_TSTAMP(ga)[16] = wall_clock64();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t B = g.sliceRows;
    const uint32_t offC = __builtin_amdgcn_readfirstlane(lp.offC), offL = __builtin_amdgcn_readfirstlane(lp.offL), offM = __builtin_amdgcn_readfirstlane(lp.offM);
    const uint32_t offA = __builtin_amdgcn_readfirstlane(lp.offA);
    int* acc = reinterpret_cast<int*>(smem + offA);                          // ONE fixed-point tile shared by the W waves
    float* vblk = reinterpret_cast<float*>(smem + __builtin_amdgcn_readfirstlane((par & 1u) ? lp.offV[1] : lp.offV[0]));     // (a select, not lp.offV[par & 1]: a dynamically indexed plan lives in scratch)
    uint32_t* flags = reinterpret_cast<uint32_t*>(smem + offC + 1280);       // [0] last arriver, [2..3] cutoff job verdict, [4] list length
    float* wbound = reinterpret_cast<float*>(smem + offC + 1344);            // [16] per-wave sums of |v_j| over the slice
    uint16_t* list = reinterpret_cast<uint16_t*>(smem + offL);
    const uint32_t* m32 = reinterpret_cast<const uint32_t*>(smem + offM);   // one dword per candidate slot (see stage_issue)
    const float* means = reinterpret_cast<const float*>(smem + offM);        // Q4: the row means as f32

    const uint32_t j0 = s * B;
    // Geometry the hand-off (E) needs, worked out NOW and parked in vector registers: the kernel runs at its scalar-register limit, and
    // hipcc, rather than hold these, re-reads the geometry from the kernel-argument segment where they are used -- behind a barrier,
    // a scalar-memory latency on the call's chain each time
    uint32_t vSlabOff = (s * g.tiles + t) * (uint32_t)(TILE_F * 4), vSlabBytes = (uint32_t)min((size_t)0xFFFFFFFFu, (size_t)g.slices * g.tiles * TILE_F * 4);
    uint32_t vSlices = g.slices, vSliceStride = g.tiles * (uint32_t)(TILE_F * 4);
    if constexpr (!PERSIST) asm volatile("" : "+v"(vSlabOff), "+v"(vSlabBytes), "+v"(vSlices), "+v"(vSliceStride));      // (the persistent instantiations have no register to spare)
    const uint32_t nb = min(B, g.inDim - j0);
    const uint32_t e = a.expNo ? a.expNo[0] : 0u;

    // ---- A. everything the selection needs lands in LDS (stage_issue): the row means of the candidate slots, the slice of v
    EFFORT_PSTAMP(1)
    if (!staged) stage_issue<FMT, W, COMPACT>(ga, ref, tid, smem, lp, par);
    EFFORT_PSTAMP(2)
    // plain grids (one item per workgroup: the accumulator region is untouched so far): the cutoff's count table, which borrows
    // that region, is cleared HERE, and the barrier its adds need behind the clearing passes under the staged loads' round trip
    // (block_find_cutoff, PREZERO) instead of on the cutoff's own chain
    constexpr bool kPreZero = !PERSIST;
    if constexpr (kPreZero) cutoff_table_zero<NT>(reinterpret_cast<uint32_t*>(smem + offA), tid);
    // Q4 outliers (phase O below): the share of this item, and -- lean kernels, v small enough for a region of its own -- what the
    // MERGED form needs, asked for now: the whole of v (eight loads per thread, written to LDS once the staged loads are awaited)
    // and the wave's first block's meta word and entry bounds.
    constexpr bool kOlMerge = FMT != kFp16 && W == 8 && (!PERSIST || kProductOnly);      // (round 6: the persistent instantiation of the SHIPPED library too -- with the stamp code gone it has the registers: 124 VGPRs, no scratch)
    const uint32_t olPer = ol_outputs_per_item<E>(g);
    bool olAny = false, olEarly = false;
    uint32_t olParts = 1u, olStride = 0u;                              // partial sums of a thin share: [parts][olStride], olStride = its blocks * 64
    uint32_t olBFirst = 0u, olNB = 0u;
    float olVx[kOlMerge ? 8 : 1]; uint32_t olMeta0 = 0u, olBeg0 = 0u, olEnd0 = 0u;
    auto ol_share = [&]() {                                            // the item's share of its tile's outputs, in blocks of 64
        const uint32_t oBeg = min(t * (uint32_t)TILE_F + s * olPer, g.outDim);
        const uint32_t oEnd = max(oBeg, min(min(t * (uint32_t)TILE_F + (s + 1u) * olPer, (t + 1u) * (uint32_t)TILE_F), g.outDim));
        olBFirst = oBeg >> 6; olNB = ((oEnd + 63u) >> 6) - olBFirst;                        // (oBeg is a multiple of 64)
        // a share with fewer blocks than the workgroup has waves splits every block's steps among `parts` waves
        olParts = (olNB && olNB < (uint32_t)W) ? (uint32_t)W / olNB : 1u;
        olStride = olNB * 64u;                                                             // (parts * blocks <= W: at most 64 * W floats)
    };
    if constexpr (FMT != kFp16) {
        olAny = a.ol.blockPtr != nullptr;                                                  // uniform per call
        if (olAny) {
            if constexpr (kOlMerge) {
                ol_share();
                olEarly = g.inDim <= kOlEarlyFloats && lp.offO != 0u;
                if (olEarly) {
#pragma unroll
                    for (int u = 0; u < 8; u++) olVx[u] = a.v[min((uint32_t)(u * NT + tid), g.inDim - 1u)];
                    if ((uint32_t)wave < olNB * olParts) {
                        const uint32_t bq = (uint32_t)wave / olParts;
                        olMeta0 = a.ol.meta[(size_t)(olBFirst + bq) * 64u + (uint32_t)lane];
                        olBeg0 = a.ol.blockPtr[olBFirst + bq]; olEnd0 = a.ol.blockPtr[olBFirst + bq + 1u];
                    }
                }
            }
        }
    }
    const float rankBound = a.rankBound[e];                       // (asked for here: its round trip runs under the staged loads')
    float vj[VPT]; uint16_t prj[VPT];
    const uint16_t* pr = a.probes + (size_t)e * kProbes;
    const bool fused = (ga.split & 1u) == 0u;                    // uniform
#pragma unroll
    for (int i = 0; i < VPT; i++) { vj[i] = 0.0f; prj[i] = 0; }
    const bool needCut = fused && cachedCall != ci;              // uniform: a persistent workgroup evaluates a call's cutoff once
    // ... or takes it from the call's cutoff job -- except for its FIRST item: the launch has just started, no job can have
    // finished, and evaluating the cutoff here (5 us, its inputs loaded beside the stage loads) beats waiting for one
    const bool viaJob = needCut && GA_CUTJOBS(ga) != 0u && !firstItem;
    // Input prologue (uniform per call): the multiply's input is v itself, or silu(v) * vAux (the FFN gate,
    // runNetwork.swift:181), or rmsNorm(v) * vAux (runNetwork.swift:121-122,173-175) -- evaluated here, per workgroup,
    // instead of in a launch of its own.
    const uint32_t pre = FUSED ? (uint32_t)a.pre : (uint32_t)kPreNone;      // FUSED is a separate instantiation: the plain multiply pays nothing
    // A prologue's operands are ALL asked for here, in one memory round trip beside the stage loads: the first 4096 raw inputs,
    // their partners from vAux (the gate's x3 as f32, the norm weights as f16: kept as raw bits), the probes, and vAux for this
    // thread's element of the slice.  (Asked for where they were used -- vAux after the norm's reduction, the slice's vAux after
    // the staged loads had landed -- they were two more dependent round trips: a fused launch cost 3-4 us more than a plain one.)
    float rawn[VPT]; uint32_t auxc[VPT]; uint32_t auxS = 0u;
    const bool preAny = FUSED && pre != (uint32_t)kPreNone;       // uniform
    const bool needCutEarly = fused && cachedCall != ci;          // (= needCut below)
#pragma unroll
    for (int i = 0; i < VPT; i++) { rawn[i] = 0.0f; auxc[i] = 0u; }
    if (preAny) {
        const bool gate = pre == (uint32_t)kPreSiluGate;
        if (needCutEarly || !gate) {
#pragma unroll
            for (int i = 0; i < VPT; i++) rawn[i] = *(a.v + tid + NT * i);
        }
        if (needCutEarly) {
#pragma unroll
            for (int i = 0; i < VPT; i++) {
                auxc[i] = gate ? __float_as_uint(*(reinterpret_cast<const float*>(a.vAux) + tid + NT * i)) : (uint32_t)reinterpret_cast<const uint16_t*>(a.vAux)[tid + NT * i];
                prj[i] = pr[tid + NT * i];
            }
        }
        const uint32_t js = j0 + min((uint32_t)tid, nb - 1u);
        auxS = gate ? __float_as_uint(*(reinterpret_cast<const float*>(a.vAux) + js)) : (uint32_t)reinterpret_cast<const uint16_t*>(a.vAux)[js];
    }
    const uint32_t lg = g.sliceLog2;
    const uint32_t nSlots = FMT == kFp16 ? (g.rowsPerIn << lg) : (nb << 3);
    float normInv = 1.0f;
    if (FUSED && pre == kPreRmsNorm) {
        // The sum of squares in EXACTLY the order add_rmsnorm_mul_kernel (decode.hip) sums it -- 1024 threads, thread u adding
        // x[u], x[u + 1024], ... in that order, a DPP scan-order sum over each 64 of them (wave_sum_f32), the sixteen wave sums in wave order -- so
        // that the normalised input, hence the cutoff and the row selection, are bit-identical with the unfused path's.  A
        // thread here stands for R = 1024 / NT of that kernel's threads: u = tid + NT * r.
        constexpr int R = 1024 / NT;
        static_assert(R >= 1 && R * NT == 1024 && R * W == 16, "fused rmsNorm: the workgroup must divide add_rmsnorm_mul_kernel's 1024 threads");
        float part[R];
#pragma unroll
        for (int r = 0; r < R; r++) part[r] = 0.0f;
#pragma unroll
        for (int i = 0; i < VPT; i++) part[i % R] += rawn[i] * rawn[i];            // element tid + NT * i = element i / R of virtual thread tid + NT * (i % R)
        for (uint32_t base = 4096u + (uint32_t)tid; base < g.inDim; base += 1024u) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t j = base + (uint32_t)(r * NT);
                if (j < g.inDim) { const float x = *(a.v + j); part[r] += x * x; }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            part[r] = wave_sum_f32(part[r]);                                   // (the glue kernel's block_sum: same DPP order)
            if (lane == 0) wbound[r * W + wave] = part[r];
        }
        __syncthreads();
        float tot = 0.0f;
#pragma unroll
        for (int w2 = 0; w2 < 16; w2++) tot += wbound[w2];
        normInv = 1.0f / sqrtf(tot / (float)g.inDim + 1e-5f);                  // aux.metal:150
        __syncthreads();                                                         // wbound is reused below
    }
    auto xform = [&](float x, uint32_t aux) -> float {             // the input prologue applied to an input and its partner from vAux (raw bits)
        if (!FUSED) return x;
        if (pre == kPreSiluGate) return __uint_as_float(aux) * x / (1.0f + expf(-x));
        if (pre == kPreRmsNorm) return (x * normInv) * half_bits_to_float((uint16_t)aux);
        return x;
    };
    auto load_cut_inputs = [&]() {
        if (preAny) {                                              // (operands in registers since the top of the item)
#pragma unroll
            for (int i = 0; i < VPT; i++) vj[i] = xform(rawn[i], auxc[i]);
        } else {
#pragma unroll
            for (int i = 0; i < VPT; i++) { vj[i] = *(a.v + tid + NT * i); prj[i] = pr[tid + NT * i]; }
        }
    };
    if constexpr (kPreZero) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (the clearing stores; NOT __syncthreads: it would await the loads in flight)
        __builtin_amdgcn_s_barrier();
    }
    if (needCut && !viaJob) load_cut_inputs();
    EFFORT_PSTAMP(3)
    // the call's cutoff job publishes ONE word: the cutoff's bits (a non-negative float) with the sign bit raised.  Thread 0
    // asks for it here, before the staged loads are awaited
    uint32_t* const cutWords = ga.queue + 9 * 16;
    uint32_t cutWord = 0;
    if (viaJob && tid == 0 && !(GA_ABLATE(ga) & 32u)) cutWord = __hip_atomic_load(&cutWords[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the staged loads have landed (each thread waits for its own; it reads back only what its own lane loaded until the
    // next barrier).  The slice's absolute sum bounds every partial sum of this workgroup (see the scale below).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    EFFORT_PSTAMP(4)
    // (plain grids) what the streaming phase (D) starts with -- the bucket buffer's descriptor, this lane's column offset, the row
    // arithmetic -- parked in vector registers like the hand-off's geometry above: re-read from the kernel-argument segment where D
    // begins, they were a scalar-memory round trip between the selection and the first row load
    uint32_t vColOK = 0, voff = 0, vRecords = 0, vBuckLo = 0, vBuckHi = 0, vRowPitch = 0, vInDim = 0, vRow0 = 0;
    auto stream_geom = [&]() {
        const uint32_t col = t * (64u * E) + (uint32_t)lane * E;
        vColOK = col < g.cols ? 1u : 0u;                           // a piece past the last column is skipped whole
        // lanes past the last column re-read the row's LAST piece -- the line their neighbours are fetching anyway.  (They used to
        // re-read column 0: one more 128-byte line per kept row for the ragged last tile, 68 MB of a 32-call launch's 824.)
        voff = (vColOK ? col : (g.cols - 1u) / (uint32_t)E * (uint32_t)E) * 2u;
        vRecords = (uint32_t)min((size_t)0xFFFFFFFFu, (size_t)g.numExperts * g.expertRows * g.rowPitch - (size_t)a.bucketsTrim);
        vBuckLo = (uint32_t)(size_t)a.buckets; vBuckHi = (uint32_t)((size_t)a.buckets >> 32);
        vRowPitch = g.rowPitch; vInDim = g.inDim; vRow0 = e * g.expertRows + (FMT == kFp16 ? j0 : j0 * 8u);
    };
    if constexpr (!PERSIST) {
        stream_geom();
        asm volatile("" : "+v"(vColOK), "+v"(voff), "+v"(vRecords), "+v"(vBuckLo), "+v"(vBuckHi), "+v"(vRowPitch), "+v"(vInDim), "+v"(vRow0));
    }
    if constexpr (kOlMerge) {
        if (olEarly) {                                                // (published by the barrier of B, with vblk)
            float* const vf = reinterpret_cast<float*>(smem + __builtin_amdgcn_readfirstlane(lp.offO));
#pragma unroll
            for (int u = 0; u < 8; u++) if ((uint32_t)(u * NT + tid) < g.inDim) vf[u * NT + tid] = olVx[u];
        }
    }
    float bound = 0.0f;
#pragma unroll
    for (int u = 0; u < (FMT == kFp16 ? 1 : 2); u++) {
        const uint32_t jl = tid + u * NT;
        if (jl < nb) {
            float x = vblk[jl];
            if (preAny) { x = xform(x, auxS); vblk[jl] = x; }       // (FP16: u == 0, jl == tid: the element auxS was loaded for)
            bound += fabsf(x);
        }
    }
    bound = wave_sum_f32(bound);                                  // (only the exponent of the slice's bound matters: any order)
    if (lane == 0) wbound[wave] = bound;
    if (tid == 0) { flags[4] = 0u; flags[5] = (uint32_t)(W * KB); }      // the list is empty; its first W batches go to the waves by index (D), the cursor starts behind them
    if (stamp) GA_TSTAMP(ga)[17] = wall_clock64();
    if (wstamp) ph[1] = wall_clock64();

    // ---- B. cutoff (findCutoff32), redundantly per workgroup; its first barrier also publishes vblk / wbound.
    //         Its lookup table borrows the accumulator + list region, which is initialised afterwards. ------
    uint32_t* tbl = reinterpret_cast<uint32_t*>(smem + offA);
    float cutoff;
    bool fromJob = false;
    if (viaJob) {
        // the call's cutoff job (head of the item queues, see bucket_mul_kernel) publishes the value and raises the flag;
        // it never waits on anything, so this wait ends -- and should it not within ~4 ms, the cutoff is evaluated here
        if (tid == 0) {
            for (int spin = 0; spin < ((GA_ABLATE(ga) & 32u) ? 0 : 20000) && !(cutWord >> 31); spin++) {      // (ablate 32: exercise the fallback)
                __builtin_amdgcn_s_sleep(8);
                cutWord = __hip_atomic_load(&cutWords[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            flags[2] = cutWord >> 31;
            flags[3] = cutWord & 0x7FFFFFFFu;
        }
        __syncthreads();                                             // publishes vblk / wbound / the list length, and the verdict
        fromJob = flags[2] != 0u;
        if (fromJob) { cutoff = __uint_as_float(flags[3]); cachedCall = ci; cachedCutoff = cutoff; }
        else load_cut_inputs();
    }
    if (fromJob) {
    } else if (needCut) {
        cutoff = block_find_cutoff<NT, kPreZero>(vj, prj, a.q, smem + offC, tbl, []() {}, stampCut ? GA_TSTAMP(ga) + 8 : nullptr);
        cachedCall = ci; cachedCutoff = cutoff;
        // (BucketMul.cutoff, bucketMul.swift:22, is stored with the slab, in E: a store here sits in front of the `s_waitcnt vmcnt(0)` that
        //  rankBound's consumer needs, and the call's first workgroup waited a microsecond for its acknowledgement)
    } else if (fused) {
        cutoff = cachedCutoff;
        __syncthreads();                                             // publishes vblk / wbound / the list length
    } else {
        // split mode: the standalone cutoff kernel ran first on this stream (cheaper in aggregate when several
        // calls overlap: one workgroup evaluates it instead of all of them)
        cutoff = a_cutoff[0];
        __syncthreads();                                             // publishes vblk / wbound / the list length
    }
    // What the selection's first block needs from LDS -- the means of its four rounds, |v| of this thread's input row, the waves'
    // bounds -- asked for HERE, together, beside the cutoff's own read-back: one LDS round trip (~250 cycles on a lone call's
    // chain) where the scheduler, left alone, made four dependent ones (a wait inside the branch around |v|, one after the first mean)
    const uint32_t slotCap = __builtin_amdgcn_readfirstlane((lp.offV[0] - lp.offM) / 4u) - 1u;      // last dword of the means region
    const uint16_t* m16 = reinterpret_cast<const uint16_t*>(m32);
    auto load_block = [&](int blk, uint32_t (&mraw)[4], float (&vq)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t c = min((uint32_t)((blk * 4 + u) * NT + tid), slotCap);
            if constexpr (COMPACT && FMT == kFp16) { mraw[u] = (uint32_t)m16[c] << 16; vq[u] = 0.0f; }      // compact means: u16 per slot, kept in the high half
            else { mraw[u] = m32[c]; vq[u] = FMT != kFp16 ? vblk[min(c >> 3, nb - 1u)] : 0.0f; }
        }
    };
    uint32_t mraw0[4]; float vq0[4];
    load_block(0, mraw0, vq0);
    const uint32_t jlSel = (uint32_t)tid & ((1u << lg) - 1u);
    float axRaw = 0.0f;
    if (FMT == kFp16) axRaw = vblk[min(jlSel, nb - 1u)];
    float wb[W];
#pragma unroll
    for (int w2 = 0; w2 < W; w2++) wb[w2] = wbound[w2];
    __builtin_amdgcn_sched_barrier(0);
    for (int i = tid; i < TILE_L; i += NT) acc[i] = 0;               // the table is dead: zero the tile (barrier in C)
    // Fixed-point scale of this workgroup's tile.  Every product is |v_j| * |w| with |w| <= (max |w| of its rank), so
    // every partial sum is bounded by L = (sum over the slice of |v_j|) * (sum over ranks of that rank's max |w|)
    // (Q4: of that rank's max row mean); rankBound comes from registration.  With 2^k * L < 2^30 no accumulator can
    // leave int32 (intermediate wrap-around would be harmless anyway: integer addition is modular), and 2^k being a
    // power of two, scaling and un-scaling are exact.
    float L = 0.0f;
#pragma unroll
    for (int w2 = 0; w2 < W; w2++) L += wb[w2];
    L *= rankBound;
    int kexp = 30 - (int)((__float_as_uint(L) >> 23) & 0xFFu) + 126;     // L < 2^(e-126)  =>  2^kexp * L < 2^30
    kexp = L > 0.0f ? max(-100, min(100, kexp)) : 0;
    const float scale = __uint_as_float((uint32_t)(127 + kexp) << 23), unscale = __uint_as_float((uint32_t)(127 - kexp) << 23);
    if (stamp) GA_TSTAMP(ga)[18] = wall_clock64();
    if (wstamp) ph[2] = wall_clock64();

#if defined(EFFORT_PAD_TEST) && EFFORT_PAD_TEST == 1            // A/B: 24 KB of code nobody executes, in the middle of the kernel
    if (ga.numCU == 0xdeadu) asm volatile(".rept 6000\n s_nop 0\n .endr" ::: "memory");
#elif defined(EFFORT_PAD_TEST) && EFFORT_PAD_TEST == 2          // A/B: 4 KB of code everybody executes (1000 cycles by itself)
    asm volatile(".rept 1000\n s_nop 0\n .endr" ::: "memory");
#endif
    // ---- C. keep test (bucketMul.metal:69 / bucketMulQ4.metal:47) + compaction -------------------
    // Every thread tests its slot of each round against the means it holds in registers (FP16: all its slots belong to ONE
    // input row, whose |v| it reads once), ballots, and a wave sums its survivors; ONE LDS atomic per wave reserves that
    // many list entries, and the survivors are written at (wave base + survivors of the wave's earlier rounds + earlier
    // lanes of the ballot).  The list order therefore depends on the order in which the waves arrive -- it does not
    // matter: every listed row is added with integer arithmetic (see D), and the reference's own list is appended with an
    // atomic counter in no particular order (bucketMul.metal:71).
    const float ax = (FMT == kFp16 && jlSel < nb) ? fabsf(axRaw) : 0.0f;
    // (the rounds go in blocks of four, a block's reads issued together; blocks wholly past the slice's slots are skipped with a
    //  uniform branch: a lone call's slices have 2048 slots -- one block of four rounds with 512 threads -- a 32-call launch's
    //  8192)
    const bool sel = !(GA_ABLATE(ga) & 8u);
    uint32_t keepMask = 0;                                  // bit r: this thread's slot of round r is kept
    uint32_t before[kRounds];                               // (wave-uniform) survivors of the wave's earlier rounds
    uint32_t wtot = 0;
#pragma unroll
    for (int blk = 0; blk < kRounds / 4; blk++) {
        if (blk > 0 && (uint32_t)(blk * 4 * NT) >= nSlots) {           // uniform
#pragma unroll
            for (int u = 0; u < 4; u++) before[blk * 4 + u] = wtot;
            continue;
        }
        uint32_t mraw[4]; float vq[4];
        if (blk == 0) {
#pragma unroll
            for (int u = 0; u < 4; u++) { mraw[u] = mraw0[u]; vq[u] = vq0[u]; }
        } else {
            load_block(blk, mraw, vq);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int rr = blk * 4 + u;
            before[rr] = wtot;
            const uint32_t c = rr * NT + tid;
            bool k;
            if (FMT == kFp16) k = (cutoff < (kCutoffScale * half_bits_to_float((uint16_t)(mraw[u] >> 16))) * ax) & ((c >> lg) < g.rowsPerIn) & sel;
            else k = (cutoff < (kCutoffScale * __uint_as_float(mraw[u])) * fabsf(vq[u])) & (c < nSlots) & sel;
            keepMask |= k ? (1u << rr) : 0u;
            wtot += (uint32_t)__popcll(__ballot(k));
        }
    }
    uint32_t wbase = 0;
    if (lane == 0) wbase = atomicAdd(&flags[4], wtot);
    wbase = __builtin_amdgcn_readfirstlane(wbase);
#pragma unroll
    for (int rr = 0; rr < kRounds; rr++) {
        if ((uint32_t)(rr * NT) >= nSlots) break;           // uniform
        const uint32_t c = rr * NT + tid;
        const bool k = (keepMask >> rr) & 1u;
        const unsigned long long m = __ballot(k);
        const uint32_t pos = wbase + before[rr] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (k) list[pos] = (uint16_t)(FMT == kFp16 ? (((c >> lg) << 12) | (c & ((1u << lg) - 1u))) : c);
    }
    __syncthreads();
    // (the list's length and this wave's FIRST batch of entries -- a static share, batch `wave` -- in one LDS round trip; entries past
    //  the length are garbage and replaced below)
    const uint32_t n = flags[4];
    const uint32_t code0 = list[(uint32_t)(wave * KB) + ((uint32_t)lane & (uint32_t)(KB - 1))];
    if (t == 0 && tid == 0) a_sliceCounts[s] = n;                  // dispatch.size = sum over slices (test hook)
    if (stamp) GA_TSTAMP(ga)[19] = wall_clock64();
    if (wstamp) ph[3] = wall_clock64();

    // ---- D. stream the kept rows, scatter-accumulate into the LDS tile ----------------
    if constexpr (PERSIST) stream_geom();                          // (the persistent instantiations have no register to park anything in)
    const bool colOK = vColOK != 0u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<uint16_t*>((size_t)(uint32_t)__builtin_amdgcn_readfirstlane(vBuckLo) | ((size_t)(uint32_t)__builtin_amdgcn_readfirstlane(vBuckHi) << 32)), 0,     // (uint32_t: readfirstlane returns int, and a sign-extended low half is another pointer)
        (int)__builtin_amdgcn_readfirstlane(vRecords), 0x00020000);
    const uint32_t nU = (GA_ABLATE(ga) & 4u) ? 0u : __builtin_amdgcn_readfirstlane(n);   // n is workgroup-uniform; keep it in an SGPR
    // The waves of a workgroup do NOT run at one speed (the older wave wins the arbitration for issue slots and the memory
    // pipeline: measured, wave 0 gets through a static share of the rows 2-3x sooner than the last wave and then idles at
    // the barrier), so the list is handed out dynamically: a wave takes the next KB entries with one LDS atomic.
    uint32_t* const cursor = flags + 5;
    auto grab = [&]() -> uint32_t {
        uint32_t bq = 0;
        if (lane == 0) bq = atomicAdd(cursor, (uint32_t)KB);
        return __builtin_amdgcn_readfirstlane(bq);
    };

    // Software pipeline: the loads of batch k+1 are issued before batch k is accumulated, so a wave keeps
    // up to 2*KB row pieces in flight.  Loads are never predicated: a batch that runs past the wave's
    // last row re-reads that last row (an L1/L2 hit) -- a branch around a load makes hipcc drain
    // vmcnt(0) -- and the accumulate step skips the surplus with a wave-uniform test.
    auto decode_code = [&](uint32_t code, uint32_t& boff, float& dv) {
        {
            uint32_t rowIdx;
            if (FMT == kFp16) { const uint32_t rank = code >> 12, jl = code & 4095u; rowIdx = rank * vInDim + jl; dv = vblk[jl] * scale; }
            else {   // entry value = v*mean (bucketMulQ4.metal:52), the one magnitude of the row; carried in fixed point
                rowIdx = code;
                dv = __int_as_float(__float2int_rn((vblk[code >> 3] * means[code]) * scale));
            }
            boff = (vRow0 + rowIdx) * vRowPitch;                    // byte offset of the bucket row (< 4 GiB, checked at registration); vRow0 = e * expertRows + the slice's first row
        }
    };
    auto decode = [&](uint32_t i0, uint32_t& boff, float& dv) {
        // lane u decodes list entry i0+u (clamped); the row loops read it back with v_readlane into SGPRs
        boff = 0; dv = 0.0f;
        if (lane < KB && nU) decode_code(list[min(i0 + (uint32_t)lane, nU - 1u)], boff, dv);
    };
    auto decode_first = [&](uint32_t i0, uint32_t& boff, float& dv) {      // the entries were asked for with the list's length; a lane past it repeats the batch's first row
        boff = 0; dv = 0.0f;
        if (lane < KB && nU) decode_code(i0 + (uint32_t)lane < nU ? code0 : (uint32_t)__builtin_amdgcn_readfirstlane((int)code0), boff, dv);
    };
    // (`aux`: the cache policy of the batch's loads, an immediate of the instruction -- std::integral_constant: the two policies are two copies of
    //  the whole streaming loop below, chosen ONCE per item; a branch per batch breaks hipcc's counted waits, see EFFORT_ROW_AUX above)
    auto issue = [&](auto aux, Piece<E> (&piece)[KB], uint32_t boff) {
#pragma unroll
        for (int u = 0; u < KB; u++) piece[u].template load<decltype(aux)::value>(rsrc, voff, EFFORT_ABLATE_NOLOAD ? 0u : __builtin_amdgcn_readlane(boff, u));
    };
    // One row piece -> E (Q4: 4E) integer LDS atomics on the workgroup's tile.  Why fixed point: a float
    // read-add-write needs a PRIVATE tile per wave (W x the LDS, so two workgroups per CU at best) and two LDS
    // instructions per element; ds_add_f32 runs at 0.5 elements/clk/CU; ds_add_u32 runs at 13 (tools/lab/microbench.hip),
    // lets all waves share one tile, and -- integer addition being associative -- makes the sum independent of the
    // order in which waves and workgroups happen to run.  Each product is rounded once to the grid 2^-k (k above: at
    // least 30 bits below the bound L), finer than the f32 rounding of a running sum; the reference's own summation
    // order is unspecified (atomics / simd_sum), see DESIGN.md.
    // The LDS byte address is built with v_bfe_u32 + v_lshl_add_u32 (hipcc otherwise spends three VALU ops on it).
    using lds_i = __attribute__((address_space(3))) int;
    const uint32_t accB = (uint32_t)(size_t)(lds_i*)(acc + lane);
    constexpr int kShift = (E == 1 ? 8 : E == 2 ? 9 : 10);      // log2(E * 64 lanes * 4 bytes)
    // (Measured and dropped: E = 6 -- 12-byte pieces, 384-column tiles, so that the 688 columns of 11008 outputs make TWO tiles and a
    //  32-call launch exactly one item per persistent workgroup -- 169 against 158 us per launch; E = 8: 204-218 against 183.)
    auto row = [&](const Piece<E>& pc, float dd) {
        if (FMT == kFp16) {
            // bucketMul.metal:100-106: v = d.x*float(w) with the position bits left in w; acc[pos] += v
#pragma unroll
            for (int j = 0; j < E; j++) {
                const uint32_t dw = pc.dword(j);                       // the dword holding element j (bits 16*(j&1)..)
                uint32_t a2; int qv;
                asm("v_bfe_u32 %0, %1, %2, 4\n\tv_lshl_add_u32 %0, %0, %3, %4" : "=&v"(a2) : "v"(dw), "n"(16 * (j & 1)), "n"(kShift), "v"(accB));
                // product = d * float(w): one v_fma_mix_f32 reads the f16 half in place (exact, as the f32 multiply of
                // the converted half is); then floor(x + 0.5) to the fixed-point grid
                if (j & 1) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]\n\tv_cvt_rpi_i32_f32 %0, %0" : "=v"(qv) : "v"(dd), "v"(dw));
                else asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]\n\tv_cvt_rpi_i32_f32 %0, %0" : "=v"(qv) : "v"(dd), "v"(dw));
                __hip_atomic_fetch_add((lds_i*)(size_t)a2 + j * 64, qv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            // bucketMulQ4.metal:76-81: low nibble first <-> sub-bucket 3,2,1,0; out += (n&8) ? -d : d.  Here: plane
            // (n&8) of slot (sub-bucket, n&7) += d; the planes are subtracted at the hand-off.
            const int di = __float_as_int(dd);
#pragma unroll
            for (int j = 0; j < E; j++) {
                const uint32_t x = pc.word(j);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t a2;
                    asm("v_bfe_u32 %0, %1, %2, 4\n\tv_lshl_add_u32 %0, %0, %3, %4" : "=&v"(a2) : "v"(x), "n"(4 * q), "n"(kShift), "v"(accB));
                    __hip_atomic_fetch_add((lds_i*)(size_t)a2 + ((3 - q) * 16 * E + j) * 64, di, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    };
    auto accumulate = [&](const Piece<E> (&piece)[KB], float dv, uint32_t nv) {
        if (EFFORT_ABLATE_NOSCATTER) {       // ablation build: keep the loads live, skip the LDS scatter
#pragma unroll
            for (int u = 0; u < KB; u++) asm volatile("" ::"v"(piece[u].word(E - 1)), "v"(dv));
            return;
        }
        if (!colOK) return;                  // lanes past the last column sit the batch out (one exec change)
        if (nv == (uint32_t)KB) {        // uniform: full batch, no per-row test
#pragma unroll
            for (int u = 0; u < KB; u++) row(piece[u], __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dv), u)));
        } else {
#pragma unroll
            for (int u = 0; u < KB; u++)
                if ((uint32_t)u < nv) row(piece[u], __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dv), u)));
        }
    };

    // Near the end of its rows (the last two rounds of the loop: a few microseconds of streaming left) a wave asks for the
    // next item: late enough that the queue still balances the workgroups, early enough that the staged loads land under
    // the remaining rows.
    bool asked = false;
    // ---- (Q4) the outlier stepper.  calcOutliers (bucketMulQ4.metal:13-21: out[o] += v[in] * value per outlier).  Registration laid
    //      the table out per block of 64 outputs in jagged-diagonal order, FOUR bytes per outlier (f16 value | input; dispatch.hip):
    //      ONE LANE OWNS AN OUTPUT and sums its products in a register -- no atomics, no fixed point -- entry after entry in the
    //      table's order, while step k of a wave (the k-th entry of every output of the block that has one) is one contiguous
    //      load.  The tile's outputs are shared out among its slice items in whole blocks; the waves of an item take the blocks of
    //      its share in turn, a thin share (a lone call's: one block) splits every block's steps among `parts` waves, whose
    //      partial sums meet in LDS.  v is gathered from an LDS copy.
    //      MERGED (lean kernels, inDim <= 4096; round 5): the streaming loop is bound by the LDS atomics of its scatter (four per
    //      16-bit word), the outlier steps by memory and VALU -- so a wave works its first block's steps INSIDE the streaming loop,
    //      kOlMerged steps per half trip: the entries of the next steps asked for with the next batch of rows, the current ones
    //      added between two batches' scatters; always executed, never branched around (a branch around a load makes hipcc drain
    //      vmcnt(0)): past the wave's last step the loads re-read a valid entry and the adds are predicated off.  What is left
    //      when the rows run out is drained after the loop.  (profiles/r05_q4_ablation.json: as a phase of its own the outliers
    //      were 19 of the 74 us of a 16-call launch.)
    using lds_f = __attribute__((address_space(3))) float;
    uint32_t olLen = 0u, olOut = 0u, olKa = 0u, olKf = 0u, olK1 = 0u, olCur = 0u, olSlot = 0xFFFFFFFFu;
    float olAcc = 0.0f;
    const lds_f* olV = nullptr;                                    // the LDS copy of v the stepper gathers from
    if constexpr (kOlMerge) {
        if (olEarly) {
            olV = (const lds_f*)(size_t)(uint32_t)(size_t)(__attribute__((address_space(3))) void*)(smem + __builtin_amdgcn_readfirstlane(lp.offO));
            if ((uint32_t)wave < olNB * olParts) {
                const uint32_t bq = (uint32_t)wave / olParts, part = (uint32_t)wave % olParts;
                olLen = olMeta0 >> 8; olOut = olMeta0 & 63u;              // lane i <-> the output of rank i: counts descend with the lane
                const uint32_t olLenAll = olLen;
                const uint32_t maxLen = (uint32_t)__builtin_amdgcn_readfirstlane((int)olLen);
                const uint32_t chunk = (maxLen + olParts - 1u) / olParts, k0 = min(maxLen, part * chunk);
                const uint32_t bBeg = __builtin_amdgcn_readfirstlane(olBeg0), bEnd = __builtin_amdgcn_readfirstlane(olEnd0);
                olK1 = bEnd > bBeg ? min(maxLen, k0 + chunk) : k0;
                olKa = olKf = k0;
                olCur = bBeg + wave_sum_u32(min(olLenAll, k0));           // entries of the steps before k0 = sum over the outputs of min(count, k0)
                olLen = min(olLenAll, olK1);                              // (this wave's steps end at olK1: one test per step, `count > step`)
                olSlot = part * olStride + bq * 64u + olOut;
            }
        }
    }
    // acc + x * (the f16 in the high half of the entry): one v_fma_mix_f32 reads the half in place (the FP16 multiply's own idiom)
    auto ol_fma = [](float x, uint32_t ent, float acc) -> float {
        float r;
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(x), "v"(ent), "v"(acc));
        return r;
    };
    auto ol_fetch = [&](uint32_t (&ent)[kOlMerged]) {              // the entries of the next kOlMerged steps asked for (branch-free: lanes past a step's population read on into the next entries)
#pragma unroll
        for (int u = 0; u < kOlMerged; u++) {
            ent[u] = EFFORT_OL_LOAD((a.ol.entry + olCur) + lane);                   // (a uniform base + 4 * lane: no address arithmetic per step; the array ends in 64 entries of padding)
            olCur += (uint32_t)__popcll(__ballot(olLen > olKf + (uint32_t)u));
        }
        olKf += (uint32_t)kOlMerged;
    };
    auto ol_add = [&](const uint32_t (&ent)[kOlMerged]) {
        float x[kOlMerged];
#pragma unroll
        for (int u = 0; u < kOlMerged; u++) x[u] = olV[ent[u] & 0xFFFFu];                       // all the gathers of the unit go out together
#pragma unroll
        for (int u = 0; u < kOlMerged; u++)                                                      // bucketMulQ4.metal:19: out[o.z] += v[o.y] * o.x
            { const float r = ol_fma(x[u], ent[u], olAcc); olAcc = olLen > olKa + (uint32_t)u ? r : olAcc; }      // (a select, not a branch)
        olKa += (uint32_t)kOlMerged;
    };
    // (measured: raising the wave priority of this loop -- s_setprio 2 -- so that a co-resident workgroup's selection does not
    //  take its issue slots is 8 % SLOWER per launch: the other workgroup's head then takes that much longer)
    auto stream_rows = [&](auto withOl, auto aux) {
        constexpr bool OL = decltype(withOl)::value;
        Piece<E> pa[KB], pb[KB];
        uint32_t entA[OL ? kOlMerged : 1], entB[OL ? kOlMerged : 1];
        uint32_t boffA, boffB; float dvA, dvB;
        uint32_t baseA = (uint32_t)(wave * KB), baseB;             // (no LDS atomic for the first batch: the cursor starts at W * KB)
        if (baseA < nU) { decode_first(baseA, boffA, dvA); issue(aux, pa, boffA); }
        if constexpr (OL) ol_fetch(entA);
        while (baseA < nU) {
            if (!asked && nU - baseA <= 4u * KB * W) asked = prefetch();
            if (wstamp && GA_TRACE(ga) && item + GA_CUTJOBS(ga) < (uint32_t)kTraceItems) {       // progress stamps (trace mode only)
                const uint32_t q0 = (4u * baseA) / nU, q1 = min(4u, (4u * (baseA + 2u * KB * W)) / nU);
                for (uint32_t q = q0 + 1u; q <= q1 && q < 4u; q++) GA_TSTAMP(ga)[kTraceOff + (size_t)kTraceItems * 8u + (size_t)(item + GA_CUTJOBS(ga)) * 4u + q - 1u] = wall_clock64();
            }
            baseB = grab();
            decode(baseB, boffB, dvB);
            issue(aux, pb, boffB);
            if constexpr (OL) ol_fetch(entB);
            accumulate(pa, dvA, __builtin_amdgcn_readfirstlane(min((uint32_t)KB, nU - baseA)));
            if constexpr (OL) ol_add(entA);
            baseA = grab();
            decode(baseA, boffA, dvA);
            issue(aux, pa, boffA);
            if constexpr (OL) ol_fetch(entA);
            accumulate(pb, dvB, __builtin_amdgcn_readfirstlane(baseB < nU ? min((uint32_t)KB, nU - baseB) : 0u));
            if constexpr (OL) ol_add(entB);
        }
        if constexpr (OL) {                                        // the rows ran out first: the rest of the wave's block (entA: asked for, not yet added)
            while (olKa < olK1) {
                ol_fetch(entB); ol_add(entA);
                if (olKa >= olK1) break;
                ol_fetch(entA); ol_add(entB);
            }
        }
    };
    // the rows' cache policy: non-temporal, or -- the host says the launches in flight read the SAME matrices (effort_set_row_reuse) -- the ordinary one
    // (the second copy lives in the kernels that serve GROUP launches -- the persistent instantiations and the E = 4 plain ones; the lean E = 1 / E = 2 kernels a
    //  lone call runs keep one policy: with two copies they were 0.1-0.3 us slower per call on the default path, profiles/r06_ab_row_reuse.txt)
    constexpr bool kRowSwitch = kRowAux != 0 && (PERSIST || E == 4);
    const bool rowsReused = kRowSwitch && (ga.split & 8u) != 0u;                   // uniform per launch
    auto stream = [&](auto withOl) {
        if constexpr (kRowSwitch) { if (rowsReused) { stream_rows(withOl, std::integral_constant<int, 0>{}); return; } }
        stream_rows(withOl, std::integral_constant<int, kRowAux>{});
    };
    if constexpr (kOlMerge) { if (olEarly) stream(std::true_type{}); else stream(std::false_type{}); }
    else stream(std::false_type{});
    // (measured too: everything BUT this loop at raised priority -- no effect, 173.9 vs 173.5 us per 32-call launch)
    if (!asked) asked = prefetch();            // (a wave without rows; wave 0 always gets an answer: it is the one that pulls)
    __syncthreads();                           // every wave's atomics have landed in the tile
    if (stamp) GA_TSTAMP(ga)[20] = wall_clock64();
    if (wstamp) ph[4] = wall_clock64();

    // ---- O. Q4 outliers, what the streaming loop did not take: everything in the generic kernels and for inDim > 4096 (v is
    //         copied to LDS here, over the staging regions that are dead by now -- or gathered from memory beyond 16384 inputs);
    //         in the merged form the sums of the waves' first blocks are handed over, and the blocks of a share with more blocks
    //         than waves worked.  (Rounds 2-4: entries interleaved by blocks of 16 outputs, two LDS atomics per outlier on
    //         two-level fixed-point sums -- 3.6 MB per call at 2.4 TB/s, 24 of the 80 us of a 16-call launch: DESIGN.md 4.1.)
    float* const olsum = reinterpret_cast<float*>(smem + offM);        // (means | vblk are dead by now) [parts][blocks of the share * 64]
    if constexpr (FMT != kFp16) {
        if (olAny) {
            const OutlierIndex& ol = a.ol;
            // (merged form: the copy of v sits in a region of its own, filled already; otherwise it is made here, over the staging regions)
            float* const vfull = olEarly ? reinterpret_cast<float*>(smem + __builtin_amdgcn_readfirstlane(lp.offO)) : olsum + (olPer > kOlSumFloats ? olPer : kOlSumFloats);
            const bool vLds = g.inDim <= kOlLdsFloats;
            if constexpr (!kOlMerge) ol_share();                                           // (the generic kernels: not held across the streaming loop)
            const uint32_t bFirst = olBFirst, nB = olNB, parts = olParts;
            if (vLds && !olEarly) {
                for (uint32_t i0 = 0; i0 < g.inDim; i0 += NT * 8u) {                           // eight loads in flight (clamped, branch-free)
                    float x[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) x[u] = a.v[min(i0 + u * NT + tid, g.inDim - 1u)];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t i = i0 + u * NT + tid;
                        if (i < g.inDim) vfull[i] = x[u];
                    }
                }
                __syncthreads();                                                           // vfull is whole
            }
            if (olSlot != 0xFFFFFFFFu) olsum[olSlot] = olAcc;                       // merged form: the wave's first block, summed under the rows
            const lds_f* const vfullL = (const lds_f*)(size_t)(uint32_t)(size_t)(__attribute__((address_space(3))) void*)vfull;
            // (v from its LDS copy or from memory: two instantiations of the loop, NOT one loop reading through `vLds ? vfull : a.v`
            //  -- hipcc turns that select into a generic pointer and every gather into a FLAT load behind s_waitcnt vmcnt(0) lgkmcnt(0))
            auto run_blocks = [&](auto fromLds) {
                for (uint32_t q = (uint32_t)wave + (olEarly ? (uint32_t)W : 0u); q < nB * parts; q += (uint32_t)W) {     // uniform per wave
                    const uint32_t bq = q / parts, part = q % parts;
                    const uint32_t meta = ol.meta[(size_t)(bFirst + bq) * 64u + (uint32_t)lane];
                    const uint32_t lenAll = meta >> 8, myOut = meta & 63u;                 // lane i <-> the output of rank i: counts descend with the lane
                    const uint32_t maxLen = (uint32_t)__builtin_amdgcn_readfirstlane((int)lenAll);
                    const uint32_t chunk = (maxLen + parts - 1u) / parts, k0 = min(maxLen, part * chunk), k1 = min(maxLen, k0 + chunk);
                    const uint32_t myLen = min(lenAll, k1);                                // (this wave's steps end at k1: one test per step, `count > step`)
                    const uint32_t bBeg = __builtin_amdgcn_readfirstlane(ol.blockPtr[bFirst + bq]), bEnd = __builtin_amdgcn_readfirstlane(ol.blockPtr[bFirst + bq + 1u]);
                    // entries of the steps before k0 = sum over the outputs of min(count, k0)
                    uint32_t cur = bBeg + wave_sum_u32(min(lenAll, k0));
                    // (f32 sums, like the reference's float atomics; one fused multiply-add per outlier.  An f64 sum was tried for
                    //  independence of how a thin share's steps are split among the waves: v_cvt_f64_f32 + v_add_f64 per step cost the
                    //  phase a quarter of its time)
                    float acc = 0.0f;
                    auto fetch = [&](uint32_t (&ent)[kOlBatch], uint32_t kk) {             // the entries of steps kk .. kk + kOlBatch - 1 asked for (clamped, branch-free)
#pragma unroll
                        for (int u = 0; u < kOlBatch; u++) {
                            ent[u] = EFFORT_OL_LOAD((ol.entry + cur) + lane);
                            cur += (uint32_t)__popcll(__ballot(myLen > kk + (uint32_t)u));
                        }
                    };
                    auto add = [&](const uint32_t (&ent)[kOlBatch], uint32_t kk) {
                        float x[kOlBatch];
#pragma unroll
                        for (int u = 0; u < kOlBatch; u++) {                              // all the gathers of the unit go out together
                            const uint32_t in = ent[u] & 0xFFFFu;
                            if constexpr (decltype(fromLds)::value) x[u] = vfullL[in]; else x[u] = a.v[in];
                        }
#pragma unroll
                        for (int u = 0; u < kOlBatch; u++)                                                  // bucketMulQ4.metal:19: out[o.z] += v[o.y] * o.x
                            { const float r = ol_fma(x[u], ent[u], acc); acc = myLen > kk + (uint32_t)u ? r : acc; }      // (a select, not a branch)
                    };
                    if (k0 < k1 && bEnd > bBeg) {
                        uint32_t entA[kOlBatch], entB[kOlBatch];
                        uint32_t kk = k0;
                        fetch(entA, kk);
                        for (;;) {
                            const bool more = kk + (uint32_t)kOlBatch < k1;
                            if (more) fetch(entB, kk + (uint32_t)kOlBatch);
                            add(entA, kk);
                            if (!more) break;
                            kk += (uint32_t)kOlBatch;
                            const bool more2 = kk + (uint32_t)kOlBatch < k1;
                            if (more2) fetch(entA, kk + (uint32_t)kOlBatch);
                            add(entB, kk);
                            if (!more2) break;
                            kk += (uint32_t)kOlBatch;
                        }
                    }
                    olsum[part * olStride + bq * 64u + myOut] = acc;                            // every (part, output of the share) is written exactly once
                }
            };
            if (vLds) run_blocks(std::true_type{}); else run_blocks(std::false_type{});
            __syncthreads();
        }
    }

    // ---- E. the tile, back in f32 -> one slab (native [slot][j][lane] order), write-through; ticket; last arriver
    //         of the tile reduces the S slabs in slice order and writes out[] -------------------------------
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a_slabs, 0, (int)__builtin_amdgcn_readfirstlane(vSlabBytes), 0x00020000);
    const uint32_t slabOff = __builtin_amdgcn_readfirstlane(vSlabOff);
    const uint32_t nSlices = __builtin_amdgcn_readfirstlane(vSlices);
    auto tile_out = [&](int o) -> float {                           // output o of the tile, native [slot][j][lane] order
        if (FMT == kFp16) return (float)acc[o] * unscale;
        const int slot = o / (E * 64), rem = o % (E * 64);          // slot = sub-bucket*8 + position
        const int p = (((slot >> 3) * 16 + (slot & 7)) * E * 64) + rem;
        float r = (float)(acc[p] - acc[p + 8 * E * 64]) * unscale;
        if (olAny) {                                                 // this item's share of the tile's outputs: + their outliers
            const uint32_t ol_o = (uint32_t)((rem & 63) * E + (rem >> 6)) * 32u + (uint32_t)slot - s * olPer;      // tile-local output, from the share's start
            if (ol_o < olPer && (uint32_t)(t * TILE_F) + s * olPer + ol_o < g.outDim) {
                float so = olsum[ol_o];
                for (uint32_t pp = 1; pp < olParts; pp++) so += olsum[pp * olStride + ol_o];     // (a thin share's partial sums, in wave order)
                r += so;
            }
        }
        return r;
    };
    if (fused && t == 0 && s == 0 && tid == 0 && !GA_CUTJOBS(ga)) a_cutoff[0] = cutoff;       // BucketMul.cutoff (bucketMul.swift:22) of a call without a cutoff job
    // (the item of tile 0, slice 0 -- not the call's first BLOCK: with a slice count that is not a multiple of 8 that block may be padding of the item grid)
    // (Round 4 built this hand-off WITHOUT ticket and drain -- a reducer named up front polling sentinel slabs -- and measured it
    //  slower on plain grids: branch `chain-launch`)
    {   // (a fixed trip count, the tile read back in ONE LDS round trip: `for (o = tid * 2; o < TILE_F; ...)` compiled to a loop of dependent ones)
        constexpr int kIt = (TILE_F + NT * 2 - 1) / (NT * 2);
        float s0[kIt], s1[kIt];
#pragma unroll
        for (int it = 0; it < kIt; it++) {
            const int o = min(tid * 2 + it * NT * 2, TILE_F - 2);
            s0[it] = tile_out(o); s1[it] = tile_out(o + 1);
        }
#pragma unroll
        for (int it = 0; it < kIt; it++) {
            const int o = tid * 2 + it * NT * 2;
            if (TILE_F % (NT * 2) != 0 && o >= TILE_F) continue;
            typedef uint32_t u2 __attribute__((ext_vector_type(2)));
            u2 pk; pk[0] = __float_as_uint(s0[it]); pk[1] = __float_as_uint(s1[it]);
            __builtin_amdgcn_raw_buffer_store_b64(pk, srs, (uint32_t)o * 4u, slabOff, kSc1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // this wave's slab stores have left the CU
    if (stamp) { GA_TSTAMP(ga)[21] = wall_clock64(); GA_TSTAMP(ga)[22] = n; }
    if (wstamp) ph[5] = wall_clock64();
    // the stamps are flushed off the critical path (after the ticket), spread over 32 cache lines
    auto flush_stamps = [&]() {
#ifndef EFFORT_PRODUCT_ONLY
        if (!wstamp) return;
        if (GA_TRACE(ga) && item + GA_CUTJOBS(ga) < (uint32_t)kTraceItems) {           // one record per item: who / where / when
            unsigned long long* rec = GA_TSTAMP(ga) + kTraceOff + (size_t)(item + GA_CUTJOBS(ga)) * 8u;
            rec[0] = (unsigned long long)(item + GA_CUTJOBS(ga)) | ((unsigned long long)blockIdx.x << 32);
            rec[1] = (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) | ((unsigned long long)(ci & 0xFu) << 4) | ((unsigned long long)(n & 0xFFFFu) << 8) | ((unsigned long long)(t & 0xFFu) << 24) |
                     ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);   // XCC_ID | call (mod 16) | kept rows | tile | HW_ID
#pragma unroll
            for (int i = 0; i < 6; i++) rec[2 + i] = ph[i];
        }
        unsigned long long* line = GA_TSTAMP(ga) + 64 + (item & 31u) * 8u;
#pragma unroll
        for (int i = 0; i < 5; i++) atomicAdd(&line[i], ph[i + 1] - ph[i]);
        atomicAdd(&line[5], 1ull);
        atomicMin(&GA_TSTAMP(ga)[0], ph[0]);
        atomicMax(&GA_TSTAMP(ga)[26], ph[0]);                                       // latest workgroup start
        atomicMax(&GA_TSTAMP(ga)[28], ph[5] - ph[0]);                               // longest workgroup (start .. slab drained)
        atomicMax(&GA_TSTAMP(ga)[29], ph[4] - ph[3]);                               // longest streaming phase
        atomicMax(&GA_TSTAMP(ga)[1], (unsigned long long)wall_clock64());
#endif
    };
    {
        __syncthreads();
        if (tid == 0) {
            const uint32_t ticket = __hip_atomic_fetch_add(&a_counters[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flags[0] = (ticket == nSlices - 1u) ? 1u : 0u;
        }
        __syncthreads();
        if (flags[0] == 0u) { flush_stamps(); return; }
    }
    if (GA_ABLATE(ga) & 2u) { if (tid == 0) __hip_atomic_store(&a_counters[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }

    // last arriver of tile t: every slab of the tile was stored write-through (sc1) and drained before its ticket;
    // read them past L1 (sc1), sum in slice order, un-permute, add the Q4 outliers, write out[].  Each thread owns
    // four adjacent tile slots and keeps up to kRed 16-byte loads in flight (each is a fabric round trip); the four
    // running sums per slot are combined in a fixed order.
    const bool rstamp = GA_TSTAMP(ga) && ci == 0 && t == 0 && tid == 0;
    if (rstamp) GA_TSTAMP(ga)[23] = wall_clock64();
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const uint32_t sliceStride = __builtin_amdgcn_readfirstlane(vSliceStride);
    // G thread groups share the slices when the workgroup has more threads than the tile has float4 columns (E = 1): each
    // group then has at most 16 slices = ONE round trip; the groups' partial sums meet in LDS and are added in group order.
    constexpr int kCols4 = TILE_F / 4;
    constexpr int G = NT / kCols4 >= 2 ? NT / kCols4 : 1;
    // [G][TILE_F] partial sums of the thread groups, BEHIND the accumulator tile (the list and the rest of the cutoff table's
    // region are free: G * TILE_F * 4 + the tile <= the table, see plan_lds)
    float* const gpart = reinterpret_cast<float*>(smem + offA + (uint32_t)TILE_L * 4u);
    static_assert(G == 1 || (uint32_t)TILE_L * 4u + (uint32_t)G * TILE_F * 4u <= (cutoff_table_bytes(NT) > (uint32_t)TILE_L * 4u ? cutoff_table_bytes(NT) : (uint32_t)TILE_L * 4u),
                  "the thread groups' partial sums must fit behind the tile");
    auto reduce_tile = [&](auto kc) {
        constexpr int kRed = decltype(kc)::value;          // 16-byte slab loads in flight per thread
        const int grp = G > 1 ? tid / kCols4 : 0;
        const uint32_t per = G > 1 ? ((nSlices + G - 1) / G + 3u) / 4u * 4u : nSlices;   // slices per group, a multiple of 4
        const uint32_t sl0 = (uint32_t)grp * per, sl1 = min(nSlices, sl0 + per);
        for (int o = (G > 1 ? tid % kCols4 : tid) * 4; o < TILE_F; o += (G > 1 ? kCols4 : NT) * 4) {
            const uint32_t vo = t * (uint32_t)(TILE_F * 4) + (uint32_t)o * 4u;
            // the residual of this thread's four outputs is asked for BEFORE the slabs (it was a dependent round trip after them:
            // +1.7 us per launch with the epilogue)
            float res[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (FUSED && a.resid && (G == 1 || grp == 0)) {
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const uint32_t oo = (uint32_t)o + h, lane2 = oo & 63u, sj = oo >> 6, j = sj % E, slot = sj / E;
                    const uint32_t c2 = t * (64u * E) + lane2 * E + j;
                    if (c2 < g.cols) res[h] = *(a.resid + c2 * NACC + slot);
                }
            }
            float sm[4][4];                                // [tile slot of this thread][slice % 4]: fixed summation order
#pragma unroll
            for (int h = 0; h < 4; h++) { sm[h][0] = 0.0f; sm[h][1] = 0.0f; sm[h][2] = 0.0f; sm[h][3] = 0.0f; }
            for (uint32_t sl = sl0; sl < sl1; sl += kRed) {
                u4v r[kRed];
#pragma unroll
                for (int i = 0; i < kRed; i++)
                    r[i] = __builtin_amdgcn_raw_buffer_load_b128(srs, vo, min(sl + i, nSlices - 1u) * sliceStride, kSc1);
#pragma unroll
                for (int i = 0; i < kRed; i++) {
                    if (sl + i < sl1) {
#pragma unroll
                        for (int h = 0; h < 4; h++) sm[h][i & 3] += __uint_as_float(r[i][h]);
                    }
                }
            }
            float tot[4];
#pragma unroll
            for (int h = 0; h < 4; h++) tot[h] = (sm[h][0] + sm[h][1]) + (sm[h][2] + sm[h][3]);
            if (G > 1) {
#pragma unroll
                for (int h = 0; h < 4; h++) gpart[grp * TILE_F + o + h] = tot[h];
                __syncthreads();                           // (every thread of the workgroup runs exactly one such iteration)
                if (grp != 0) continue;
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    tot[h] = gpart[o + h];
                    for (int g2 = 1; g2 < G; g2++) tot[h] += gpart[g2 * TILE_F + o + h];
                }
            }
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const uint32_t oo = (uint32_t)o + h;
                const uint32_t lane2 = oo & 63u, sj = oo >> 6, j = sj % E, slot = sj / E;
                const uint32_t c2 = t * (64u * E) + lane2 * E + j;
                if (c2 < g.cols) {
                    const uint32_t oi = c2 * NACC + slot;
                    const float val = (FUSED && a.resid) ? res[h] + tot[h] : tot[h];
                    a.out[oi] = val;
                }
            }
        }
    };
    // (the four running sums take slices i%4; chunk sizes are multiples of 4, so the order does not depend on the chunk.
    //  Up to 16 slices per thread group the whole reduction is ONE memory round trip per thread.)
    if (nSlices <= 8u * G) reduce_tile(std::integral_constant<int, 8>{});
    else reduce_tile(std::integral_constant<int, 16>{});
    // (G > 1: thread group 0 reads the partial sums in the accumulator region after reduce_tile's only barrier; a persistent
    //  workgroup's next item -- or cutoff job -- zeroes its count table there, so the region must be quiescent first)
    if (PERSIST && G > 1) __syncthreads();
    if (rstamp) GA_TSTAMP(ga)[24] = wall_clock64();
    if (tid == 0) {
        __hip_atomic_store(&a_counters[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next call
#if !defined(EFFORT_CUT_FINE) && !defined(EFFORT_PRODUCT_ONLY)
        if (GA_TSTAMP(ga)) {
            flush_stamps();
            const uint32_t done = __hip_atomic_fetch_add(ga.groupDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (done == ga.totalTiles - 1u) {                                  // whole kernel finished: fold the stamps
                const unsigned long long t0 = __hip_atomic_load(&GA_TSTAMP(ga)[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long t1 = __hip_atomic_load(&GA_TSTAMP(ga)[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                GA_TSTAMP(ga)[2] += t1 - t0; GA_TSTAMP(ga)[3] += 1;
                GA_TSTAMP(ga)[27] += __hip_atomic_load(&GA_TSTAMP(ga)[26], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - t0;   // dispatch ramp
                __hip_atomic_store(&GA_TSTAMP(ga)[26], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&GA_TSTAMP(ga)[0], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&GA_TSTAMP(ga)[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ga.groupDone, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#endif
    }
}

// A cutoff job: findCutoff32 of call ci, once for the launch (persistent launches: the first items of every queue, so
// the first workgroups to run take them; they wait on nothing).  The value goes to ga.cutoff[ci] write-through, then the
// flag is raised; the call's items pick it up (mul_item, phase B).
template <int NT>
__device__ __forceinline__ void cutoff_job(const GroupKArgs& ga, uint32_t ci, char* smem) {
    constexpr int VPT = 4096 / NT;
    const CallDesc& a = ga.call[ci];
    int tid0 = threadIdx.x;
    asm volatile("" : "+v"(tid0));
    const int tid = tid0;
    const uint32_t e = a.expNo ? a.expNo[0] : 0u;
    const uint16_t* pr = a.probes + (size_t)e * kProbes;
    float vj[VPT]; uint16_t prj[VPT];
#pragma unroll
    for (int i = 0; i < VPT; i++) { vj[i] = a.v[tid + NT * i]; prj[i] = pr[tid + NT * i]; }
    [[maybe_unused]] constexpr bool PERSIST = true;              // (cutoff jobs exist in persistent launches only)
    const unsigned long long tj0 = (GA_TSTAMP(ga) && GA_TRACE(ga)) ? wall_clock64() : 0ull;
    const float cutoff = block_find_cutoff<NT>(vj, prj, a.q, smem + cutoff_table_bytes(NT), reinterpret_cast<uint32_t*>(smem), []() {}, nullptr);
    if (tid == 0 && GA_TSTAMP(ga) && GA_TRACE(ga)) {
        unsigned long long* rec = GA_TSTAMP(ga) + kTraceOff + (size_t)ci * 8u;
        rec[0] = (unsigned long long)ci | ((unsigned long long)blockIdx.x << 32) | (1ull << 63);
        rec[1] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
        rec[2] = tj0; rec[3] = wall_clock64();
    }
    if (tid == 0) {
        __hip_atomic_store(&ga.queue[9 * 16 + ci], __float_as_uint(cutoff) | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // value and "ready" in one word
        __hip_atomic_store(reinterpret_cast<uint32_t*>(ga.cutoff + ci), __float_as_uint(cutoff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // BucketMul.cutoff, for the host
    }
    __syncthreads();                                               // the table region is free for the next item
}

// The kernel: every workgroup pulls items until the queue of its XCD is dry.  Launched with one item per workgroup it
// is a plain grid; launched with fewer workgroups than items (ga.persistent) the workgroups are PERSISTENT: the
// dispatcher only places ~40 workgroups/us chip-wide and spreads a large grid unevenly over the CUs (measured with
// tools/lab/microbench.hip: residency probe), so a group launch sizes its grid to the chip -- R workgroups per CU,
// R fixed by the LDS each one asks for -- and balances the work itself.
// (Round 4's CHAIN instantiation -- a layer's dependent multiplies as stages of ONE launch, measured 22 % slower than the launches of
//  their own -- lives on branch `chain-launch`: DESIGN.md 4.5.)
template <int FMT, int E, int W, bool FUSED, bool COMPACT = false, bool PERSIST = true>
__global__ __launch_bounds__(64 * W, (W <= 8 ? EFFORT_MIN_WAVES_PER_EU : 4)) void bucket_mul_kernel(const GroupKArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t s_item;
    __shared__ unsigned long long s_next;                  // the item after the current one | the generation (item count) it was pulled in << 32: ONE word,
                                                           // so that a wave polling mid-loop can never pair a new generation with a stale item
#ifdef EFFORT_CUT_FINE
    if (GA_TSTAMP(ga) && blockIdx.x == 0 && threadIdx.x == 0) GA_TSTAMP(ga)[26] = clock64();
#endif
#ifndef EFFORT_NO_TOUCH
    // Plain grids: a workgroup's first ~10 scalar loads -- item range, call descriptor, geometry, pointers -- depend on one another,
    // and each first touch of a 64-byte line of the kernel-argument segment misses the scalar cache: the misses queue up behind one
    // another on the call's chain.  One dword of every line a small group's prologue reads (header, ranges, geometries, the first
    // three descriptors) is asked for HERE, all at once; the values go nowhere -- the lines are warm when the chain asks for them.
    if constexpr (!PERSIST) {
        const uint32_t* const kw = reinterpret_cast<const uint32_t*>(&ga);
        uint32_t touch[12];
#pragma unroll
        for (int i = 0; i < 12; i++) touch[i] = kw[i * 16];
#pragma unroll
        for (int i = 0; i < 12; i++) asm volatile("" :: "s"(touch[i]));
    }
#endif
#ifdef EFFORT_CUT_FINE
#define EFFORT_ESTAMP(i) if (GA_TSTAMP(ga) && blockIdx.x == 0 && threadIdx.x == 0) GA_TSTAMP(ga)[16 + (i)] = clock64();
#else
#define EFFORT_ESTAMP(i)
#endif
    EFFORT_ESTAMP(0)
    // A CU's two persistent workgroups start together and -- their items being alike -- stay IN STEP: both in their heads (staging, cutoff wait,
    // selection), both streaming, both in their tails (outliers, hand-off, reduction), so the pipe that bounds the stream (Q4: the LDS atomics; per-item
    // traces profiles/r06_q4_timelines.txt) idles through every head and tail.  The workgroups placed second (block >= numCU: the dispatcher fills the
    // CUs round-robin -- speed only, never correctness) start `staggerSleeps` x ~0.9 us late and the pair runs out of step from then on.  Measured, Q4 alone on the
    // chip: 32 calls per launch 109.5 -> 99.8 us at 6-12 us, 16 calls as 768 items 85.4 -> 69.0; with four launches in flight it LOSES 2-5 % (the CU is
    // busy anyway, the wait is just a wait), so the host asks for it only where launches do not overlap (api.hip: a context without lanes).
#ifdef EFFORT_Q4_STAGGER_US                 // (lab A/B builds: a fixed delay whatever the host says)
    const uint32_t staggerSleeps = (uint32_t)((EFFORT_Q4_STAGGER_US) * 10 / 9);
#else
    const uint32_t staggerSleeps = ga.staggerSleeps;
#endif
    if constexpr (PERSIST) {                    // (counted sleeps, not a clock: the shipped kernels read no clock at all)
        if (GA_PERSISTENT(ga) && staggerSleeps && blockIdx.x >= ga.numCU)
            for (uint32_t i = 0; i < staggerSleeps; i++) __builtin_amdgcn_s_sleep(32);
    }
    const uint32_t total = ga.totalItems + GA_CUTJOBS(ga);
    uint32_t cachedCall = 0xFFFFFFFFu; float cachedCutoff = 0.0f;
    const uint32_t x = blockIdx.x & 7u;                    // block b sits on XCD b%8; item i wants XCD i%8
    uint32_t dry = 0;                                      // (thread 0) queues found empty: when its own XCD's queue is dry a workgroup takes items of the others
    // thread 0: the next item of this XCD's queue, or -- that one dry -- of the next queue that still has one; `total` when all are dry
    // (Measured and dropped: queues handing out positions from both ends -- the fast workgroup of every CU, the one placed
    //  first, from the front where the full column tiles are, the slow one from the back where the ragged last tiles were
    //  put, one for each slow workgroup of a 32-call launch.  176.8 vs 172.6 us per launch: the launch does not end on the
    //  slow workgroups' item size but on the chip streaming with one workgroup per CU for its last 30 us either way.)
    auto pull = [&]() -> uint32_t {
        uint32_t got = total;
        for (uint32_t tries = 0; tries < 8u && got >= total; tries++) {
            const uint32_t q = (x + tries) & 7u;
            if ((dry >> q) & 1u) continue;
            const uint32_t cand = __hip_atomic_fetch_add(&ga.queue[q * 16u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 8u + q;
            if (cand < total) got = cand; else dry |= 1u << q;
            if (GA_ABLATE(ga) & 64u) break;                    // (ablate 64: no stealing)
        }
        return got;
    };
    auto pull_sync = [&]() -> uint32_t {                   // ... handed to the whole workgroup
        if (threadIdx.x == 0) s_item = pull();
        __syncthreads();
        const uint32_t it = __builtin_amdgcn_readfirstlane(s_item);   // uniform by construction: keep everything derived from it scalar
        __syncthreads();
        return it;
    };
    // The loop is software-pipelined over the items of a persistent workgroup: near the end of item i's streaming phase wave 0
    // pulls item i+1 from the queue and publishes it in LDS (s_next: item, generation); every wave that finds it there
    // issues its share of item i+1's stage loads (row means + slice of v, stage_issue) under its remaining rows.  A wave that
    // finishes its rows before the answer is there stages its share at the start of item i+1 instead.
    constexpr bool kPipe = !FUSED && FMT == kFp16;         // (a fused input prologue transforms v in place; Q4's outlier phase reuses the staging region)
    const LdsPlan lp = ga.lp;                              // (the launcher's plan_lds<FMT, E, W> over ga.geom)
    uint32_t par = 0;                                      // vblk buffer of the current item
    bool staged = false;                                   // (per wave) its share of the item's stage loads is already in flight / landed
    uint32_t gen = 0;                                      // items this workgroup has worked
    if (threadIdx.x == 0) __hip_atomic_store(&s_next, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t item = GA_PERSISTENT(ga) ? pull_sync() : blockIdx.x;
    while (item < total) {
        if (item < GA_CUTJOBS(ga)) {                           // uniform: a cutoff job (the queues hand these out first)
            if (item < ga.count) cutoff_job<64 * W>(ga, item, smem);
            item = GA_PERSISTENT(ga) ? pull_sync() : total;
            continue;
        }
        ItemRef ref;
        EFFORT_ESTAMP(1)
        if (!locate_item(ga, item - GA_CUTJOBS(ga), ref)) {    // padding of the item grid (slices are dealt in rounds of 8)
            item = GA_PERSISTENT(ga) ? pull_sync() : total;
            continue;
        }
        EFFORT_ESTAMP(2)
        gen++;
        bool stagedNext = false;
        auto prefetch = [&]() -> bool {
            if (!GA_PERSISTENT(ga)) return true;
            if (threadIdx.x == 0)                                  // wave 0's first call: pull, publish (item and generation in one store)
                __hip_atomic_store(&s_next, ((unsigned long long)gen << 32) | (unsigned long long)pull(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned long long nx = __hip_atomic_load(&s_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__builtin_amdgcn_readfirstlane((uint32_t)(nx >> 32)) != gen) return false;   // no answer yet
            const uint32_t next = __builtin_amdgcn_readfirstlane((uint32_t)nx);
            ItemRef nref;
            if (kPipe && !(GA_ABLATE(ga) & 128u) && next >= GA_CUTJOBS(ga) && next < total && locate_item(ga, next - GA_CUTJOBS(ga), nref)) {
                int tid0 = threadIdx.x;
                asm volatile("" : "+v"(tid0));
                stage_issue<FMT, W, COMPACT>(ga, nref, tid0, smem, lp, par ^ 1u);
                stagedNext = true;
            }
            return true;
        };
        const bool firstOwn = gen == 1u && (GA_ABLATE(ga) & 512u) != 0u;     /* (measured: evaluating the first item's cutoff locally instead of waiting for the job is 1 us slower per 32-call launch; kept as an ablation) */
        mul_item<FMT, E, W, FUSED, COMPACT, PERSIST>(ga, item - GA_CUTJOBS(ga), ref, smem, lp, cachedCall, cachedCutoff, par, staged, firstOwn, prefetch);
        // (mul_item's barriers lie between wave 0's publication and this read)
        item = GA_PERSISTENT(ga) ? __builtin_amdgcn_readfirstlane((uint32_t)__hip_atomic_load(&s_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) : total;
        staged = stagedNext;
        par ^= 1u;
    }
    if (GA_PERSISTENT(ga) && threadIdx.x == 0) {               // the last workgroup out rewinds the queues (and flags) for the next launch
        const uint32_t gone = __hip_atomic_fetch_add(&ga.queue[8 * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gone == gridDim.x - 1u) {
            for (int i = 0; i <= 8; i++) __hip_atomic_store(&ga.queue[i * 16], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = 0; i < kMaxGroup; i++) __hip_atomic_store(&ga.queue[9 * 16 + i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- host side ------------------------------------------------------------------------------
template <int FMT, int E, int W>
static hipError_t launch_mul_t(const GroupKArgs& gaIn, hipStream_t st) {
    GroupKArgs ga = gaIn;
    // plain grids of the product path (no stamps, no ablation switches) run the lean instantiation (PERSIST = false, see above);
    // built for 8-wave workgroups, the only size the heuristics choose
    constexpr bool kLean = W == 8;
#ifdef EFFORT_LEAN_STAMPS
    const bool lean = kLean && !ga.persistent && !ga.ablate;
#else
    const bool lean = kLean && !ga.persistent && !ga.tstamp && !ga.ablate;
#endif
    const LdsPlan lp = ga.lp = plan_lds<FMT, E, W>(ga.geom, kMaxGeoms, lean || (kLean && kProductOnly));
    uint32_t lds = lp.total;
    if (lp.offA > 65536u && FMT == kFp16) return hipErrorInvalidValue;     // stage_issue's destinations sit below offA
    for (uint32_t i = 0; i < ga.count; i++) {
        const MulGeom& g = ga.geom[ga.call[i].geom];
        if (g.slots > (uint32_t)kRounds * 64 * W || g.slots != (FMT == kFp16 ? (g.rowsPerIn << g.sliceLog2) : g.sliceRows * 8u)) return hipErrorInvalidValue;
        if (FMT == kFp16 ? (1u << g.sliceLog2) > 64u * W : g.sliceRows > 128u * W) return hipErrorInvalidValue;       // stage_issue: a thread lands one (Q4: two) inputs of the slice
        if (((uint32_t)ga.wgEnd8[i] - (i ? (uint32_t)ga.wgEnd8[i - 1] : 0u)) * 8u != align_up(g.tiles * g.slices, 8)) return hipErrorInvalidValue;
    }
    if (ga.totalItems != (uint32_t)ga.wgEnd8[ga.count - 1] * 8u) return hipErrorInvalidValue;
    uint32_t grid = ga.totalItems;
    if (ga.persistent) {
        // R workgroups per CU, exactly: ask for enough LDS that R+1 cannot share a CU
        const uint32_t R = ga.persistent;
        grid = ga.numCU * R;
        const uint32_t force = (160u * 1024u) / (R + 1u) + 512u;
        if (lds < force && force <= (160u * 1024u) / R) lds = force;
    }
    bool fusedAny = false;
    for (uint32_t i = 0; i < ga.count; i++) fusedAny = fusedAny || ga.call[i].pre || ga.call[i].resid;
    if (fusedAny && FMT != kFp16) return hipErrorInvalidValue;
    // (every instantiation may use the whole LDS: bucket_mul_prepare_device, once per device at effort_create)
    if (lds > kMaxLdsBytes) return hipErrorInvalidValue;
    const bool compact = (ga.split & 4u) != 0u;                   // (api.hip: persistent FP16 launches of plain calls)
    if (compact && (FMT != kFp16 || (fusedAny && !lean))) return hipErrorInvalidValue;
    const dim3 gd(grid), bd(64 * W);
    if constexpr (kLean) {
        if (lean && compact && fusedAny) { hipLaunchKernelGGL((bucket_mul_kernel<kFp16, E, W, true, true, false>), gd, bd, lds, st, ga); return hipGetLastError(); }
        if (lean && compact) { hipLaunchKernelGGL((bucket_mul_kernel<kFp16, E, W, false, true, false>), gd, bd, lds, st, ga); return hipGetLastError(); }
        if (lean && fusedAny) { hipLaunchKernelGGL((bucket_mul_kernel<kFp16, E, W, true, false, false>), gd, bd, lds, st, ga); return hipGetLastError(); }
        if (lean) { hipLaunchKernelGGL((bucket_mul_kernel<FMT, E, W, false, false, false>), gd, bd, lds, st, ga); return hipGetLastError(); }
    }
    if (fusedAny) hipLaunchKernelGGL((bucket_mul_kernel<kFp16, E, W, true>), gd, bd, lds, st, ga);
    else if (compact) hipLaunchKernelGGL((bucket_mul_kernel<kFp16, E, W, false, true>), gd, bd, lds, st, ga);
    else hipLaunchKernelGGL((bucket_mul_kernel<FMT, E, W, false>), gd, bd, lds, st, ga);
    return hipGetLastError();
}

#define EFFORT_GEOMS(X) X(16, 1) X(16, 2) X(16, 4) X(8, 1) X(8, 2) X(8, 4) X(4, 1) X(4, 2) X(4, 4) X(2, 4)

// Lets every instantiation of the kernel ask for up to the whole LDS of a CU as dynamic shared memory, on the CURRENT device.
// Done once per device when its first context is created (api.hip), not lazily at launch: a function attribute belongs to a
// device, launches may come from several threads and devices, and a launch may sit inside a hipGraph capture.
template <int FMT, int E, int W>
static hipError_t prepare_t() {
    auto set = [&](const void* f) {            // (the kernel's static words come out of the same 160 KB)
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, f);
        if (e != hipSuccess) return e;
        if (fa.sharedSizeBytes > 1024u) return hipErrorInvalidValue;
        return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160u * 1024u - (uint32_t)fa.sharedSizeBytes));
    };
    constexpr bool kLean = W == 8;
    hipError_t err = set(reinterpret_cast<const void*>(&bucket_mul_kernel<FMT, E, W, false>));
    if constexpr (kLean) if (err == hipSuccess) err = set(reinterpret_cast<const void*>(&bucket_mul_kernel<FMT, E, W, false, false, false>));
    if constexpr (FMT == kFp16) {
        if (err == hipSuccess) err = set(reinterpret_cast<const void*>(&bucket_mul_kernel<kFp16, E, W, true>));
        if (err == hipSuccess) err = set(reinterpret_cast<const void*>(&bucket_mul_kernel<kFp16, E, W, false, true>));
        if constexpr (kLean) if (err == hipSuccess) err = set(reinterpret_cast<const void*>(&bucket_mul_kernel<kFp16, E, W, true, false, false>));
        if constexpr (kLean) if (err == hipSuccess) err = set(reinterpret_cast<const void*>(&bucket_mul_kernel<kFp16, E, W, false, true, false>));
        if constexpr (kLean) if (err == hipSuccess) err = set(reinterpret_cast<const void*>(&bucket_mul_kernel<kFp16, E, W, true, true, false>));
    }
    return err;
}
hipError_t bucket_mul_prepare_device() {
    hipError_t err = hipSuccess;
#define EFFORT_CASE(w, e) if (err == hipSuccess) err = prepare_t<kFp16, e, w>(); if (err == hipSuccess) err = prepare_t<kQ4, e, w>();
    EFFORT_GEOMS(EFFORT_CASE)
#undef EFFORT_CASE
    return err;
}

template <int FMT>
static hipError_t launch_mul_fmt(int W, int E, const GroupKArgs& a, hipStream_t st) {
#define EFFORT_CASE(w, e) if (W == w && E == e) return launch_mul_t<FMT, e, w>(a, st);
    EFFORT_GEOMS(EFFORT_CASE)
#undef EFFORT_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_bucket_mul(Format fmt, int W, int E, const GroupKArgs& a, hipStream_t st) {
    return fmt == kFp16 ? launch_mul_fmt<kFp16>(W, E, a, st) : launch_mul_fmt<kQ4>(W, E, a, st);
}

size_t bucket_mul_lds_bytes(Format fmt, int W, int E, const MulGeom& g, bool lean) {
#define EFFORT_CASE(w, e)                                                                     \
    if (W == w && E == e)                                                                     \
        return fmt == kFp16 ? plan_lds<kFp16, e, w>(&g, 1, lean).total : plan_lds<kQ4, e, w>(&g, 1, lean).total;
    EFFORT_GEOMS(EFFORT_CASE)
#undef EFFORT_CASE
    return 0;
}

// Resident workgroups per CU the runtime grants an instantiation at a given dynamic LDS size (0: unsupported).
int bucket_mul_occupancy(Format fmt, int W, int E, size_t ldsBytes) {
    int n = 0;
#define EFFORT_CASE(w, e)                                                                                              \
    if (W == w && E == e) {                                                                                            \
        const void* f = fmt == kFp16 ? reinterpret_cast<const void*>(&bucket_mul_kernel<kFp16, e, w, false>)                  \
                                     : reinterpret_cast<const void*>(&bucket_mul_kernel<kQ4, e, w, false>);                   \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 64 * w, ldsBytes) != hipSuccess) n = 0;                \
    }
    EFFORT_GEOMS(EFFORT_CASE)
#undef EFFORT_CASE
    return n;
}

uint32_t bucket_mul_max_candidates(int W) { return (uint32_t)kRounds * 64u * (uint32_t)W; }

}  // namespace effort
