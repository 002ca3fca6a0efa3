// Internal declarations shared by the HIP translation units of libeffort_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <utility>

namespace effort {

constexpr int kWave = 64;            // CDNA4 wavefront
constexpr int kProbes = 4096;        // bucketMul.swift:17
constexpr float kCutoffScale = 100000.0f;   // CUTOFF_SCALE, bucketMul.metal:33

enum Format : int { kFp16 = 0, kQ4 = 1 };

// Geometry of one multiply launch (see DESIGN.md "bucket_mul kernel").
struct MulGeom {
    uint32_t inDim;        // GEMV input size
    uint32_t outDim;       // GEMV output size held by this handle (a column shard in multi-GPU)
    uint32_t cols;         // u16 columns per bucket row: outDim/16 (FP16) or outDim/32 (Q4)
    uint32_t rowsPerIn;    // bucket rows per input row that are present: percentLoad (FP16) or 8 (Q4)
    uint32_t expertRows;   // rowsPerIn * inDim  (= expertSize, loader.swift:50)
    uint32_t tiles;        // T: column tiles of 64*E u16 columns
    uint32_t slices;       // S: row slices actually used (every slice owns sliceRows input rows)
    uint32_t sliceRows;    // B: input rows per slice
    uint32_t sliceLog2;    // ceil(log2(B)): FP16 candidate slots are laid out [rank][2^sliceLog2]
    uint32_t slots;        // candidate slots per slice: rowsPerIn << sliceLog2 (FP16) or 8*B (Q4)
    uint32_t rowPitch;     // bytes from one bucket row to the next: 2*cols as converted, or padded to whole 128-byte lines (effort_weights_align_rows)
    uint32_t numExperts;   // experts stacked in the buffers (bounds of the buffer descriptor)
    uint32_t elems;        // E: u16 columns per lane of this call's tiles (= the launch's template parameter)
};

// The Q4 outliers, indexed at registration (dispatch.hip): FOUR bytes per outlier (f16 value << 16 | input: inDim <= 65536), in a
// jagged-diagonal layout per block of 64 consecutive outputs -- the outputs of a block ranked by entry count (descending), entry k of
// rank i at blockPtr[b] + (entries of steps before k) + i -- so that ONE lane owns an output (its sum lives in a register, no
// atomics) and a wave's step is one contiguous load.  meta[b * 64 + i] = entry count of rank i << 8 | its output within the block.
struct OutlierIndex {
    const uint32_t* blockPtr;  // [ceil(outDim / 64) + 1] entry bounds by block of 64 outputs (nullptr: no outliers)
    const uint32_t* entry;     // [n]  f16 value << 16 | input
    const uint32_t* meta;      // [ceil(outDim / 64) * 64]  entry count << 8 | output in block, by rank
};

// One launch = a GROUP of up to kMaxGroup independent bucketMul calls (own weights, v, out, effort, scratch):
// the decode loop's Wq|Wk|Wv and W1|W3 (runNetwork.swift:132-134,178-182) are such groups.  A lone call's
// workgroups spend most of their life in dependent fixed-latency steps, so one call cannot load the chip; in a
// group the workgroups of different calls overlap on the CUs inside ONE kernel, no stream juggling involved.
// Everything a launch needs travels BY VALUE in the kernel arguments (3.6 KB of the 4 KB a kernel may take; arguments
// are copied when a launch is captured into a hipGraph): a compact descriptor per call, the few distinct launch
// geometries of the group, and the bases of the context scratch the calls index into.
#ifndef EFFORT_MAX_GROUP
#define EFFORT_MAX_GROUP 32
#endif
constexpr int kMaxGroup = EFFORT_MAX_GROUP;        // calls per launch (the macro: A/B builds of the kernel-argument size, tools/ only)
constexpr int kMaxGeoms = 4;
constexpr int kQueueWords = 11 * 16;               // layout of GroupKArgs::queue (words)
constexpr int kTraceOff = 512, kTraceItems = 4096;   // per-item trace records: u64 index into the stamp buffer / capacity         // distinct (shape, slicing) geometries per launch
enum Prologue : uint16_t { kPreNone = 0, kPreSiluGate = 1, kPreRmsNorm = 2 };
struct CallDesc {                    // 112 bytes
    const uint16_t* buckets;
    const void* stats;         // f16x4 (FP16) or f32x2 (Q4) per bucket row
    const float* rankBound;    // [numExperts] sum over ranks of the rank's max |w| (Q4: max row mean): fixed-point bound, from registration
    const uint16_t* probes;    // f16 [numExperts][4096]
    const float* v;
    const uint32_t* expNo;     // nullable
    float* out;                // f32 [outDim]
    OutlierIndex ol;
    uint16_t q;                // Int(4095*(1-effort)), bucketMul.swift:39
    uint16_t bucketsTrim;      // a column shard's `buckets` points rank*cols*2 bytes INTO the full handle's rows: the buffer descriptor ends that many bytes early,
                               // at the end of the full allocation (a ragged last tile of the last row must not read past it)
    uint32_t slabOff;          // this call's partial-tile slabs [slices][tiles][tileFloats]: offset into GroupKArgs::slabs, in units of 64 floats
    uint16_t tileOff;          // ... arrival tickets [tiles]: offset into GroupKArgs::counters
    uint16_t sliceOff;         // ... kept rows per slice [slices] (sum = dispatch.size): offset into GroupKArgs::sliceCounts
    uint16_t geom;             // index into GroupKArgs::geom
    uint16_t pre;              // Prologue: how the kernel derives its input from v (and vAux)
    const void* vAux;          // kPreSiluGate: x3 f32 [inDim], input = x3 * v / (1 + exp(-v)) (silu(x1, x3), matrix.metal:25-35);
                               // kPreRmsNorm: norm weights f16 [inDim], input = v / sqrt(mean(v^2) + 1e-5) * w (rmsNormFast + mul(by:))
    const float* resid;        // nullable epilogue: out = resid + product (h.add(by:), runNetwork.swift:172,183); may alias out
};
// Where the regions of a workgroup's dynamic LDS start (plan_lds in bucket_mul.hip; computed by the launcher, not by every workgroup).
struct LdsPlan { uint32_t offM, offV[2], offA, offL, offC, offO, total; };     // offO: Q4, the whole input vector for the outlier phase (0: none)
// Layout: what EVERY workgroup reads first comes first and together -- the scalars and the LDS plan share the first 64-byte line
// of the kernel-argument block, the scratch pointers the second -- so that a workgroup's first scalar loads are two lines, not a
// chain of dependent ones (a plain grid's workgroup runs its prologue once, on the dependent chain of the call).
struct GroupKArgs {
    uint32_t count;
    uint32_t totalTiles;           // sum of tiles: the workgroup that finishes the last tile folds the timing stamps
    uint32_t persistent;           // 0: one workgroup per item; R > 0: numCU*R persistent workgroups pull items from the queues
    uint32_t numCU;
    uint32_t ablate;               // profiling only (env EFFORT_ABLATE): 2 = no last-arriver reduce, 4 = no row streaming, 8 = no selection, 32 = never wait for a cutoff job
    uint32_t split;                // bit 0: the cutoffs were evaluated by find_cutoff_group_kernel (split mode), else in the multiply kernel;
                                   // bit 2: FP16 calls' `stats` point at the compact row means (u16 per bucket row), not at the f16x4 stats
                                   // bit 3: the bucket rows will be read again soon (effort_set_row_reuse): stream them with the ordinary cache policy, not nt
    uint32_t cutJobs;              // persistent launches: the first cutJobs items (a multiple of 8 >= count) are cutoff jobs, one per call
    uint32_t trace;                // profiling only: 1 = every item leaves a 64-byte record (who ran it, where, its phase stamps) at tstamp + kTraceOff
    LdsPlan lp;                    // of the instantiation launched, over geom[] (launch_mul_t)
    uint32_t totalItems;           // = 8 * wgEnd8[count - 1]: items of the launch (without the cutoff jobs)
    uint32_t staggerSleeps;        // persistent launches: the workgroups placed second on their CU (block >= numCU) start this many `s_sleep 32` (2048 cycles, ~0.9 us each) late; 0: at once -- bucket_mul_kernel
    uint32_t* groupDone;           // counter of finished tiles (zero between launches)
    uint32_t* queue;               // [9][16]: per-XCD item cursors (one cache line each), line 8 = exit counter;
                                   // then [32] cutoff words (value | ready bit) of the calls; all zero between launches
    float* slabs;                  // context scratch the calls index into
    uint32_t* counters;
    uint32_t* sliceCounts;
    float* cutoff;                 // [count]: BucketMul.cutoff of every call
    unsigned long long* tstamp;    // nullable profiling stamps: [0]=min start, [1]=max end, [2]=sum, [3]=launches, [16..] phases
    uint16_t wgEnd8[kMaxGroup];    // exclusive end of each call's item range, in units of 8 items (the ranges are multiples of 8)
    MulGeom geom[kMaxGeoms];
    CallDesc call[kMaxGroup];
};
static_assert(sizeof(CallDesc) == 112 && sizeof(GroupKArgs) <= 4000, "the launch descriptor must fit the kernel-argument segment");

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float half_bits_to_float(uint16_t h) { return __half2float(__ushort_as_half(h)); }

// f32 -> bfloat (round to nearest even) -> f32; bit-for-bit what Metal's bfloat() conversion does.
__device__ __forceinline__ float bf16_round(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) u |= 0x00400000u;       // quiet NaN, keep payload top bits
    else u += 0x7FFFu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xFFFF0000u);
}

// Two of them at once, as 16-bit patterns (x in the low half): gfx950's v_cvt_pk_bf16_f32 rounds to nearest even in ONE instruction
// where the integer form above takes six per value -- the cutoff's value pass converts 2 x 4096 numbers per workgroup, and every
// instruction of a lone call's workgroup is four cycles of the call's dependent chain.  (Finite inputs: identical bits; NaNs are
// quieted by the hardware's own rule.)
typedef __bf16 effort_bf16x2 __attribute__((ext_vector_type(2)));
typedef float effort_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf16_pack2(float x, float y) {
    effort_f32x2 v; v[0] = x; v[1] = y;
    const effort_bf16x2 r = __builtin_convertvector(v, effort_bf16x2);
    uint32_t u; __builtin_memcpy(&u, &r, 4); return u;
}

// ---- wave64 reductions and scans on DPP ------------------------------------------------------
// __shfl_xor / __shfl_down compile to ds_bpermute_b32 on gfx950: an LDS round trip per step, six dependent ones per reduction
// (~400 cycles).  row_shr:1/2/4/8 + row_bcast:15/31 fold into the VALU instruction itself (v_add_u32_dpp ...): six
// instructions, the inclusive scan in every lane and the total in lane 63.
#define EFFORT_DPP_SCAN(v, OP, IDENT)                                                                 \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x111, 0xf, 0xf, false)); /* row_shr:1 */         \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x112, 0xf, 0xf, false)); /* row_shr:2 */         \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x114, 0xf, 0xf, false)); /* row_shr:4 */         \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x118, 0xf, 0xf, false)); /* row_shr:8 */         \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x142, 0xa, 0xf, false)); /* row_bcast:15 */      \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x143, 0xc, 0xf, false)); /* row_bcast:31 */
__device__ __forceinline__ int dpp_op_add(int a, int b) { return a + b; }
__device__ __forceinline__ int dpp_op_umin(int a, int b) { return (int)min((uint32_t)a, (uint32_t)b); }
__device__ __forceinline__ int dpp_op_umax(int a, int b) { return (int)max((uint32_t)a, (uint32_t)b); }
// inclusive prefix sum over the wave's lanes (lane i: v_0 + ... + v_i)
__device__ __forceinline__ uint32_t wave_prefix_sum_u32(uint32_t x) { int v = (int)x; EFFORT_DPP_SCAN(v, dpp_op_add, 0) return (uint32_t)v; }
// the wave's total / minimum / maximum, uniform
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) { int v = (int)x; EFFORT_DPP_SCAN(v, dpp_op_add, 0) return (uint32_t)__builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) { int v = (int)x; EFFORT_DPP_SCAN(v, dpp_op_umin, -1) return (uint32_t)__builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) { int v = (int)x; EFFORT_DPP_SCAN(v, dpp_op_umax, 0) return (uint32_t)__builtin_amdgcn_readlane(v, 63); }
// (f32 sum: the additions run in scan order -- lanes 0..15 left to right within a row, then the rows -- not in the xor
//  butterfly's; use only where the order is free, i.e. not for the rmsNorm sums the glue kernel and the fused prologue share)
__device__ __forceinline__ float wave_sum_f32(float x) {
    int v = __float_as_int(x);
#define EFFORT_FADD_(a, b) __float_as_int(__int_as_float(a) + __int_as_float(b))
    EFFORT_DPP_SCAN(v, EFFORT_FADD_, 0)
#undef EFFORT_FADD_
    return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}

// maximum of any floats (v_max_f32 on DPP), uniform
__device__ __forceinline__ float wave_max_f32(float x) {
    int v = __float_as_int(x);
#define EFFORT_FMAX_(a, b) __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b)))
    EFFORT_DPP_SCAN(v, EFFORT_FMAX_, (int)0xFF800000)      /* -inf */
#undef EFFORT_FMAX_
    return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}
// sum over each row of 16 lanes; the row's total lands in its lane 15 (row_shr only: rows do not mix)
__device__ __forceinline__ float row16_sum_f32(float x) {
    int v = __float_as_int(x);
#define EFFORT_FADD_(a, b) __float_as_int(__int_as_float(a) + __int_as_float(b))
    v = EFFORT_FADD_(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
    v = EFFORT_FADD_(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
    v = EFFORT_FADD_(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
    v = EFFORT_FADD_(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
#undef EFFORT_FADD_
    return __int_as_float(v);
}

// A kernel may ask for up to a CU's whole LDS (less its static words) as dynamic shared memory: set ONCE per (kernel, device) --
// a function attribute belongs to a device, launches may come from several host threads and devices -- not lazily behind a
// per-process static.
inline hipError_t allow_full_lds(const void* fn) {
    static std::mutex m;
    static std::map<std::pair<const void*, int>, bool> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(m);
    bool& d = done[std::make_pair(fn, dev)];
    if (d) return hipSuccess;
    hipFuncAttributes fa;
    e = hipFuncGetAttributes(&fa, fn);
    if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160u * 1024u - (uint32_t)fa.sharedSizeBytes));
    if (e == hipSuccess) d = true;
    return e;
}

// ---- launchers (one per translation unit) ---------------------------------------------------
bool dense_gemv_supported(uint32_t inDim, uint32_t outDim);
hipError_t launch_dense_gemv(const uint16_t* W_f16, const float* v, float* out, uint32_t inDim, uint32_t outDim, hipStream_t st);
hipError_t launch_find_cutoff(const float* v, const uint16_t* probes, const uint32_t* expNo, uint32_t q,
                              float* cutoff, uint32_t* dispatchCount, unsigned long long* tstamp, hipStream_t st);

// Returns hipErrorInvalidValue for unsupported (fmt, W, E).
hipError_t launch_bucket_mul(Format fmt, int wavesPerGroup, int elemsPerLane, const GroupKArgs& ga, hipStream_t st);
hipError_t bucket_mul_prepare_device();     // once per device: the kernels may use the whole LDS (hipFuncSetAttribute)
hipError_t launch_find_cutoff_group(const GroupKArgs& ga, hipStream_t st);    // ga.cutoff[i] of every call
size_t bucket_mul_lds_bytes(Format fmt, int wavesPerGroup, int elemsPerLane, const MulGeom& g, bool lean);    // lean: with the lean Q4 kernels' region for v (plan_lds)
uint32_t bucket_mul_max_candidates(int wavesPerGroup);
int bucket_mul_occupancy(Format fmt, int wavesPerGroup, int elemsPerLane, size_t ldsBytes);
  // rowsPerIn*sliceRows must not exceed this

hipError_t launch_calc_dispatch(Format fmt, const void* stats, const float* v, const uint32_t* expNo,
                                const float* cutoff, const MulGeom& g, float* dispatch, uint32_t* count,
                                uint32_t* ctxCount, uint32_t* blockScratch, hipStream_t st);

hipError_t launch_convert_fp16(const uint16_t* W, uint32_t outDim, uint32_t inDim, uint16_t* buckets, uint32_t pitchCols,
                               uint16_t* stats, uint16_t* probes, uint16_t* scratchVals, int* status, hipStream_t st);

hipError_t launch_convert_q4(const uint16_t* core2, uint32_t inDim, uint32_t outDim, uint32_t cnt, uint16_t* buckets, float* stats, uint16_t* probes,
                             float* outliers, int numCU, hipStream_t st);

hipError_t launch_compact_means(const void* stats_f16x4, uint16_t* means, uint32_t rows, hipStream_t st);
hipError_t launch_rank_bound(Format fmt, const uint16_t* buckets, uint32_t pitchCols, const void* stats, uint32_t numExperts, uint32_t rowsPerIn,
                             uint32_t inDim, uint32_t cols, float* rowScratch, float* rankBound, hipStream_t st);
// decode-loop glue (decode.hip)
hipError_t launch_add_rmsnorm_mul(float* h, const float* delta, const uint16_t* w, float* out, uint32_t n, hipStream_t st);
hipError_t launch_rope_kv(const float* xq, const float* xk, const float* xv, float* qOut, float* kCache, float* vCache,
                          const uint32_t* pos, uint32_t numHeads, uint32_t numHeadsKV, uint32_t headDim, float ropeBase, uint32_t maxTokens,
                          int* status, hipStream_t st);
hipError_t launch_attention(const float* q, const float* kCache, const float* vCache, const uint32_t* pos, float* out,
                            uint32_t numHeads, uint32_t headDim, uint32_t maxTokens, hipStream_t st);
hipError_t launch_rope_attention(const float* xq, const float* xk, const float* xv, float* kCache, float* vCache, const uint32_t* pos,
                                 float* out, uint32_t numHeads, uint32_t numHeadsKV, uint32_t headDim, uint32_t maxTokens, float ropeBase,
                                 int* status, hipStream_t st);
hipError_t launch_silu_mul(const float* x1, const float* x3, float* out, uint32_t n, hipStream_t st);
hipError_t launch_fetch_row(const uint16_t* emb, const uint32_t* id, float* out, uint32_t n, hipStream_t st);
hipError_t launch_top2_softmax(const float* gate, uint32_t n, uint32_t* idx, float* val, hipStream_t st);
hipError_t launch_mix2(const float* f0, const float* f1, const float* val, float* out, uint32_t n, hipStream_t st);
hipError_t launch_argmax(const float* logits, uint32_t n, uint32_t* idOut, uint32_t* pos, uint32_t* history, uint32_t historyLen,
                         int* status, hipStream_t st);

hipError_t launch_f32_to_f16(const float* in, uint16_t* out, uint32_t n, hipStream_t st);
hipError_t launch_cosine(const float* a, const float* b, uint32_t n, float* out3, hipStream_t st);
hipError_t launch_validate_outliers(const float* outliers, uint64_t n, uint32_t inDim, uint32_t outDim, int* bad, hipStream_t st);
hipError_t launch_build_outlier_index(const float* outliers, uint64_t n, uint32_t inDim, uint32_t outDim, uint32_t* rowPtr,
                                      uint32_t* blockPtr, uint32_t* entry, uint32_t* meta, uint32_t* tmp, hipStream_t st);
hipError_t launch_sort_pairs_u32(const uint32_t* keysIn, uint32_t* keysOut, const uint32_t* valsIn, uint32_t* valsOut, uint64_t n, hipStream_t st);   // stable (convert_q4.hip)

}  // namespace effort
