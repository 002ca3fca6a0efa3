// bucket_mul -- the hot kernel.  Replaces prepareDispatch + roundUp + zeroRange32 + bucketMul
// (bucketMul.metal:11-117) and prepareDispatchQ4 + bucketMulQ4 (bucketMulQ4.metal:25-92) with ONE
// fused launch; bucketIntegrate (bucketMul.metal:122-137) becomes the small `integrate` kernel below.
//
// Work decomposition (MI355X-first, not the reference's [cols x 32] grid of 32-thread groups):
//   grid   = T column tiles x S row slices, one workgroup of W wave64 each (W = 16 by default),
//            block id -> (tile, slice) remapped so all tiles of a slice sit on one XCD (same L2:
//            they share the stats lines and the 128-B lines that straddle two tiles).
//   slice  = a contiguous block of B input rows (all ranks of them) -> its kept bucket rows are
//            neighbours in HBM within each rank plane.
//   tile   = 64*E u16 columns; lane l of every wave owns columns l*E .. l*E+E-1 of the tile, so one
//            wave load instruction reads one contiguous 128*E-byte piece of one bucket row.
// Per workgroup:
//   1. test its 16*B (Q4: 8*B) candidate rows against the cutoff exactly as prepareDispatch does
//      (cutoff < (1e5*mean)*|v|) and compact the survivors, in ascending bucket-row order, into an
//      LDS list with wave ballots + mbcnt prefix (no global atomics, deterministic);
//   2. waves take list entries round-robin and stream those rows from HBM (16 loads in flight per
//      lane), scattering each product into a PRIVATE per-wave LDS accumulator tile with ds_add_f32:
//      acc[slot][j][lane], slot = the 4 position bits of the f16 weight (Q4: sub-bucket*8 + the 3
//      position bits of the nibble).  The layout puts lane l on LDS bank l%32 whatever the slot, so
//      the scatter is bank-conflict free; private tiles make the f32 summation order fixed.
//   3. the W private tiles are summed in wave order and written as one partial "slab".
// `integrate` then sums the S slabs per output in slice order (and, for Q4, adds the outliers in
// table order) and writes out[] -- deterministic end to end.
#include "effort_internal.h"

namespace effort {

constexpr int kBatch = 16;   // bucket rows in flight per wave (one VGPR-resident load each)

template <int FMT> struct Fmt;
template <> struct Fmt<kFp16> { static constexpr int kAcc = 16; };
template <> struct Fmt<kQ4> { static constexpr int kAcc = 32; };

template <int E> struct LoadT;
template <> struct LoadT<1> { using type = uint16_t; };
template <> struct LoadT<2> { using type = uint32_t; };
template <> struct LoadT<4> { using type = uint2; };

__device__ __forceinline__ uint32_t word_of(uint16_t w, int) { return w; }
__device__ __forceinline__ uint32_t word_of(uint32_t w, int j) { return (w >> (16 * j)) & 0xFFFFu; }
__device__ __forceinline__ uint32_t word_of(uint2 w, int j) { return ((j < 2 ? w.x : w.y) >> (16 * (j & 1))) & 0xFFFFu; }

__device__ __forceinline__ void lds_add(float* p, float x) {
    // wave-private slot: a plain LDS read-modify-write instruction (ds_add_f32), no return value
    __hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

__host__ __device__ inline uint32_t align_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

// LDS carve (bytes): [W][tileFloats] f32 | vblk[B] f32 | wcnt[2][16] u32 | list[rows*B] u16 | dlist (Q4) f32
template <int FMT, int E, int W>
__host__ __device__ inline uint32_t lds_layout(uint32_t B, uint32_t rowsPerIn, uint32_t* offV, uint32_t* offC,
                                               uint32_t* offL, uint32_t* offD) {
    uint32_t o = (uint32_t)W * Fmt<FMT>::kAcc * E * 64 * 4;
    *offV = o; o += align_up(B * 4, 16);
    *offC = o; o += 2 * 16 * 4;
    *offL = o; o += align_up(rowsPerIn * B * 2, 16);
    *offD = o; if (FMT == kQ4) o += align_up(rowsPerIn * B * 4, 16);
    return o;
}

template <int FMT, int E, int W>
__global__ __launch_bounds__(64 * W) void bucket_mul_kernel(const MulArgs a) {
    constexpr int NACC = Fmt<FMT>::kAcc;
    constexpr int TILE_F = NACC * E * 64;
    using LT = typename LoadT<E>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const MulGeom& g = a.g;
    // XCD-aware id -> (tile, slice): the dispatcher places block b on XCD b%8 (speed only).
    const uint32_t b = blockIdx.x, xcd = b & 7u, k = b >> 3;
    const uint32_t s = (k / g.tiles) * 8u + xcd, t = k % g.tiles;
    if (s >= g.slices) return;

    const int tid = threadIdx.x, lane = tid & 63;
    if (a.tstamp && tid == 0) atomicMin(&a.tstamp[0], (unsigned long long)wall_clock64());
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t B = g.sliceRows;
    uint32_t offV, offC, offL, offD;
    lds_layout<FMT, E, W>(B, g.rowsPerIn, &offV, &offC, &offL, &offD);
    float* acc = reinterpret_cast<float*>(smem);
    float* vblk = reinterpret_cast<float*>(smem + offV);
    uint32_t* wcnt = reinterpret_cast<uint32_t*>(smem + offC);
    uint16_t* list = reinterpret_cast<uint16_t*>(smem + offL);
    float* dlist = reinterpret_cast<float*>(smem + offD);
    float* myacc = acc + wave * TILE_F + lane;

    const uint32_t j0 = s * B;
    const uint32_t nb = min(B, g.inDim - j0);
    const uint32_t e = a.expNo ? a.expNo[0] : 0u;
    const float cutoff = a.cutoff[0];

#pragma unroll
    for (int i = 0; i < NACC * E; i++) myacc[i * 64] = 0.0f;
    for (uint32_t jl = tid; jl < nb; jl += 64 * W) vblk[jl] = a.v[j0 + jl];
    __syncthreads();

    // ---- 1. dispatch: keep test + ordered compaction into LDS ---------------------------------
    const uint32_t NC = g.rowsPerIn * nb;
    uint32_t n = 0;
    int round = 0;
    for (uint32_t c0 = 0; c0 < NC; c0 += 64 * W, round++) {
        const uint32_t c = c0 + tid;
        bool keep = false;
        uint32_t code = 0;
        float dval = 0.0f;
        if (c < NC) {
            if (FMT == kFp16) {
                // bucket row i = rank*inDim + j (rank-major, convert.metal:83-100); v index = i % inDim
                const uint32_t rank = c / nb, jl = c - rank * nb;
                const size_t row = (size_t)e * g.expertRows + (size_t)rank * g.inDim + j0 + jl;
                const float mean = half_bits_to_float(reinterpret_cast<const uint16_t*>(a.stats)[row * 4 + 3]);
                const float x = vblk[jl];
                keep = cutoff < (kCutoffScale * mean) * fabsf(x);         // bucketMul.metal:69
                code = (rank << 12) | jl;
            } else {
                // bucket row i = j*8 + rank (input-major, bucketMulQ4.metal:46); entry value = v*mean (:52)
                const uint32_t jl = c >> 3, rank = c & 7u;
                const size_t row = (size_t)e * g.expertRows + (size_t)(j0 + jl) * 8u + rank;
                const float mean = reinterpret_cast<const float*>(a.stats)[row * 2 + 1];
                const float x = vblk[jl];
                keep = cutoff < (kCutoffScale * mean) * fabsf(x);         // bucketMulQ4.metal:47
                dval = x * mean;
                code = (jl << 3) | rank;
            }
        }
        const unsigned long long m = __ballot(keep);
        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane == 0) wcnt[(round & 1) * 16 + wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int w2 = 0; w2 < W; w2++) {
            const uint32_t cw = wcnt[(round & 1) * 16 + w2];
            woff += (w2 < wave) ? cw : 0u;
            tot += cw;
        }
        if (keep) {
            list[n + woff + pre] = (uint16_t)code;
            if (FMT == kQ4) dlist[n + woff + pre] = dval;
        }
        n += tot;
    }
    __syncthreads();
    if (t == 0 && tid == 0 && n) atomicAdd(a.dispatchCount, n);   // dispatch.size (test hook)

    // ---- 2. stream the kept rows, scatter-accumulate into the private LDS tile ----------------
    const uint32_t col = t * (64u * E) + (uint32_t)lane * E;
    const bool colOK = col < g.cols;
    const uint16_t* __restrict__ wbase = a.buckets + (colOK ? col : 0u);
    const uint32_t myRows = (n > (uint32_t)wave) ? (n - wave + W - 1) / W : 0u;   // entries wave, wave+W, ...

    for (uint32_t i0 = 0; i0 < myRows; i0 += kBatch) {
        // lane u of the wave decodes entry i0+u; the row loop below reads it back with v_readlane.
        // A short last batch re-reads its last valid row (an L1/L2 hit) so that all kBatch loads stay
        // unconditional -- a branch around a load makes hipcc drain vmcnt(0) per load.
        const uint32_t nv = min((uint32_t)kBatch, myRows - i0);
        uint32_t eoff = 0; float dv = 0.0f;
        if (lane < kBatch) {
            const uint32_t kk = wave + W * (i0 + min((uint32_t)lane, nv - 1u));
            const uint32_t code = list[kk];
            uint32_t rowIdx;
            if (FMT == kFp16) { const uint32_t rank = code >> 12, jl = code & 4095u; rowIdx = rank * g.inDim + j0 + jl; dv = vblk[jl]; }
            else { const uint32_t jl = code >> 3, rank = code & 7u; rowIdx = (j0 + jl) * 8u + rank; dv = dlist[kk]; }
            eoff = (e * g.expertRows + rowIdx) * g.cols;
        }
        LT wreg[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            const uint32_t ro = __builtin_amdgcn_readlane(eoff, u);
            wreg[u] = *reinterpret_cast<const LT*>(wbase + (size_t)ro);     // lanes past the last column read column 0
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            const float dd = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dv), u));
            if ((uint32_t)u < nv && colOK) {
#pragma unroll
                for (int j = 0; j < E; j++) {
                    const uint32_t x = word_of(wreg[u], j);
                    if (FMT == kFp16) {
                        // bucketMul.metal:100-106: v = d.x*float(w) with the position bits left in w
                        const float val = dd * half_bits_to_float((uint16_t)x);
                        lds_add(myacc + ((x & 15u) * E + j) * 64, val);
                    } else {
                        // bucketMulQ4.metal:76-81: low nibble first <-> sub-bucket 3,2,1,0
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t nib = (x >> (4 * q)) & 15u;
                            const float val = (nib & 8u) ? -dd : dd;
                            lds_add(myacc + (((3 - q) * 8 + (nib & 7u)) * E + j) * 64, val);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- 3. sum the W private tiles in wave order -> one slab (native [slot][j][lane] order) ---
    float* slab = a.slabs + ((size_t)s * g.tiles + t) * TILE_F;
    for (int o = tid; o < TILE_F; o += 64 * W) {
        float sum = acc[o];
#pragma unroll
        for (int w2 = 1; w2 < W; w2++) sum += acc[w2 * TILE_F + o];
        slab[o] = sum;
    }
    if (a.tstamp && tid == 0) atomicMax(&a.tstamp[1], (unsigned long long)wall_clock64());
}

// integrate: out[(col)*NACC + slot] = sum over slices of slab[slice][tile][slot][j][lane]; Q4 adds the
// outliers of that output afterwards (calcOutliers, bucketMulQ4.metal:13-21) in table order.
template <int FMT, int E>
__global__ __launch_bounds__(256) void integrate_kernel(const float* __restrict__ slabs, const MulGeom g,
                                                        float* __restrict__ out, const OutlierIndex ol,
                                                        const float* __restrict__ v, int hasOutliers,
                                                        unsigned long long* __restrict__ tstamp) {
    constexpr int NACC = Fmt<FMT>::kAcc;
    constexpr int TILE_F = NACC * E * 64;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (tstamp && gid == 0 && tstamp[1] > tstamp[0]) { tstamp[2] += tstamp[1] - tstamp[0]; tstamp[3] += 1; }
    if (gid >= g.tiles * TILE_F) return;
    const uint32_t t = gid / TILE_F, o = gid % TILE_F;
    const uint32_t lane = o & 63u, sj = o >> 6, j = sj % E, slot = sj / E;
    const uint32_t col = t * (64u * E) + lane * E + j;
    if (col >= g.cols) return;
    const float* p = slabs + (size_t)t * TILE_F + o;
    const size_t stride = (size_t)g.tiles * TILE_F;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    uint32_t sl = 0;
    for (; sl + 4 <= g.slices; sl += 4) {
        s0 += p[(size_t)(sl + 0) * stride];
        s1 += p[(size_t)(sl + 1) * stride];
        s2 += p[(size_t)(sl + 2) * stride];
        s3 += p[(size_t)(sl + 3) * stride];
    }
    for (; sl < g.slices; sl++) s0 += p[(size_t)sl * stride];
    float sum = (s0 + s1) + (s2 + s3);
    const uint32_t oi = col * NACC + slot;
    if (FMT == kQ4 && hasOutliers) {
        for (uint32_t q = ol.rowPtr[oi]; q < ol.rowPtr[oi + 1]; q++) sum += v[ol.inIdx[q]] * ol.value[q];
    }
    out[oi] = sum;
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
    const uint32_t grid = a.g.tiles * align_up(a.g.slices, 8);
    hipLaunchKernelGGL((bucket_mul_kernel<FMT, E, W>), dim3(grid), dim3(64 * W), lds, st, a);
    return hipGetLastError();
}

template <int FMT>
static hipError_t launch_mul_fmt(int W, int E, const MulArgs& a, hipStream_t st) {
#define EFFORT_CASE(w, e) if (W == w && E == e) return launch_mul_t<FMT, e, w>(a, st);
    EFFORT_CASE(16, 1) EFFORT_CASE(16, 2) EFFORT_CASE(8, 1) EFFORT_CASE(8, 2) EFFORT_CASE(8, 4)
    EFFORT_CASE(4, 1) EFFORT_CASE(4, 2) EFFORT_CASE(4, 4)
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
    EFFORT_CASE(16, 1) EFFORT_CASE(16, 2) EFFORT_CASE(8, 1) EFFORT_CASE(8, 2) EFFORT_CASE(8, 4)
    EFFORT_CASE(4, 1) EFFORT_CASE(4, 2) EFFORT_CASE(4, 4)
#undef EFFORT_CASE
    return 0;
}

hipError_t launch_integrate(Format fmt, int E, const float* slabs, const MulGeom& g, float* out,
                            const OutlierIndex* ol, const float* v, unsigned long long* tstamp, hipStream_t st) {
    OutlierIndex o = ol ? *ol : OutlierIndex{nullptr, nullptr, nullptr};
    const int has = ol && ol->rowPtr ? 1 : 0;
    const uint32_t total = g.tiles * g.tileFloats;
    const dim3 grid((total + 255) / 256), block(256);
#define EFFORT_CASE(f, e) if (fmt == f && E == e) { hipLaunchKernelGGL((integrate_kernel<f, e>), grid, block, 0, st, slabs, g, out, o, v, has, tstamp); return hipGetLastError(); }
    EFFORT_CASE(kFp16, 1) EFFORT_CASE(kFp16, 2) EFFORT_CASE(kFp16, 4)
    EFFORT_CASE(kQ4, 1) EFFORT_CASE(kQ4, 2) EFFORT_CASE(kQ4, 4)
#undef EFFORT_CASE
    return hipErrorInvalidValue;
}

}  // namespace effort
