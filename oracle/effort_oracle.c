/*
 * effort_oracle.c -- CPU restatement of kolinko/effort's bucketMul hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (effort_amd/, libeffort_hip.so) never links, imports or calls it.
 *
 * Parity status: "parity unpinned" for the FP16 path -- the reference is Swift + Metal
 * and can be neither compiled nor run in this image, and its own tests hold no golden
 * vectors for bucketMul.  Every function below restates one Metal
 * kernel / Swift host routine literally, citing the reference file:line it follows
 * (paths relative to /root/reference).  The Q4 layout/multiply IS pinned: see
 * oracle/q4_layout.py and tests/golden/ (fixtures generated from the importable
 * q4_draft.py).
 *
 * Arithmetic conventions (the reference's MTL_FAST_MATH may reassociate; we take the
 * source literally): left-to-right fp32 products, no FMA contraction (build with
 * -ffp-contract=off), round-to-nearest-even f32->f16 / f32->bf16.
 *
 * Where the reference is non-deterministic (atomic append order in prepareDispatch,
 * simd_sum order, atomic float adds) this file fixes ONE order and says so.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <omp.h>

#define EO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ fp16 / bf16 */

static inline float h2f(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) {
            u = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            man &= 0x3FFu;
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        u = sign | 0x7F800000u | (man << 13);
    } else {
        u = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float f; memcpy(&f, &u, 4); return f;
}

static inline uint16_t f2h(float f) { /* round-to-nearest-even */
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? 0x200u : 0));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);        /* overflows to inf */
    if (x < 0x33000001u) return (uint16_t)sign;                      /* rounds to zero  */
    int32_t e = (int32_t)(x >> 23) - 127;
    uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
    if (e < -14) { /* subnormal half */
        int shift = -14 - e + 13;                                    /* 14..24 */
        uint32_t r = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3FFu);
    uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (uint16_t)(sign | r);
}

static inline float bf16r(float f) { /* f32 -> bfloat (RNE) -> f32 */
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) { u |= 0x00400000u; u &= 0xFFFF0000u; }
    else { u += 0x7FFFu + ((u >> 16) & 1u); u &= 0xFFFF0000u; }
    memcpy(&f, &u, 4); return f;
}

EO_API float eo_h2f(uint16_t h) { return h2f(h); }
EO_API uint16_t eo_f2h(float f) { return f2h(f); }
EO_API float eo_bf16r(float f) { return bf16r(f); }

/* The multiplies convert one half per weight: a 65536-entry table of h2f() (built once, from h2f itself, so every
 * product is bit-identical with calling it) instead of ~15 integer operations per element. */
static float H2F_LUT[65536];
static int h2f_lut_ready = 0;
static void h2f_lut_init(void) {
    if (h2f_lut_ready) return;
    #pragma omp critical(eo_lut)
    {
        if (!h2f_lut_ready) {
            for (uint32_t i = 0; i < 65536u; i++) H2F_LUT[i] = h2f((uint16_t)i);
            #pragma omp flush
            h2f_lut_ready = 1;
        }
    }
}

/* ------------------------------------------------------------------ converter (FP16 layout) */

/* idxsBitonicSortAbs (convert.metal:315-342) driven by Vector.sortAbs (model.swift:660-683):
 * for p in 0..<logn, q in 0...p: one launch of n threads; thread gid with (gid & distance)==0
 * swaps (gid, gid|distance) iff (|f[gid]| < |f[partner]|) == direction, where
 * direction = ((gid >> p) & 2) == 0.  Net effect: descending by |x|; ties land wherever the
 * network puts them, which is what we reproduce.  n must be a power of two. */
static void bitonic_sort_abs(uint16_t* vals, uint16_t* idxs, uint32_t n) {
    int logn = 0; while ((1u << logn) < n) logn++;
    for (int p = 0; p < logn; p++) {
        for (int q = 0; q <= p; q++) {
            uint32_t distance = 1u << (p - q);
            for (uint32_t gid = 0; gid < n; gid++) {
                if (gid & distance) continue;
                uint32_t partner = gid | distance;
                int direction = (((gid >> p) & 2u) == 0);
                /* half abs compare == integer compare of the magnitude bits (no NaNs) */
                int less = (vals[gid] & 0x7FFFu) < (vals[partner] & 0x7FFFu);
                if (less == direction) {
                    uint16_t t = vals[gid]; vals[gid] = vals[partner]; vals[partner] = t;
                    t = idxs[gid]; idxs[gid] = idxs[partner]; idxs[partner] = t;
                }
            }
        }
    }
}

/* Vector.sortAbs (model.swift:660-683): non-power-of-two rows are copied into a zero-filled
 * buffer of size 2^(floor(log2 n)+1), sorted, and the first n entries copied back. */
static void sort_abs_row(uint16_t* vals, uint16_t* idxs, uint32_t n, uint16_t* pv, uint16_t* pi) {
    uint32_t l2 = 0; while ((2u << l2) <= n) l2++;        /* floor(log2 n) */
    if ((1u << l2) == n) { bitonic_sort_abs(vals, idxs, n); return; }
    uint32_t padded = 2u << l2;
    memset(pv, 0, padded * 2); memset(pi, 0, padded * 2);
    memcpy(pv, vals, n * 2); memcpy(pi, idxs, n * 2);
    bitonic_sort_abs(pv, pi, padded);
    memcpy(vals, pv, n * 2); memcpy(idxs, pi, n * 2);
}

static inline uint16_t half_to_ushort(uint16_t h) { /* Metal half -> ushort: truncate toward 0 */
    float f = h2f(h);
    if (!(f > 0.0f)) return 0;
    if (f >= 65535.0f) return 65535;
    return (uint16_t)f;
}

/* preBucketize (convert.metal:43-78) for ONE input row, literal, including the counter kept as
 * a half in slot 0 of each bucket and the unguarded write at slot 1+counter.  bv is the row's
 * zero-initialised [C][bSize+1] scratch (fresh MTLBuffers are zero-filled).  Returns the number
 * of writes that fell outside the row's own scratch (reference UB; dropped here). */
static int prebucketize_row(const uint16_t* wv, const uint16_t* wi, uint16_t* bv,
                            uint32_t outDim, uint32_t bSize) {
    uint32_t C = outDim / bSize, lim = C * (bSize + 1);
    int oob = 0;
    for (uint32_t i = 0; i < outDim; i++) {
        uint16_t val = wv[i], idx = wi[i];
        uint16_t bucket = (uint16_t)(idx / bSize), posId = (uint16_t)(idx % bSize);
        uint16_t mask = (uint16_t)(0xFFFFu ^ (bSize - 1));            /* :66-70 */
        val = (uint16_t)((val & mask) | posId);
        uint16_t bOffset = (uint16_t)(bucket * (bSize + 1));
        uint16_t counter = half_to_ushort(bv[bOffset]);
        uint32_t slot = (uint32_t)bOffset + 1u + counter;
        if (slot < lim) bv[slot] = val; else oob++;
        bv[bOffset] = f2h(h2f(bv[bOffset]) + 1.0f);                    /* half += 1 */
    }
    return oob;
}

/* Sort + preBucketize one already-transposed row; exposed for the docs/bucketmul.html KAT
 * (12 columns, bucket size 4).  ranked[rank*C + b] receives bVals[b][1+rank]. */
EO_API int eo_bucketize_row(const uint16_t* row_vals, uint32_t outDim, uint32_t bSize,
                            uint16_t* ranked /* [bSize][outDim/bSize] */) {
    uint32_t C = outDim / bSize;
    uint16_t* v = (uint16_t*)malloc(outDim * 2), *ix = (uint16_t*)malloc(outDim * 2);
    uint16_t* pv = (uint16_t*)malloc(2 * outDim * 2 + 64), *pi = (uint16_t*)malloc(2 * outDim * 2 + 64);
    uint16_t* bv = (uint16_t*)calloc(C * (bSize + 1), 2);
    memcpy(v, row_vals, outDim * 2);
    for (uint32_t i = 0; i < outDim; i++) ix[i] = (uint16_t)i;
    sort_abs_row(v, ix, outDim, pv, pi);
    int oob = prebucketize_row(v, ix, bv, outDim, bSize);
    for (uint32_t b = 0; b < C; b++)
        for (uint32_t r = 0; r < bSize; r++) ranked[r * C + b] = bv[b * (bSize + 1) + 1 + r];
    free(v); free(ix); free(pv); free(pi); free(bv);
    return oob;
}

/* bucketize() FP16 part (convert.swift:209-260) = getProbes (convert.metal:14-22) ->
 * prepareValsIdxs (:27-41) -> inDim x sortAbs -> preBucketize (:43-78) -> bucketize (:83-100)
 * -> makeStats (:105-119).  W is the HF matrix f16 [outDim, inDim] row-major.
 * Returns <0 on violated preconditions (convert.swift:210-215,239), else the number of
 * out-of-row writes dropped (0 for well-defined inputs). */
EO_API int eo_convert_fp16(const uint16_t* W, uint32_t outDim, uint32_t inDim, uint32_t bSize,
                           uint16_t* buckets /* [inDim*bSize, outDim/bSize] */,
                           uint16_t* stats /* [inDim*bSize, 4] */,
                           uint16_t* probes /* [4096] */) {
    if (!(outDim >= 4096 || (outDim > 0 && 4096 % outDim == 0))) return -1;
    if (inDim < 4096) return -2;
    if (outDim > 32000 || inDim > 32000) return -3;
    if (bSize == 0 || outDim % bSize != 0) return -4;
    uint32_t C = outDim / bSize;

    /* getProbes: probes[id*rep+i] = w[id+i + id*cols] */
    uint32_t rep = outDim >= 4096 ? 1 : 4096 / outDim;
    for (uint32_t id = 0; id < 4096 / rep; id++)
        for (uint32_t i = 0; i < rep; i++) probes[id * rep + i] = W[(size_t)id * inDim + id + i];

    int oob_total = 0;
    #pragma omp parallel reduction(+:oob_total)
    {
        uint16_t* v = (uint16_t*)malloc(outDim * 2), *ix = (uint16_t*)malloc(outDim * 2);
        uint16_t* pv = (uint16_t*)malloc(2 * outDim * 2 + 64), *pi = (uint16_t*)malloc(2 * outDim * 2 + 64);
        uint16_t* bv = (uint16_t*)malloc((size_t)C * (bSize + 1) * 2);
        #pragma omp for schedule(dynamic, 16)
        for (uint32_t row = 0; row < inDim; row++) {
            /* prepareValsIdxs: vals[i*srcRows + rowId] = w[rowId*srcCols + i]; idxs = rowId */
            for (uint32_t r = 0; r < outDim; r++) { v[r] = W[(size_t)r * inDim + row]; ix[r] = (uint16_t)r; }
            sort_abs_row(v, ix, outDim, pv, pi);
            memset(bv, 0, (size_t)C * (bSize + 1) * 2);
            oob_total += prebucketize_row(v, ix, bv, outDim, bSize);
            /* bucketize: buckets[bucketNo + rowId*C + i*inDim*C] = bVals[row][bucketNo][1+i] */
            for (uint32_t b = 0; b < C; b++)
                for (uint32_t i = 0; i < bSize; i++)
                    buckets[((size_t)i * inDim + row) * C + b] = bv[b * (bSize + 1) + 1 + i];
        }
        free(v); free(ix); free(pv); free(pi); free(bv);
    }
    /* makeStats: f32 sequential sum of |half| over the bucket row, / bCols, stored to all 4 lanes */
    #pragma omp parallel for schedule(static)
    for (uint32_t r = 0; r < inDim * bSize; r++) {
        float sum = 0;
        for (uint32_t i = 0; i < C; i++) sum += fabsf(h2f(buckets[(size_t)r * C + i]));
        uint16_t m = f2h(sum / (float)C);
        stats[r * 4 + 0] = m; stats[r * 4 + 1] = m; stats[r * 4 + 2] = m; stats[r * 4 + 3] = m;
    }
    return oob_total;
}

/* ------------------------------------------------------------------ findCutoff32 */

/* findCutoff32 (bucketMul.metal:141-247), launched with 1024 threads in one threadgroup
 * (bucketMul.swift:41): thread id owns values 4*id..4*id+3; simdgroup g = 32 threads = 128
 * consecutive values.  q is the Swift-side Int(Double(4095)*(1-effort)) (bucketMul.swift:39).
 * threadgroup initialisers are taken at face value (maxCount = 0). */
EO_API void eo_find_cutoff(const float* v, const uint16_t* probes, uint32_t expNo, uint32_t q,
                           float* cutoff_out, int* loops_out) {
    const float CUTOFF_SCALE = 100000.0f;                       /* int 100000 * float -> float */
    uint32_t effort = 4096u - q;                                /* :154 */
    float lvals[4096];
    float tgMin[32], tgMax[32];
    for (int g = 0; g < 32; g++) {
        float sgMin = 999.0f, sgMax = -999.0f;                  /* :155-156, reduced :166-167 */
        for (int k = 0; k < 128; k++) {
            int j = g * 128 + k;
            float t = CUTOFF_SCALE * v[j];                      /* :160, left to right */
            float u = t * bf16r(h2f(probes[j + expNo * 4096u]));
            float a = bf16r(fabsf(u));
            lvals[j] = a;
            sgMax = fmaxf(sgMax, a); sgMin = fminf(sgMin, a);
        }
        tgMin[g] = bf16r(sgMin); tgMax[g] = bf16r(sgMax);       /* :179-180, stored as bfloat */
    }
    float minBound = tgMin[0], maxBound = tgMax[0];
    for (int g = 1; g < 32; g++) { minBound = fminf(minBound, tgMin[g]); maxBound = fmaxf(maxBound, tgMax[g]); }
    float newBound = (minBound + maxBound) / 2;                 /* :195 */
    int loops = 0; int minCount = 4096, maxCount = 0;           /* :175-176,198 */
    for (;;) {
        loops += 1;
        uint32_t countAbove = 0;
        #pragma omp simd reduction(+:countAbove)
        for (int j = 0; j < 4096; j++) countAbove += (lvals[j] > newBound) ? 1u : 0u;   /* :204-212 (one thread, vectorised: 4096 compares a round) */
        if (countAbove < effort) { maxBound = newBound; maxCount = (int)countAbove; }
        else { minBound = newBound; minCount = (int)countAbove; }
        newBound = (maxBound + minBound) / 2;                   /* :222 */
        if (countAbove == effort || (maxBound - minBound < 0.00001f) || (abs(maxCount - minCount) < 3)) break;
        if (loops > 100) break;                                 /* :236 */
    }
    *cutoff_out = newBound;
    if (loops_out) *loops_out = loops;
}

/* bucketMul.swift:39 */
EO_API uint32_t eo_effort_to_q(double effort) { return (uint32_t)(int)((double)(4096 - 1) * (1 - effort)); }

/* ------------------------------------------------------------------ dispatch */

/* prepareDispatch (bucketMul.metal:47-79) with chunkSize 4, nThreads = stats.rows/4
 * (bucketMul.swift:43-45).  The reference appends via atomic_fetch_add, i.e. in
 * non-deterministic order; the oracle fixes ascending bucket-row order.
 * dispatch is float2[] = {v[i % rowsCount], float(i*colsCount)}.  Returns the count. */
EO_API uint32_t eo_prepare_dispatch(const float* v, const uint16_t* stats_h4, uint32_t expNo, float cutoff,
                                    uint32_t statsRows, uint32_t rowsCount, uint32_t colsCount,
                                    uint32_t expertSize, float* dispatch) {
    /* two passes over fixed chunks of rows (count, then write at the chunk's prefix): the list comes out in ascending
     * bucket-row order whatever the thread count */
    enum { CH = 256 };
    uint32_t off = expertSize * expNo, cnt[CH + 1];
    uint32_t per = (statsRows + CH - 1) / CH;
    h2f_lut_init();
    #pragma omp parallel for schedule(static)
    for (int c = 0; c < CH; c++) {
        uint32_t b = off + (uint32_t)c * per, e = b + per, k = 0;
        if (e > off + statsRows) e = off + statsRows;
        for (uint32_t i = b; i < e; i++)
            k += (cutoff < 100000.0f * H2F_LUT[stats_h4[(size_t)i * 4 + 3]] * fabsf(v[i % rowsCount])) ? 1u : 0u;   /* :69 */
        cnt[c + 1] = k;
    }
    cnt[0] = 0;
    for (int c = 0; c < CH; c++) cnt[c + 1] += cnt[c];
    #pragma omp parallel for schedule(static)
    for (int c = 0; c < CH; c++) {
        uint32_t b = off + (uint32_t)c * per, e = b + per, n = cnt[c];
        if (e > off + statsRows) e = off + statsRows;
        for (uint32_t i = b; i < e; i++) {
            float val = v[i % rowsCount];
            if (cutoff < 100000.0f * H2F_LUT[stats_h4[(size_t)i * 4 + 3]] * fabsf(val)) {
                dispatch[2 * n] = val; dispatch[2 * n + 1] = (float)(uint32_t)(i * colsCount); n++;
            }
        }
    }
    return cnt[CH];
}

/* prepareDispatchQ4 (bucketMulQ4.metal:25-57), single thread (bucketMulQ4.swift:44-47):
 * stats are float2 (uses .y), val = v[i/8], entry = {val*mean, float(i*colsCount)}. */
EO_API uint32_t eo_prepare_dispatch_q4(const float* v, const float* stats_f2, uint32_t expNo, float cutoff,
                                       uint32_t statsRows, uint32_t colsCount, uint32_t expertSize,
                                       float* dispatch) {
    uint32_t n = 0, off = expertSize * expNo;
    for (uint32_t i = off; i < off + statsRows; i++) {
        float s1 = stats_f2[(size_t)i * 2 + 1];
        float val = v[i / 8];
        if (cutoff < 100000.0f * s1 * fabsf(val)) {
            dispatch[2 * n] = val * s1; dispatch[2 * n + 1] = (float)(uint32_t)(i * colsCount); n++;
        }
    }
    return n;
}

/* roundUp (bucketMul.metal:22-31) + zeroRange32 (:11-20) as deployed in bucketMul.swift:57-58:
 * size = (1 + size/2048)*2048 (always grows), new tail entries = {0,0}; zeroRange32 runs 2048
 * threads so it covers the whole added range. */
EO_API uint32_t eo_round_up_pad(float* dispatch, uint32_t size) {
    uint32_t prev = size, ns = (1u + size / 2048u) * 2048u;
    for (uint32_t id = 0; id < 2048; id++) { uint32_t p = prev + id; if (p < ns) { dispatch[2 * p] = 0; dispatch[2 * p + 1] = 0; } }
    return ns;
}

/* ------------------------------------------------------------------ bucketMul / integrate */

/* bucketMul (bucketMul.metal:83-117), grid [cols, groups]: thread (x,y) walks dispatch slice
 * y*D/groups .. in order, w = weights[int(d.y)+x], acc[bits(w)&15] += d.x*float(w) (the
 * position bits stay in the multiplied value); writes tmp[y*16384 + x*16 + i].
 * The reference's 16-way select adds +0 to the 15 other slots, which never changes them. */
EO_API void eo_bucket_mul(const uint16_t* weights, const float* dispatch, uint32_t dispatchSize,
                          uint32_t cols, uint32_t groups, float* tmp /* [groups][16384] */) {
    /* A task = (group y, block of XB columns); inside it the dispatch slice is walked row by row and every column of the
     * block adds its product: per thread (x, y) of the reference that is the same sequence of additions, r ascending, so
     * the sums are bit-identical with the literal loop nest -- but a bucket row is read as a contiguous piece instead of one
     * half per 64-byte line (the column-wise walk ran at 1.8 GB/s on 256 cores). */
    /* column blocks as wide as the thread count allows (about four tasks per thread): a block is a contiguous run of each row,
     * and narrow blocks turn the walk into one cache line per row again once the matrices stop fitting the last-level cache */
    enum { XBMAX = 1024 };
    int threads = omp_get_max_threads();
    uint32_t want = (uint32_t)((4 * threads + (int)groups - 1) / (int)groups);
    uint32_t nblk = want < 1 ? 1 : want;
    if (nblk > (cols + 31) / 32) nblk = (cols + 31) / 32;
    uint32_t XB = (cols + nblk - 1) / nblk;
    if (XB > XBMAX) { XB = XBMAX; }
    nblk = (cols + XB - 1) / XB;
    uint32_t per = dispatchSize / groups;
    h2f_lut_init();
    #pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (uint32_t y = 0; y < groups; y++) {
        for (uint32_t xb = 0; xb < nblk; xb++) {
            float acc[XBMAX][16];
            uint32_t x0 = xb * XB, nx = cols - x0 < XB ? cols - x0 : XB;
            memset(acc, 0, (size_t)nx * 16 * sizeof(float));
            uint32_t rowOffset = y * dispatchSize / groups;
            for (uint32_t r = 0; r < per; r++) {
                float d0 = dispatch[2 * (size_t)(rowOffset + r)], d1 = dispatch[2 * (size_t)(rowOffset + r) + 1];
                const uint16_t* row = weights + (size_t)(int)d1 + x0;
                for (uint32_t x = 0; x < nx; x++) {
                    uint16_t w = row[x];
                    acc[x][w & 15u] += d0 * H2F_LUT[w];
                }
            }
            for (uint32_t x = 0; x < nx; x++)
                for (int i = 0; i < 16; i++) tmp[(size_t)y * 16384 + (x0 + x) * 16 + i] = acc[x][i];
        }
    }
}

/* bucketIntegrate (bucketMul.metal:122-137): out[i] = simd_sum over the 32 groups.  simd_sum's
 * order is unspecified; the oracle uses the xor-butterfly tree (16,8,4,2,1). */
EO_API void eo_bucket_integrate(const float* tmp, float* out, uint32_t outDim) {
    #pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < outDim; i++) {
        float s[32];
        for (int l = 0; l < 32; l++) s[l] = tmp[i + (size_t)l * 16384];
        for (int d = 16; d >= 1; d >>= 1) for (int l = 0; l < d; l++) s[l] = s[l] + s[l + d];
        out[i] = s[0];
    }
}

/* bucketMulQ4 (bucketMulQ4.metal:61-92): ushort w holds 4 nibbles; the loop runs i = 3..0 taking
 * the LOW nibble first, acc[(w&7) + i*8] += (w&8) ? -d.x : d.x; then atomically adds the 32
 * partials into out[x*32 + k] (order across groups unspecified; the oracle adds y = 0..31).
 * out must be pre-zeroed by the caller (expertMul.swift:27). */
EO_API void eo_bucket_mul_q4(const uint16_t* weights, const float* dispatch, uint32_t dispatchSize,
                             uint32_t cols, uint32_t groups, float* out) {
    /* (column blocks in parallel; per column the groups y = 0..31 in order, per group the rows in order: the literal sums) */
    enum { XB = 16 };
    uint32_t per = dispatchSize / groups, nblk = (cols + XB - 1) / XB;
    #pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t xb = 0; xb < nblk; xb++) {
        uint32_t x0 = xb * XB, nx = cols - x0 < XB ? cols - x0 : XB;
        for (uint32_t y = 0; y < groups; y++) {
            float acc[XB][32];
            memset(acc, 0, sizeof(acc));
            uint32_t rowOffset = y * dispatchSize / groups;
            for (uint32_t r = 0; r < per; r++) {
                float d0 = dispatch[2 * (size_t)(rowOffset + r)], d1 = dispatch[2 * (size_t)(rowOffset + r) + 1];
                const uint16_t* row = weights + (size_t)(int)d1 + x0;
                for (uint32_t x = 0; x < nx; x++) {
                    uint16_t w = row[x];
                    for (int i = 3; i >= 0; i--) {
                        float val = (w & 8u) ? -d0 : d0;
                        acc[x][(w & 7u) + i * 8] += val;
                        w >>= 4;
                    }
                }
            }
            for (uint32_t x = 0; x < nx; x++)
                for (int i = 0; i < 32; i++) out[(x0 + x) * 32 + i] += acc[x][i];
        }
    }
}

/* calcOutliers (bucketMulQ4.metal:13-21): out[uint(o.z)] += v[uint(o.y)] * o.x, one atomic per
 * outlier (order unspecified; the oracle goes in table order). */
EO_API void eo_calc_outliers(const float* v, const float* outliers_f4, uint64_t n, float* out) {
    for (uint64_t k = 0; k < n; k++) {
        const float* o = outliers_f4 + 4 * k;
        out[(uint32_t)o[2]] += v[(uint32_t)o[1]] * o[0];
    }
}

/* ------------------------------------------------------------------ full calls (host orchestration) */

/* BucketMul.fullMul (bucketMul.swift:54-70) = calcDispatch (:34-47) -> roundUp/zeroRange32 ->
 * mul (:72-88).  dispatch scratch must hold 2*(statsRows+2048) floats; tmp 32*16384 floats.
 * Returns the dispatch count before padding. */
EO_API int64_t eo_bucketmul_full(const float* v, const uint16_t* buckets, const uint16_t* stats_h4,
                                 const uint16_t* probes, uint32_t expNo, double effort,
                                 uint32_t inDim, uint32_t outDim, uint32_t percentLoad,
                                 float* out, float* dispatch, float* tmp, float* cutoff_out) {
    if (outDim % 16 || (outDim / 16) % 4 || outDim > 16384) return -1;      /* bucketMul.swift:73-76, :52 */
    uint32_t cols = outDim / 16, statsRows = inDim * percentLoad, expertSize = percentLoad * inDim;
    float cutoff; eo_find_cutoff(v, probes, expNo, eo_effort_to_q(effort), &cutoff, 0);
    uint32_t n = eo_prepare_dispatch(v, stats_h4, expNo, cutoff, statsRows, inDim, cols, expertSize, dispatch);
    uint32_t D = eo_round_up_pad(dispatch, n);
    eo_bucket_mul(buckets, dispatch, D, cols, 32, tmp);
    eo_bucket_integrate(tmp, out, outDim);
    if (cutoff_out) *cutoff_out = cutoff;
    return n;
}

/* expertMul Q4 branch (expertMul.swift:25-28) + BucketMulQ4.fullMul (bucketMulQ4.swift:54-63). */
EO_API int64_t eo_bucketmul_q4_full(const float* v, const uint16_t* buckets, const float* stats_f2,
                                    const uint16_t* probes, const float* outliers_f4, uint64_t nOutliers,
                                    uint32_t expNo, double effort, uint32_t inDim, uint32_t outDim,
                                    float* out, float* dispatch, float* cutoff_out) {
    if (outDim % 32 || (outDim / 16) % 4) return -1;
    uint32_t cols = outDim / 32, statsRows = inDim * 8, expertSize = 8 * inDim;
    memset(out, 0, (size_t)outDim * 4);                                    /* out.zero() */
    float cutoff; eo_find_cutoff(v, probes, expNo, eo_effort_to_q(effort), &cutoff, 0);
    uint32_t n = eo_prepare_dispatch_q4(v, stats_f2, expNo, cutoff, statsRows, cols, expertSize, dispatch);
    uint32_t D = eo_round_up_pad(dispatch, n);
    eo_bucket_mul_q4(buckets, dispatch, D, cols, 32, out);
    if (outliers_f4 && nOutliers) eo_calc_outliers(v, outliers_f4, nOutliers, out);
    if (cutoff_out) *cutoff_out = cutoff;
    return n;
}

/* ------------------------------------------------------------------ dense baseline + metric */

/* basicMul (matrix.metal:150-162): out[row] = sum_i v[i]*float(m[row][i]), f32 sequential.
 * round_v_to_f16 != 0 applies helpers/mps.swift:19's v.asFloat16() first (what the MPS path sees). */
EO_API void eo_dense_gemv(const uint16_t* W, const float* v, float* out, uint32_t outDim, uint32_t inDim,
                          int round_v_to_f16) {
    float* vv = (float*)malloc((size_t)inDim * 4);
    for (uint32_t i = 0; i < inDim; i++) vv[i] = round_v_to_f16 ? h2f(f2h(v[i])) : v[i];
    #pragma omp parallel for schedule(static)
    for (uint32_t r = 0; r < outDim; r++) {
        float sum = 0; const uint16_t* m = W + (size_t)r * inDim;
        for (uint32_t i = 0; i < inDim; i++) sum += vv[i] * h2f(m[i]);
        out[r] = sum;
    }
    free(vv);
}

/* cosineSimilarityTo (model.swift:511-519; aux.metal:293-312): dot/(sqrt(|a|^2)*sqrt(|b|^2)), f32
 * accumulation (atomic order unspecified; the oracle is sequential). */
EO_API float eo_cosine(const float* a, const float* b, uint32_t n) {
    float dot = 0, ma = 0, mb = 0;
    for (uint32_t i = 0; i < n; i++) { dot += a[i] * b[i]; ma += a[i] * a[i]; mb += b[i] * b[i]; }
    return dot / (sqrtf(ma) * sqrtf(mb));
}
