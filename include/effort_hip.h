/*
 * effort_hip.h -- C ABI of the MI355X (gfx950) implementation of Effort's bucketMul hot path.
 *
 * This is the drop-in boundary: every entry point below replaces one piece of the reference's
 * Swift/Metal interface for the path (file:line relative to kolinko/effort @ 2024_08_07).  A
 * Swift shim that keeps `bucketMul(v:by:expNo:out:effort:)` / `bucketMulQ4(...)` and forwards
 * here is shown in INTEGRATION.md.  Plain pointers and sizes only; all `*_dev` pointers are
 * device (HBM) addresses owned by the caller.  Nothing here depends on PyTorch.
 *
 * Conventions (reference: helpers/gpu.swift:109-196)
 *   - Calls ENQUEUE work on the context's HIP stream and return; results are valid after
 *     effort_sync() (= gpu.eval(), helpers/gpu.swift:109-119) or any later work on that stream.
 *   - The reference aborts on violated preconditions (assert/precondition); this ABI returns a
 *     negative error code instead and enqueues nothing.
 *   - One effort_ctx per host thread / stream (the reference's BucketMul.shared singleton,
 *     bucketMul.swift:24, is not re-entrant; a context is its re-entrant equivalent and owns the
 *     same scratch: cutoff, dispatch counter, partial tiles).
 */
#ifndef EFFORT_HIP_H
#define EFFORT_HIP_H

#include <stdint.h>

#if defined(__GNUC__)
#define EFFORT_API __attribute__((visibility("default")))
#else
#define EFFORT_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct effort_ctx effort_ctx;
typedef struct effort_w effort_w;      /* = class ExpertWeights, loader.swift:46-167 */

enum {
    EFFORT_OK = 0,
    EFFORT_ERR_ARG = -1,          /* null pointer / bad handle                                        */
    EFFORT_ERR_SHAPE = -2,        /* inDim >= 4096, outDim % 32 == 0, outDim <= 16384 (bucketMul.swift:36,52,73) */
    EFFORT_ERR_EFFORT = -3,       /* effort outside [0,1]                                              */
    EFFORT_ERR_HIP = -4,          /* a HIP runtime call failed (see effort_last_error)                 */
    EFFORT_ERR_KIND = -5,         /* FP16 weights passed to the Q4 call or vice versa                  */
    EFFORT_ERR_CONVERT = -6,      /* bucketize() preconditions (convert.swift:210-215,239)             */
    EFFORT_ERR_BLAS = -7,         /* rocBLAS failure in effort_dense_gemv                              */
    EFFORT_ERR_COMM = -8          /* an RCCL call failed (see effort_last_error)                       */
};

/* ---- context = class Gpu + the BucketMul / BucketMulQ4 singletons ------------------------------ */

/* helpers/gpu.swift:36-60 (device, queue) + bucketMul.swift:24-32 (scratch).  `stream` is a
 * hipStream_t (NULL = the default stream).  Returns NULL on failure. */
EFFORT_API effort_ctx* effort_create(int device, void* stream);
EFFORT_API void effort_destroy(effort_ctx* ctx);
EFFORT_API int effort_set_stream(effort_ctx* ctx, void* stream);
/* gpu.eval() -- helpers/gpu.swift:109-119: block until everything enqueued so far has finished. */
EFFORT_API int effort_sync(effort_ctx* ctx);
/* Overlap of independent launches (the reference's command queue lets independent kernels overlap too, helpers/gpu.swift:
 * 135-196).  lanes = 1 (default): every call is enqueued on the context's stream, in order.  lanes = 2..4: the context owns
 * that many internal streams, each with its own scratch, and every effort_bucketmul* call goes to one of them, ordered
 *   - after everything enqueued ON THE CONTEXT'S STREAM before the call -- exactly that: outside a capture the lane forks from the
 *     stream only if the stream still has work (hipStreamQuery); when it answers "idle" everything enqueued on it has completed and
 *     no edge is recorded.  Work the caller keeps on OTHER streams is ordered against a multiply only through the context's stream:
 *     make that stream wait for it (hipStreamWaitEvent) before the call -- a wait enqueued on the stream keeps it "busy" until
 *     the event fires, so the fork happens -- and do not rely on an earlier, already completed wait.  Inside a capture the fork
 *     is always a graph edge; a lane's completion event is recorded lazily, when something first waits for the lane (a join, a
 *     dependent launch on another lane), which may be inside a capture that began after the lane's launches: the event then becomes a
 *     node of that graph and orders the graph's work after launches enqueued BEFORE the capture (a join at the top of the
 *     capture is the explicit form) -- and
 *   - after earlier multiplies of this context whose outputs it reads or overwrites, or whose inputs it overwrites (address
 *     ranges of v / expNo / aux / resid / out are compared);
 * otherwise it runs beside the multiplies still in flight: the head of one launch (staging, cutoffs, selection: HBM idle)
 * hides under the streaming of the others (one 32-call launch after another, each on its own matrices: 0.60 -> 0.70 of the
 * HBM roofline).  Results are bit-identical.  The multiplies become visible to the context's stream at effort_join (enqueues waits, returns at
 * once), and implicitly at effort_sync, at any other effort_* call on the context and at the test hooks.  A caller who
 * enqueues its OWN work on the stream to consume an output calls effort_join first; inside a hipGraph capture call
 * effort_join before ending the capture (the lanes fork from and must rejoin the capturing stream).
 * Costs 64 MiB of scratch per extra lane.
 * Runtime note (HIP 7.0.x as bundled by torch 2.10+rocm7.0, found in round 6): DESTROYING a hipGraph that was captured across several
 * streams -- which every capture of a context with lanes > 1 is -- corrupts the HIP runtime's heap (reproducible with plain torch ops,
 * tools/lab/graph_event_repro.py); keep such graphs for the life of the process, or capture with lanes = 1. */
EFFORT_API int effort_set_overlap(effort_ctx* ctx, int lanes);
EFFORT_API int effort_join(effort_ctx* ctx);
/* Cache policy of the bucket-row stream.  reuse = 0 (default): the kept rows are read NON-TEMPORALLY -- a row is read once per call and a
 * model's weights are tens of times the chip's caches (Mistral-7B at 25 % effort reads 3.5 GB per token against 32 MB of L2 and 256 MB of
 * Infinity Cache), so the stream should not push the row means, the partial tiles and the other launches' lines out on its way through
 * (one 32-call launch 160 -> 150 us, four in flight 133 -> 127, a lone call 18.1 -> 17.75: DESIGN.md 4.1).  reuse = 1: the ordinary policy,
 * for a caller whose launches in flight read the SAME matrices within a few hundred MB of one another (a batch of inputs on one set of
 * weights, through effort_set_overlap lanes or several contexts): the later launches then hit in the Infinity Cache (four launches in flight
 * on one set of 32 matrices: 117 us per launch against 130).  Speed only: results are bit-identical either way.  Takes effect with the next
 * launch (a captured launch keeps the policy it was captured with).  The second policy lives in the kernels that serve GROUP launches (persistent grids,
 * and plain grids of 256-column tiles): lone calls, pairs and plain launches of narrower tiles ignore the hint (two copies of the loop cost them 0.1-0.3 us
 * per call on the default path). */
EFFORT_API int effort_set_row_reuse(effort_ctx* ctx, int reuse);
/* Text of the context's last error; with ctx == NULL: why the last effort_create returned NULL ("null context" if none did). */
EFFORT_API const char* effort_last_error(effort_ctx* ctx);
EFFORT_API const char* effort_version(void);

/* ---- weights = ExpertWeights ------------------------------------------------------------------- */

/* FP16 bundle (loader.swift:46-167, layout written by convert.swift:209-260):
 *   buckets f16 [numExperts][inDim*percentLoad][outDim/16]   rank-major bucket rows, low 4 mantissa
 *                                                            bits of each weight = output position
 *   stats   f16 [numExperts][inDim*percentLoad][4]           mean|row| in all four lanes (.w is read)
 *   probes  f16 [numExperts][4096]
 * percentLoad (1..16) = rank slices present per expert (loader.swift:50, expertSize = percentLoad*inDim).
 * Pointers are BORROWED: the caller keeps the buffers alive while the handle is in use.  Registration reads the
 * buckets once (max |w| of every bucket row, which bounds the multiply's fixed-point accumulators): fill the
 * buffers BEFORE registering, and register again if their contents change. */
EFFORT_API effort_w* effort_weights_fp16(effort_ctx* ctx, const void* buckets_dev, const void* stats_dev,
                              const void* probes_dev, int inDim, int outDim, int percentLoad,
                              int numExperts);

/* The same with bucket rows `rowPitchBytes` apart (0 = 2*outDim/16, the reference's dense layout; otherwise a multiple of 4
 * bytes >= that: rows are read as dwords.  The CONVERTER is stricter -- effort_convert_fp16_pitched writes pitches that are
 * multiples of 8 -- and effort_aligned_row_pitch returns multiples of 128; one shape rule everywhere: outDim % 32 == 0).
 * The reference's rows are 2*cols bytes apart (1376 for 11008 outputs): the 512-byte row pieces the multiply streams then
 * straddle 128-byte lines and HBM delivers 5.4-5.6 TB/s where line-aligned rows reach 6.1-6.9.  A loader that places the rows
 * effort_aligned_row_pitch(outDim) bytes apart (effort_convert_fp16_pitched writes them so) gets the fast stream with no
 * second copy of the buckets (+2.3 % bytes for 11008 outputs).  Results are bit-identical for any pitch. */
EFFORT_API effort_w* effort_weights_fp16_pitched(effort_ctx* ctx, const void* buckets_dev, int rowPitchBytes, const void* stats_dev,
                                      const void* probes_dev, int inDim, int outDim, int percentLoad, int numExperts);
/* 2*outDim/16 rounded up to whole 128-byte lines; EFFORT_ERR_SHAPE for an outDim registration would refuse (outDim % 32, <= 16384). */
EFFORT_API int effort_aligned_row_pitch(int outDim);

/* Q4 bundle (layout written by q4_draft.py:70-322; loaded by loader.swift:70,98,124):
 *   buckets  u16 [numExperts][inDim*8][outDim/32]   4 nibbles per word, nibble = sign<<3 | pos,
 *                                                   bucket row i = inRow*8 + rank (input-major)
 *   stats    f32 [numExperts][inDim*8][2]           (mean|row|, same); .y is read
 *   probes   f16 [numExperts][4096]
 *   outliers f32 [nOutliers][4]                     (value, inIdx, outIdx, 0); may be NULL / 0
 * Borrowed like the FP16 bundle, except the outlier table: registration turns it into an index in HBM of FOUR bytes per
 * outlier (f16 value | output within a block of 2^(16 - bits of inDim) outputs | input: hence inDim, outDim <= 65536 when there
 * are outliers) and does not read outliers_dev again.  Every entry must name an element of this matrix and carry a value that
 * is an f16 number (the table comes from an f16 matrix, q4_draft.py:58-67): otherwise NULL is returned (EFFORT_ERR_ARG). */
EFFORT_API effort_w* effort_weights_q4(effort_ctx* ctx, const void* buckets_dev, const void* stats_dev,
                            const void* probes_dev, const void* outliers_dev, int64_t nOutliers,
                            int inDim, int outDim, int numExperts);
EFFORT_API void effort_weights_free(effort_w* w);
/* The multiply accumulates in fixed point; its scale comes from a per-expert bound on the weights (sum over ranks of the
 * rank's largest |w|; Q4: largest row mean) read at registration.  If the borrowed buffers are REWRITTEN afterwards (the
 * reference's loader.swift buffers are mutable) call effort_weights_refresh before the next multiply: with a stale bound
 * the sums of larger weights wrap.  get/set copy the bound ([numExperts] floats, host memory): a column shard of a
 * multi-GPU split can take the full matrix's bound, so that every rank rounds its products on the same grid.
 * FP16 handles also keep a compact copy of the row means (stats lane .w, 2 bytes per bucket row: +0.15 % of the buckets'
 * size) that big launches stage instead of the 8-byte stats; effort_weights_refresh re-reads it with the bound. */
EFFORT_API int effort_weights_refresh(effort_w* w);
/* Row pitch.  The converter's bucket rows are 2*cols bytes apart (1376 for 11008 outputs), so the row pieces the multiply
 * streams straddle 128-byte lines and HBM delivers 5.4-5.6 TB/s instead of 6.1-6.9 (measured on line-aligned shapes).
 * effort_weights_align_rows gives a handle registered on the dense layout its OWN device copy of the buckets with every row
 * on a 128-byte boundary and reads that from then on (bit-identical results; the caller's buffer is no longer read by
 * multiplies and may be freed -- keep it if effort_weights_refresh will be needed).  Call it BEFORE capturing launches of
 * the handle into a hipGraph (a captured launch keeps the pointers it was given).  No-op if the pitch is aligned already
 * (buckets converted or loaded with effort_aligned_row_pitch: no second copy).  effort_weights_row_pitch: bytes. */
EFFORT_API int effort_weights_align_rows(effort_w* w);
EFFORT_API int effort_weights_row_pitch(const effort_w* w);
EFFORT_API int effort_weights_get_bound(effort_w* w, float* host_out);
EFFORT_API int effort_weights_set_bound(effort_w* w, const float* host_in);

/* ---- the hot path ------------------------------------------------------------------------------ */

/* func bucketMul(v:by:expNo:out:effort:) -- bucketMul.swift:11-15 -> BucketMul.fullMul (:54-70):
 * findCutoff32 + prepareDispatch + roundUp/zeroRange32 + bucketMul + bucketIntegrate
 * (bucketMul.metal:11-247).  v_dev f32[inDim], out_dev f32[outDim] (fully overwritten),
 * expNo_dev = device u32 expert index (NULL = expert 0; expertMul.swift:18-22), effort in [0,1].
 * Row selection (cutoff, dispatch set) is bit-exact with the reference's arithmetic.  Products are accumulated in
 * per-workgroup fixed point: order-free, hence bit-identical run to run; within 2e-5 * max|out| of
 * an f32 accumulation (measured 1-4e-6). */
EFFORT_API int effort_bucketmul(effort_ctx* ctx, const effort_w* w, const float* v_dev, const uint32_t* expNo_dev,
                     float* out_dev, double effort);

/* func bucketMulQ4(...) -- bucketMulQ4.swift:11-17 -> fullMul (:54-63), INCLUDING the caller's
 * out.zero() (expertMul.swift:27) and the calcOutliers pass (bucketMulQ4.metal:13-21).
 * Determinism: the bucket rows are accumulated in fixed point like effort_bucketmul's (order-free).  The OUTLIER sums are f32, as the
 * reference's float atomics are: an output's entries are added in the table's order, and an item whose share of the outputs is thin
 * splits them among the waves of its workgroup, the partial sums added in wave order.  How they are split follows the launch geometry
 * (group size, the device's CU count, a column shard against the full handle), so a bundle WITH outliers gives bit-identical results
 * run to run for one geometry, and results that may differ in the last bits between a lone call, a grouped call and a column shard
 * (all within the 2e-5 * max|out| bar).  Without outliers Q4 is as order-free as FP16. */
EFFORT_API int effort_bucketmul_q4(effort_ctx* ctx, const effort_w* w, const float* v_dev, const uint32_t* expNo_dev,
                        float* out_dev, double effort);

/* func basicMul(v:by:out:) -- helpers/mps.swift:14-47: dense out = W * f16(v), W f16 [outDim,inDim]
 * row-major, f32 result; the "100 % effort dense" baseline.  A streaming HIP kernel (csrc/gemv.hip) by default;
 * effort_set_dense_backend(ctx, 1) routes it through rocBLAS' hssgemv instead (the library the north star names: the
 * bench reports both). */
EFFORT_API int effort_dense_gemv(effort_ctx* ctx, const void* W_f16_dev, const float* v_dev, float* out_dev,
                      int inDim, int outDim);
EFFORT_API int effort_set_dense_backend(effort_ctx* ctx, int rocblas);

/* A GROUP of n (1..32) independent bucketMul calls in ONE kernel launch: call i multiplies vs[i] by ws[i] at
 * efforts[i] into outs[i] (expNos may be NULL, or hold NULL entries = expert 0).  Same row selection as n
 * effort_bucketmul calls, exactly; outputs equal up to the f32 rounding of the per-slice partial sums (the slicing
 * depends on how many calls share the launch; bit for bit when it is pinned with effort_hip_debug.h's effort_set_tuning).  The point is
 * throughput: the decode loop issues such groups back to back on unchanged
 * input -- Wq|Wk|Wv (runNetwork.swift:132-134) and W1|W3 (runNetwork.swift:178-182) -- and the reference's command
 * buffer lets them overlap; here their workgroups share the CUs inside one launch.  All handles of a group are of
 * the same kind; shapes may differ (a launch carries four distinct shapes; a group with more is issued as consecutive
 * launches).  effort_group_dispatch_count / effort_group_cutoff read call idx's hooks. */
EFFORT_API int effort_bucketmul_group(effort_ctx* ctx, int n, const effort_w* const* ws, const float* const* vs_dev,
                           const uint32_t* const* expNos_dev, float* const* outs_dev, const double* efforts);
/* effort_bucketmul_group with the decode loop's neighbouring element-wise steps folded into the launch (FP16 bundles):
 *   prologues[i] (NULL = all 0): how call i derives its input from vs[i] --
 *     EFFORT_PRE_NONE      input = v
 *     EFFORT_PRE_SILU_GATE input = x3 * v / (1 + exp(-v)), x3 = v_aux[i] f32 [inDim]: silu(x1, x3, out: x2) feeding w2
 *                          (runNetwork.swift:181-182, matrix.metal:25-35)
 *     EFFORT_PRE_RMSNORM   input = v / sqrt(mean(v^2) + 1e-5) * w, w = v_aux[i] f16 [inDim]: rmsNormFast + mul(by:)
 *                          feeding wq|wk|wv and w1|w3 (runNetwork.swift:121-122,173-175; aux.metal:113-152)
 *   resids[i] (NULL array or entry = none): out = resid + product -- h.add(by:) after wo and w2 (runNetwork.swift:172,183);
 *     resid may alias outs[i] (h += product in place).
 * Cutoff, dispatch and accumulation see the derived input exactly as if it had been materialised first.  A decoder layer
 * is then five launches instead of eight (Mistral-7B shapes at 25 % effort: 307 against 300 tokens/s, bit-identical logits). */
#define EFFORT_PRE_NONE 0
#define EFFORT_PRE_SILU_GATE 1
#define EFFORT_PRE_RMSNORM 2
EFFORT_API int effort_bucketmul_group_fused(effort_ctx* ctx, int n, const effort_w* const* ws, const float* const* vs_dev,
                                 const uint32_t* const* expNos_dev, float* const* outs_dev, const double* efforts,
                                 const int* prologues, const void* const* v_aux_dev, const float* const* resids_dev);
EFFORT_API int effort_bucketmul_q4_group(effort_ctx* ctx, int n, const effort_w* const* ws, const float* const* vs_dev,
                              const uint32_t* const* expNos_dev, float* const* outs_dev, const double* efforts);
EFFORT_API int effort_group_dispatch_count(effort_ctx* ctx, int idx, uint32_t* host_out);
EFFORT_API int effort_group_cutoff(effort_ctx* ctx, int idx, float* host_out);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ------------------------------------------------
 * The reference is single-device (helpers/gpu.swift:36-38); this is what a host needs to spread the path over the GPUs of a
 * node.  Two partitions: whole MATRICES on different ranks (nothing below but the communicator and the gather of the output
 * vectors), or bucket COLUMNS of one matrix across the ranks:
 *   effort_weights_column_shard(full, rank, world) -- rank's columns [rank*C/world, (rank+1)*C/world) of every bucket row of a
 *     registered bundle, i.e. outputs [rank*outDim/world, ...), as a VIEW of the full handle's buffers (no copy; C/world must be
 *     even); stats and probes are shared -- they are row-global, so every rank computes the same cutoff and selects the same rows
 *     -- and the shard takes the full matrix's fixed-point bound.  Q4: with the slice of the outlier index on those outputs.
 *     Multiply it like any handle; free it BEFORE the full handle; after rewriting weights in place refresh the FULL handle and
 *     shard again (effort_weights_refresh refuses a shard).
 *   effort_comm_unique_id: rank 0 makes the 128-byte id, the host program ships it to the other ranks by its own means;
 *   effort_comm_create(ctx, rank, world, id): collective over the world (ncclCommInitRank); one communicator per context;
 *   effort_allgather_outputs(ctx, send, recv, count): recv f32 [world][count] = every rank's `count` outputs, enqueued on the
 *     context's stream after the multiplies that wrote them (send may be recv + rank*count).  Matrices sharing an input
 *     vector put their shards' outputs side by side and gather ONCE (the messages are KB-scale: latency-bound over xGMI). */
#define EFFORT_COMM_ID_BYTES 128
EFFORT_API effort_w* effort_weights_column_shard(const effort_w* full, int rank, int world);
EFFORT_API int effort_comm_unique_id(void* id_out_128_bytes);
EFFORT_API int effort_comm_create(effort_ctx* ctx, int rank, int world, const void* id_128_bytes);
EFFORT_API int effort_comm_destroy(effort_ctx* ctx);
EFFORT_API int effort_comm_rank(effort_ctx* ctx);
EFFORT_API int effort_comm_world(effort_ctx* ctx);
EFFORT_API int effort_allgather_outputs(effort_ctx* ctx, const float* send_dev, float* recv_dev, int count);

/* ---- reference-visible state / test hooks ------------------------------------------------------ */

/* dispatch.size after calcDispatch (bucketMul.swift:46-47): number of bucket rows selected by the most
 * recent effort_bucketmul / _q4 / effort_calc_dispatch, before padding.  Synchronises the stream. */
EFFORT_API int effort_last_dispatch_count(effort_ctx* ctx, uint32_t* host_out);
/* BucketMul.shared.cutoff (bucketMul.swift:22).  Synchronises the stream. */
EFFORT_API int effort_last_cutoff(effort_ctx* ctx, float* host_out);

/* BucketMul.calcDispatch (bucketMul.swift:34-47) materialised in the reference's format: float2
 * entries {v[row % inDim] (Q4: v[row/8]*mean), float(row*cols)} in ASCENDING bucket-row order (the
 * reference's atomic append order is unspecified), count written to *count_dev.  dispatch_dev must
 * hold 2*inDim*percentLoad floats.  Not used by the fused hot path; kept for parity tests. */
EFFORT_API int effort_calc_dispatch(effort_ctx* ctx, const effort_w* w, const float* v_dev, const uint32_t* expNo_dev,
                         double effort, float* dispatch_dev, uint32_t* count_dev);

/* ---- decode-loop glue around the multiplies (runNetwork.swift:68-316) -----------------------------
 * The callers either side of the hot path: what a token step does between two expertMul calls.  The token position
 * and the token id live in DEVICE memory (pos_dev, id_dev), so a whole token step can be replayed from one hipGraph.
 * All vectors f32 on the device (VectorFloat), norm weights / embeddings f16. */

/* h += delta (delta may be NULL); out = h / sqrt(mean(h^2) + 1e-5) * w -- rmsNormFast + mul(by:) + add(by:)
 * (aux.metal:113-152,268-274; runNetwork.swift:121-122,170-173). */
EFFORT_API int effort_add_rmsnorm_mul(effort_ctx* ctx, float* h_dev, const float* delta_dev, const void* w_f16_dev, float* out_dev, int n);
/* rope_mx on q and on k, repeat4x32 of k and v over the query heads, stored at cache row *pos_dev
 * (runNetwork.swift:128-149, aux.metal:218-261, createFreqsCis2 model.swift:693-717; caches f32 [maxTokens][numHeads][headDim]). */
EFFORT_API int effort_rope_kv(effort_ctx* ctx, const float* xq_dev, const float* xk_dev, const float* xv_dev, float* q_out_dev,
                   float* k_cache_dev, float* v_cache_dev, const uint32_t* pos_dev, int numHeads, int numHeadsKV, int headDim,
                   int maxTokens, float ropeBase);
/* calcScores (/sqrt(headDim)) + softmax + sumScores over tokens 0..*pos_dev (runNetwork.swift:151-163, aux.metal:185-198,379-447). */
EFFORT_API int effort_attention(effort_ctx* ctx, const float* q_dev, const float* k_cache_dev, const float* v_cache_dev,
                     const uint32_t* pos_dev, float* out_dev, int numHeads, int headDim, int maxTokens);
/* effort_rope_kv + effort_attention in one launch (one workgroup per query head; the newest token is attended from LDS). */
EFFORT_API int effort_rope_attention(effort_ctx* ctx, const float* xq_dev, const float* xk_dev, const float* xv_dev, float* k_cache_dev,
                          float* v_cache_dev, const uint32_t* pos_dev, float* out_dev, int numHeads, int numHeadsKV, int headDim,
                          int maxTokens, float ropeBase);
/* silu(x1, x3, out:) = x3 * x1 / (1 + exp(-x1)) (matrix.metal:25-35). */
EFFORT_API int effort_silu_mul(effort_ctx* ctx, const float* x1_dev, const float* x3_dev, float* out_dev, int n);
/* tokEmbeddings.fetchRow(id, out:) (aux.metal:355): row *id_dev of an f16 [vocab][n] table as f32. */
EFFORT_API int effort_fetch_row(effort_ctx* ctx, const void* emb_f16_dev, const uint32_t* id_dev, float* out_dev, int n);
/* Mixtral routing (runNetwork.swift:185-199): idx2_dev = the two largest gate logits' experts (mpsTopK(topK: 2)),
 * val2_dev = softmax over those two logits; effort_mix2: out = f0 * val2[0] + f1 * val2[1]. */
EFFORT_API int effort_top2_softmax(effort_ctx* ctx, const float* gate_dev, int n, uint32_t* idx2_dev, float* val2_dev);
EFFORT_API int effort_mix2(effort_ctx* ctx, const float* f0_dev, const float* f1_dev, const float* val2_dev, float* out_dev, int n);
/* greedy pick: *id_out_dev = argmax(logits) (the reference takes mpsTopK[0], helpers/mps.swift:52-84); if history_dev is
 * given, history_dev[*pos_dev] = the pick; then *pos_dev += 1. */
EFFORT_API int effort_argmax(effort_ctx* ctx, const float* logits_dev, int n, uint32_t* id_out_dev, uint32_t* pos_dev, uint32_t* history_dev,
                  int historyLen);
/* The position lives in device memory, so the glue cannot refuse a step with a return code: a step at *pos_dev >= maxTokens
 * (effort_rope_kv, effort_rope_attention) or >= historyLen (effort_argmax) writes NOTHING to the cache / history and raises
 * bit 0 of the context's decode status; an argmax over NaN logits returns token 0 and raises bit 1.  Reads and clears it. */
EFFORT_API int effort_decode_status(effort_ctx* ctx, int* host_out);

/* ---- weight layout converter ------------------------------------------------------------------- */

/* func bucketize(_:outTensorsPref:tensors:goQ8:false) -- convert.swift:209-260 with kernels getProbes,
 * prepareValsIdxs, idxsBitonicSortAbs, preBucketize, bucketize, makeStats (convert.metal:14-119,
 * 315-342).  W_f16_dev is the HF matrix [outDim, inDim]; outputs as in effort_weights_fp16 with
 * percentLoad 16, one expert.  Runs on the GPU (all device pointers), enqueued on the stream. */
EFFORT_API int effort_convert_fp16(effort_ctx* ctx, const void* W_f16_dev, int outDim, int inDim,
                        void* buckets_dev, void* stats_dev, void* probes_dev);
/* The same, writing bucket rows `rowPitchBytes` apart (0 = dense; otherwise a multiple of 8 bytes >= 2*outDim/16 -- the stats pass
 * reads 8 bytes at a time; see effort_weights_fp16_pitched; the padding is left untouched). */
EFFORT_API int effort_convert_fp16_pitched(effort_ctx* ctx, const void* W_f16_dev, int outDim, int inDim,
                                void* buckets_dev, int rowPitchBytes, void* stats_dev, void* probes_dev);
/* Elements the last effort_convert_fp16 calls could not place: the reference's preBucketize (convert.metal:40-61) drops an
 * element whose bucket is already full, which happens when zero padding of a non-power-of-two row ties with real zeros;
 * the converter reproduces that and counts the drops here.  Reads and clears the count (0 = every element was placed). */
EFFORT_API int effort_convert_status(effort_ctx* ctx, int* host_out);

/* q4_draft.convert(core2) (q4_draft.py:70-322; driver q4_convert.py:41-81) for one matrix, on the GPU.  core2_f16_dev is
 * W.T: f16 [inDim][outDim], outDim % 32 == 0.  Outputs (device buffers of the caller): buckets u16 [inDim*8][outDim/32],
 * stats f32 [inDim*8][2], probes f16 [min(inDim, outDim)] (the diagonal after outlier removal), outliers f32 [n][4] =
 * (value, inIdx, outIdx, 0) with n = effort_q4_outlier_count(inDim, outDim, perc) = int(inDim*outDim*perc), ordered by |w|
 * descending, flat index ascending on ties (the reference's unstable argsort leaves ties unspecified).  Bit-identical to
 * the reference's outputs (tests/golden/q4_*.npz).  Synchronises the stream; allocates its scratch (4*inDim*outDim bytes
 * + the sort's) for the duration of the call. */
EFFORT_API int64_t effort_q4_outlier_count(int inDim, int outDim, double perc);
EFFORT_API int effort_convert_q4(effort_ctx* ctx, const void* core2_f16_dev, int inDim, int outDim, double perc, void* buckets_dev,
                      void* stats_dev, void* probes_dev, void* outliers_dev);

/* VectorFloat.cosineSimilarityTo (model.swift:511-519; aux.metal:293-312).  Synchronises. */
EFFORT_API int effort_cosine(effort_ctx* ctx, const float* a_dev, const float* b_dev, int n, float* host_out);

/* Tuning knobs, profiling, tracing and ablation hooks live in effort_hip_debug.h: they are not part of the drop-in surface. */

#ifdef __cplusplus
}
#endif
#endif /* EFFORT_HIP_H */
