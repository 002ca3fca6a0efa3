// Glue kernels of the decode loop around bucketMul (runNetwork.swift:68-316) -- the callers either side of the hot
// path, SURVEY section 8f row 1.  Everything a token needs between two multiplies, with the token position and the
// token id kept in DEVICE memory so that a whole token step can be replayed from one hipGraph (the reference's loop
// spends ~15 ms/token in gaps between its ~25 dispatches per layer, runNetwork.swift:91-103).
//   add_rmsnorm_mul  h += delta; out = h / sqrt(mean(h^2) + 1e-5) * w     rmsNorm32fast + mulVec32by16 + add (aux.metal:113-152,268-274)
//   rope_kv          rope_mx on q and on the 4x repeated k, repeat4x32 of k and v, written at cache[pos]   (aux.metal:218-261)
//   attention        dotSetScore2 (/sqrt(headDim)) + softmax + sumScores32 over tokens 0..pos, one workgroup per head (aux.metal:185-198,379-447)
//   silu_mul         x3 * x1 / (1 + exp(-x1))                              silu32b (matrix.metal:25-35)
//   fetch_row        tok_embeddings row (f16) -> f32                       fetchRow16to32 (aux.metal:355)
//   top2_softmax     Mixtral gate: top-2 experts + softmax of their logits (mpsTopK + softmax, runNetwork.swift:186-189); mix2: weighted sum
//   argmax           greedy pick of the next token (the reference takes mpsTopK[0], helpers/mps.swift:52-84), pos += 1
#include "effort_internal.h"

namespace effort {

__device__ __forceinline__ float block_sum(float x, float* red /* [17] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    x = wave_sum_f32(x);                              // (DPP scan order; the multiply's fused rmsNorm prologue sums the same way)
    __syncthreads();                                  // red may still be read from a previous call
    if (lane == 0) red[wave] = x;
    __syncthreads();
    float s = 0.0f;
    for (int w = 0; w < nw; w++) s += red[w];         // same order in every thread: identical result everywhere
    return s;
}

__device__ __forceinline__ float block_max(float x, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    x = wave_max_f32(x);
    __syncthreads();
    if (lane == 0) red[wave] = x;
    __syncthreads();
    float s = red[0];
    for (int w = 1; w < nw; w++) s = fmaxf(s, red[w]);
    return s;
}

__global__ __launch_bounds__(1024) void add_rmsnorm_mul_kernel(float* __restrict__ h, const float* __restrict__ delta,
                                                               const uint16_t* __restrict__ w, float* __restrict__ out, uint32_t n) {
    __shared__ float red[17];
    // One memory round trip: every load (h, delta, the norm weights) is issued before anything is used, the updated state
    // stays in registers for the second half (n <= 4 * 1024 elements per register slot; a longer state takes the loop
    // below).  This kernel sits on the decode loop's dependent chain twice per layer.
    constexpr int K = 4;
    if (n <= K * 1024u) {
        float x[K], d[K], wf[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t i = min(k * 1024u + threadIdx.x, n - 1u);               // clamped, branch-free
            x[k] = h[i];
            d[k] = delta ? delta[i] : 0.0f;
            wf[k] = half_bits_to_float(w[i]);
        }
        float ss = 0.0f;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool live = k * 1024u + threadIdx.x < n;
            x[k] += d[k];
            if (live && delta) h[k * 1024u + threadIdx.x] = x[k];
            ss += live ? x[k] * x[k] : 0.0f;
        }
        const float inv = 1.0f / sqrtf(block_sum(ss, red) / (float)n + 1e-5f);            // aux.metal:150
#pragma unroll
        for (int k = 0; k < K; k++)
            if (k * 1024u + threadIdx.x < n) out[k * 1024u + threadIdx.x] = (x[k] * inv) * wf[k];
        return;
    }
    float ss = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        float x = h[i];
        if (delta) { x += delta[i]; h[i] = x; }
        ss += x * x;
    }
    const float inv = 1.0f / sqrtf(block_sum(ss, red) / (float)n + 1e-5f);                // aux.metal:150
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) out[i] = (h[i] * inv) * half_bits_to_float(w[i]);
}

// grid = numHeads, block = headDim.  freq_j = base^(-2j/headDim) (createFreqsCis2, model.swift:693-717: logspace with
// base 1e-6 <=> theta 1e6), rotate_half convention of rope_mx.
__global__ void rope_kv_kernel(const float* __restrict__ xq, const float* __restrict__ xk, const float* __restrict__ xv,
                               float* __restrict__ qOut, float* __restrict__ kCache, float* __restrict__ vCache,
                               const uint32_t* __restrict__ posPtr, uint32_t numHeads, uint32_t kvRepeats, float logBase,
                               uint32_t maxTokens, int* __restrict__ status) {
    const uint32_t head = blockIdx.x, d = threadIdx.x, headDim = blockDim.x, half = headDim / 2, pos = posPtr[0];
    if (pos >= maxTokens) {                            // past the cache: nothing is written, the context's status word says so
        if (head == 0 && d == 0) atomicOr(status, 1);
        return;
    }
    const uint32_t j = d % half;
    const float freq = (float)exp((double)logBase * (-(double)j / (double)half));          // Float(freq), model.swift:707
    const float angle = (float)pos * freq;
    const float c = cosf(angle), s = sinf(angle);
    const float* q = xq + head * headDim;
    const float* k = xk + (head / kvRepeats) * headDim;                                    // repeat4x32: kv head y feeds heads 4y..4y+3
    const float qr = d < half ? q[d] * c - q[d + half] * s : q[d] * c + q[d - half] * s;
    const float kr = d < half ? k[d] * c - k[d + half] * s : k[d] * c + k[d - half] * s;
    const size_t slot = ((size_t)pos * numHeads + head) * headDim + d;
    qOut[head * headDim + d] = qr;
    kCache[slot] = kr;
    vCache[slot] = xv[(head / kvRepeats) * headDim + d];
}

// grid = numHeads, block = 256.  Tokens 0..pos.  exp(x - max) / sum exp(x - max) == the reference's exp(x) / sum exp(x).
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, const float* __restrict__ kCache,
                                                        const float* __restrict__ vCache, const uint32_t* __restrict__ posPtr,
                                                        float* __restrict__ out, uint32_t numHeads, uint32_t headDim, uint32_t maxTokens) {
    extern __shared__ float sc[];                      // [maxTokens] scores, then probabilities
    __shared__ float red[17];
    const uint32_t head = blockIdx.x, tid = threadIdx.x, nTok = min(posPtr[0] + 1u, maxTokens);
    const float* qh = q + head * headDim;
    const float scale = 1.0f / sqrtf((float)headDim);
    const int lane = tid & 63, wave = tid >> 6;
    // one wave per token: lanes stride the head dimension
    for (uint32_t t = wave; t < nTok; t += 4) {
        const float* kh = kCache + ((size_t)t * numHeads + head) * headDim;
        float dot = 0.0f;
        for (uint32_t d = lane; d < headDim; d += 64) dot += qh[d] * kh[d];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off);
        if (lane == 0) sc[t] = dot * scale;
    }
    __syncthreads();
    float m = -INFINITY;
    for (uint32_t t = tid; t < nTok; t += 256) m = fmaxf(m, sc[t]);
    m = block_max(m, red);
    float sum = 0.0f;
    for (uint32_t t = tid; t < nTok; t += 256) { const float e = expf(sc[t] - m); sc[t] = e; sum += e; }
    sum = block_sum(sum, red);
    const float inv = 1.0f / sum;
    // sumScores32: out[head][d] = sum_t p_t * v[t][head][d]; threads = (token phase, d)
    __shared__ float part[256];
    const uint32_t d = tid % headDim, ph = tid / headDim, nph = 256 / headDim;
    float acc = 0.0f;
    if (ph < nph)
        for (uint32_t t = ph; t < nTok; t += nph) acc += sc[t] * vCache[((size_t)t * numHeads + head) * headDim + d];
    part[tid] = acc;
    __syncthreads();
    if (tid < headDim) {
        float s2 = 0.0f;
        for (uint32_t p = 0; p < nph; p++) s2 += part[p * headDim + tid];
        out[head * headDim + tid] = s2 * inv;
    }
}

// rope_kv + attention in ONE launch (one workgroup per query head): the head ropes its q and its kv head's k, stores k / v
// at cache[pos], and attends over tokens 0..pos, the newest token straight from LDS.  Saves a dependent launch per layer.
__global__ __launch_bounds__(256) void rope_attention_kernel(const float* __restrict__ xq, const float* __restrict__ xk,
                                                             const float* __restrict__ xv, float* __restrict__ kCache,
                                                             float* __restrict__ vCache, const uint32_t* __restrict__ posPtr,
                                                             float* __restrict__ out, uint32_t numHeads, uint32_t kvRepeats,
                                                             uint32_t headDim, uint32_t maxTokens, float logBase, int* __restrict__ status) {
    extern __shared__ float sc[];                      // [maxTokens] scores, then probabilities
    __shared__ float red[17];
    __shared__ float qs[256], ks[256], vs[256];        // this head's roped q, roped k and v of the newest token
    const uint32_t head = blockIdx.x, tid = threadIdx.x, half = headDim / 2;
    // this head's q / k / v do not depend on the position: asked for beside it, not after it (one dependent round trip less)
    float qa = 0.0f, qb = 0.0f, ka = 0.0f, kb = 0.0f, vv = 0.0f, freq = 0.0f;
    if (tid < headDim) {
        const uint32_t d = tid, j = d % half, dp = d < half ? d + half : d - half;
        const float* q = xq + head * headDim;
        const float* k = xk + (head / kvRepeats) * headDim;
        qa = q[d]; qb = q[dp]; ka = k[d]; kb = k[dp];
        vv = xv[(head / kvRepeats) * headDim + d];
        freq = (float)exp((double)logBase * (-(double)j / (double)half));
    }
    if (posPtr[0] >= maxTokens) {                      // past the cache (uniform): nothing is written, the status word says so
        if (head == 0 && tid == 0) atomicOr(status, 1);
        return;
    }
    const uint32_t pos = posPtr[0], nTok = pos + 1u;
    // the first pass of key rows needs only the position: asked for here, under the rotation's arithmetic and its barrier
    const uint32_t lpt = headDim / 8u, tokPerPass = 256u / lpt, sub = tid % lpt, tk = tid / lpt;
    float4 ka0[4], kb0[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t t = min((uint32_t)u * tokPerPass + tk, pos);
        const float4* kh = reinterpret_cast<const float4*>(kCache + ((size_t)t * numHeads + head) * headDim + sub * 8u);
        ka0[u] = kh[0]; kb0[u] = kh[1];
    }
    if (tid < headDim) {
        const uint32_t d = tid;
        const float angle = (float)pos * freq;
        const float c = cosf(angle), s = sinf(angle);
        const float qr = d < half ? qa * c - qb * s : qa * c + qb * s;
        const float kr = d < half ? ka * c - kb * s : ka * c + kb * s;
        const size_t slot = ((size_t)pos * numHeads + head) * headDim + d;
        qs[d] = qr; ks[d] = kr; vs[d] = vv;
        kCache[slot] = kr; vCache[slot] = vv;
    }
    __syncthreads();
    // scores: headDim/8 threads per token, 8 dims each as two float4 loads; four token passes are issued before any is
    // consumed, so the cache reads overlap instead of costing one memory round trip per token
    const float scale = 1.0f / sqrtf((float)headDim);
    float q8[8];
#pragma unroll
    for (int i = 0; i < 8; i++) q8[i] = qs[sub * 8u + i];
    for (uint32_t t0 = 0; t0 < nTok; t0 += tokPerPass * 4u) {
        float4 ka[4], kb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t t = min(t0 + u * tokPerPass + tk, pos);           // clamped: surplus lanes re-read the newest row
            const float4* kh = reinterpret_cast<const float4*>(kCache + ((size_t)t * numHeads + head) * headDim + sub * 8u);
            if (t0 == 0u) { ka[u] = ka0[u]; kb[u] = kb0[u]; }                  // (uniform: the first pass was asked for before the rotation)
            else { ka[u] = kh[0]; kb[u] = kh[1]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t t = t0 + u * tokPerPass + tk;
            float dot;
            if (t >= pos) {                                                   // the newest token: from LDS (its cache row is being written)
                dot = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; i++) dot += q8[i] * ks[sub * 8u + i];
            } else {
                dot = q8[0] * ka[u].x + q8[1] * ka[u].y + q8[2] * ka[u].z + q8[3] * ka[u].w +
                      q8[4] * kb[u].x + q8[5] * kb[u].y + q8[6] * kb[u].z + q8[7] * kb[u].w;
            }
            if (lpt == 16u) {                                                  // (uniform) headDim 128: a token's 16 lanes are one DPP row -- no LDS round trips
                dot = row16_sum_f32(dot);                                      // the row's total sits in its lane 15
                if (sub == 15u && t < nTok) sc[t] = dot * scale;
            } else {
                for (uint32_t off = lpt / 2u; off >= 1u; off >>= 1) dot += __shfl_xor(dot, (int)off);
                if (sub == 0 && t < nTok) sc[t] = dot * scale;
            }
        }
    }
    // the first eight tokens' value rows of this thread's phase do not depend on the scores: asked for HERE, their round trip runs
    // under the softmax instead of after it (a decode step's attention is a chain of dependent round trips: position, q/k/v,
    // keys, values)
    const uint32_t tpr = headDim / 4u, nph = 256u / tpr, d4 = (tid % tpr) * 4u, ph = tid / tpr;
    float4 vv0[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const uint32_t tc = min(ph + (uint32_t)u * nph, pos);
        vv0[u] = *reinterpret_cast<const float4*>(vCache + ((size_t)tc * numHeads + head) * headDim + d4);
    }
    __syncthreads();
    float m = -INFINITY;
    for (uint32_t t = tid; t < nTok; t += 256) m = fmaxf(m, sc[t]);
    m = block_max(m, red);
    float sum = 0.0f;
    for (uint32_t t = tid; t < nTok; t += 256) { const float e = expf(sc[t] - m); sc[t] = e; sum += e; }
    sum = block_sum(sum, red);
    const float inv = 1.0f / sum;
    // weighted sum: headDim/4 threads per token phase (one float4 of the head each), 256/(headDim/4) phases; eight
    // tokens' loads in flight per thread
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (uint32_t t0 = ph; t0 < nTok; t0 += nph * 8u) {
        float4 vv[8]; float p8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t t = t0 + u * nph;
            const uint32_t tc = min(t, pos);
            if (t0 == ph) vv[u] = vv0[u];                                   // (the first pass was asked for before the softmax)
            else vv[u] = *reinterpret_cast<const float4*>(vCache + ((size_t)tc * numHeads + head) * headDim + d4);
            p8[u] = t < nTok ? sc[t] : 0.0f;
            if (t == pos) vv[u] = make_float4(vs[d4], vs[d4 + 1], vs[d4 + 2], vs[d4 + 3]);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { acc.x += p8[u] * vv[u].x; acc.y += p8[u] * vv[u].y; acc.z += p8[u] * vv[u].z; acc.w += p8[u] * vv[u].w; }
    }
    __shared__ float4 part4[256];
    part4[tid] = acc;
    __syncthreads();
    if (tid < tpr) {
        float4 s2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (uint32_t p = 0; p < nph; p++) { const float4 x = part4[p * tpr + tid]; s2.x += x.x; s2.y += x.y; s2.z += x.z; s2.w += x.w; }
        *reinterpret_cast<float4*>(out + head * headDim + d4) = make_float4(s2.x * inv, s2.y * inv, s2.z * inv, s2.w * inv);
    }
}

__global__ void silu_mul_kernel(const float* __restrict__ x1, const float* __restrict__ x3, float* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x3[i] * x1[i] / (1.0f + expf(-x1[i]));
}

__global__ void fetch_row_kernel(const uint16_t* __restrict__ emb, const uint32_t* __restrict__ idPtr, float* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = half_bits_to_float(emb[(size_t)idPtr[0] * n + i]);
}

// one workgroup: greedy next token = index of the largest logit (lowest index on ties); advances the position
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, uint32_t n, uint32_t* __restrict__ idOut,
                                                      uint32_t* __restrict__ posPtr, uint32_t* __restrict__ history,
                                                      uint32_t historyLen, int* __restrict__ status) {
    __shared__ float bv[16];
    __shared__ uint32_t bi[16];
    float best = -INFINITY; uint32_t idx = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float x = logits[i];
        if (x > best || (x == best && i < idx)) { best = x; idx = i; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off); const uint32_t oi = (uint32_t)__shfl_xor((int)idx, off);
        if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < (blockDim.x >> 6); w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        if (idx >= n) { idx = 0; atomicOr(status, 2); }           // every logit NaN: a valid id all the same (the next fetch_row reads row idx)
        idOut[0] = idx;
        if (history) {
            if (posPtr[0] < historyLen) history[posPtr[0]] = idx;
            else atomicOr(status, 1);                                 // past the history buffer: not written
        }
        posPtr[0] += 1u;
    }
}

// Mixtral routing (runNetwork.swift:185-199): mpsTopK(topK: 2) of the gate logits, softmax over the two picked values.
__global__ void top2_softmax_kernel(const float* __restrict__ gate, uint32_t n, uint32_t* __restrict__ idx, float* __restrict__ val) {
    if (threadIdx.x != 0) return;
    uint32_t i0 = 0, i1 = 0xFFFFFFFFu; float v0 = -INFINITY, v1 = -INFINITY;
    for (uint32_t i = 0; i < n; i++) {
        const float x = gate[i];
        if (x > v0) { v1 = v0; i1 = i0; v0 = x; i0 = i; }
        else if (x > v1) { v1 = x; i1 = i; }
    }
    if (i1 == 0xFFFFFFFFu) { i1 = i0; v1 = v0; }
    const float e1 = expf(v1 - v0), inv = 1.0f / (1.0f + e1);
    idx[0] = i0; idx[1] = i1; val[0] = inv; val[1] = e1 * inv;
}
// out = f0 * val[0] + f1 * val[1]   (ffnOut[i].mul(by: gateVals[i]); h.add(by: ffnOut[i]) -- the add rides in the next norm)
__global__ void mix2_kernel(const float* __restrict__ f0, const float* __restrict__ f1, const float* __restrict__ val,
                            float* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f0[i] * val[0] + f1[i] * val[1];
}

hipError_t launch_top2_softmax(const float* gate, uint32_t n, uint32_t* idx, float* val, hipStream_t st) {
    hipLaunchKernelGGL(top2_softmax_kernel, dim3(1), dim3(64), 0, st, gate, n, idx, val);
    return hipGetLastError();
}
hipError_t launch_mix2(const float* f0, const float* f1, const float* val, float* out, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(mix2_kernel, dim3((n + 255) / 256), dim3(256), 0, st, f0, f1, val, out, n);
    return hipGetLastError();
}
hipError_t launch_add_rmsnorm_mul(float* h, const float* delta, const uint16_t* w, float* out, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(add_rmsnorm_mul_kernel, dim3(1), dim3(1024), 0, st, h, delta, w, out, n);
    return hipGetLastError();
}
hipError_t launch_rope_kv(const float* xq, const float* xk, const float* xv, float* qOut, float* kCache, float* vCache,
                          const uint32_t* pos, uint32_t numHeads, uint32_t numHeadsKV, uint32_t headDim, float ropeBase, uint32_t maxTokens,
                          int* status, hipStream_t st) {
    hipLaunchKernelGGL(rope_kv_kernel, dim3(numHeads), dim3(headDim), 0, st, xq, xk, xv, qOut, kCache, vCache, pos, numHeads,
                       numHeads / numHeadsKV, logf(ropeBase), maxTokens, status);
    return hipGetLastError();
}
hipError_t launch_attention(const float* q, const float* kCache, const float* vCache, const uint32_t* pos, float* out,
                            uint32_t numHeads, uint32_t headDim, uint32_t maxTokens, hipStream_t st) {
    hipLaunchKernelGGL(attention_kernel, dim3(numHeads), dim3(256), maxTokens * sizeof(float), st, q, kCache, vCache, pos, out, numHeads, headDim, maxTokens);
    return hipGetLastError();
}
hipError_t launch_rope_attention(const float* xq, const float* xk, const float* xv, float* kCache, float* vCache, const uint32_t* pos,
                                 float* out, uint32_t numHeads, uint32_t numHeadsKV, uint32_t headDim, uint32_t maxTokens, float ropeBase,
                                 int* status, hipStream_t st) {
    hipLaunchKernelGGL(rope_attention_kernel, dim3(numHeads), dim3(256), maxTokens * sizeof(float), st, xq, xk, xv, kCache, vCache, pos, out,
                       numHeads, numHeads / numHeadsKV, headDim, maxTokens, logf(ropeBase), status);
    return hipGetLastError();
}
hipError_t launch_silu_mul(const float* x1, const float* x3, float* out, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(silu_mul_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x1, x3, out, n);
    return hipGetLastError();
}
hipError_t launch_fetch_row(const uint16_t* emb, const uint32_t* id, float* out, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(fetch_row_kernel, dim3((n + 255) / 256), dim3(256), 0, st, emb, id, out, n);
    return hipGetLastError();
}
hipError_t launch_argmax(const float* logits, uint32_t n, uint32_t* idOut, uint32_t* pos, uint32_t* history, uint32_t historyLen,
                         int* status, hipStream_t st) {
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, st, logits, n, idOut, pos, history, historyLen, status);
    return hipGetLastError();
}

}  // namespace effort
