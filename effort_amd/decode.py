"""The decode loop around bucketMul -- host mirror of ``runNetwork`` (runNetwork.swift:68-316) for Mistral-7B-shaped
models (the caller of the hot path).

One token step = per layer: rmsNorm * attnNorm -> wq | wk | wv (ONE grouped bucketMul launch: three independent calls on
the same input) -> rope + 4x kv repeat + cache -> scores / softmax / weighted sum -> wo -> residual + rmsNorm * ffnNorm
-> w1 | w3 (one grouped launch) -> silu -> w2 -> residual; then the output norm, the dense LM head (``basicMul``,
runNetwork.swift:222) and a greedy pick.  The glue runs in the HIP kernels of effort_amd/csrc/decode.hip through the C
ABI; the token position and the current token id live in device memory, so a whole step is captured once into a hipGraph
and replayed per token -- no host work, no per-kernel launch gaps (the reference spends ~15 ms/token in such gaps,
runNetwork.swift:91-103).  torch is used for buffers and graph capture only.

``dense=True`` routes every projection through ``basicMul`` (rocBLAS GEMV on the f16 cores): the baseline the
reference compares against (``effort`` 100 % vs dense, KL divergence of the logits).
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass

import torch

from . import _lib
from .bucket_mul import basicMul, bucketMul, bucketMulGroup
from .runtime import gpu as _gpu
from .weights import ExpertWeights


@dataclass
class MistralConfig:                      # main.swift:45-77
    stateDim: int = 4096
    hiddenDim: int = 14336
    numLayers: int = 32
    numHeads: int = 32
    numHeadsKV: int = 8
    headDim: int = 128
    vocab: int = 32000
    ropeBase: float = 1e6                 # createFreqsCis2: logspace base 1e-6 (model.swift:700)
    numExperts: int = 1                   # > 1: Mixtral -- a dense gate picks 2 experts per token and layer (runNetwork.swift:185-199)


class Layer:                              # loader.swift Layer: norms + seven ExpertWeights
    __slots__ = ("attnNorm", "ffnNorm", "wq", "wk", "wv", "wo", "w1", "w2", "w3", "ffnGate")


class Model:
    def __init__(self, cfg: MistralConfig):
        self.cfg = cfg
        self.layers: list[Layer] = []
        self.norm = None                  # f16 [stateDim]
        self.output = None                # f16 [vocab, stateDim]  (output.core)
        self.tokEmbeddings = None         # f16 [vocab, stateDim]  (tok_embeddings.core)

    @classmethod
    def random(cls, cfg: MistralConfig, seed: int = 0, device="cuda", scale: float = 0.02, keep_cores: bool = True,
               structured: bool = False) -> "Model":
        """Random-init weights of the architecture (no checkpoints here), converted by the GPU bucketizer.
        ``structured``: the statistics trained transformers show and i.i.d. Gaussians lack -- heavy-tailed weights with
        per-input-channel and per-output scale spread, norm weights with a few outlier channels (so the normalised state
        a multiply sees is heavy-tailed), a peaked output distribution -- see ``structured_matrix``.  The reference's quality
        figures (cos-sim 0.99 at 25 % effort, docs/ryc/ryc0.3.png) are for such weights; on Gaussian ones 25 % gives 0.94."""
        m = cls(cfg)
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)

        def mat(o, i, s=scale):
            if structured:
                return structured_matrix(o, i, gen, device, s)
            return (torch.randn((o, i), generator=gen, device=device, dtype=torch.float32) * s).to(torch.float16)

        def vec(n):
            if structured:
                return structured_norm_weights(n, gen, device)
            return (1.0 + 0.1 * torch.randn(n, generator=gen, device=device, dtype=torch.float32)).to(torch.float16)

        kv = cfg.numHeadsKV * cfg.headDim

        def bundle(o, i, experts=1):
            ews = []
            for _ in range(experts):
                ew = ExpertWeights.from_core(mat(o, i))
                if not keep_cores:
                    ew.core = None
                ews.append(ew)
            if experts == 1:
                ew = ews[0]
            else:                                         # Mixtral: all experts in one buffer, picked by expNo (loader.swift:113-166)
                ew = ExpertWeights.stack(ews)
                ew.core = torch.stack([e.core for e in ews]) if keep_cores else None      # [E, out, in] for the dense path
            ew.handle
            return ew

        for _ in range(cfg.numLayers):
            L = Layer()
            L.attnNorm, L.ffnNorm = vec(cfg.stateDim), vec(cfg.stateDim)
            for name, (o, i) in (("wq", (cfg.stateDim, cfg.stateDim)), ("wk", (kv, cfg.stateDim)), ("wv", (kv, cfg.stateDim)),
                                 ("wo", (cfg.stateDim, cfg.stateDim))):
                setattr(L, name, bundle(o, i))
            for name, (o, i) in (("w1", (cfg.hiddenDim, cfg.stateDim)), ("w3", (cfg.hiddenDim, cfg.stateDim)), ("w2", (cfg.stateDim, cfg.hiddenDim))):
                setattr(L, name, bundle(o, i, cfg.numExperts))
            L.ffnGate = mat(cfg.numExperts, cfg.stateDim) * 10 if cfg.numExperts > 1 else None     # f16 [numExperts, stateDim]
            m.layers.append(L)
        m.norm = vec(cfg.stateDim)
        m.output = mat(cfg.vocab, cfg.stateDim, 0.08 if structured else scale)      # structured: logits a few units wide (a peaked next-token distribution)
        m.tokEmbeddings = (torch.randn((cfg.vocab, cfg.stateDim), generator=gen, device=device, dtype=torch.float32)).to(torch.float16)
        return m

    @classmethod
    def load(cls, loader, cfg: MistralConfig, percentLoad: int = 16, device="cuda") -> "Model":
        """From a bucketed model on disk (effort_amd.bucketfile, names of convert.swift:70-105)."""
        from .bucketfile import loadExpertWeights
        m = cls(cfg)
        for n in range(cfg.numLayers):
            L = Layer()
            L.attnNorm = loader[f"layers.{n}.attention_norm"].to(device=device, dtype=torch.float16)
            L.ffnNorm = loader[f"layers.{n}.ffn_norm"].to(device=device, dtype=torch.float16)
            for s in "qkvo":
                setattr(L, "w" + s, loadExpertWeights(loader, f"layers.{n}.attention.w{s}", device=device))
            for w, (o, i) in (("w1", (cfg.hiddenDim, cfg.stateDim)), ("w3", (cfg.hiddenDim, cfg.stateDim)), ("w2", (cfg.stateDim, cfg.hiddenDim))):
                setattr(L, w, loadExpertWeights(loader, f"layers.{n}.feed_forward.experts.", w, inDim=i, outDim=o, numExperts=cfg.numExperts,
                                                percentLoad=percentLoad, device=device))
            gate = f"layers.{n}.feed_forward.gate"
            L.ffnGate = loader[gate].to(device=device, dtype=torch.float16) if cfg.numExperts > 1 and loader.hasTensor(gate) else None
            m.layers.append(L)
        m.norm = loader["model.norm"].to(device=device, dtype=torch.float16)
        m.output = loader["output.core"].to(device=device, dtype=torch.float16)
        m.tokEmbeddings = loader["tok_embeddings.core"].to(device=device, dtype=torch.float16)
        return m


def structured_matrix(o: int, i: int, gen, device, scale: float = 0.02) -> torch.Tensor:
    """Synthetic f16 [o, i] matrix with the structure of trained weights: heavy-tailed entries (a Gaussian times a
    log-normal, sigma 0.5), a log-normal scale per input channel (sigma 0.7) and per output (sigma 0.3); overall std =
    ``scale``.  With a heavy-tailed input this puts single-multiply cos-sim at 25 % effort near 0.99 (bench.py: sweep_structured)."""
    w = torch.randn((o, i), generator=gen, device=device, dtype=torch.float32)
    w *= torch.exp(0.5 * torch.randn((o, i), generator=gen, device=device, dtype=torch.float32))
    w *= torch.exp(0.7 * torch.randn((1, i), generator=gen, device=device, dtype=torch.float32))
    w *= torch.exp(0.3 * torch.randn((o, 1), generator=gen, device=device, dtype=torch.float32))
    w *= scale / w.std()
    return w.to(torch.float16)


def structured_norm_weights(n: int, gen, device) -> torch.Tensor:
    """rmsNorm weights f16 [n]: log-normal spread (sigma 0.6) with 0.5 % outlier channels eight times larger."""
    w = torch.exp(0.6 * torch.randn(n, generator=gen, device=device, dtype=torch.float32))
    big = torch.rand(n, generator=gen, device=device) < 0.005
    w = torch.where(big, w * 8.0, w)
    return (w / w.pow(2).mean().sqrt()).to(torch.float16)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Decoder:
    """State of one sequence (the globals of main.swift:78-140: h, xq, KV caches, scores ...) + the token step."""

    def __init__(self, model: Model, maxTokens: int = 256, fused_attention: bool = True, fused_glue=True,
                 world: int = 1, rank: int = 0, sharded: bool | None = None, emulate_world: bool = False):
        cfg = self.cfg = model.cfg
        # (Round 4's `chain=True` -- a layer's dependent multiplies as ONE launch of resident workgroups -- measured 252 against 308
        #  tokens/s and lives on branch `chain-launch`: DESIGN.md 4.5.)
        self.fused_attention = bool(fused_attention)      # rope + cache + attention in one launch per layer (else two)
        # rmsNorm, silu and the residual adds folded into the multiplies (effort_bucketmul_group_fused): 5 launches per layer
        # instead of 8.  Dense-FFN models only; the dense baseline keeps the separate glue kernels.  On by default since round 3:
        # with every operand of a prologue asked for at the top of the item (they were two more dependent round trips) and the
        # residual asked for before the slabs, all three folded are 306 against 300 tokens/s at 25 % effort (253 against 247 at
        # 50 %); the gate alone or the residuals alone still lose 1 % (tools/lab/decode_ab.py --fused-glue ...).  Bit-identical logits.
        # fused_glue may also name WHICH steps fold into the multiplies: any of "norm" (rmsNorm into wq|wk|wv and w1|w3), "gate"
        # (silu into w2), "resid" (the residual adds after wo and w2); True = all three
        parts = ("norm", "gate", "resid") if fused_glue is True else tuple(fused_glue or ())
        self.fuse = frozenset(parts) if all(L.ffnGate is None for L in model.layers) else frozenset()
        self.fused_glue = bool(self.fuse)
        self.model, self.maxTokens = model, int(maxTokens)
        dev = model.norm.device
        self.g = _gpu(dev.index)
        # Column-sharded decode (BASELINE config 4: "all 32 layers' Wq/Wk/Wv/Wo + FFN matrices sharded across 8 x MI355X, RCCL
        # all-gather"; the reference is single-device, runNetwork.swift:121-183 is what this wraps): every bundle becomes this rank's
        # column shard, every launch group is followed by ONE effort_allgather_outputs (in place on h for wo / w2), the state vectors
        # stay replicated, and the glue (attention, the LM head) runs replicated on every rank.  ``sharded=True`` with world 1 runs
        # the same path through a communicator of one rank; ``emulate_world`` computes every rank's launches in THIS process (no
        # collective): how the split is validated, and its launches timed, on one GPU.
        self.world, self.rank = int(world), int(rank)
        self.sharded = bool(sharded) if sharded is not None else (self.world > 1)
        self.groups = None
        if self.sharded or emulate_world:
            if self.fuse != {"norm", "gate", "resid"}:
                raise ValueError("the column-sharded decode loop needs the glue folded into the multiplies (fused_glue=True, dense FFN)")
            if not emulate_world and not self.g.has_comm:
                raise RuntimeError("Decoder(sharded=True): give the device's context its communicator first (effort_amd.sharded.init_comm / Gpu.comm_create)")
            if not emulate_world and (self.g.comm_world != self.world or self.g.comm_rank != self.rank):
                raise ValueError("Decoder: world / rank differ from the context's communicator")
            from .sharded import ColumnShardedGroups
            self.sharded = True
            self.groups = ColumnShardedGroups(self.world, self.rank, emulate=emulate_world, gpu=self.g)
        f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)      # noqa: E731
        q, kv = cfg.numHeads * cfg.headDim, cfg.numHeadsKV * cfg.headDim
        self.h, self.h_norm, self.fxn, self.outNormed = f(cfg.stateDim), f(cfg.stateDim), f(cfg.stateDim), f(cfg.stateDim)
        self.xq_temp, self.xk_temp, self.xv_temp, self.xq = f(q), f(kv), f(kv), f(q)
        self.attnOutput, self.attnFfnOut, self.ffnOut = f(q), f(cfg.stateDim), f(cfg.stateDim)
        self.x1, self.x3, self.x2 = f(cfg.hiddenDim), f(cfg.hiddenDim), f(cfg.hiddenDim)
        if cfg.numExperts > 1:                                                      # second routed expert + the gate (runNetwork.swift:185-199)
            self.x1b, self.x3b, self.x2b, self.ffnOutB = f(cfg.hiddenDim), f(cfg.hiddenDim), f(cfg.hiddenDim), f(cfg.stateDim)
            self.ffnMix = f(cfg.stateDim)
            self.gateOut, self.gateVals = f(cfg.numExperts), f(2)
            self.gateIdxs = torch.zeros(2, dtype=torch.int32, device=dev)
        self.logits = f(cfg.vocab)
        self.kCache = [f(self.maxTokens, cfg.numHeads, cfg.headDim) for _ in range(cfg.numLayers)]     # xkLayerTokenHead
        self.vCache = [f(self.maxTokens, cfg.numHeads, cfg.headDim) for _ in range(cfg.numLayers)]     # xvLayerToken
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.tokId = torch.zeros(1, dtype=torch.int32, device=dev)
        self.history = torch.zeros(self.maxTokens, dtype=torch.int32, device=dev)
        self._graphs: dict = {}

    # -- one token: everything between fetching the embedding and picking the next token ----------------------------
    def token_step(self, effort: float = 0.25, dense: bool = False):
        cfg, g, lib, m = self.cfg, self.g, _lib.lib(), self.model
        g._bind_stream()
        ck = lambda rc, what: g.check(rc, what)                                     # noqa: E731

        def muls(v, pairs):
            if dense:
                for ew, out in pairs:
                    basicMul(v, ew.core, out)
            elif len(pairs) == 1:
                bucketMul(v, pairs[0][0], None, pairs[0][1], effort)
            else:
                bucketMulGroup([(v, ew, None, out, effort) for ew, out in pairs])

        ck(lib.effort_fetch_row(g.ctx, _p(m.tokEmbeddings), _p(self.tokId), _p(self.h), cfg.stateDim), "fetch_row")
        delta = None
        if self.sharded and not dense:
            G = self.groups
            for n, L in enumerate(m.layers):
                an, fnw = {"norm": L.attnNorm}, {"norm": L.ffnNorm}
                G.mul(self.h, [(L.wq, self.xq_temp, an), (L.wk, self.xk_temp, an), (L.wv, self.xv_temp, an)], effort)      # :121-134
                ck(lib.effort_rope_attention(g.ctx, _p(self.xq_temp), _p(self.xk_temp), _p(self.xv_temp), _p(self.kCache[n]), _p(self.vCache[n]),
                                             _p(self.pos), _p(self.attnOutput), cfg.numHeads, cfg.numHeadsKV, cfg.headDim, self.maxTokens,
                                             C.c_float(cfg.ropeBase)), "rope_attention")
                G.mul(self.attnOutput, [(L.wo, self.h, {"resid": self.h})], effort)                                     # :170-172, in place on h
                G.mul(self.h, [(L.w1, self.x1, fnw), (L.w3, self.x3, fnw)], effort)                                      # :173-179
                G.mul(self.x1, [(L.w2, self.h, {"gate": self.x3, "resid": self.h})], effort)                             # :181-183, in place on h
            ck(lib.effort_add_rmsnorm_mul(g.ctx, _p(self.h), None, _p(m.norm), _p(self.outNormed), cfg.stateDim), "rmsnorm")
            basicMul(self.outNormed, m.output, self.logits)                                           # :222 (replicated: every rank picks the same token)
            ck(lib.effort_argmax(g.ctx, _p(self.logits), cfg.vocab, _p(self.tokId), _p(self.pos), _p(self.history), int(self.history.numel())), "argmax")
            return
        if self.fused_glue and not dense:
            fn, fg, fr = "norm" in self.fuse, "gate" in self.fuse, "resid" in self.fuse
            for n, L in enumerate(m.layers):
                if fn and delta is None:                       # (a pending residual add needs the glue kernel: "norm" folds fully only with "resid")   :121-134
                    bucketMulGroup([(self.h, L.wq, None, self.xq_temp, effort, {"norm": L.attnNorm}),
                                    (self.h, L.wk, None, self.xk_temp, effort, {"norm": L.attnNorm}),
                                    (self.h, L.wv, None, self.xv_temp, effort, {"norm": L.attnNorm})])
                else:
                    ck(lib.effort_add_rmsnorm_mul(g.ctx, _p(self.h), _p(delta), _p(L.attnNorm), _p(self.h_norm), cfg.stateDim), "rmsnorm")
                    delta = None
                    muls(self.h_norm, [(L.wq, self.xq_temp), (L.wk, self.xk_temp), (L.wv, self.xv_temp)])
                ck(lib.effort_rope_attention(g.ctx, _p(self.xq_temp), _p(self.xk_temp), _p(self.xv_temp), _p(self.kCache[n]), _p(self.vCache[n]),
                                             _p(self.pos), _p(self.attnOutput), cfg.numHeads, cfg.numHeadsKV, cfg.headDim, self.maxTokens,
                                             C.c_float(cfg.ropeBase)), "rope_attention")
                if fr:
                    bucketMulGroup([(self.attnOutput, L.wo, None, self.h, effort, {"resid": self.h})])         # :170-172, h += wo(attn)
                    d2 = None
                else:
                    muls(self.attnOutput, [(L.wo, self.attnFfnOut)])
                    d2 = self.attnFfnOut
                if fn and d2 is None:
                    bucketMulGroup([(self.h, L.w1, None, self.x1, effort, {"norm": L.ffnNorm}),              # :173-179
                                    (self.h, L.w3, None, self.x3, effort, {"norm": L.ffnNorm})])
                else:
                    ck(lib.effort_add_rmsnorm_mul(g.ctx, _p(self.h), _p(d2), _p(L.ffnNorm), _p(self.fxn), cfg.stateDim), "rmsnorm")
                    muls(self.fxn, [(L.w1, self.x1), (L.w3, self.x3)])
                if fg:
                    src, extra = self.x1, {"gate": self.x3}
                else:
                    ck(lib.effort_silu_mul(g.ctx, _p(self.x1), _p(self.x3), _p(self.x2), cfg.hiddenDim), "silu")
                    src, extra = self.x2, {}
                if fr:
                    bucketMulGroup([(src, L.w2, None, self.h, effort, dict(extra, resid=self.h))])            # :181-183, h += w2(silu)
                    delta = None
                else:
                    if extra:
                        bucketMulGroup([(src, L.w2, None, self.ffnOut, effort, extra)])
                    else:
                        muls(src, [(L.w2, self.ffnOut)])
                    delta = self.ffnOut
            ck(lib.effort_add_rmsnorm_mul(g.ctx, _p(self.h), _p(delta), _p(m.norm), _p(self.outNormed), cfg.stateDim), "rmsnorm")
            basicMul(self.outNormed, m.output, self.logits)                                           # :222
            ck(lib.effort_argmax(g.ctx, _p(self.logits), cfg.vocab, _p(self.tokId), _p(self.pos), _p(self.history), int(self.history.numel())), "argmax")
            return
        for n, L in enumerate(m.layers):
            ck(lib.effort_add_rmsnorm_mul(g.ctx, _p(self.h), _p(delta), _p(L.attnNorm), _p(self.h_norm), cfg.stateDim), "rmsnorm")
            muls(self.h_norm, [(L.wq, self.xq_temp), (L.wk, self.xk_temp), (L.wv, self.xv_temp)])      # runNetwork.swift:132-134
            if self.fused_attention:
                ck(lib.effort_rope_attention(g.ctx, _p(self.xq_temp), _p(self.xk_temp), _p(self.xv_temp), _p(self.kCache[n]), _p(self.vCache[n]),
                                             _p(self.pos), _p(self.attnOutput), cfg.numHeads, cfg.numHeadsKV, cfg.headDim, self.maxTokens,
                                             C.c_float(cfg.ropeBase)), "rope_attention")
            else:
                ck(lib.effort_rope_kv(g.ctx, _p(self.xq_temp), _p(self.xk_temp), _p(self.xv_temp), _p(self.xq), _p(self.kCache[n]),
                                      _p(self.vCache[n]), _p(self.pos), cfg.numHeads, cfg.numHeadsKV, cfg.headDim, self.maxTokens, C.c_float(cfg.ropeBase)), "rope_kv")
                ck(lib.effort_attention(g.ctx, _p(self.xq), _p(self.kCache[n]), _p(self.vCache[n]), _p(self.pos), _p(self.attnOutput),
                                        cfg.numHeads, cfg.headDim, self.maxTokens), "attention")
            muls(self.attnOutput, [(L.wo, self.attnFfnOut)])                                          # :170
            ck(lib.effort_add_rmsnorm_mul(g.ctx, _p(self.h), _p(self.attnFfnOut), _p(L.ffnNorm), _p(self.fxn), cfg.stateDim), "rmsnorm")
            if L.ffnGate is None:
                muls(self.fxn, [(L.w1, self.x1), (L.w3, self.x3)])                                    # :178-179
                ck(lib.effort_silu_mul(g.ctx, _p(self.x1), _p(self.x3), _p(self.x2), cfg.hiddenDim), "silu")
                muls(self.x2, [(L.w2, self.ffnOut)])                                                  # :182
                delta = self.ffnOut
            else:
                # Mixtral (:185-199): dense gate -> top-2 experts -> softmax of the two; the expert numbers stay on the
                # device (expNo).  Both experts' w1|w3 share the input: ONE grouped launch of four calls; their w2 another.
                basicMul(self.fxn, L.ffnGate, self.gateOut)
                ck(lib.effort_top2_softmax(g.ctx, _p(self.gateOut), cfg.numExperts, _p(self.gateIdxs), _p(self.gateVals)), "top2")
                e0, e1 = self.gateIdxs[0:1], self.gateIdxs[1:2]
                if dense:
                    self._dense_experts(L, e0, e1)
                else:
                    bucketMulGroup([(self.fxn, L.w1, e0, self.x1, effort), (self.fxn, L.w3, e0, self.x3, effort),
                                    (self.fxn, L.w1, e1, self.x1b, effort), (self.fxn, L.w3, e1, self.x3b, effort)])
                    ck(lib.effort_silu_mul(g.ctx, _p(self.x1), _p(self.x3), _p(self.x2), cfg.hiddenDim), "silu")
                    ck(lib.effort_silu_mul(g.ctx, _p(self.x1b), _p(self.x3b), _p(self.x2b), cfg.hiddenDim), "silu")
                    bucketMulGroup([(self.x2, L.w2, e0, self.ffnOut, effort), (self.x2b, L.w2, e1, self.ffnOutB, effort)])
                ck(lib.effort_mix2(g.ctx, _p(self.ffnOut), _p(self.ffnOutB), _p(self.gateVals), _p(self.ffnMix), cfg.stateDim), "mix2")
                delta = self.ffnMix
        ck(lib.effort_add_rmsnorm_mul(g.ctx, _p(self.h), _p(delta), _p(m.norm), _p(self.outNormed), cfg.stateDim), "rmsnorm")
        basicMul(self.outNormed, m.output, self.logits)                                               # :222
        ck(lib.effort_argmax(g.ctx, _p(self.logits), cfg.vocab, _p(self.tokId), _p(self.pos), _p(self.history), int(self.history.numel())), "argmax")

    def _dense_experts(self, L, e0, e1):
        """Dense baseline of the routed FFN: the picked experts' cores are gathered on the device (index_select keeps the
        step graph-capturable), then plain basicMul."""
        cfg, g, lib = self.cfg, self.g, _lib.lib()
        for e, x1, x3, x2, out in ((e0, self.x1, self.x3, self.x2, self.ffnOut), (e1, self.x1b, self.x3b, self.x2b, self.ffnOutB)):
            basicMul(self.fxn, torch.index_select(L.w1.core, 0, e)[0], x1)
            basicMul(self.fxn, torch.index_select(L.w3.core, 0, e)[0], x3)
            g.check(lib.effort_silu_mul(g.ctx, _p(x1), _p(x3), _p(x2), cfg.hiddenDim), "silu")
            basicMul(x2, torch.index_select(L.w2.core, 0, e)[0], out)

    def _graph(self, effort: float, dense: bool):
        key = ("dense",) if dense else (float(effort),)
        if key not in self._graphs:
            self.reset()
            self.token_step(effort, dense)                       # warm (handles, kernel attributes, rocBLAS)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                self.token_step(effort, dense)
            self.g._bind_stream()
            self._graphs[key] = gr
        return self._graphs[key]

    def status(self) -> int:
        """Device-side conditions of the steps since the last call (effort_decode_status): bit 0 = a step ran past the cache or
        the history buffer (it wrote nothing there), bit 1 = argmax over NaN logits.  Reads and clears."""
        st = C.c_int(0)
        self.g._bind_stream()
        self.g.check(_lib.lib().effort_decode_status(self.g.ctx, C.byref(st)), "decode_status")
        return int(st.value)

    def reset(self):
        self.pos.zero_()
        self.tokId.zero_()
        self.history.zero_()

    def run(self, tokenIds: list[int], numTokens: int, effort: float = 0.25, dense: bool = False, forced: bool = False,
            collect_logits: bool = False):
        """runNetwork(tokens:effort:): feed the prompt one token per step, then continue greedily until ``numTokens``
        steps have run.  ``forced``: every step's input comes from ``tokenIds`` (teacher forcing, for the KL measurement).
        Returns (token ids picked at every step, seconds per step measured from the 3rd step on like the reference,
        logits per step if asked)."""
        assert 1 <= len(tokenIds) and numTokens <= self.maxTokens
        gr = self._graph(effort, dense)
        self.reset()
        logits = []
        t0, timed = None, 0
        for step in range(numTokens):
            if step < len(tokenIds):
                self.tokId.fill_(int(tokenIds[step]))            # prompt (or forced) token; otherwise the previous pick stays
            elif forced:
                break
            if step == 2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            gr.replay()
            if collect_logits:
                logits.append(self.logits.clone())
            if step >= 2:
                timed += 1
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / timed if t0 is not None and timed else float("nan")
        steps = min(numTokens, len(tokenIds)) if forced else numTokens
        picked = self.history[:steps].cpu().tolist()
        st = self.status()
        if st:
            raise RuntimeError(f"decode loop: device status {st} (1: a step past maxTokens / the history buffer, 2: NaN logits)")
        return picked, dt, (torch.stack(logits) if collect_logits else None)


def kl_divergence(logits_ref: torch.Tensor, logits_test: torch.Tensor) -> float:
    """mean over positions of KL(softmax(ref) || softmax(test))."""
    a = torch.log_softmax(logits_ref.double(), -1)
    b = torch.log_softmax(logits_test.double(), -1)
    return float((a.exp() * (a - b)).sum(-1).mean())
