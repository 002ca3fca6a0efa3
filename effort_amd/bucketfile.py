"""On-disk bucket format and the whole-model converter driver.

Mirrors, for the hot path's data format:

* ``TensorSaver`` / ``TensorLoader`` -- helpers/safetensors.swift:22-83,87-216: a model is a set of safetensors shards
  ``<model>-%05d-of-%05d.safetensors`` plus ``<model>.safetensors.index.json`` = ``{"weight_map": {tensor: shard}}``;
  every shard carries ``__metadata__.description`` (helpers/safetensors.swift:256).  16-bit tensors are stored as
  ``F16`` (bucket words keep their bit pattern: position bits / nibbles are part of it), f32 tensors as ``F32``.
* ``convertMistral`` -- convert.swift:59-127 (FP16: every attention and FFN projection bucketized, attention cores
  kept) and q4_convert.py:29-81 (Q4: wq and w1..w3 bucketized, every core kept): tensor names
  ``layers.N.attention.w{q,k,v,o}.{buckets,bucket.stats,probes,core}``,
  ``layers.N.feed_forward.experts.0.w{1,2,3}.{buckets,bucket.stats,probes[,outliers,core]}``,
  ``layers.N.{attention_norm,ffn_norm}``, ``model.norm``, ``output.core``, ``tok_embeddings.core``.
  The reference loops ``inDim`` bitonic launches per matrix from the host (minutes per layer); here a matrix is one
  GPU converter call (effort_convert_fp16: 6 ms for 4096x11008) or one batched tensor program (Q4).
* ``loadExpertWeights`` -- ``ExpertWeights.init(elName:)`` and ``init(prefix, wId, inDim:outDim:numExperts:percentLoad:)``
  (loader.swift:60-167): single bundles, Mixtral-style expert stacks, and ``percentLoad`` < 16 = only the first
  ``inDim*percentLoad`` bucket rows of every expert are read (``copyFrom(..., mySize: true)``, loader.swift:157-159).
"""
from __future__ import annotations

import json
import os
from typing import Callable, Mapping

import torch

DESCRIPTION = "Bucket weights format, see mixtral-kolinko at github"      # helpers/safetensors.swift:256


def _as_stored(t: torch.Tensor) -> torch.Tensor:
    """The dtype a tensor is written with: 16-bit words as F16 (bit pattern preserved), everything else F32."""
    t = t.detach()
    if t.dtype in (torch.int16, torch.uint16):
        t = t.view(torch.float16)
    elif t.dtype not in (torch.float16, torch.float32):
        t = t.to(torch.float32)
    return t.contiguous().cpu()


class TensorSaver:
    """helpers/safetensors.swift:22-83: ``saver[i]`` is the dict of shard i; ``save()`` writes shards + index."""

    def __init__(self, path: str, model: str = "model", pad_total: bool = True):
        self.path, self.model, self.pad_total = path, model, pad_total
        self.files: list[dict[str, torch.Tensor]] = []

    def __getitem__(self, index: int) -> dict:
        while len(self.files) <= index:
            self.files.append({})
        return self.files[index]

    def __setitem__(self, index: int, val: dict):
        while len(self.files) <= index:
            self.files.append({})
        self.files[index] = val

    def _fname(self, i: int) -> str:
        total = f"{len(self.files):05d}" if self.pad_total else f"{len(self.files)}"      # q4_convert.py:70 leaves it unpadded
        return f"{self.model}-{i + 1:05d}-of-{total}.safetensors"

    def save(self) -> str:
        from safetensors.torch import save_file
        os.makedirs(self.path, exist_ok=True)
        weight_map = {}
        for i, tensors in enumerate(self.files):
            save_file({k: _as_stored(v) for k, v in tensors.items()}, os.path.join(self.path, self._fname(i)),
                      metadata={"description": DESCRIPTION})
            for k in tensors:
                weight_map[k] = self._fname(i)
        index = os.path.join(self.path, f"{self.model}.safetensors.index.json")
        with open(index, "w") as f:
            json.dump({"weight_map": weight_map}, f, indent=2)
        return index


class TensorLoader:
    """helpers/safetensors.swift:87-216: index-driven, uncached tensor access (``loader[name]``, ``hasTensor``)."""

    def __init__(self, path: str = "./", model: str = "model", device: str | torch.device = "cpu"):
        self.path, self.device = path, device
        with open(os.path.join(path, f"{model}.safetensors.index.json")) as f:
            self.index: dict[str, str] = json.load(f)["weight_map"]

    def hasTensor(self, name: str) -> bool:
        return name in self.index

    def __contains__(self, name: str) -> bool:
        return name in self.index

    def keys(self):
        return self.index.keys()

    def __getitem__(self, name: str) -> torch.Tensor:
        from safetensors import safe_open
        if name not in self.index:
            raise KeyError(name)
        with safe_open(os.path.join(self.path, self.index[name]), framework="pt", device="cpu") as f:
            t = f.get_tensor(name)
        return t.to(self.device)

    def rows(self, name: str, n: int) -> torch.Tensor:
        """The first ``n`` rows only (copyFrom(mySize: true)): a partial read, not load-then-slice."""
        from safetensors import safe_open
        with safe_open(os.path.join(self.path, self.index[name]), framework="pt", device="cpu") as f:
            t = f.get_slice(name)[:n]
        return t.to(self.device)

    def vector(self, name: str, assertShape=None) -> torch.Tensor:
        t = self[name]
        assert assertShape is None or list(t.shape) == list(assertShape), f"wrong shape loaded! {name} has shape {list(t.shape)}"
        return t

    matrix = vector


# ------------------------------------------------------------------------------------------------ conversion driver
def _bucketize_fp16(core: torch.Tensor) -> dict:
    from .convert import bucketize
    out: dict[str, torch.Tensor] = {}
    bucketize(core, "", out, goQ8=False)                # GPU converter, convert.swift:209-260
    return out


def _bucketize_q4(core: torch.Tensor) -> dict:
    from .q4 import convert
    return convert(core.t().contiguous())                # q4_convert.py:54,63: convert(W.T)


def convertMistral(tensors: Mapping[str, torch.Tensor], saver: TensorSaver, numLayers: int = 32, q4: bool = False,
                   device: str | torch.device = "cuda", bucketize_fp16: Callable = _bucketize_fp16,
                   bucketize_q4: Callable = _bucketize_q4, log: Callable = lambda *a: None) -> TensorSaver:
    """HF Mistral tensors (``model.layers.N.self_attn.q_proj.weight`` ...) -> bucketed model in ``saver``.

    FP16 follows convert.swift:59-127 (shard N holds layer N; the globals ride in shard 0); Q4 follows
    q4_convert.py:29-81 (shard 0 holds the globals, shard N+1 layer N; only wq and the FFN are bucketized, ``stats``
    is dropped as the reference's loader never reads it).  Call ``saver.save()`` afterwards."""
    def dev16(name):
        return tensors[name].to(device=device, dtype=torch.float16)

    glob = {"model.norm": tensors["model.norm.weight"], "output.core": tensors["lm_head.weight"],
            "tok_embeddings.core": tensors["model.embed_tokens.weight"]}
    if q4:
        saver[0] = dict(glob)
    for layerNo in range(numLayers):
        log(f"converting Mistral's layer {layerNo}")
        out = saver[layerNo + 1] if q4 else saver[layerNo]
        if not q4 and layerNo == 0:
            out.update(glob)
        pre = f"model.layers.{layerNo}."
        out[f"layers.{layerNo}.attention_norm"] = tensors[pre + "input_layernorm.weight"]
        out[f"layers.{layerNo}.ffn_norm"] = tensors[pre + "post_attention_layernorm.weight"]
        for s in ("k", "o", "q", "v"):
            old = pre + f"self_attn.{s}_proj.weight"
            new = f"layers.{layerNo}.attention.w{s}."
            out[new + "core"] = tensors[old]
            if not q4:
                for k, t in bucketize_fp16(dev16(old)).items():
                    out[new + k] = t
            elif s == "q":                                    # q4_convert.py:57: only wq among the attention matrices
                for k, t in bucketize_q4(dev16(old)).items():
                    if k != "stats":
                        out[new + k] = t
        for oldName, newName in (("gate_proj", "w1"), ("down_proj", "w2"), ("up_proj", "w3")):
            old = pre + f"mlp.{oldName}.weight"
            new = f"layers.{layerNo}.feed_forward.experts.0.{newName}."
            if q4:
                out[new + "core"] = tensors[old]
                for k, t in bucketize_q4(dev16(old)).items():
                    if k != "stats":
                        out[new + k] = t
            else:
                for k, t in bucketize_fp16(dev16(old)).items():
                    out[new + k] = t
        for k in list(out):
            out[k] = _as_stored(out[k])                       # off the GPU: a 7B model does not have to fit at once
    return saver


# ------------------------------------------------------------------------------------------------ loading
def loadExpertWeights(loader: TensorLoader, prefix: str, wId: str | None = None, *, inDim: int | None = None,
                      outDim: int | None = None, numExperts: int = 1, percentLoad: int | None = None, q4: bool = False,
                      device: str | torch.device = "cuda"):
    """``ExpertWeights(elName:)`` when ``wId`` is None (names ``prefix.{core,probes,buckets,bucket.stats,outliers}``,
    shapes from the core, loader.swift:60-111) else ``ExpertWeights(prefix, wId, inDim:outDim:numExperts:percentLoad:)``
    (names ``prefix{e}.{wId}.*``, experts stacked, only ``inDim*percentLoad`` rows per expert read, :113-166)."""
    from .weights import ExpertWeights
    full = 8 if q4 else 16
    pl = full if percentLoad is None else int(percentLoad)
    assert 1 <= pl <= full and (not q4 or pl == 8), "percentLoad: 1..16 rank slices (Q4 bundles are loaded whole)"
    names = [prefix + "."] if wId is None else [f"{prefix}{e}.{wId}." for e in range(numExperts)]
    core = loader[names[0] + "core"].to(device) if loader.hasTensor(names[0] + "core") else None
    if inDim is None or outDim is None:
        assert core is not None, "shapes come from the core matrix (loader.swift:61-63) or must be given"
        outDim, inDim = core.shape
    outliers = loader[names[0] + "outliers"].to(device) if loader.hasTensor(names[0] + "outliers") else None
    if not loader.hasTensor(names[0] + "probes"):
        raise KeyError(f"buckets not loaded for {names[0]} (loader.swift:104-107): dense fallback only")
    rows = inDim * pl
    probes = torch.stack([loader[n + "probes"].to(device)[:4096] for n in names])
    buckets = torch.stack([loader.rows(n + "buckets", rows).to(device) for n in names])
    stats = torch.stack([loader.rows(n + "bucket.stats", rows).to(device) for n in names])
    return ExpertWeights(buckets, stats, probes, inSize=inDim, outSize=outDim, percentLoad=pl, numExperts=len(names),
                         outliers=outliers, core=core, q4=q4)
