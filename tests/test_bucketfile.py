"""On-disk bucket format (effort_amd/bucketfile.py) on the CPU: shard / index naming, dtypes and bit patterns, tensor names
of the converter driver (convert.swift:59-127, q4_convert.py:29-81), partial-row reads.  The conversions themselves are
injected (the real ones need the GPU: tests/test_gpu_parity.py::test_model_file_roundtrip)."""
import json
import os

import numpy as np
import pytest
import torch

from effort_amd import bucketfile as bf


def fake_model(numLayers, hidden=32, ffn=48, kv=16, vocab=10):
    g = torch.Generator().manual_seed(1)
    t = {"model.norm.weight": torch.randn(hidden, generator=g).half(), "lm_head.weight": torch.randn(vocab, hidden, generator=g).half(),
         "model.embed_tokens.weight": torch.randn(vocab, hidden, generator=g).half()}
    for n in range(numLayers):
        p = f"model.layers.{n}."
        t[p + "input_layernorm.weight"] = torch.randn(hidden, generator=g).half()
        t[p + "post_attention_layernorm.weight"] = torch.randn(hidden, generator=g).half()
        for s, o in (("q", hidden), ("k", kv), ("v", kv), ("o", hidden)):
            t[p + f"self_attn.{s}_proj.weight"] = torch.randn(o, hidden, generator=g).half()
        t[p + "mlp.gate_proj.weight"] = torch.randn(ffn, hidden, generator=g).half()
        t[p + "mlp.up_proj.weight"] = torch.randn(ffn, hidden, generator=g).half()
        t[p + "mlp.down_proj.weight"] = torch.randn(hidden, ffn, generator=g).half()
    return t


def fake_fp16(core):
    o, i = core.shape
    words = (torch.arange(16 * i * (o // 16), dtype=torch.int32) % 65536).to(torch.uint16).view(torch.int16).reshape(16 * i, o // 16)
    return {"buckets": words, "bucket.stats": torch.ones(16 * i, 4, dtype=torch.float16), "probes": torch.zeros(4096, dtype=torch.float16)}


def fake_q4(core):
    o, i = core.shape
    return {"buckets": torch.full((8 * i, max(1, o // 32)), -2, dtype=torch.int16), "bucket.stats": torch.ones(8 * i, 2),
            "stats": torch.ones(8 * i, 2), "probes": torch.zeros(min(i, o), dtype=torch.float16), "outliers": torch.ones(3, 4)}


def test_saver_loader_roundtrip(tmp_path):
    s = bf.TensorSaver(str(tmp_path), "buckets-FP16")
    words = torch.tensor([[0x3C01, 0xBC0F], [0x7BFF, 0x0001]], dtype=torch.int32).to(torch.int16)   # bit patterns must survive
    s[0]["a.buckets"] = words
    s[2]["c.stats"] = torch.arange(6, dtype=torch.float32).reshape(3, 2)
    s[2]["c.rows"] = torch.arange(40, dtype=torch.float16).reshape(10, 4)
    index = s.save()
    assert os.path.basename(index) == "buckets-FP16.safetensors.index.json"
    wm = json.load(open(index))["weight_map"]
    assert wm == {"a.buckets": "buckets-FP16-00001-of-00003.safetensors", "c.stats": "buckets-FP16-00003-of-00003.safetensors",
                  "c.rows": "buckets-FP16-00003-of-00003.safetensors"}
    assert os.path.exists(tmp_path / "buckets-FP16-00002-of-00003.safetensors")          # empty shards are written too
    from safetensors import safe_open
    with safe_open(str(tmp_path / wm["a.buckets"]), framework="pt") as f:
        assert f.metadata() == {"description": bf.DESCRIPTION}
        assert f.get_tensor("a.buckets").dtype == torch.float16
    L = bf.TensorLoader(str(tmp_path), "buckets-FP16")
    assert L.hasTensor("a.buckets") and not L.hasTensor("nope") and "c.stats" in L
    assert L["a.buckets"].view(torch.int16).tolist() == words.tolist()
    assert L["c.stats"].dtype == torch.float32 and L.matrix("c.stats", [3, 2]).tolist() == s[2]["c.stats"].tolist()
    assert L.rows("c.rows", 4).tolist() == s[2]["c.rows"][:4].tolist()
    with pytest.raises(AssertionError):
        L.vector("c.stats", [2, 3])
    with pytest.raises(KeyError):
        L["nope"]


def test_convert_mistral_fp16_names(tmp_path):
    """convert.swift:59-127: shard N = layer N, globals in shard 0, all seven projections bucketized, attention cores kept."""
    src = fake_model(2)
    s = bf.convertMistral(src, bf.TensorSaver(str(tmp_path), "buckets-FP16"), numLayers=2, device="cpu", bucketize_fp16=fake_fp16)
    s.save()
    L = bf.TensorLoader(str(tmp_path), "buckets-FP16")
    want = {"model.norm", "output.core", "tok_embeddings.core"}
    for n in range(2):
        want |= {f"layers.{n}.attention_norm", f"layers.{n}.ffn_norm"}
        for w in "qkvo":
            want |= {f"layers.{n}.attention.w{w}.{k}" for k in ("buckets", "bucket.stats", "probes", "core")}
        for w in "123":
            want |= {f"layers.{n}.feed_forward.experts.0.w{w}.{k}" for k in ("buckets", "bucket.stats", "probes")}
    assert set(L.keys()) == want
    assert L.index["model.norm"] == L.index["layers.0.attention.wq.buckets"] == "buckets-FP16-00001-of-00002.safetensors"
    assert L.index["layers.1.ffn_norm"] == "buckets-FP16-00002-of-00002.safetensors"
    assert torch.equal(L["layers.1.attention.wk.core"], src["model.layers.1.self_attn.k_proj.weight"])
    assert torch.equal(L["layers.0.feed_forward.experts.0.w2.buckets"].view(torch.int16), fake_fp16(src["model.layers.0.mlp.down_proj.weight"])["buckets"])


def test_convert_mistral_q4_names(tmp_path):
    """q4_convert.py:29-81: shard 0 = globals, shard N+1 = layer N, only wq and the FFN bucketized, every core kept."""
    src = fake_model(1)
    s = bf.convertMistral(src, bf.TensorSaver(str(tmp_path), "model", pad_total=False), numLayers=1, q4=True, device="cpu", bucketize_q4=fake_q4)
    s.save()
    L = bf.TensorLoader(str(tmp_path), "model")
    assert L.index["model.norm"] == "model-00001-of-2.safetensors" and L.index["layers.0.attention_norm"] == "model-00002-of-2.safetensors"
    keys = set(L.keys())
    for w in "kvo":
        assert {k for k in keys if k.startswith(f"layers.0.attention.w{w}.")} == {f"layers.0.attention.w{w}.core"}
    for pre in ("layers.0.attention.wq.", "layers.0.feed_forward.experts.0.w1.", "layers.0.feed_forward.experts.0.w3."):
        assert {k[len(pre):] for k in keys if k.startswith(pre)} == {"core", "buckets", "bucket.stats", "probes", "outliers"}
    assert L["layers.0.feed_forward.experts.0.w2.outliers"].dtype == torch.float32
    assert L["layers.0.attention.wq.buckets"].dtype == torch.float16 and (L["layers.0.attention.wq.buckets"].view(torch.int16) == -2).all()
