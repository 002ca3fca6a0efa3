"""Decode loop (effort_amd/decode.py + csrc/decode.hip) against a plain PyTorch fp32 restatement of runNetwork.swift's
math on the same random weights.  The Swift loop cannot run here, so this row is pinned by our own dense path:
bars -- dense path: logits within 2e-3 * max|logit| of the torch reference at every step and identical greedy tokens;
effort 1.0 (bucketMul over all rows = the dense product up to the position bits left in the weights' low mantissa):
cos-sim of the logits > 0.999 and identical greedy tokens on this model."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def torch_reference(model, tokenIds, numTokens):
    """fp32 PyTorch restatement (runNetwork.swift:104-222, aux.metal glue): returns (picked ids, logits per step)."""
    cfg = model.cfg
    f = lambda t: t.float()                                                     # noqa: E731
    half = cfg.headDim // 2
    freqs = torch.tensor([math.exp(math.log(cfg.ropeBase) * (-j / half)) for j in range(half)], dtype=torch.float32, device=DEV)
    kc = [[] for _ in model.layers]
    vc = [[] for _ in model.layers]
    picked, logits_all = [], []
    tok = tokenIds[0]

    def rms(x, w):
        return x / torch.sqrt((x * x).mean() + 1e-5) * f(w)

    def rope(x, pos):                                                             # [heads, headDim], rotate_half convention
        ang = pos * freqs
        c, s = torch.cos(ang).repeat(2), torch.sin(ang).repeat(2)
        rot = torch.cat([-x[:, half:], x[:, :half]], dim=1)
        return x * c + rot * s

    for step in range(numTokens):
        if step < len(tokenIds):
            tok = tokenIds[step]
        h = f(model.tokEmbeddings[tok])
        for n, L in enumerate(model.layers):
            hn = rms(h, L.attnNorm).half().float()                               # basicMul feeds v.asFloat16() (helpers/mps.swift:19)
            xq = (f(L.wq.core) @ hn).view(cfg.numHeads, cfg.headDim)
            xk = (f(L.wk.core) @ hn).view(cfg.numHeadsKV, cfg.headDim).repeat_interleave(cfg.numHeads // cfg.numHeadsKV, 0)
            xv = (f(L.wv.core) @ hn).view(cfg.numHeadsKV, cfg.headDim).repeat_interleave(cfg.numHeads // cfg.numHeadsKV, 0)
            kc[n].append(rope(xk, step))
            vc[n].append(xv)
            q = rope(xq, step)
            K, V = torch.stack(kc[n]), torch.stack(vc[n])                          # [T, heads, dim]
            sc = torch.einsum("hd,thd->ht", q, K) / math.sqrt(cfg.headDim)
            p = torch.softmax(sc, dim=1)
            att = torch.einsum("ht,thd->hd", p, V).reshape(-1)
            h = h + f(L.wo.core) @ att.half().float()
            fx = rms(h, L.ffnNorm).half().float()
            if L.ffnGate is None:
                x1, x3 = f(L.w1.core) @ fx, f(L.w3.core) @ fx
                x2 = x3 * x1 / (1 + torch.exp(-x1))
                h = h + f(L.w2.core) @ x2.half().float()
            else:                                                                  # Mixtral routing, runNetwork.swift:185-199
                gate = f(L.ffnGate) @ fx
                vals, idxs = torch.topk(gate, 2)
                w = torch.softmax(vals, 0)
                mix = 0
                for k in range(2):
                    e = int(idxs[k])
                    x1, x3 = f(L.w1.core[e]) @ fx, f(L.w3.core[e]) @ fx
                    x2 = x3 * x1 / (1 + torch.exp(-x1))
                    mix = mix + w[k] * (f(L.w2.core[e]) @ x2.half().float())
                h = h + mix
        lg = f(model.output) @ rms(h, model.norm).half().float()
        logits_all.append(lg)
        tok = int(torch.argmax(lg))
        picked.append(tok)
    return picked, torch.stack(logits_all)


@pytest.fixture(scope="module")
def small_model(hip_lib_built):
    import effort_amd  # noqa: F401
    from effort_amd.decode import MistralConfig, Model
    cfg = MistralConfig(stateDim=4096, hiddenDim=4096, numLayers=2, numHeads=32, numHeadsKV=8, headDim=128, vocab=512)
    return Model.random(cfg, seed=5)


def test_dense_path_matches_torch_reference(small_model):
    from effort_amd.decode import Decoder
    prompt, steps = [3, 77, 130], 10
    want_ids, want_logits = torch_reference(small_model, prompt, steps)
    dec = Decoder(small_model, maxTokens=16)
    ids, dt, logits = dec.run(prompt, steps, dense=True, collect_logits=True)
    assert ids == want_ids
    err = float((logits - want_logits).abs().max() / want_logits.abs().max())
    assert err < 2e-3, err
    ids2, _, logits2 = dec.run(prompt, steps, dense=True, collect_logits=True)        # replay after reset: same bits
    assert ids2 == ids and torch.equal(logits, logits2)
    two = Decoder(small_model, maxTokens=16, fused_attention=False)                     # rope_kv + attention as two launches
    ids3, _, logits3 = two.run(prompt, steps, dense=True, collect_logits=True)
    # different f32 summation orders, then f16 rounding of the vectors fed to rocBLAS: same bar as against the reference
    assert ids3 == ids and float((logits3 - logits).abs().max() / logits.abs().max()) < 2e-3
    assert float((logits3 - want_logits).abs().max() / want_logits.abs().max()) < 2e-3


def test_effort_one_tracks_dense_and_low_effort_degrades(small_model):
    from effort_amd.decode import Decoder, kl_divergence
    prompt, steps = [3, 77, 130], 10
    dec = Decoder(small_model, maxTokens=16, fused_glue=False)
    ids_d, _, lg_d = dec.run(prompt, steps, dense=True, collect_logits=True)
    forced = prompt + ids_d[len(prompt) - 1:-1]                                   # the inputs the dense run saw
    ids_1, dt, lg_1 = dec.run(forced, steps, effort=1.0, forced=True, collect_logits=True)
    cos = torch.nn.functional.cosine_similarity(lg_1, lg_d, dim=1)
    assert float(cos.min()) > 0.999, cos
    assert ids_1 == ids_d
    kl_1 = kl_divergence(lg_d, lg_1)
    _, _, lg_q = dec.run(forced, steps, effort=0.25, forced=True, collect_logits=True)
    kl_q = kl_divergence(lg_d, lg_q)
    assert 0.0 <= kl_1 < kl_q, (kl_1, kl_q)                                       # less effort, further from dense
    ids_g, dt_g, _ = dec.run(prompt, steps, effort=1.0)                             # free-running greedy at effort 1
    assert ids_g == ids_d and dt_g > 0
    # the same loop with rmsNorm, silu and the residual adds folded into the multiplies: same tokens, logits to rounding
    sep = Decoder(small_model, maxTokens=16, fused_glue=True)                       # the loop with the glue folded into the multiplies (the default since round 3)
    assert sep.fused_glue and not dec.fused_glue and Decoder(small_model, maxTokens=16).fused_glue
    ids_s, _, lg_s = sep.run(forced, steps, effort=1.0, forced=True, collect_logits=True)
    assert ids_s == ids_1 and float((lg_s - lg_1).abs().max() / lg_1.abs().max()) < 2e-3


def test_mixtral_routing(hip_lib_built):
    """numExperts > 1: dense gate -> top-2 experts -> softmax weights, experts picked on the device through expNo
    (runNetwork.swift:185-199).  Dense path vs the torch restatement; effort 1.0 vs dense."""
    from effort_amd.decode import Decoder, MistralConfig, Model
    cfg = MistralConfig(stateDim=4096, hiddenDim=4096, numLayers=2, numHeads=32, numHeadsKV=8, headDim=128, vocab=512, numExperts=4)
    model = Model.random(cfg, seed=9)
    prompt, steps = [5, 9], 8
    want_ids, want_logits = torch_reference(model, prompt, steps)
    dec = Decoder(model, maxTokens=16)
    ids, _, logits = dec.run(prompt, steps, dense=True, collect_logits=True)
    assert ids == want_ids
    assert float((logits - want_logits).abs().max() / want_logits.abs().max()) < 2e-3
    forced = prompt + ids[len(prompt) - 1:-1]
    ids_1, _, lg_1 = dec.run(forced, steps, effort=1.0, forced=True, collect_logits=True)
    assert float(torch.nn.functional.cosine_similarity(lg_1, logits, dim=1).min()) > 0.999
    assert ids_1 == ids


def test_model_load_from_bucket_files(hip_lib_built, tmp_path):
    """HF-named tensors -> convertMistral -> shards on disk -> Model.load -> the decode loop gives the same logits, bit for
    bit, as a model bucketized in memory from the same matrices (same converter, same layout, same kernels)."""
    from effort_amd import bucketfile as bf
    from effort_amd.decode import Decoder, Layer, MistralConfig, Model
    from effort_amd.weights import ExpertWeights
    cfg = MistralConfig(stateDim=4096, hiddenDim=4096, numLayers=1, numHeads=32, numHeadsKV=8, headDim=128, vocab=64)
    g = torch.Generator().manual_seed(3)
    mat = lambda o, i: (torch.randn(o, i, generator=g) * 0.02).half()                    # noqa: E731
    vec = lambda n: (1 + 0.1 * torch.randn(n, generator=g)).half()                       # noqa: E731
    kv = cfg.numHeadsKV * cfg.headDim
    src = {"model.norm.weight": vec(4096), "lm_head.weight": mat(cfg.vocab, 4096), "model.embed_tokens.weight": torch.randn(cfg.vocab, 4096, generator=g).half(),
           "model.layers.0.input_layernorm.weight": vec(4096), "model.layers.0.post_attention_layernorm.weight": vec(4096),
           "model.layers.0.self_attn.q_proj.weight": mat(4096, 4096), "model.layers.0.self_attn.k_proj.weight": mat(kv, 4096),
           "model.layers.0.self_attn.v_proj.weight": mat(kv, 4096), "model.layers.0.self_attn.o_proj.weight": mat(4096, 4096),
           "model.layers.0.mlp.gate_proj.weight": mat(4096, 4096), "model.layers.0.mlp.up_proj.weight": mat(4096, 4096),
           "model.layers.0.mlp.down_proj.weight": mat(4096, 4096)}
    bf.convertMistral(src, bf.TensorSaver(str(tmp_path), "buckets-FP16"), numLayers=1).save()
    loaded = Model.load(bf.TensorLoader(str(tmp_path), "buckets-FP16"), cfg)
    direct = Model(cfg)
    L = Layer()
    L.attnNorm, L.ffnNorm, L.ffnGate = src["model.layers.0.input_layernorm.weight"].to(DEV), src["model.layers.0.post_attention_layernorm.weight"].to(DEV), None
    for name, key in (("wq", "self_attn.q_proj"), ("wk", "self_attn.k_proj"), ("wv", "self_attn.v_proj"), ("wo", "self_attn.o_proj"),
                      ("w1", "mlp.gate_proj"), ("w3", "mlp.up_proj"), ("w2", "mlp.down_proj")):
        setattr(L, name, ExpertWeights.from_core(src[f"model.layers.0.{key}.weight"].to(DEV)))
    direct.layers.append(L)
    direct.norm, direct.output, direct.tokEmbeddings = src["model.norm.weight"].to(DEV), src["lm_head.weight"].to(DEV), src["model.embed_tokens.weight"].to(DEV)
    a = Decoder(loaded, maxTokens=16).run([1, 2, 3], 8, effort=0.5, collect_logits=True)
    b = Decoder(direct, maxTokens=16).run([1, 2, 3], 8, effort=0.5, collect_logits=True)
    assert a[0] == b[0] and torch.equal(a[2], b[2])
    assert loaded.layers[0].w1.core is None and loaded.layers[0].wq.core is not None              # convert.swift keeps attention cores only


@pytest.mark.parametrize("fused_attention", [True, False])
def test_steps_past_the_cache_write_nothing_and_are_reported(small_model, fused_attention):
    """A token step at pos >= maxTokens (a graph replayed once too often, or token_step called directly) must not write past
    the key/value cache or the history buffer: the glue skips the writes and raises the context's decode status."""
    from effort_amd.decode import Decoder
    dec = Decoder(small_model, maxTokens=4, fused_attention=fused_attention)
    ids, _, _ = dec.run([3, 77], 4, dense=True)
    assert len(ids) == 4 and dec.status() == 0
    guard_k = [k.clone() for k in dec.kCache]
    hist = dec.history.clone()
    dec.token_step(0.25, True)                          # pos == maxTokens now
    dec.g.eval()
    assert dec.status() == 1 and dec.status() == 0      # reported once, then cleared
    assert all(torch.equal(a, b) for a, b in zip(guard_k, dec.kCache)) and torch.equal(hist, dec.history)
    dec.logits.fill_(float("nan"))                      # argmax over NaN: a valid token id all the same
    import ctypes as C
    from effort_amd import _lib
    p = lambda t: C.c_void_p(t.data_ptr())              # noqa: E731
    dec.pos.zero_()
    dec.g.check(_lib.lib().effort_argmax(dec.g.ctx, p(dec.logits), small_model.cfg.vocab, p(dec.tokId), p(dec.pos), p(dec.history), 4), "argmax")
    dec.g.eval()
    assert int(dec.tokId.item()) == 0 and dec.status() == 2


def test_in_graph_multiplies_match_the_oracle(small_model, oracle_cpu):
    """The effort path of the decode loop against the CPU oracle, multiply by multiply: after a replayed token step the
    decoder's buffers hold the LAST layer's inputs and outputs -- h_norm -> xq|xk|xv (the grouped launch), attnOutput -> wo,
    fxn -> x1|x3, x2 -> w2 -- and every one of those seven in-graph multiplies must be the oracle's bucketMul of the GPU's
    own input vector (same cutoff and rows, or the outputs would be far apart; products within the multiply's tolerance).
    Then the same with rmsNorm folded into the launches (fused_glue): the fused launch's input is what the glue kernel
    writes, bit for bit, so the oracle is fed that."""
    import numpy as np

    from effort_amd.decode import Decoder
    from tests.test_gpu_parity import close
    cfg, L = small_model.cfg, small_model.layers[-1]

    def host(ew):
        return (ew.buckets[0].contiguous().cpu().numpy().view(np.uint16), ew.stats[0].cpu().numpy().view(np.uint16),
                ew.probes[0].cpu().numpy().view(np.uint16))

    def check(ew, vin, out, effort, what):
        want, n, cutoff = oracle_cpu.bucket_mul(vin.cpu().numpy(), *host(ew), ew.inSize, ew.outSize, effort)
        assert close(out.cpu().numpy(), want), what
        return n
    for effort in (0.25, 0.6):
        dec = Decoder(small_model, maxTokens=16, fused_glue=False)
        dec.run([3, 77, 130, 9], 4, effort=effort, forced=True)                  # four replayed steps; the buffers hold the last one
        check(L.wq, dec.h_norm, dec.xq_temp, effort, "wq")
        check(L.wk, dec.h_norm, dec.xk_temp, effort, "wk")
        check(L.wv, dec.h_norm, dec.xv_temp, effort, "wv")
        check(L.wo, dec.attnOutput, dec.attnFfnOut, effort, "wo")
        check(L.w1, dec.fxn, dec.x1, effort, "w1")
        check(L.w3, dec.fxn, dec.x3, effort, "w3")
        n2 = check(L.w2, dec.x2, dec.ffnOut, effort, "w2")
        assert dec.g.last_dispatch_count() == n2                                 # the last multiply launch of the step: exact row count
        # rmsNorm and the residual adds folded into the launches: the same single f32 add per element and the glue kernel's own
        # summation order, so the fused launches see the same input bits
        fus = Decoder(small_model, maxTokens=16, fused_glue=("norm", "resid"))
        fus.run([3, 77, 130, 9], 4, effort=effort, forced=True)
        assert torch.equal(fus.x1, dec.x1) and torch.equal(fus.x3, dec.x3) and torch.equal(fus.logits, dec.logits)   # same input bits, same everything


def test_column_sharded_decoder(small_model):
    """BASELINE config 4's product path as far as one GPU reaches: the decode loop with every bundle column-sharded
    (Decoder(world=G): ColumnShardedGroups, one effort_allgather_outputs per launch group, in place on h for wo / w2).
    (a) a world of ONE through RCCL (effort_comm_create on the device's context): shards are whole-matrix views, the launch
    geometry is the unsharded one -- logits bit-identical with the plain decoder, from a hipGraph with the collectives in it;
    (b) worlds of 2 and 8 EMULATED in this process (every rank's launches, their slices written where the gather puts them):
    same tokens on teacher-forced inputs, logits within the multiply's bar (a shard's slicing -- its rounding grid -- differs,
    and a 2e-5 difference can flip a row selection downstream: hence 2e-3 on the logits, the bar of the other decode tests)."""
    import effort_amd as ea
    from effort_amd.decode import Decoder
    prompt, steps = [3, 77, 130], 8
    plain = Decoder(small_model, maxTokens=16)
    ids_p, _, lg_p = plain.run(prompt, steps, effort=0.5, collect_logits=True)
    forced = prompt + ids_p[len(prompt) - 1:-1]
    g = ea.gpu(0)
    with pytest.raises(RuntimeError):
        Decoder(small_model, maxTokens=16, sharded=True)                                # no communicator yet
    g.comm_create(0, 1, ea.Gpu.comm_unique_id())
    try:
        one = Decoder(small_model, maxTokens=16, world=1, rank=0, sharded=True)
        assert one.sharded and one.groups is not None
        ids_1, _, lg_1 = one.run(prompt, steps, effort=0.5, collect_logits=True)
        assert ids_1 == ids_p and torch.equal(lg_1, lg_p)
        assert one.status() == 0
    finally:
        g.comm_destroy()
    for world in (2, 8):
        emu = Decoder(small_model, maxTokens=16, world=world, emulate_world=True)
        _, _, lg_e = emu.run(forced, steps, effort=0.5, forced=True, collect_logits=True)
        assert float((lg_e - lg_p).abs().max() / lg_p.abs().max()) < 2e-3, world
        assert lg_e.argmax(-1).tolist() == lg_p.argmax(-1).tolist(), world
    with pytest.raises(ValueError):
        Decoder(small_model, maxTokens=16, world=2, emulate_world=True, fused_glue=("norm",))
