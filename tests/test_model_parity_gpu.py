"""The patched model at BASELINE.json's sizes against the CPU oracle (model-level parity had only been checked
at 4096 tokens x 2 layers, and the deferred schedules only against each other).

configs[1] / configs[3]: the Llama-3-8B and the Mistral-7B geometry, a 32,768-token prompt, budget 2048, TSP layer 15, TSP length
2048 -- at 17 layers, so that the TSP layer closes the second group of eight of the default schedule and one layer runs on the 2048
survivors behind it.  For BOTH schedules the bench times -- the product default (FASTKV_DEFER=1, FASTKV_DEFER_HOLD=8: groups of
eight 32k layers through `fastkv_update_kv_ptrs_f16`, the post-TSP layer flushed at the end) and the reference's call pattern
(FASTKV_DEFER=0: one `update_kv` per layer inside the attention forward, /root/reference/baselines/fastkv/llama_model.py:136-145)
-- every layer's K, V and query WINDOW rows (the only query rows the operator reads, utils.py:93) are captured where the attention
module hands them over, replayed through `OracleFastKVCluster` on the CPU, and compared bit for bit with
  * the rows that ended up in every layer's cache (what decode will attend over),
  * `tsp_idx` of the TSP layer (llama_model.py:252-254),
  * the rewired position ids (llama_model.py:254, :368-371) and the length of the sequence behind the TSP layer.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

S, LAYERS, TSP_IDX, BUDGET, TSP_LEN, W = 32768, 17, 15, 2048, 2048, 8


def _build(family, monkeypatch, defer, extra=(), S=S, LAYERS=LAYERS, TSP_IDX=TSP_IDX):
    from baselines.monkeypatch import replace_llama, replace_mistral, set_model
    from benchmark import prefill
    monkeypatch.setenv("FASTKV_DEFER", defer)
    monkeypatch.setenv("FASTKV_DEFER_HOLD", "8")
    monkeypatch.setenv("FASTKV_SLAB_CACHE", "0")
    name = {"llama": "llama3-8b", "mistral": "mistral-7b", "ministral": "ministral-8b"}[family]
    a = prefill.parse_args(["--model_path", name, "--num_layers", str(LAYERS), "--device", "cuda", "--save_txt", "", "--method", "fastkv",
                            "--max_capacity_prompts", str(BUDGET), "--tsp_len", str(TSP_LEN), "--tsp_idx", str(TSP_IDX),
                            "--pooling", "maxpool", *extra])
    a.save_txt = False
    a.context_lengths = [S]
    (replace_llama if family == "llama" else replace_mistral)("fastkv")             # (Ministral-8B is a MistralForCausalLM)
    torch.manual_seed(101)
    model = prefill.build_model(a, "cuda")
    set_model(model, a)
    return model, a


def _cfg_of(cl):
    return dict(window_size=cl.window_size, max_capacity_prompt=cl.max_capacity_prompt, kernel_size=cl.kernel_size, pooling=cl.pooling,
                tsp_layer=cl.tsp_layer, tsp_length=cl.tsp_length, tsp_rate=cl.tsp_rate, retain_rate=cl.retain_rate,
                eviction_mode=cl.eviction_mode)


def _full_q(q_win, S_):
    """[B,H,S,D] logical query tensor with the captured window rows at its end (the oracle, like the reference, indexes the last
    W rows of the whole tensor); [B,S,H,D] storage as the attention module produces it.  The other rows are never read."""
    B, H, Wn, D = q_win.shape
    full = torch.zeros(B, S_, H, D, dtype=q_win.dtype).transpose(1, 2)
    full[:, :, S_ - Wn:] = q_win
    return full


@pytest.mark.parametrize("defer", ["1", "0"])
@pytest.mark.parametrize("family", ["llama", "mistral"])
def test_32k_prefill_every_layer_against_the_oracle(family, defer, monkeypatch):
    _prefill_against_the_oracle(family, defer, monkeypatch, S, LAYERS, TSP_IDX, [8, 8, 1])


# The reference's PUBLISHED recipe -- the only setting its scripts ship (/root/reference/scripts/eval_prefill.sh:4-12: Llama-3.1-8B,
# `--tsp_idx 15 --tsp_rate 0.2 --retain_rate 0.1 --eviction_mode proportional`; /root/reference/scripts2/eval_prefill.sh:37-47:
# Ministral-8B-Instruct-2410, `--tsp_idx 17`, same rates; pooling / window / kernel at the harness defaults maxpool / 8 / 7).  At 32,768
# tokens: every layer up to the TSP layer keeps int(32768 * 0.1) = 3276 rows, the TSP layer passes int(32768 * 0.2) = 6553 tokens on,
# and the layers behind it -- retain_rate 0.1 / 0.2 = 0.5 (utils.py:41-46) -- keep int(6553 * 0.5) = 3276 of 6553.  Unlike the constant
# budget, the layers behind the TSP layer are LONGER than FASTKV_DEFER_MAX_LEN: the default schedule runs them as "long" layers in
# groups of eight (cluster.py DeferredCompression.add) under FASTKV_DEFER_MAX_LEN=4096 (rounds 2-4); since round 5 the default is 8192 and
# they wait for the end of the forward pass like the 2048-token layers of the constant budget: the whole-depth cases below drive
# [8, 8, 16] (Llama: 16 + 16 layers; the library scores the 16 with ONE rolling launch, eight entries on the chip at a time) and
# [8, 8, 2, 8] (Ministral geometry, 26 of its 36 layers: 18 in front of / including the TSP layer 17, 8 behind it) entry calls.
RECIPE = ("--eviction_mode", "proportional", "--retain_rate", "0.1", "--tsp_rate", "0.2")


@pytest.mark.parametrize("family,layers,tsp_idx,defer,calls", [
    ("llama", 32, 15, "1", [8, 8, 16]),
    ("llama", 18, 15, "0", []),
    ("ministral", 26, 17, "1", [8, 8, 2, 8]),
    ("ministral", 20, 17, "0", []),
])
def test_published_recipe_32k_every_layer_against_the_oracle(family, layers, tsp_idx, defer, calls, monkeypatch):
    _prefill_against_the_oracle(family, defer, monkeypatch, S, layers, tsp_idx, calls, extra=RECIPE, tsp_len=int(S * 0.2),
                                caps=(int(S * 0.1), int(int(S * 0.2) * (0.1 / 0.2))))


def test_70k_prefill_every_layer_against_the_oracle(monkeypatch):
    """A prompt beyond what a regular fused scoring launch holds (70,000 tokens): the three layers in front of and including the TSP
    layer are compressed TOGETHER -- one rolling launch whose entries are halves of a layer's KV heads (csrc/fused.hip) -- and every
    layer's cache equals the oracle's."""
    _prefill_against_the_oracle("llama", "1", monkeypatch, 70000, 4, 2, [3, 1])


def _prefill_against_the_oracle(family, defer, monkeypatch, S, LAYERS, TSP_IDX, deferred_calls, extra=(), tsp_len=TSP_LEN, caps=None):
    """`caps` = (capacity of the layers up to and including the TSP layer, capacity of the layers behind it); default: the constant
    budget everywhere.  `extra`: more harness flags (the published recipe's --eviction_mode proportional ...)."""
    TSP_LEN_, caps = tsp_len, caps or (BUDGET, BUDGET)
    from fastkv_amd import cluster as C
    from fastkv_amd import ops
    from oracle.fastkv_oracle import OracleFastKVCluster

    model, a = _build(family, monkeypatch, defer, extra=extra, S=S, LAYERS=LAYERS, TSP_IDX=TSP_IDX)
    captured = {}

    def grab(layer_idx, cl, k, q, v):
        # the operator's inputs exactly as handed over (logical [B,H,S,D] views of [B,S,H,D] storage); only the window rows of q
        captured[layer_idx] = (_cfg_of(cl), k.detach().cpu(), q[:, :, q.shape[2] - cl.window_size:].detach().cpu(), v.detach().cpu())

    real_add, real_add_tsp, real_update = C.DeferredCompression.add, C.DeferredCompression.add_tsp_layer, C.FastKVCluster.update_kv

    def add(self, layer_idx, cl, k, q, v, out_factory=None):
        grab(layer_idx, cl, k, q, v)
        return real_add(self, layer_idx, cl, k, q, v, out_factory=out_factory)

    def add_tsp(self, layer_idx, cl, k, q, v, out_factory=None):
        grab(layer_idx, cl, k, q, v)
        return real_add_tsp(self, layer_idx, cl, k, q, v, out_factory=out_factory)

    def update_kv(self, k, q, v, mask, groups, layer_idx, **kw):
        grab(layer_idx, self, k, q, v)
        return real_update(self, k, q, v, mask, groups, layer_idx, **kw)

    # class-level patches: the instances stay plain FastKVCluster objects, so the deferred schedule stays eligible (an instance
    # whose update_kv was wrapped would be called layer by layer: cluster.py `eligible`)
    monkeypatch.setattr(C.DeferredCompression, "add", add)
    monkeypatch.setattr(C.DeferredCompression, "add_tsp_layer", add_tsp)
    monkeypatch.setattr(C.FastKVCluster, "update_kv", update_kv)
    entry_calls = []
    real_entries = ops.update_kv_entries
    monkeypatch.setattr(ops, "update_kv_entries", lambda *x, **k: (entry_calls.append(len(x[0])), real_entries(*x, **k))[1])

    ids = torch.randint(0, model.config.vocab_size, (1, S), generator=torch.Generator().manual_seed(103)).cuda()
    seen = []
    hooks = [l.register_forward_hook(lambda m, i, o: seen.append(o.shape[1])) for l in model.model.layers]
    with torch.no_grad():
        out = model(ids, attention_mask=torch.ones_like(ids))
    torch.cuda.synchronize()
    ops.raise_if_aborted("test")
    for h in hooks:
        h.remove()
    # the schedule under test really ran: two groups of eight (the TSP layer closing the second) + the post-TSP layer at the end
    assert entry_calls == (deferred_calls if defer == "1" else []), entry_calls
    assert sorted(captured) == list(range(LAYERS))
    assert [captured[i][1].shape[2] for i in range(LAYERS)] == [S] * (TSP_IDX + 1) + [TSP_LEN_] * (LAYERS - TSP_IDX - 1)
    assert seen == [S] * TSP_IDX + [TSP_LEN_] * (LAYERS - TSP_IDX)           # the TSP layer's OUTPUT is already gathered

    pkv = out.past_key_values
    tsp_model = model.model.layers[TSP_IDX].self_attn.tsp_idx
    bad = []
    for i in range(LAYERS):
        cfg, k, qw, v = captured[i]
        oc = OracleFastKVCluster(**cfg)
        groups = qw.shape[1] // k.shape[1]
        wk, wv, wt = oc.update_kv(k, _full_q(qw, k.shape[2]), v, None, groups, i)
        ck, cv = pkv.layers[i].keys.cpu(), pkv.layers[i].values.cpu()
        if ck.shape != wk.shape or not (torch.equal(ck.view(torch.int16), wk.view(torch.int16)) and torch.equal(cv.view(torch.int16), wv.view(torch.int16))):
            bad.append(f"layer {i}: cache rows differ from the oracle's")
        if i == TSP_IDX:
            if wt is None or tsp_model is None or not torch.equal(tsp_model.cpu(), wt):
                bad.append("tsp_idx differs from the oracle's")
            if not torch.equal(model.model.layers[i].new_position_ids.cpu(), wt):      # positions = arange: gather(positions, idx) == idx
                bad.append("rewired position ids differ")
        elif wt is not None:
            bad.append(f"layer {i}: the oracle produced a TSP index on a non-TSP layer")
        # the state the call leaves on the cluster object (proportional mode rewrites max_capacity_prompt on every call and tsp_length
        # on the TSP layer, utils.py:86-87, :123-124; `compress_fastkv` and later prompts read them): product object == oracle object
        cl = model.model.layers[i].self_attn.kv_cluster
        if (cl.max_capacity_prompt, cl.tsp_length, cl.tsp_layer) != (oc.max_capacity_prompt, oc.tsp_length, oc.tsp_layer):
            bad.append(f"layer {i}: cluster state {(cl.max_capacity_prompt, cl.tsp_length)} != the oracle's {(oc.max_capacity_prompt, oc.tsp_length)}")
        if cl.max_capacity_prompt != caps[0 if i <= TSP_IDX else 1]:
            bad.append(f"layer {i}: capacity {cl.max_capacity_prompt}")
        captured[i] = None
    assert not bad, bad
    assert model.model.layers[TSP_IDX].self_attn.kv_cluster.tsp_length == TSP_LEN_
    assert tuple(pkv.layers[0].keys.shape) == (1, 8, caps[0], 128) and tuple(pkv.layers[LAYERS - 1].keys.shape) == (1, 8, caps[1], 128)
    assert out.logits.shape[:2] == (1, 1) and bool(torch.isfinite(out.logits).all())


def test_mistral_sliding_window_prefill_on_gpu_against_the_stock_eager_attention(monkeypatch):
    """Mistral-7B geometry (2 layers) WITH sliding_window = 4096 (v0.1; /root/reference/baselines/fastkv/mistral_model.py:143-153
    hands the window to the prefill's attention call), a 6000-token prompt (longer than the window), budget 512.
    No TSP (tsp_len above the prompt length): the patched model's last-token logits equal those of the STOCK MistralForCausalLM
    -- unpatched classes, HF's own eager sliding-window attention -- on the same weights to fp16 tolerance, and differ from the same
    model without a window (so the comparison would notice a dropped window); every layer's cache rows equal the oracle
    cluster's on the captured q / k / v (the operator sees un-windowed K / V: the window is the attention's business)."""
    from baselines.monkeypatch import replace_mistral, set_model
    from benchmark import prefill
    from fastkv_amd import cluster as C
    from oracle.fastkv_oracle import OracleFastKVCluster
    Sw, win = 6000, 4096
    monkeypatch.setenv("FASTKV_DEFER", "0")
    monkeypatch.setenv("FASTKV_SLAB_CACHE", "0")

    def build(method, window, impl):
        a = prefill.parse_args(["--model_path", "mistral-7b", "--num_layers", "2", "--device", "cuda", "--save_txt", "", "--method", method,
                                "--max_capacity_prompts", "512", "--tsp_len", "8192", "--tsp_idx", "0", "--pooling", "avgpool",
                                "--sliding_window", str(window), "--attn_implementation", impl])
        a.save_txt = False
        a.context_lengths = [Sw]
        replace_mistral(method)
        torch.manual_seed(111)
        m = prefill.build_model(a, "cuda")
        set_model(m, a)
        return m

    model = build("fastkv", win, "sdpa")
    assert model.config.sliding_window == win and type(model.model.layers[0].self_attn).__name__ == "MistralFastKVAttention"
    captured = {}
    real_update = C.FastKVCluster.update_kv

    def update_kv(self, k, q, v, mask, groups, layer_idx, **kw):
        captured[layer_idx] = (_cfg_of(self), k.detach().cpu(), q.detach().cpu(), v.detach().cpu())
        return real_update(self, k, q, v, mask, groups, layer_idx, **kw)

    monkeypatch.setattr(C.FastKVCluster, "update_kv", update_kv)
    ids = torch.randint(0, model.config.vocab_size, (1, Sw), generator=torch.Generator().manual_seed(113)).cuda()
    with torch.no_grad():
        out = model(ids, attention_mask=torch.ones_like(ids))
    torch.cuda.synchronize()
    assert sorted(captured) == [0, 1] and model.model.layers[0].self_attn.tsp_idx is None
    for i in (0, 1):
        cfg, k, q, v = captured[i]
        wk, wv, wt = OracleFastKVCluster(**cfg).update_kv(k, q, v, None, 4, i)
        assert wt is None and wk.shape[2] == 512
        # transformers' sliding-window cache layer keeps the last window - 1 rows of what it is handed: all 512 here
        assert torch.equal(out.past_key_values.layers[i].keys.cpu(), wk) and torch.equal(out.past_key_values.layers[i].values.cpu(), wv), i
    got = out.logits.float().cpu()
    sd = {k_: v_.clone() for k_, v_ in model.state_dict().items()}
    del model, out
    torch.cuda.empty_cache()
    try:
        res = {}
        for tag, window in (("window", win), ("no_window", 0)):
            stock = build("fullkv", window, "eager")
            assert type(stock.model.layers[0].self_attn).__name__ == "MistralAttention"
            stock.load_state_dict(sd)
            with torch.no_grad():
                res[tag] = stock(ids, attention_mask=torch.ones_like(ids)).logits[:, -1:].float().cpu()
            del stock
            torch.cuda.empty_cache()
    finally:
        replace_mistral("fastkv")
    scale = float(res["window"].abs().max())
    err = float((got - res["window"]).abs().max())
    assert err <= 1e-2 * scale, (err, scale)
    assert float((res["no_window"] - res["window"]).abs().max()) > 5 * max(err, 1e-3 * scale)   # the window matters at this length
