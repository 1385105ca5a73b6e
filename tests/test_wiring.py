"""Plugin boundary (baselines.monkeypatch) and model wiring on CPU: BASELINE.json configs[0] "CPU plumbing".

The product cluster has no CPU path, so these tests put the oracle-backed cluster (test infrastructure) into the patched
model -- what is under test is the wiring: class swap before construction, compressed K/V into the cache, TSP gather +
position rewire + rotary refresh, last-token cut, decode after prefill, the fullkv arm, and the reference's recorded
decoder-layer behaviour (tests/golden/decoder_layer_tsp.npz)."""
import types

import numpy as np
import pytest
import torch

from helpers import GOLDEN


def _oracle_cluster(old):
    """Oracle-backed stand-in with the same attributes; the CPU model runs in fp32, the oracle (like the path) in fp16."""
    from oracle.fastkv_oracle import OracleFastKVCluster

    class Fp32Adapter(OracleFastKVCluster):
        def update_kv(self, key_states, query_states, value_states, attention_mask, num_key_value_groups, layer_idx):
            dt = key_states.dtype
            if dt == torch.float16:
                return super().update_kv(key_states, query_states, value_states, attention_mask, num_key_value_groups, layer_idx)
            if query_states.shape[2] < (int(query_states.shape[2] * self.retain_rate) if self.eviction_mode == "proportional"
                                        else self.max_capacity_prompt):
                return key_states, value_states, None
            k, v, t = super().update_kv(key_states.half(), query_states.half(), value_states.half(), attention_mask,
                                        num_key_value_groups, layer_idx)
            return k.to(dt), v.to(dt), t

    return Fp32Adapter(old.window_size, old.max_capacity_prompt, old.kernel_size, old.pooling, old.tsp_layer,
                       old.tsp_length, old.tsp_rate, old.retain_rate, old.eviction_mode)


def _args(**kw):
    from benchmark import prefill
    a = prefill.parse_args(["--model_path", "tiny", "--device", "cpu", "--save_txt", ""])
    a.save_txt = False
    a.cluster_factory = _oracle_cluster
    a.random_tokens = True
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_fastkv_prefill_plumbing_cpu():
    from benchmark import prefill
    a = _args(method="fastkv", context_lengths=[300], max_capacity_prompts=64, tsp_len=96, tsp_idx=1, num_warmups=0)
    res = prefill.run(a)[0]
    # every layer caches exactly `capacity` rows: layers 0..tsp_idx from 300 tokens, later ones from the 96 TSP survivors
    assert res["cache_lens"] == [64, 64, 64, 64]
    assert res["logits_shape"][:2] == [1, 1]                         # last-token cut (llama_model.py:392)


def test_tsp_reduces_sequence_after_tsp_layer_and_decode_works():
    from baselines.monkeypatch import replace_llama, set_model
    from benchmark import prefill
    a = _args(method="fastkv", max_capacity_prompts=40, tsp_len=80, tsp_idx=0)
    a.context_lengths = [200]
    replace_llama("fastkv")
    model = prefill.build_model(a, "cpu")
    from transformers.models.llama import modeling_llama
    assert type(model.model.layers[0].self_attn).__name__ == "LlamaFastKVAttention"
    assert isinstance(model.model.layers[0].self_attn, modeling_llama._fastkv_stock_attention)
    set_model(model, a)
    for layer in model.model.layers:
        layer.self_attn.kv_cluster = _oracle_cluster(layer.self_attn.kv_cluster)
    seen = []
    hooks = [l.register_forward_hook(lambda m, i, o: seen.append(o.shape[1])) for l in model.model.layers]
    ids = torch.randint(0, 1000, (1, 200))
    with torch.no_grad():
        out = model(ids, attention_mask=torch.ones_like(ids))
    for h in hooks:
        h.remove()
    assert seen == [80, 80, 80, 80]                                  # layer 0 is the TSP layer: its OUTPUT is already gathered
    tsp = model.model.layers[0].self_attn.tsp_idx
    assert tsp.shape == (1, 80) and bool((tsp[:, 1:] > tsp[:, :-1]).all()) and tsp[0, -1] == 199
    assert torch.equal(model.model.layers[0].new_position_ids, tsp)
    # decode one token on top of the compressed cache: cache grows by one row per layer
    pkv = out.past_key_values
    before = [int(pkv.layers[i].keys.shape[-2]) for i in range(4)]
    nxt = out.logits[:, -1].argmax(-1, keepdim=True)
    with torch.no_grad():
        out2 = model(nxt, past_key_values=pkv, position_ids=torch.tensor([[200]]))
    after = [int(out2.past_key_values.layers[i].keys.shape[-2]) for i in range(4)]
    assert after == [b + 1 for b in before] and out2.logits.shape[:2] == (1, 1)
    assert model.model.layers[0].self_attn.tsp_idx is None            # decode: no TSP (llama_model.py:143-145)


def test_no_compression_equals_fullkv():
    """With budgets above the prompt length FastKV takes the early-out everywhere: logits must equal the fullkv arm."""
    from baselines.monkeypatch import replace_llama, set_model
    from benchmark import prefill
    ids = torch.randint(0, 1000, (1, 50))
    outs = {}
    for method in ("fastkv", "fullkv"):
        a = _args(method=method, max_capacity_prompts=512, tsp_len=2048, tsp_idx=1)
        a.context_lengths = [50]
        replace_llama(method)
        torch.manual_seed(7)
        model = prefill.build_model(a, "cpu")
        set_model(model, a)
        with torch.no_grad():
            outs[method] = model(ids, attention_mask=torch.ones_like(ids)).logits
    assert outs["fastkv"].shape == outs["fullkv"].shape == (1, 1, 1024)
    assert torch.equal(outs["fastkv"], outs["fullkv"])
    replace_llama("fastkv")


def test_mistral_patch_runs():
    from baselines.monkeypatch import replace_mistral, set_model
    from transformers import MistralConfig, MistralForCausalLM
    replace_mistral("fastkv")
    cfg = MistralConfig(hidden_size=128, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2, intermediate_size=256,
                        vocab_size=500, head_dim=32, sliding_window=None, max_position_embeddings=512)
    cfg._attn_implementation = "sdpa"
    model = MistralForCausalLM(cfg).eval()
    assert type(model.model.layers[0].self_attn).__name__ == "MistralFastKVAttention"
    a = types.SimpleNamespace(method="fastkv", window_size=8, kernel_size=5, pooling="avgpool", max_capacity_prompts=32, tsp_len=64,
                              tsp_rate=0.2, eviction_mode="constant", tsp_idx=1, retain_rate=0.1)
    set_model(model, a)
    assert a.window_size == [8, 8, 8]                                 # listified like monkeypatch.py:121-139
    for layer in model.model.layers:
        layer.self_attn.kv_cluster = _oracle_cluster(layer.self_attn.kv_cluster)
    ids = torch.randint(0, 500, (1, 150))
    with torch.no_grad():
        out = model(ids)
    assert [int(out.past_key_values.layers[i].keys.shape[-2]) for i in range(3)] == [32, 32, 32]
    assert out.logits.shape == (1, 1, 500)


@pytest.mark.parametrize("tsp", [False, True])
def test_mistral_sliding_window_prefill_matches_the_stock_sliding_window_attention(tsp):
    """A Mistral config WITH `sliding_window` (v0.1): the reference hands the window to the flash-attention call of the prefill
    (/root/reference/baselines/fastkv/mistral_model.py:143-153); here it reaches the mask builder and the attention interface.
    Prompt longer than the window, budgets that compress every layer's cache:
      * no TSP (tsp_len >= prompt): the logits equal those of the STOCK MistralForCausalLM (unpatched classes, eager sliding-window
        attention) on the same weights -- compression only changes what is cached;
      * TSP at layer 0: the layers behind it attend causally + windowed over the surviving TOKENS; cache rows of every layer
        equal the oracle cluster's on the captured q / k / v (the wiring hands the operator un-windowed K/V: the window is the
        attention's business, utils.py:93-132 never sees it)."""
    from baselines.monkeypatch import replace_mistral, set_model
    from transformers import MistralConfig, MistralForCausalLM
    S, win = 200, 48
    cfg = MistralConfig(hidden_size=128, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2, intermediate_size=256,
                        vocab_size=500, head_dim=32, sliding_window=win, max_position_embeddings=512)
    ids = torch.randint(0, 500, (1, S), generator=torch.Generator().manual_seed(29))
    a = types.SimpleNamespace(method="fastkv", window_size=8, kernel_size=5, pooling="avgpool", max_capacity_prompts=32,
                              tsp_len=100 if tsp else 4096, tsp_rate=0.2, eviction_mode="constant", tsp_idx=0, retain_rate=0.1)
    replace_mistral("fastkv")
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(31)
    model = MistralForCausalLM(cfg).eval()
    assert type(model.model.layers[0].self_attn).__name__ == "MistralFastKVAttention"
    set_model(model, a)
    captured = []
    for layer in model.model.layers:
        cl = layer.self_attn.kv_cluster = _oracle_cluster(layer.self_attn.kv_cluster)
        orig = cl.update_kv

        def spy(k, q, v, m, g, li, _orig=orig):
            out = _orig(k, q, v, m, g, li)
            captured.append((li, k.shape[2], out))
            return out

        cl.update_kv = spy
    with torch.no_grad():
        out = model(ids, attention_mask=torch.ones_like(ids))
    assert [c[1] for c in captured] == ([S, 100, 100] if tsp else [S, S, S])
    for li, _, (kc, vc, t) in captured:
        # (transformers' sliding-window cache layer keeps the LAST window - 1 rows of what `update` was handed)
        have = out.past_key_values.layers[li].keys
        assert have.shape[-2] == min(32, win - 1) and torch.equal(have, kc[:, :, -have.shape[-2]:])
    if tsp:
        assert captured[0][2][2].shape == (1, 100) and torch.isfinite(out.logits).all()
        return
    replace_mistral("fullkv")
    try:
        cfg2 = MistralConfig(**{**cfg.to_dict(), "sliding_window": win})
        cfg2._attn_implementation = "eager"
        stock = MistralForCausalLM(cfg2).eval()
        assert type(stock.model.layers[0].self_attn).__name__ == "MistralAttention"
        stock.load_state_dict(model.state_dict())
        with torch.no_grad():
            want = stock(ids, attention_mask=torch.ones_like(ids)).logits[:, -1:]
            cfg3 = MistralConfig(**{**cfg.to_dict(), "sliding_window": None})
            cfg3._attn_implementation = "eager"
            nowin = MistralForCausalLM(cfg3).eval()
            nowin.load_state_dict(model.state_dict())
            other = nowin(ids, attention_mask=torch.ones_like(ids)).logits[:, -1:]
    finally:
        replace_mistral("fastkv")
    assert torch.allclose(out.logits, want, atol=2e-4, rtol=1e-4), float((out.logits - want).abs().max())
    assert not torch.allclose(other, want, atol=2e-3)                # (the window matters at this length: the test would see it missing)


def test_unsupported_methods_are_refused():
    from baselines.monkeypatch import replace_llama
    with pytest.raises(NotImplementedError):
        replace_llama("snapkv")


def test_decoder_layer_matches_reference_golden():
    """llama_decoderlayer_forward_fastkv of the reference, recorded on a duck-typed layer (make_golden.py)."""
    from baselines.fastkv.llama_model import llama_decoderlayer_forward_fastkv
    g = np.load(f"{GOLDEN}/decoder_layer_tsp.npz")
    hid = torch.from_numpy(g["hidden_in"]).view(torch.float16)
    tsp, pos = torch.from_numpy(g["tsp_idx"]), torch.from_numpy(g["position_ids"])

    class Attn:
        def __init__(self, idx, tsp_layer):
            self.tsp_idx, self.kv_cluster = idx, types.SimpleNamespace(tsp_layer=tsp_layer)

        def __call__(self, hidden_states=None, **kw):
            return hidden_states * 0.5, None

    for tag, tsp_layer, idx in (("tsp", True, tsp), ("not_tsp_layer", False, tsp), ("tsp_none", True, None)):
        me = types.SimpleNamespace(input_layernorm=lambda x: x, post_attention_layernorm=lambda x: x, mlp=lambda x: x * 0.25,
                                   self_attn=Attn(idx, tsp_layer))
        out = llama_decoderlayer_forward_fastkv(me, hid, position_ids=pos)
        assert torch.equal(out.view(torch.int16), torch.from_numpy(g["hidden_out_" + tag])), tag
        assert (me.new_position_ids is not None) == bool(g["has_new_pos_" + tag]), tag
        if me.new_position_ids is not None:
            assert torch.equal(me.new_position_ids, torch.from_numpy(g["new_pos_" + tag]))


def test_e2e_driver_cpu():
    """benchmark/e2e.py: prefill + greedy decode over the compressed cache (counterpart of the reference driver)."""
    from benchmark import e2e
    a = _args(method="fastkv", context_lengths=[200], max_capacity_prompts=48, tsp_len=80, tsp_idx=1, num_warmups=0, genlen=5)
    res = e2e.run(a)[0]
    assert res["final_cache_len_layer0"] == 48 + 4                   # budget + the 4 decoded tokens
    assert res["throughput_tok_s"] > 0


def test_slab_cache_matches_dynamic_cache(monkeypatch):
    """FASTKV_SLAB_CACHE=1 (fastkv_amd/cache.py: pre-sized per-layer slabs, decode appends in place) must be invisible:
    prefill + three greedy decode steps give the same logits and cache lengths as with DynamicCache."""
    from baselines.monkeypatch import replace_llama, set_model
    from benchmark import prefill
    from fastkv_amd.cache import FastKVSlabCache, SlabLayer
    ids = torch.randint(0, 1000, (1, 150))
    runs = {}
    for slab in ("0", "1"):
        monkeypatch.setenv("FASTKV_SLAB_CACHE", slab)
        monkeypatch.setenv("FASTKV_SLAB_RESERVE", "2")                  # forces a slab growth during the decode steps
        a = _args(method="fastkv", max_capacity_prompts=40, tsp_len=80, tsp_idx=1)
        a.context_lengths = [150]
        replace_llama("fastkv")
        torch.manual_seed(11)
        model = prefill.build_model(a, "cpu")
        set_model(model, a)
        for layer in model.model.layers:
            layer.self_attn.kv_cluster = _oracle_cluster(layer.self_attn.kv_cluster)
        logits, lens = [], []
        with torch.no_grad():
            out = model(ids, attention_mask=torch.ones_like(ids))
            pkv = out.past_key_values
            logits.append(out.logits)
            for step in range(3):
                nxt = logits[-1][:, -1].argmax(-1, keepdim=True)
                out = model(nxt, past_key_values=pkv, position_ids=torch.tensor([[150 + step]]))
                logits.append(out.logits)
                lens.append([int(pkv.layers[i].keys.shape[-2]) for i in range(4)])
        runs[slab] = (logits, lens, pkv)
    assert isinstance(runs["1"][2], FastKVSlabCache) and isinstance(runs["1"][2].layers[0], SlabLayer)
    assert not isinstance(runs["0"][2], FastKVSlabCache)
    assert runs["0"][1] == runs["1"][1] and runs["1"][1][-1] == [43, 43, 43, 43]
    for x, y in zip(runs["0"][0], runs["1"][0]):
        assert torch.equal(x, y)
    assert runs["1"][2].get_seq_length() == 43


def test_eager_attention_stays_causal_after_the_tsp_layer():
    """`--attn_implementation eager` masks only when it is handed a mask: after the TSP gather the model loop rebuilds the
    causal mask for the surviving tokens (the reference's flash-attention call is causal=True on whatever sequence it gets,
    /root/reference/baselines/fastkv/llama_model.py:181-183).  Same weights, same prompt: eager == sdpa."""
    from baselines.monkeypatch import replace_llama, set_model
    from benchmark import prefill
    ids = torch.randint(0, 1000, (1, 160), generator=torch.Generator().manual_seed(11))
    outs = {}
    for impl in ("sdpa", "eager"):
        a = _args(method="fastkv", max_capacity_prompts=40, tsp_len=64, tsp_idx=1, attn_implementation=impl)
        a.context_lengths = [160]
        replace_llama("fastkv")
        torch.manual_seed(13)
        model = prefill.build_model(a, "cpu")
        set_model(model, a)
        for layer in model.model.layers:
            layer.self_attn.kv_cluster = _oracle_cluster(layer.self_attn.kv_cluster)
        with torch.no_grad():
            outs[impl] = model(ids, attention_mask=torch.ones_like(ids)).logits
    assert torch.allclose(outs["eager"], outs["sdpa"], atol=2e-4, rtol=1e-4), float((outs["eager"] - outs["sdpa"]).abs().max())


@pytest.mark.parametrize("install_override", [False, True])
def test_generate_decodes_at_true_positions_after_a_compressed_prefill(install_override):
    """`model.generate()` on top of a compressed prefill: decode steps must run at the prompt's TRUE positions (200, 201, ...),
    not at positions derived from the 40-row cache -- what the reference's prepare_inputs_for_generation override provides on
    transformers 4.45 (/root/reference/baselines/monkeypatch.py:280-288) and what `replace_llama("fastkv")` installs here too
    (`:55-56`).  The installed transformers derives the positions by itself as well: same result with the stock preparation;
    `replace_llama("fullkv")` puts the stock one back."""
    from transformers import LlamaForCausalLM
    from baselines import monkeypatch as MP
    from benchmark import prefill
    a = _args(method="fastkv", max_capacity_prompts=40, tsp_len=80, tsp_idx=0)
    a.context_lengths = [200]
    MP.replace_llama("fastkv")
    # replace_llama installs the override for every compressing method, as the reference does (monkeypatch.py:55-56) ...
    assert LlamaForCausalLM.prepare_inputs_for_generation is MP.prepare_inputs_for_generation_llama
    stock = LlamaForCausalLM.prepare_inputs_for_generation
    if not install_override:                                       # ... and the stock preparation gives the same positions here
        LlamaForCausalLM.prepare_inputs_for_generation = MP._STOCK_PREPARE["llama"]
    try:
        torch.manual_seed(17)
        model = prefill.build_model(a, "cpu")
        MP.set_model(model, a)
        for layer in model.model.layers:
            layer.self_attn.kv_cluster = _oracle_cluster(layer.self_attn.kv_cluster)
        seen = []
        inner = type(model.model).forward

        def spy(self, *args, **kw):
            seen.append((kw["position_ids"][0, -1].item(), kw["past_key_values"].get_seq_length()))
            return inner(self, *args, **kw)

        type(model.model).forward = spy
        try:
            ids = torch.randint(0, 1000, (1, 200), generator=torch.Generator().manual_seed(19))
            out = model.generate(ids, attention_mask=torch.ones_like(ids), max_new_tokens=4, do_sample=False)
        finally:
            type(model.model).forward = inner
        assert out.shape == (1, 204)
        assert seen == [(199, 0), (200, 40), (201, 41), (202, 42)]
    finally:
        LlamaForCausalLM.prepare_inputs_for_generation = stock
    MP.replace_llama("fullkv")
    assert LlamaForCausalLM.prepare_inputs_for_generation is MP._STOCK_PREPARE["llama"]
    MP.replace_llama("fastkv")


def test_static_decode_is_refused_when_a_sliding_window_is_smaller_than_the_slab():
    """the HIP decode attention attends over the whole slab; a model with `sliding_window` set limits attention in the
    eager path.  Static decode is therefore only allowed while the slabs cannot hold more rows than the window."""
    import types
    from baselines.fastkv._wiring import _window_allows_static
    slab = lambda rows: types.SimpleNamespace(kslab=torch.empty(1, 2, rows, 4))
    cache = types.SimpleNamespace(layers=[slab(300), slab(200), types.SimpleNamespace(kslab=None)])
    assert _window_allows_static(types.SimpleNamespace(sliding_window=None), cache)
    assert _window_allows_static(types.SimpleNamespace(sliding_window=4096), cache)
    assert not _window_allows_static(types.SimpleNamespace(sliding_window=256), cache)
    assert _window_allows_static(types.SimpleNamespace(), cache)              # (Llama configs have no such attribute)
