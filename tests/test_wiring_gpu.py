"""Model wiring on the MI355X with the product cluster (HIP kernels): the slab cache -- compaction writes straight into the
layer's pre-sized cache buffer (fastkv_update_kv_strided_f16), decode appends in place -- against DynamicCache."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(slab, monkeypatch):
    from baselines.monkeypatch import replace_llama, set_model
    from benchmark import prefill
    monkeypatch.setenv("FASTKV_SLAB_CACHE", slab)
    # two layers of the Llama-3-8B geometry (head_dim 128: the HIP path has no head_dim-32 "tiny" variant)
    a = prefill.parse_args(["--model_path", "llama3-8b", "--num_layers", "2", "--device", "cuda", "--save_txt", "", "--method",
                            "fastkv", "--max_capacity_prompts", "128", "--tsp_len", "256", "--tsp_idx", "0"])
    a.save_txt = False
    a.context_lengths = [700]
    replace_llama("fastkv")
    torch.manual_seed(3)
    model = prefill.build_model(a, "cuda")
    set_model(model, a)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 1000, (1, 700), generator=g).cuda()
    logits = []
    with torch.no_grad():
        out = model(ids, attention_mask=torch.ones_like(ids))
        pkv = out.past_key_values
        logits.append(out.logits.float().cpu())
        for step in range(4):
            nxt = out.logits[:, -1].argmax(-1, keepdim=True)
            out = model(nxt, past_key_values=pkv, position_ids=torch.tensor([[700 + step]], device="cuda"))
            logits.append(out.logits.float().cpu())
    torch.cuda.synchronize()
    return logits, pkv


def test_slab_cache_in_place_compaction_matches_dynamic_cache(monkeypatch):
    from fastkv_amd.cache import FastKVSlabCache
    dyn, pkv_d = _run("0", monkeypatch)
    slab, pkv_s = _run("1", monkeypatch)
    assert isinstance(pkv_s, FastKVSlabCache) and not isinstance(pkv_d, FastKVSlabCache)
    layer = pkv_s.layers[0]
    assert layer.keys.data_ptr() == layer.kslab.data_ptr() and layer.len == 128 + 4          # rows were written in place
    for i in range(2):
        assert torch.equal(pkv_s.layers[i].keys, pkv_d.layers[i].keys) and torch.equal(pkv_s.layers[i].values, pkv_d.layers[i].values)
    for x, y in zip(dyn, slab):
        assert torch.equal(x, y)


# ------------------------------------------------------------------------------------------------ whole patched model vs the oracle
def _capture_and_compare(family):
    """Two decoder layers of the family's 7-8B geometry (head_dim 128, 32 query / 8 KV heads) on the GPU with the PRODUCT
    cluster, a 4096-token prompt (BASELINE.json configs[0]'s shape: budget 512, TSP length 2048; layer 0 is the TSP layer so
    that layer 1 runs on the 2048 survivors).  Every layer's update_kv inputs are captured and replayed through the ORACLE
    cluster on the CPU: the rows that went into the cache, tsp_idx and the rewired position ids must be identical."""
    from baselines.monkeypatch import replace_llama, replace_mistral, set_model
    from benchmark import prefill
    from oracle.fastkv_oracle import OracleFastKVCluster
    name = {"llama": "llama3-8b", "mistral": "mistral-7b"}[family]
    a = prefill.parse_args(["--model_path", name, "--num_layers", "2", "--device", "cuda", "--save_txt", "", "--method", "fastkv",
                            "--max_capacity_prompts", "512", "--tsp_len", "2048", "--tsp_idx", "0", "--pooling", "maxpool"])
    a.save_txt = False
    a.context_lengths = [4096]
    (replace_llama if family == "llama" else replace_mistral)("fastkv")
    torch.manual_seed(21)
    model = prefill.build_model(a, "cuda")
    assert type(model.model.layers[0].self_attn).__name__ == ("LlamaFastKVAttention" if family == "llama" else "MistralFastKVAttention")
    set_model(model, a)
    captured = []
    for layer in model.model.layers:
        cl = layer.self_attn.kv_cluster
        orig = cl.update_kv

        def spy(key_states, query_states, value_states, attention_mask, groups, layer_idx, _orig=orig, _cl=cl, **kw):
            out = _orig(key_states, query_states, value_states, attention_mask, groups, layer_idx, **kw)
            cfg = dict(window_size=_cl.window_size, max_capacity_prompt=_cl.max_capacity_prompt, kernel_size=_cl.kernel_size,
                       pooling=_cl.pooling, tsp_layer=_cl.tsp_layer, tsp_length=_cl.tsp_length, tsp_rate=_cl.tsp_rate,
                       retain_rate=_cl.retain_rate, eviction_mode=_cl.eviction_mode)
            captured.append((cfg, groups, layer_idx, tuple(t.detach().cpu() for t in (key_states, query_states, value_states)),
                             tuple(None if t is None else t.detach().cpu() for t in out)))
            return out

        cl.update_kv = spy
    ids = torch.randint(0, 1000, (1, 4096), generator=torch.Generator().manual_seed(23)).cuda()
    with torch.no_grad():
        out = model(ids, attention_mask=torch.ones_like(ids))
    torch.cuda.synchronize()
    assert len(captured) == 2 and [c[3][0].shape[2] for c in captured] == [4096, 2048]
    for cfg, groups, layer_idx, (k, q, v), (kc, vc, tsp) in captured:
        oc = OracleFastKVCluster(cfg["window_size"], cfg["max_capacity_prompt"], cfg["kernel_size"], cfg["pooling"], cfg["tsp_layer"],
                                 cfg["tsp_length"], cfg["tsp_rate"], cfg["retain_rate"], cfg["eviction_mode"])
        wk, wv, wt = oc.update_kv(k, q, v, None, groups, layer_idx)
        assert torch.equal(kc, wk) and torch.equal(vc, wv), layer_idx
        assert (tsp is None and wt is None) or torch.equal(tsp, wt), layer_idx
        cache = out.past_key_values.layers[layer_idx]
        assert torch.equal(cache.keys.cpu(), wk) and torch.equal(cache.values.cpu(), wv), layer_idx      # what attention will decode over
    tsp0 = captured[0][4][2]
    assert tsp0 is not None and tsp0.shape == (1, 2048) and captured[1][4][2] is None
    assert torch.equal(model.model.layers[0].new_position_ids.cpu(), tsp0)                              # positions = arange -> gather == idx
    assert out.logits.shape[:2] == (1, 1) and bool(torch.isfinite(out.logits).all())


def test_llama_wiring_on_gpu_matches_oracle_cluster():
    _capture_and_compare("llama")


def test_mistral_wiring_on_gpu_matches_oracle_cluster():
    """BASELINE.json configs[3]'s wiring (/root/reference/baselines/fastkv/mistral_model.py:100-107, 217-224)."""
    _capture_and_compare("mistral")


def test_deferred_compression_of_the_post_tsp_layers_changes_nothing(monkeypatch):
    """FASTKV_DEFER (default on, DynamicCache): the layers behind the TSP layer are compressed together after the last layer
    (fastkv_amd.cluster.DeferredCompression -> ops.update_kv_entries, one launch sequence) instead of one by one inside their
    attention forward.  Same kernels, same inputs: caches and logits are bit-identical to the layer-by-layer run, and the
    deferred path really ran (its entry point was called once with the 5 layers behind the TSP layer)."""
    from baselines.monkeypatch import replace_llama, set_model
    from benchmark import prefill
    from fastkv_amd import ops

    def run(defer, slab="0", tsp_idx="0", S=3000, B=1, hold="8", max_len="8192"):
        monkeypatch.setenv("FASTKV_DEFER", defer)
        monkeypatch.setenv("FASTKV_DEFER_HOLD", hold)
        monkeypatch.setenv("FASTKV_DEFER_MAX_LEN", max_len)
        monkeypatch.setenv("FASTKV_SLAB_CACHE", slab)
        a = prefill.parse_args(["--model_path", "llama3-8b", "--num_layers", "6", "--device", "cuda", "--save_txt", "", "--method",
                                "fastkv", "--max_capacity_prompts", "512", "--tsp_len", "1024", "--tsp_idx", tsp_idx, "--pooling", "maxpool"])
        a.save_txt = False
        a.context_lengths = [S]
        replace_llama("fastkv")
        torch.manual_seed(11)
        model = prefill.build_model(a, "cuda")
        set_model(model, a)
        ids = torch.randint(0, 1000, (B, S), generator=torch.Generator().manual_seed(12)).cuda()
        with torch.no_grad():
            out = model(ids, attention_mask=torch.ones_like(ids))
            nxt = out.logits[:, -1].argmax(-1, keepdim=True)
            out2 = model(nxt, past_key_values=out.past_key_values)
        torch.cuda.synchronize()
        pkv = out.past_key_values
        return out.logits.float().cpu(), out2.logits.float().cpu(), [(l.keys.clone(), l.values.clone()) for l in pkv.layers]

    calls = []
    real = ops.update_kv_entries
    monkeypatch.setattr(ops, "update_kv_entries", lambda *a, **k: (calls.append(len(a[0])), real(*a, **k))[1])
    l1, d1, c1 = run("1")
    assert calls == [5]
    l0, d0, c0 = run("0")
    assert calls == [5]
    assert torch.equal(l1, l0) and torch.equal(d1, d0) and len(c1) == len(c0) == 6
    for (k1, v1), (k0, v0) in zip(c1, c0):
        assert torch.equal(k1, k0) and torch.equal(v1, v0)
    # FASTKV_DEFER_HOLD=2, FASTKV_DEFER_MAX_LEN=4096: long layers in front of the TSP layer wait for ONE peer: layers 0 and 1 of a
    # 5000-token prompt run as a pair when layer 1 arrives, layer 2 waits and is taken along by the TSP layer (3), layers 4 and 5 behind it
    # run together at the end
    del calls[:]
    lp, dp, cp = run("1", tsp_idx="3", S=5000, hold="2", max_len="4096")
    assert calls == [2, 2, 2]
    lq, dq, cq = run("0", tsp_idx="3", S=5000)
    assert calls == [2, 2, 2] and torch.equal(lp, lq) and torch.equal(dp, dq)
    for (k1, v1), (k0, v0) in zip(cp, cq):
        assert torch.equal(k1, k0) and torch.equal(v1, v0)
    # the default of round 5 (layers of up to 8192 tokens wait for the end of the pass, or for the TSP layer): the three layers in front of
    # the TSP layer go along with it, the two behind it run at the end -- same caches, same logits
    del calls[:]
    lr, dr, cr = run("1", tsp_idx="3", S=5000)
    assert calls == [4, 2] and torch.equal(lr, lq) and torch.equal(dr, dq)
    for (k1, v1), (k0, v0) in zip(cr, cq):
        assert torch.equal(k1, k0) and torch.equal(v1, v0)
    del calls[:]
    # the default (groups of up to eight): layers 0-2 wait, the TSP layer (3) takes all three along (one call of four entries: the library
    # scores them in as many fused launches as fit the chip, selects and copies once), layers 4 and 5 run together at the end
    lr, dr, cr = run("1", tsp_idx="3", S=5000)
    assert calls == [4, 2] and torch.equal(lr, lq) and torch.equal(dr, dq)
    for (k1, v1), (k0, v0) in zip(cr, cq):
        assert torch.equal(k1, k0) and torch.equal(v1, v0)
    del calls[:]
    # a batch of two (unpadded) prompts: every batch row of a layer is an entry of the deferred launch sequence (VERDICT r02 weak
    # #10: batches used to drop to layer by layer silently)
    lb, db, cb = run("1", B=2)
    assert calls == [5]
    lc, dc, cc = run("0", B=2)
    assert calls == [5] and lb.shape[0] == 2 and torch.equal(lb, lc) and torch.equal(db, dc)
    for (k1, v1), (k0, v0) in zip(cb, cc):
        assert k1.shape[0] == 2 and torch.equal(k1, k0) and torch.equal(v1, v0)
    del calls[:]
    # ... and over the slab cache: the deferred launch writes every layer's rows straight into that layer's slab
    ls, ds, cs = run("1", slab="1")
    assert calls == [5]
    assert torch.equal(ls, l0) and len(cs) == 6
    for (k1, v1), (k0, v0) in zip(cs, c0):
        assert torch.equal(k1[:, :, :k0.shape[2]], k0) and torch.equal(v1[:, :, :v0.shape[2]], v0)


# ------------------------------------------------------------------------------------------------ asynchronous errors reach the user
_ABORT_CHILD = """
import os, sys, time, torch
sys.path.insert(0, 'tests')
from baselines.monkeypatch import replace_llama, set_model
from benchmark import prefill
from fastkv_amd import ops
from fastkv_amd._lib import load, FastKVNativeError, FASTKV_EABORTED
from fastkv_amd.cluster import DeferredCompression
L = load()
a = prefill.parse_args(['--model_path', 'llama3-8b', '--num_layers', '3', '--device', 'cuda', '--save_txt', '', '--method', 'fastkv',
                        '--max_capacity_prompts', '512', '--tsp_len', '2048', '--tsp_idx', '2'])
a.save_txt = False
replace_llama('fastkv')
torch.manual_seed(3)
model = prefill.build_model(a, 'cuda')
set_model(model, a)
# 8192 tokens: the pair's fused launch has 512 workgroups, two per compute unit -- the launches that need all their workgroups resident
# (a launch of up to 256 workgroups is dispatched unit by unit and gets through a partly occupied GPU: its units complete one after the other)
ids = torch.randint(0, 1000, (1, 8192), generator=torch.Generator().manual_seed(5)).cuda()
with torch.no_grad():
    ref = model(ids)                                                       # idle GPU: layers 0 and 1 run as a deferred PAIR
torch.cuda.synchronize()
assert L.fastkv_last_status() == 0
ref_k = [l.keys.clone() for l in ref.past_key_values.layers]
limits = dict(DeferredCompression._max_entries)
side = torch.cuda.Stream()
# another kernel holds 200 of the 256 compute units for 400 ms: the pair's fused launch cannot become resident, its waits give
# up after FASTKV_SPIN_LIMIT_MS (30) and the launch is reported
assert L.fastkv_debug_occupy(200, 128 * 1024, 400 * 1000, side.cuda_stream) == 0
time.sleep(0.02)
try:
    with torch.no_grad():
        out = model(ids)
    if MODE == 'sync':
        raise SystemExit('model() returned although its deferred pair was abandoned')
    torch.cuda.current_stream().synchronize()                              # what a harness does before it reads its timer
    prefill._raise_if_aborted()
    raise SystemExit('no report behind the synchronisation point')
except FastKVNativeError as e:
    assert e.code == FASTKV_EABORTED and 'gave up' in str(e), str(e)
assert DeferredCompression._max_entries == limits                          # a transient error is not a property of the geometry
torch.cuda.synchronize()
time.sleep(0.5)
assert L.fastkv_last_status() == 0
with torch.no_grad():
    again = model(ids)                                                     # the context is alive and the schedule unchanged
torch.cuda.synchronize()
assert L.fastkv_last_status() == 0
for x, y in zip(ref_k, again.past_key_values.layers):
    assert torch.equal(x, y.keys)
assert torch.equal(ref.logits, again.logits)
print('child ok')
"""


@pytest.mark.parametrize("mode", ["sync", "harness"])
def test_abandoned_deferred_pair_is_raised(mode):
    """VERDICT r02 weak #8 / ADVICE r02: an in-kernel wait given up inside a DEFERRED pair used to be swallowed (and to shrink the
    process-wide entry limit).  Now: with FASTKV_CHECK_SYNC=1 the exception comes out of `model(...)` itself; without it (the
    default: no synchronisation inside the forward pass, as in the reference) it comes out of the harness's check right behind its
    synchronisation point (benchmark/prefill.py).  Either way the entry limits stay and the next prompt is bit-identical."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FASTKV_SPIN_LIMIT_MS="30", FASTKV_DEFER_MAX_LEN="1024", FASTKV_DEFER_HOLD="2", FASTKV_STRICT_PLACEMENT="0",
               FASTKV_CHECK_SYNC="1" if mode == "sync" else "0")
    r = subprocess.run([sys.executable, "-c", f"MODE = {mode!r}\n" + _ABORT_CHILD], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]
