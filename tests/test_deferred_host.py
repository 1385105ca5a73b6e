"""Host logic of fastkv_amd.cluster.DeferredCompression on the CPU: which layers run together, when, and what happens when a
launch sequence refuses more entries than it can hold.  The device calls are replaced by recorders (the product has no CPU
path; the kernels themselves are checked on the GPU: tests/test_hip_parity.py, tests/test_wiring_gpu.py)."""
import pytest
import torch

from fastkv_amd import cluster as C
from fastkv_amd._lib import FastKVNativeError


class Recorder:
    def __init__(self, max_entries):
        self.max_entries, self.calls = max_entries, []

    def entries(self, qs, ks, vs, window, ksize, pooling, cap, tsp_len=0, order="score", outs=None, return_indices=False, q_window=False):
        assert not q_window or all(q.shape[2] == window for q in qs)     # (q_window: a waiting layer keeps the window rows of q only)
        self.q_window_seen = q_window
        if len(qs) > self.max_entries:
            raise FastKVNativeError("unsupported configuration", code=-4)
        self.calls.append(("entries", len(qs), tsp_len))
        n = len(qs)
        ko = [torch.full((1, 2, cap, 4), float(k[0, 0, 0, 0])) for k in ks]
        tsp = torch.arange(n * tsp_len).view(n, tsp_len) if tsp_len else None
        return ko, [t.clone() for t in ko], tsp

    def single(self, q, k, v, window, ksize, pooling, cap, tsp_len=0, order="score", out=None, **kw):
        self.calls.append(("single", 1, tsp_len))
        ko = torch.full((1, 2, cap, 4), float(k[0, 0, 0, 0]))
        return ko, ko.clone(), (torch.arange(tsp_len)[None] if tsp_len else None)


def _layer(i, S):
    k = torch.full((1, 2, S, 4), float(i), dtype=torch.float16)
    return torch.zeros(1, 8, S, 4, dtype=torch.float16), k, k.clone()


@pytest.fixture
def rec(monkeypatch):
    r = Recorder(max_entries=2)
    monkeypatch.setattr(C.ops, "update_kv_entries", r.entries)
    monkeypatch.setattr(C.ops, "update_kv", r.single)
    monkeypatch.setattr(C.DeferredCompression, "_max_entries", {})
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    return r


def _cluster(tsp_layer=False, cap=16, tsp_length=64):
    c = C.FastKVCluster()
    c.max_capacity_prompt, c.tsp_layer, c.tsp_length, c.window_size, c.kernel_size = cap, tsp_layer, tsp_length, 8, 7
    return c


def test_long_layers_pair_up_and_the_tsp_layer_takes_its_peer_along(rec):
    d = C.DeferredCompression(max_len=100, hold_long=2)
    done = {}
    for i in range(5):                                             # layers 0..4 long, layer 5 = TSP layer
        q, k, v = _layer(i, 1000)
        ready = d.add(i, _cluster(), k, q, v)
        for j, ko, vo in ready:
            done[j] = float(ko[0, 0, 0, 0])
    assert rec.calls == [("entries", 2, 0), ("entries", 2, 0)] and sorted(done) == [0, 1, 2, 3]     # layer 4 waits
    q, k, v = _layer(5, 1000)
    ko, vo, tsp, ready = d.add_tsp_layer(5, _cluster(tsp_layer=True), k, q, v)
    assert rec.calls[-1] == ("entries", 2, 64) and [r[0] for r in ready] == [4]
    assert float(ko[0, 0, 0, 0]) == 5.0 and float(ready[0][1][0, 0, 0, 0]) == 4.0                  # own rows / the peer's rows
    assert tsp.shape == (1, 64) and int(tsp[0, 0]) == 64                                             # the TSP layer's row of the pair
    assert d.flush() == []
    for j, val in done.items():
        assert val == float(j)


@pytest.mark.parametrize("q_window", [False, True])
def test_groups_of_four_and_the_tsp_layer_takes_every_waiting_peer(rec, q_window):
    rec.max_entries = 16
    d = C.DeferredCompression(max_len=100, hold_long=4, q_window=q_window)
    done = {}
    for i in range(7):                                             # layers 0..6 long, layer 7 = TSP layer
        q, k, v = _layer(i, 1000)
        for j, ko, vo in d.add(i, _cluster(), k, q, v):
            done[j] = float(ko[0, 0, 0, 0])
    assert rec.calls == [("entries", 4, 0)] and sorted(done) == [0, 1, 2, 3]                         # layers 4-6 wait
    q, k, v = _layer(7, 1000)
    ko, vo, tsp, ready = d.add_tsp_layer(7, _cluster(tsp_layer=True), k, q, v)
    assert rec.calls[-1] == ("entries", 4, 64) and [r[0] for r in ready] == [4, 5, 6]
    assert float(ko[0, 0, 0, 0]) == 7.0 and [float(r[1][0, 0, 0, 0]) for r in ready] == [4.0, 5.0, 6.0]
    assert tsp.shape == (1, 64) and int(tsp[0, 0]) == 3 * 64                                           # the LAST entry's row: the TSP layer's
    assert d.flush() == [] and rec.q_window_seen == q_window


def test_short_layers_wait_for_the_end_and_shrink_to_what_fits(rec):
    rec.max_entries = 3
    d = C.DeferredCompression(max_len=4096, hold_long=2)
    for i in range(7):
        q, k, v = _layer(i, 200)
        assert d.add(i, _cluster(), k, q, v) == []
    out = d.flush()
    assert [o[0] for o in out] == list(range(7)) and all(float(o[1][0, 0, 0, 0]) == float(o[0]) for o in out)
    assert rec.calls == [("entries", 3, 0), ("entries", 3, 0), ("entries", 1, 0)]                     # 7 refused, 3 fits: 3 + 3 + 1
    # the limit is remembered per geometry: the next prompt asks for 3 at once
    rec.calls.clear()
    for i in range(3):
        q, k, v = _layer(i, 200)
        d.add(i, _cluster(), k, q, v)
    d.flush()
    assert rec.calls == [("entries", 3, 0)]


def test_layers_that_keep_everything_are_not_taken(rec):
    d = C.DeferredCompression()
    q, k, v = _layer(0, 10)
    assert d.add(0, _cluster(cap=512), k, q, v) is None and d.add_tsp_layer(1, _cluster(True, cap=512), k, q, v) is None
    assert rec.calls == [] and d.flush() == []
    # a TSP layer without a waiting peer runs alone, at once
    q, k, v = _layer(2, 1000)
    ko, vo, tsp, ready = d.add_tsp_layer(2, _cluster(tsp_layer=True), k, q, v)
    assert rec.calls == [("single", 1, 64)] and ready == [] and tsp.shape == (1, 64)


def test_an_aborted_launch_is_raised_not_absorbed(rec, monkeypatch):
    """FASTKV_EABORTED (an EARLIER launch of the process gave up a wait; the library reports it once, from the next call) and
    FASTKV_ELAUNCH are errors of the run, not "does not fit in one launch sequence": they reach the caller, the per-geometry
    entry limit is left alone, and the entries are still pending for a caller that handles the error and flushes again
    (ADVICE r02 / VERDICT r02 weak #8: they used to be swallowed and to shrink `_max_entries` for the whole process)."""
    state = {"fail": -5}

    def entries(qs, *a, **kw):
        if state["fail"]:
            code, state["fail"] = state["fail"], 0
            raise FastKVNativeError("an earlier fused launch gave up", code=code)
        return rec.entries(qs, *a, **kw)

    monkeypatch.setattr(C.ops, "update_kv_entries", entries)
    rec.max_entries = 16
    d = C.DeferredCompression(max_len=4096, hold_long=2)
    for i in range(4):
        q, k, v = _layer(i, 200)
        d.add(i, _cluster(), k, q, v)
    with pytest.raises(FastKVNativeError) as ei:
        d.flush()
    assert ei.value.code == -5 and C.DeferredCompression._max_entries == {}
    out = d.flush()                                                # the report was consumed: the same entries run now, all four at once
    assert [o[0] for o in out] == [0, 1, 2, 3] and rec.calls == [("entries", 4, 0)]
    # the same for the TSP pair: the error is not taken as "run the TSP layer alone"
    state["fail"] = -3
    q, k, v = _layer(7, 1000)
    d.add(7, _cluster(), k, q, v)                                  # (hold_long = 2: waits for a peer)
    q, k, v = _layer(8, 1000)
    with pytest.raises(FastKVNativeError) as ei:
        d.add_tsp_layer(8, _cluster(tsp_layer=True), k, q, v)
    assert ei.value.code == -3


@pytest.mark.parametrize("layers,tsp_idx,max_len,want", [(32, 15, 4096, [8, 8, 8, 8]), (36, 17, 4096, [8, 8, 2, 8, 8, 2]),
                                                          (32, 15, 8192, [8, 8, 16]), (36, 17, 8192, [8, 8, 2, 18])])
def test_published_recipe_schedule_and_state(rec, monkeypatch, layers, tsp_idx, max_len, want):
    """The reference's published recipe (/root/reference/scripts/eval_prefill.sh:4-12; scripts2/eval_prefill.sh:37-47 for the 36-layer
    Ministral-8B with TSP layer 17) at 32,768 tokens, host logic only: `compress_fastkv` pushes the rates (utils.py:25-46), every call
    rewrites `max_capacity_prompt` (3276 in front of and behind the TSP layer: int(32768 * 0.1) and int(6553 * 0.5)) and the TSP layer
    its `tsp_length` (6553) (utils.py:86-87, :123-124) -- and the 6553-token layers behind the TSP layer, longer than
    FASTKV_DEFER_MAX_LEN = 4096 (rounds 2-4), run as LONG layers in groups of eight like the ones in front of it (the constant budget
    never does that); under the default of round 5 (8192) they wait for the end of the forward pass and run as ONE launch sequence."""
    import types
    rec.max_entries = 32
    S = 32768
    mods = [types.SimpleNamespace(self_attn=types.SimpleNamespace(kv_cluster=C.FastKVCluster())) for _ in range(layers)]
    model = types.SimpleNamespace(model=types.SimpleNamespace(layers=mods))
    args = types.SimpleNamespace(window_size=[8] * layers, kernel_size=[7] * layers, pooling="maxpool", max_capacity_prompts=512, tsp_len=2048,
                                 tsp_rate=0.2, eviction_mode="proportional", tsp_idx=tsp_idx, retain_rate=0.1)
    C.compress_fastkv(model, args)
    d = C.DeferredCompression(max_len=max_len, hold_long=8)
    done, s_now = {}, S
    for i, m in enumerate(mods):
        cl = m.self_attn.kv_cluster
        q, k, v = _layer(i, s_now)
        if cl.tsp_layer:
            ko, vo, tsp, ready = d.add_tsp_layer(i, cl, k, q, v)
            done[i] = ko
            s_now = tsp.shape[1]
        else:
            ready = d.add(i, cl, k, q, v)
        for j, ko, vo in ready:
            done[j] = ko
    for j, ko, vo in d.flush():
        done[j] = ko
    assert [c[1] for c in rec.calls] == want and all(c[0] == "entries" for c in rec.calls)
    assert s_now == 6553 and [c[2] for c in rec.calls if c[2]] == [6553]
    assert sorted(done) == list(range(layers)) and all(t.shape[2] == 3276 and float(t[0, 0, 0, 0]) == float(j) for j, t in done.items())
    assert [m.self_attn.kv_cluster.max_capacity_prompt for m in mods] == [3276] * layers
    assert mods[tsp_idx].self_attn.kv_cluster.tsp_length == 6553
    assert [m.self_attn.kv_cluster.retain_rate for m in mods] == [0.1] * (tsp_idx + 1) + [0.5] * (layers - tsp_idx - 1)


def test_group_size_is_capped_by_the_bytes_the_waiting_layers_hold(monkeypatch):
    """FASTKV_DEFER_HOLD = 8 no longer scales the held q / k / v with the prompt and the batch without bound."""
    import types
    from baselines.fastkv._wiring import defer_hold_for
    cfg = types.SimpleNamespace(num_attention_heads=32, num_key_value_heads=8, head_dim=128, hidden_size=4096)
    monkeypatch.delenv("FASTKV_DEFER_HOLD", raising=False)
    monkeypatch.delenv("FASTKV_DEFER_HOLD_GIB", raising=False)
    assert defer_hold_for(cfg, 1, 32768) == 8 and defer_hold_for(cfg, 1, 8192) == 8
    assert defer_hold_for(cfg, 1, 131072) == 3 and defer_hold_for(cfg, 4, 32768) == 3
    assert defer_hold_for(cfg, 16, 131072) == 1
    monkeypatch.setenv("FASTKV_DEFER_HOLD", "2")
    assert defer_hold_for(cfg, 1, 32768) == 2
    monkeypatch.setenv("FASTKV_DEFER_HOLD", "16")
    monkeypatch.setenv("FASTKV_DEFER_HOLD_GIB", "64")
    assert defer_hold_for(cfg, 1, 32768) == 16


def test_the_end_of_pass_regime_is_capped_by_bytes_too(monkeypatch):
    import types
    from baselines.fastkv._wiring import defer_max_len_for
    cfg = types.SimpleNamespace(num_attention_heads=32, num_key_value_heads=8, head_dim=128, hidden_size=4096, num_hidden_layers=32)
    monkeypatch.delenv("FASTKV_DEFER_MAX_LEN", raising=False)
    monkeypatch.delenv("FASTKV_DEFER_HOLD_GIB", raising=False)
    assert defer_max_len_for(cfg, 1) == 8192                       # 32 layers x 8192 tokens x 12 KiB = 3.2 GB <= 4 GiB
    assert defer_max_len_for(cfg, 2) == 5461                       # two batch rows: the regime shrinks instead of holding 6.4 GB
    monkeypatch.setenv("FASTKV_DEFER_MAX_LEN", "4096")
    assert defer_max_len_for(cfg, 1) == 4096
