#!/usr/bin/env python3
"""bench.py -- FastKV hot path on MI355X: prefill hot-path throughput + per-kernel roofline + CPU baseline.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 is launched by torch.distributed.run, one rank per GPU).
One STEP = the whole FastKV hot path of ONE Llama-3-8B prefill at 32k context (BASELINE.json configs[1]:
TSP layer 15, budget 2048, window 8, kernel 7, maxpool, fp16):
    layers 0..15   update_kv at S=32768 (score -> select -> compact), layer 15 also produces the TSP index
    TSP propagation  hidden [1,32768,4096] -> [1,2048,4096] row gather (llama_model.py:252-259)
    layers 16..31  update_kv at S=2048 (k == n permutation case) -- handed to DeferredCompression and run as ONE launch sequence
                   after the last layer, as baselines/fastkv/_wiring.py does by default; layers 0..15 likewise in groups of 8
                   (`layer_by_layer` in the line: the 32 sequential calls of the reference's schedule, FASTKV_DEFER=0;
                   `hold_2`: the round-2 schedule, pairs)
on synthetic fp16 Q/K/V (seeded torch.randn on the device, one distinct tensor set per layer so nothing is
cache-resident across layers), inputs already in HBM.  `value` = prompt tokens / hot-path time, summed over ranks.
N>1: every rank runs its own prompt (independent prompts shard with no exchange) -> weak scaling.

Extra objects in the JSON line:
  roofline      dominant launch = the scoring of a group of `FASTKV_DEFER_HOLD` (8) 32k layers in ONE rolling launch of score_fused (csrc/fused.hip):
                algorithmic bytes (K once + Q window per layer) / average launch duration, measured with HIP events on the launch stream on
                rotating layer sets; `roofline_pair_launch` / `roofline_one_layer_launch`: the launches of two layers / one layer beside it.
  kernels       every kernel: launches per step, average microseconds (same instrumented replay).
  compact       the KV gather/compact kernel: per-layer latency at this config and GB/s at the "roofline shape"
                (same row geometry, 32 layers' worth in one launch, beyond the 256 MiB Infinity Cache).
  fp32_pipe_view  the same dominant-kernel launches priced against the FP32 matrix pipe (the contraction is an fp32 fma chain).
  seq_sharded_weak / seq_sharded_128k / tp  (N>1 only, by default) the paths with a real exchange step, each run in a freshly
                spawned child process per rank (own rendezvous port), so that a failing collective cannot take this line down:
                ONE prompt of N*32k tokens sharded on the sequence axis; ONE 128k prompt over N shards; Llama-3-70B heads over N ranks.
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's update_kv) timed on this box's host cores on a
                bounded sample of the same workload.
  ttft          (N=1) whole-model numbers around the path, random-init Llama-3-8B geometry: prefill TTFT at 32k, FastKV vs the
                full-KV arm (benchmark/prefill.py), and `decode`: ms per greedy token over the compressed cache (benchmark/e2e.py).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch

CFG = dict(model="Llama-3-8B (geometry only)", H=32, Hkv=8, D=128, hidden=4096, layers=32, S=32768, window=8, kernel=7,
           pooling="maxpool", budget=2048, tsp_len=2048, tsp_idx=15)
FP32_MATRIX_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32, = the fp32 vector peak)
HBM_PEAK_GBPS = 8000.0        # MI355X spec (guides/MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def make_layer_inputs(S, gen, dev):
    """[B,S,H,D]-physical / [B,H,S,D]-logical fp16 tensors, as the attention module produces them."""
    H, Hkv, D = CFG["H"], CFG["Hkv"], CFG["D"]
    q = torch.randn(1, S, H, D, generator=gen, device=dev, dtype=torch.float16).transpose(1, 2)
    k = torch.randn(1, S, Hkv, D, generator=gen, device=dev, dtype=torch.float16).transpose(1, 2)
    v = torch.randn(1, S, Hkv, D, generator=gen, device=dev, dtype=torch.float16).transpose(1, 2)
    return q, k, v


# The reference's PUBLISHED recipe (/root/reference/scripts/eval_prefill.sh:4-12: --tsp_idx 15 --tsp_rate 0.2 --retain_rate 0.1 --eviction_mode
# proportional; pooling / window / kernel at the harness defaults): at 32k every layer up to the TSP layer keeps int(32768 * 0.1) = 3276 rows,
# the TSP layer hands int(32768 * 0.2) = 6553 tokens on, the 16 layers behind it keep int(6553 * 0.5) = 3276 of 6553 (utils.py:41-46, :86-87, :123-124)
RECIPE = dict(eviction_mode="proportional", retain_rate=0.1, tsp_rate=0.2)


class HotPathPrefill:
    def __init__(self, dev, seed, recipe=False):
        from fastkv_amd import FastKVCluster, compress_fastkv
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        L, S = CFG["layers"], CFG["S"]
        self.recipe = recipe
        self.tsp_len = int(S * RECIPE["tsp_rate"]) if recipe else CFG["tsp_len"]       # tokens behind the TSP layer
        self.layers_in = []
        for i in range(L):
            s_i = S if i <= CFG["tsp_idx"] else self.tsp_len
            self.layers_in.append(make_layer_inputs(s_i, gen, dev))
        self.hidden = torch.randn(1, S, CFG["hidden"], generator=gen, device=dev, dtype=torch.float16)
        self.position_ids = torch.arange(S, device=dev)[None]
        # configuration pushed exactly like benchmark/prefill.py -> set_model -> compress_fastkv (utils.py:25-46)
        layers = [types.SimpleNamespace(self_attn=types.SimpleNamespace(kv_cluster=FastKVCluster())) for _ in range(L)]
        self.model = types.SimpleNamespace(model=types.SimpleNamespace(layers=layers))
        args = types.SimpleNamespace(window_size=[CFG["window"]] * L, kernel_size=[CFG["kernel"]] * L, pooling=CFG["pooling"],
                                     max_capacity_prompts=CFG["budget"], tsp_len=CFG["tsp_len"], tsp_rate=0.2,
                                     eviction_mode="constant", tsp_idx=CFG["tsp_idx"], retain_rate=0.1)
        if recipe:
            args.max_capacity_prompts, args.eviction_mode = 512, RECIPE["eviction_mode"]     # (the budget flag is at its default and unused)
            args.retain_rate, args.tsp_rate = RECIPE["retain_rate"], RECIPE["tsp_rate"]
        compress_fastkv(self.model, args)
        self.clusters = [l.self_attn.kv_cluster for l in layers]
        self.defer = os.environ.get("FASTKV_DEFER", "1") != "0"
        self.defer_max_len = int(os.environ.get("FASTKV_DEFER_MAX_LEN", "8192"))     # (baselines/fastkv/_wiring.py defer_max_len_for)
        self.defer_hold = int(os.environ.get("FASTKV_DEFER_HOLD", "8"))

    def step(self):
        """The calls the patched model makes during one prefill (baselines/fastkv/_wiring.py), without the model around them:
        per-layer `update_kv` for the layers that see the whole prompt, the TSP gather, and -- as the wiring does by default
        over the reference's cache type -- the 16 layers behind the TSP layer handed to DeferredCompression and compressed in
        one launch sequence at the end (`self.defer = False`: layer by layer, as the reference)."""
        from fastkv_amd import ops
        from fastkv_amd.cluster import DeferredCompression
        G = CFG["H"] // CFG["Hkv"]
        cache = [None] * len(self.layers_in)
        hidden = None
        defer = DeferredCompression(max_len=self.defer_max_len, hold_long=self.defer_hold) if self.defer else None
        for i, (q, k, v) in enumerate(self.layers_in):
            cl = self.clusters[i]
            if defer is not None and defer.eligible(cl, k, q) and not cl.tsp_layer:
                ready = defer.add(i, cl, k, q, v)
                if ready is None:
                    cache[i] = (k, v)
                else:
                    for j, ko, vo in ready:
                        cache[j] = (ko, vo)
                continue
            if defer is not None and defer.eligible(cl, k, q) and cl.tsp_layer:
                res = defer.add_tsp_layer(i, cl, k, q, v)                    # runs now, with the waiting layer 14 as its peer
                if res is None:
                    cache[i] = (k, v)
                    continue
                ko, vo, tsp, ready = res
                cache[i] = (ko, vo)
                for j, kr, vr in ready:
                    cache[j] = (kr, vr)
                if tsp is not None:
                    hidden, _pos = ops.tsp_propagate(self.hidden, self.position_ids, tsp)     # llama_model.py:254-257, one launch (as the wiring)
                continue
            ko, vo, tsp = cl.update_kv(k, q, v, None, G, i)
            cache[i] = (ko, vo)
            if cl.tsp_layer and tsp is not None:
                hidden, _pos = ops.tsp_propagate(self.hidden, self.position_ids, tsp)         # llama_model.py:254-257, one launch (as the wiring)
        if defer is not None:
            for i, ko, vo in defer.flush():
                cache[i] = (ko, vo)
        return cache, hidden


def profile_read(lib):
    n = lib.fastkv_profile_kernels()
    counts = (ctypes.c_int64 * n)()
    ms = (ctypes.c_double * n)()
    rc = lib.fastkv_profile_read(counts, ms)
    assert rc == 0
    return {lib.fastkv_profile_kernel_name(i).decode(): (int(counts[i]), float(ms[i])) for i in range(n)}


def compact_roofline_shape(lib, dev, steps):
    """Same row geometry as the 32k config (256-B rows at 2 KiB pitch, 2040+8 rows per head) but 32 'layers' in one
    launch: 2 x 2 GiB sources, 541 MB of algorithmic traffic, nothing served from the 256 MiB Infinity Cache.
    Both row orders: `index` (rows in ascending position: a pure gather) and `score` (the reference's order, utils.py:113,
    what the product's default path produces: with this shape's 256 heads the winners' slots come from one grouping pass per
    head, `rank_group_kernel`, launched inside the same timed bracket as the copy; a single layer's 8 heads are ranked by
    comparison counting inside the copy kernel instead -- `compact.per_layer_avg_us`).
    THREE disjoint source sets are rotated (12 GiB in all) and every call's kernel bracket (HIP events on the launch stream) is
    read on its own: the line carries min / median / max over the calls, so that a call-to-call spread is visible."""
    import statistics
    from fastkv_amd import ops
    B, Hkv, S, D, W, cap = 32, CFG["Hkv"], CFG["S"], CFG["D"], CFG["window"], CFG["budget"]
    sets = [(torch.randn(B, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2),
             torch.randn(B, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)) for _ in range(3)]
    sc = torch.rand(B * Hkv, S - W, device=dev).half()
    idx = ops.select(sc, cap - W, "index").view(B, Hkv, cap - W).contiguous()     # realistic per-head index sets
    sc3 = sc.view(B, Hkv, S - W)
    nbytes = 2 * (2 * B * Hkv * cap * D * 2) + B * Hkv * (cap - W) * 8
    res = {"shape": f"B={B} (32 layers stacked), Hkv={Hkv}, S={S}, D={D}, cap={cap}", "bytes": nbytes, "source_sets_rotated": len(sets)}
    reps = max(4, steps // 2)
    for order, kw in (("index", {}), ("score", {"scores": sc3})):
        for k, v in sets:
            ops.compact(k, v, idx, W, **kw)
        torch.cuda.synchronize()
        us = []
        for _ in range(reps):
            for k, v in sets:
                profile_read(lib)
                lib.fastkv_profile_enable(1)
                ops.compact(k, v, idx, W, **kw)
                torch.cuda.synchronize()
                lib.fastkv_profile_enable(0)
                cnt, ms = profile_read(lib)["compact_kv"]                    # (score order: the grouping pass + the copy, one bracket)
                us.append(ms / cnt * 1e3)
        us.sort()
        med = statistics.median(us)
        res[order] = {"avg_us": round(med, 2), "min_us": round(us[0], 2), "median_us": round(med, 2), "max_us": round(us[-1], 2), "calls": len(us),
                      "achieved_GBps": round(nbytes / (med * 1e-6) / 1e9, 1), "frac_of_8TBps": round(nbytes / (med * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                      "frac_of_8TBps_best": round(nbytes / (us[0] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                      "frac_of_8TBps_worst": round(nbytes / (us[-1] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}
    del sets
    return res


def _effective_cpus():
    """(logical CPUs in this process's affinity mask, CPU quota of its cgroup in cores or None).  A container may see all of the host's
    CPUs in its mask and still be throttled to a few cores' worth of time by cpu.max: OpenMP threads beyond the quota are descheduled
    in turns, and every barrier then waits for the slowest."""
    # (FASTKV_BENCH_NCPU: the mask as the PARENT saw it -- with OMP_PROC_BIND set, libgomp binds a process's initial thread to its
    # first place when it is loaded, i.e. at `import torch`: the child's own mask then reads 1 CPU)
    n = int(os.environ.get("FASTKV_BENCH_NCPU", "0")) or (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    quota = None
    try:
        f = open("/sys/fs/cgroup/cpu.max").read().split()                     # cgroup v2: "<quota> <period>" or "max <period>"
        if f and f[0] != "max":
            quota = float(f[0]) / float(f[1])
    except (OSError, ValueError, IndexError):
        try:
            q_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())      # cgroup v1
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q_ > 0 and p_ > 0:
                quota = q_ / p_
        except (OSError, ValueError):
            pass
    return n, quota


def cpu_baseline_child_main() -> int:
    """`bench.py --cpu-baseline-child`: the CPU leg in its OWN process -- started by rank 0 of the N = 1 run before it creates its
    workload, with OMP_PROC_BIND=close / OMP_PLACES=cores in the environment and no GPU work anywhere in it.  Nothing else of the bench
    shares the process: no torch intra-op pool spinning beside the oracle's OpenMP team (torch is pinned to ONE thread here and only
    makes the input tensors), no allocator state of the GPU run.  Prints one `CPU_JSON {...}` line.

    What is timed: the oracle (oracle/fastkv_oracle.c, a port of utils.py:80-134) on synthetic fp16 inputs of the step's shapes:
    layers 0, 1 (S = 32768), 15 (S = 32768, TSP), 16, 17 (S = 2048) + the hidden gather; 1 warm-up + 5 timed calls each, the MEDIAN
    scaled to the step's 15 + 1 + 16 layers (`best` beside it).  Contraction contracts: "fmaf" -- the fp32 fma chain, a CPU's native
    arithmetic, bit for bit the reference's own CPU matmul and the contract the GPU headline runs since round 6 -- is the baseline's
    `value`; "mfma16" -- the integer restatement of the gfx950 matrix instruction, the GPU's opt-in fast mode -- is timed beside it on
    layers 0 and 16."""
    import statistics
    torch.set_num_threads(1)
    from oracle import fastkv_oracle as O
    from oracle.fastkv_oracle import OracleFastKVCluster
    H, Hkv, D, S, W = CFG["H"], CFG["Hkv"], CFG["D"], CFG["S"], CFG["window"]
    G = H // Hkv
    ncpu, quota = _effective_cpus()
    load0 = os.getloadavg()[0] if hasattr(os, "getloadavg") else None
    gen = torch.Generator().manual_seed(4321)

    def layer(s_i):
        q = torch.zeros(1, s_i, H, D, dtype=torch.float16)                   # (only the last W rows of q are ever read: utils.py:93)
        q[:, s_i - W:] = torch.randn(1, W, H, D, generator=gen).half()
        k = torch.randn(1, s_i, Hkv, D, generator=gen).half()
        v = torch.randn(1, s_i, Hkv, D, generator=gen).half()
        return q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)

    q0, k0, v0 = layer(S)
    O.set_contraction("fmaf")
    limit = ncpu if quota is None else max(1, min(ncpu, int(quota)))
    cands = sorted({min(limit, x) for x in (16, 32, 64, 128)})
    picks = {}
    for nt in cands:
        O.set_threads(nt)
        O.update_kv(q0, k0, v0, W, CFG["kernel"], CFG["pooling"], CFG["budget"], 0, "score")
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.update_kv(q0, k0, v0, W, CFG["kernel"], CFG["pooling"], CFG["budget"], 0, "score")
            ts.append(time.perf_counter() - t0)
        picks[nt] = round(statistics.median(ts) * 1e3, 2)
    cores = min(picks, key=picks.get)
    O.set_threads(cores)

    def timed(inp, kw, runs=5):
        q, k, v = inp
        OracleFastKVCluster(**kw).update_kv(k, q, v, None, G, 0)             # warm-up (page-in, thread team)
        ts, out = [], None
        for _ in range(runs):
            oc = OracleFastKVCluster(**kw)
            t0 = time.perf_counter()
            out = oc.update_kv(k, q, v, None, G, 0)
            ts.append(time.perf_counter() - t0)
        return ts, out

    base = dict(window_size=W, max_capacity_prompt=CFG["budget"], kernel_size=CFG["kernel"], pooling=CFG["pooling"], tsp_length=CFG["tsp_len"])
    inputs = {0: (q0, k0, v0), 1: layer(S), CFG["tsp_idx"]: layer(S), 16: layer(CFG["tsp_len"]), 17: layer(CFG["tsp_len"])}
    per_layer, tsp_out = {}, None
    for i, inp in inputs.items():
        ts, out = timed(inp, dict(base, tsp_layer=(i == CFG["tsp_idx"])))
        per_layer[i] = ts
        if i == CFG["tsp_idx"]:
            tsp_out = out[2]
    hid = torch.randn(S, CFG["hidden"], generator=gen).half()
    tg = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.gather_rows(hid, tsp_out[0].contiguous())
        tg.append(time.perf_counter() - t0)

    def step_of(f):
        return 15 * (f(per_layer[0]) + f(per_layer[1])) / 2 + f(per_layer[CFG["tsp_idx"]]) + 16 * (f(per_layer[16]) + f(per_layer[17])) / 2 + f(tg)

    step_med, step_best = step_of(statistics.median), step_of(min)
    # the GPU's opt-in fast contract (restated in integer arithmetic: 22x slower on a CPU), on two of the sampled layers
    O.set_contraction("mfma16")
    m_pre, _ = timed(inputs[0], dict(base, tsp_layer=False), runs=3)
    m_post, _ = timed(inputs[16], dict(base, tsp_layer=False), runs=3)
    O.set_contraction("fmaf")
    m_step = 16 * statistics.median(m_pre) + 16 * statistics.median(m_post) + statistics.median(tg)
    res = {"value": round(S / step_med, 1), "unit": "tokens/s", "cores": cores, "kind": "port",
           "ms_per_step": round(step_med * 1e3, 1), "ms_per_step_best": round(step_best * 1e3, 1),
           "contraction": "fmaf (fp32 fma chain: the CPU's native arithmetic, bit for bit the reference's own CPU matmul, the GPU headline's contract)",
           "per_layer_ms": {str(i): {"median": round(statistics.median(t) * 1e3, 2), "best": round(min(t) * 1e3, 2), "runs": len(t)}
                            for i, t in per_layer.items()},
           "hidden_gather_ms": round(statistics.median(tg) * 1e3, 2),
           "thread_candidates_ms_layer0": {str(k_): v_ for k_, v_ in picks.items()},
           "mfma16_contract": {"contraction": "mfma16 (integer restatement of v_mfma_f32_32x32x16_f16)", "ms_per_step": round(m_step * 1e3, 1),
                                         "tokens_per_s": round(S / m_step, 1), "layer0_ms": round(statistics.median(m_pre) * 1e3, 2),
                                         "layer16_ms": round(statistics.median(m_post) * 1e3, 2),
                                         "sample": "layers 0 (S=32768) and 16 (S=2048), 1 warm-up + 3 timed, median, scaled to 16 + 16 layers + the gather"},
           "host": {"logical_cpus_in_mask": ncpu, "cgroup_cpu_quota_cores": quota, "loadavg_1m_before": load0,
                    "loadavg_1m_after": os.getloadavg()[0] if hasattr(os, "getloadavg") else None,
                    "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES")},
           "sample": "own process, before the GPU run starts; oracle update_kv on layers 0,1 (S=32768), 15 (TSP), 16,17 (S=2048) + hidden gather, "
                     "1 warm-up + 5 timed each, MEDIAN scaled to 15+1+16 layers; OpenMP threads picked from "
                     f"{cands} (median of 3 on layer 0) on {ncpu} logical CPUs" + (f", cgroup quota {quota:.1f} cores" if quota else "")}
    print("CPU_JSON " + json.dumps(res), flush=True)
    return 0


def cpu_baseline(timeout_s=240):
    """Runs the CPU leg (cpu_baseline_child_main) in a child process and returns its object.  Called by rank 0 of the N = 1 run before
    it has created its workload or touched the GPU with anything heavy: the host cores are idle, no thread pool of this process
    exists in the child."""
    import subprocess
    env = {k_: v_ for k_, v_ in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") and
           not k_.startswith("TORCHELASTIC_")}
    env["FASTKV_BENCH_NCPU"] = str(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    env.setdefault("OMP_PROC_BIND", "close")
    env.setdefault("OMP_PLACES", "cores")
    env.pop("OMP_NUM_THREADS", None)                                          # (the launcher pins it to 8 for the ranks; the oracle sets its own)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child"], env=env, capture_output=True, text=True,
                           timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"the CPU leg did not finish within {timeout_s} s", "kind": "port"}
    for line in r.stdout.splitlines():
        if line.startswith("CPU_JSON "):
            return json.loads(line[len("CPU_JSON "):])
    return {"error": f"the CPU leg exited with code {r.returncode}", "stderr_tail": r.stderr[-400:], "kind": "port"}


def whole_model_ttft(work):
    """BASELINE.json's other half: prefill TTFT of the whole model at 32k through the reference's own harness shape
    (benchmark/prefill.py: monkeypatch -> model(input_ids, attention_mask), device events), random-init Llama-3-8B geometry
    (no checkpoints on the box), PyTorch-ROCm SDPA attention; FastKV (TSP layer 15, budget 2048) vs the full-KV arm."""
    import gc
    del work.layers_in[:]
    work.hidden = None
    gc.collect()
    torch.cuda.empty_cache()
    from benchmark import prefill
    res = {}
    for method in ("fastkv", "fullkv"):
        args = prefill.parse_args(["--model_path", "llama3-8b", "--method", method, "--max_capacity_prompts", str(CFG["budget"]),
                                   "--tsp_len", str(CFG["tsp_len"]), "--tsp_idx", str(CFG["tsp_idx"]), "--context_lengths",
                                   str(CFG["S"]), "--num_warmups", "1", "--num_runs", "3", "--pooling", CFG["pooling"]])
        args.save_txt = False
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):          # the harness prints its own report; stdout carries ONE JSON line
            r = prefill.run(args)[0]
        res[method] = {"ttft_ms": round(r["ttft_s_mean"] * 1e3, 2), "tokens_per_s": round(r["tokens_per_s"], 1),
                       "max_mem_GiB": round(r["max_mem_GiB"], 2)}
        gc.collect()
        torch.cuda.empty_cache()
    res["speedup_vs_fullkv"] = round(res["fullkv"]["ttft_ms"] / res["fastkv"]["ttft_ms"], 3)
    res["config"] = "random-init Llama-3-8B geometry, 32768 all-ones token ids, B=1, fp16, SDPA attention, 1 warm-up + 3 runs"
    # decode over the compressed cache (the reference's benchmark/e2e.py:72-93): slab cache, the step's HIP
    # kernels, one captured step replayed
    try:
        from benchmark import e2e
        os.environ["FASTKV_SLAB_CACHE"] = "1"
        a = prefill.parse_args(["--model_path", "llama3-8b", "--method", "fastkv", "--max_capacity_prompts", str(CFG["budget"]),
                                "--tsp_len", str(CFG["tsp_len"]), "--tsp_idx", str(CFG["tsp_idx"]), "--context_lengths", str(CFG["S"]),
                                "--num_warmups", "1", "--num_runs", "2", "--pooling", CFG["pooling"], "--genlen", "64", "--random_tokens"])
        a.save_txt = False
        with contextlib.redirect_stdout(sys.stderr):
            r = e2e.run(a)[0]
        res["decode"] = {"ms_per_token": round(r["decode_ms_per_token"], 3), "prefill_ms": round(r["prefill_ms"], 1), "genlen": 64,
                         "cache_rows_layer0": r["final_cache_len_layer0"], "path": r["decode_path"],
                         "note": "greedy decode after a 32k prefill compressed to the budget; 16 GB of fp16 weights per token = 2.0 ms at 8 TB/s"}
    except Exception as e:   # noqa: BLE001 -- an extra; the contract line does not depend on it
        res["decode"] = {"error": repr(e)[:200]}
    finally:
        os.environ.pop("FASTKV_SLAB_CACHE", None)
        gc.collect()
        torch.cuda.empty_cache()
    return res


# ------------------------------------------------------------------------------------------------- multi-GPU legs
LEGS = ("seq_sharded_weak", "seq_sharded_128k", "tp", "sp_ttft_128k")
N_LEG_LAYERS = CFG["tsp_idx"] + 1          # the 16 layers that see the whole prompt; the last one is the TSP layer


def _leg_inputs(H, Hkv, S, dev, gen):
    D = CFG["D"]
    return [(torch.randn(1, S, H, D, generator=gen, device=dev, dtype=torch.float16).transpose(1, 2),
             torch.randn(1, S, Hkv, D, generator=gen, device=dev, dtype=torch.float16).transpose(1, 2),
             torch.randn(1, S, Hkv, D, generator=gen, device=dev, dtype=torch.float16).transpose(1, 2)) for _ in range(N_LEG_LAYERS)]


def run_leg(name, steps, rank, world, dev, dist):
    """One multi-GPU leg inside an initialised process group; returns the leg's JSON object (every rank computes it,
    rank 0 reports it).  Timing: barrier + synchronize on both sides of exactly `steps` steps, max over ranks."""
    import fastkv_amd.dist as FD
    gen = torch.Generator(device=dev)
    gen.manual_seed(2000 + rank)
    W, ks, pooling, cap, tsp = CFG["window"], CFG["kernel"], CFG["pooling"], CFG["budget"], CFG["tsp_len"]
    if name == "sp_ttft_128k":
        return run_sp_ttft(steps, rank, world, dev, dist)
    if name.startswith("seq_sharded"):
        S_r = CFG["S"] if name == "seq_sharded_weak" else 131072 // world
        S_glob = S_r * world
        layers = _leg_inputs(CFG["H"], CFG["Hkv"], S_r, dev, gen)
        lo = FD.HipLocalOps()
        lens = [S_r] * world

        def step():
            for i, (q, k, v) in enumerate(layers):
                FD.sp_update_kv(k, q, v, window_size=W, kernel_size=ks, pooling=pooling, capacity=cap,
                                tsp_len=tsp if i == N_LEG_LAYERS - 1 else 0, order="score", local_ops=lo, shard_lengths=lens)
        info = {"prompt_tokens": S_glob, "tokens_per_rank": S_r, "scaling": "weak" if name == "seq_sharded_weak" else "strong",
                "collectives_per_layer": 4, "kv_rows": "stay on the rank that owns them (no K/V bytes on the fabric)"}
    else:
        if 8 % world:
            return {"skipped": f"8 KV heads do not split over {world} ranks"}
        Hkv_l, H_l, S_glob = 8 // world, 64 // world, CFG["S"]
        layers = _leg_inputs(H_l, Hkv_l, S_glob, dev, gen)
        lo = FD.HipTPOps()

        def step():
            for i, (q, k, v) in enumerate(layers):
                FD.tp_update_kv(k, q, v, window_size=W, kernel_size=ks, pooling=pooling, capacity=cap,
                                tsp_len=tsp if i == N_LEG_LAYERS - 1 else 0, order="score", local_ops=lo)
        info = {"prompt_tokens": S_glob, "geometry": f"Llama-3-70B heads (H=64, Hkv=8) over {world} ranks: {H_l} query / {Hkv_l} KV heads each",
                "scaling": "strong", "collectives_per_step": 1}

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    c0 = sum(FD.COLLECTIVES.values())
    for _ in range(2):
        step()
    barrier()
    c1 = sum(FD.COLLECTIVES.values())
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ms = float(tt.item()) / steps * 1e3
    info.update({"layers": N_LEG_LAYERS, "steps": steps, "ms_per_step": round(ms, 3), "tokens_per_s": round(S_glob / (ms * 1e-3), 1),
                 "collectives_per_step_measured": (c1 - c0) // 2, "backend": dist.get_backend()})
    return info


def run_sp_ttft(steps, rank, world, dev, dist):
    """BASELINE.json configs[2] as a WHOLE-MODEL number: TTFT of ONE 131,072-token prompt, random-init Llama-3-8B geometry,
    sequence-parallel prefill over `world` ranks (fastkv_amd/sp_model.py: head-parallel attention between two all-to-alls with the
    head-local fused update_kv -- or, when the KV heads do not split over the ranks, K/V all-gather + lower-right causal attention
    + the sequence-sharded update_kv --, TSP re-shard at layer 15, replicated layers behind it).  1 warm-up + 2 timed runs."""
    from baselines.monkeypatch import replace_llama, set_model
    from benchmark import prefill
    from fastkv_amd.sp_model import SPContext, sp_prefill
    S = 131072
    if S % world:
        return {"skipped": f"131072 tokens do not split over {world} ranks"}
    a = prefill.parse_args(["--model_path", "llama3-8b", "--method", "fastkv", "--max_capacity_prompts", str(CFG["budget"]),
                            "--tsp_len", str(CFG["tsp_len"]), "--tsp_idx", str(CFG["tsp_idx"]), "--pooling", CFG["pooling"],
                            "--device", "cuda", "--save_txt", ""])
    a.save_txt = False
    a.context_lengths = [S]
    replace_llama("fastkv")
    torch.manual_seed(4242)                                       # the same random weights on every rank
    with torch.cuda.device(dev):
        model = prefill.build_model(a, dev)
    set_model(model, a)
    lens = [S // world] * world
    ids = torch.ones(1, lens[rank], dtype=torch.int64, device=dev)       # the reference's all-ones prompt (prefill.py:55)
    times = []
    for it in range(3):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            ctx = SPContext(shard_lengths=lens)
            out = sp_prefill(model, ids, ctx)
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if it:
            times.append(float(tt.item()))
        del out
    ms = sum(times) / len(times) * 1e3
    return {"prompt_tokens": S, "tokens_per_rank": lens[0], "model": "random-init Llama-3-8B geometry, fp16, 32 layers", "ttft_ms": round(ms, 2),
            "tokens_per_s": round(S / (ms * 1e-3), 1), "scaling": "strong", "runs": len(times), "backend": dist.get_backend(),
            "max_mem_GiB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
            "layout": ctx.layout(model.config.num_attention_heads, model.config.num_key_value_heads),
            "note": "whole-model TTFT (not the hot path alone).  layout 'heads': all-to-all to head shards, causal attention over the whole "
                    "prompt for H/P query heads per rank (1/P of the work each), head-local fused update_kv, all-to-all back: 2 collectives per "
                    "layer up to the TSP layer (+1 on it), none behind it; 'gather' (KV heads do not split over the ranks): K/V all-gather, 5 per layer"}


def leg_child_main(a):
    """`bench.py --leg NAME`: one rank of one leg, started by a rank of the main run (same RANK / WORLD_SIZE, own port)."""
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend == "nccl":
        try:
            dist.init_process_group("nccl", device_id=dev)
        except (TypeError, RuntimeError, ValueError):
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("nccl")
    else:
        dist.init_process_group(backend)
    try:
        res = run_leg(a.leg, a.steps, rank, world, dev, dist)
    finally:
        dist.destroy_process_group()
    if rank == 0:
        print("LEG_JSON " + json.dumps(res), flush=True)


def spawn_leg(name, idx, steps, rank, timeout_s=240):
    """Start this rank's child for leg `name` and wait for it; returns the leg object (rank 0) or a status."""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 101 + idx)
    for key in list(env):
        # the child is NOT a worker of the launcher's agent: with TORCHELASTIC_USE_AGENT_STORE inherited it would look for the
        # agent's store on the new port and wait for ever; without it rank 0's child hosts a fresh store there
        if key.startswith("TORCHELASTIC_"):
            env.pop(key)
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", name, "--gpus", os.environ.get("WORLD_SIZE", "1"), "--steps", str(steps)]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        so, se = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        proc.kill()                                                  # exactly the child this rank started
        so, se = proc.communicate()
        return {"error": f"leg timed out after {timeout_s} s on rank {rank}", "stderr_tail": se[-400:]}
    if proc.returncode != 0:
        return {"error": f"leg exited with code {proc.returncode} on rank {rank}", "stderr_tail": se[-600:]}
    for line in so.splitlines():
        if line.startswith("LEG_JSON "):
            return json.loads(line[len("LEG_JSON "):])
    return {"status": "ok (no report on this rank)"}


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(a) -> int:
    """`python bench.py --gpus N` without a launcher around it (no RANK in the environment): start the N ranks the way the
    contract's own command does -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- as a CHILD process, pass its output through and exit with its code.  This
    process makes no HIP call before or after (a process that has initialised the GPU must not be replaced or forked from),
    so nothing is exec'ed: the launcher is an ordinary subprocess.  The legs' rendezvous ports are MASTER_PORT + 101.. (spawn_leg)."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT") or 0) or _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")              # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:                                        # ranks print nothing but rank 0's ONE JSON line on stdout
        ln = ln.rstrip("\n")
        if ln.startswith("{") and '"metric"' in ln and line is None:
            line = ln
            print(line, flush=True)                              # at once: the long legs behind it report on stderr
        elif ln:
            print(ln, file=sys.stderr)
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks exited cleanly but printed no contract line", file=sys.stderr)
        rc = 1
    return rc


def rehearse(a, rank, world) -> int:
    """What tests/test_bench_contract.py runs on a box without a GPU: the launcher, the rendezvous and the one-line protocol of
    an N-rank run, with nothing measured (value null, "rehearsal": true -- not a bench line)."""
    seen, backend = 1, "none"
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("BENCH_BACKEND", "gloo")
        dist.init_process_group(backend)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        assert int(t.item()) == world * (world + 1) // 2
        seen = dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "prefill_hotpath_tokens_per_s", "value": None, "rehearsal": True, "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ranks_seen": seen, "backend": backend}), flush=True)
    return 0


def default_contraction() -> str:
    """The arithmetic contract the library runs when nothing is forced (csrc/capi.hip default_contract_f16): the fp32 fma chain -- the
    contraction that reproduces the reference's logits bit for bit -- unless FASTKV_CONTRACTION=mfma16 opts into the fast mode."""
    return "mfma16" if os.environ.get("FASTKV_CONTRACTION", "fmaf")[:1] in ("m", "M") else "fmaf"


def roofline_of_the_contract(rf, contraction, flop_per_launch):
    """`rf` = the HBM view of the dominant launch (algorithmic K bytes / HIP-event duration against 8 TB/s).  Under the mfma16 contract
    that IS the bound (the contraction is ~2 % of the fp16 matrix peak).  Under the fp32-fma-chain contract (the default since round
    6) the launch is bound by the FP32 lanes the matrix and the vector pipe share: 32 flop per K byte on v_mfma_f32_32x32x2_f32 at a
    dense peak of 157.3 TFLOP/s (guides/MI355X_MICROARCH.md: = the fp32 vector rate; vector work does not overlap it,
    tools/probes/probe_overlap.hip) -- `bound` = "mfma", achieved = algorithmic flops of the contraction (2 x H*W query rows x D x S per
    layer) / the same duration, and the HBM view stays beside it as `hbm_view` (with the PMC traffic)."""
    if rf is None or contraction != "fmaf":
        return rf
    us = rf["avg_launch_us"]
    tf = flop_per_launch / (us * 1e-6) / 1e12
    return {"kernel": rf["kernel"], "bound": "mfma", "achieved": round(tf, 2), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / FP32_MATRIX_PEAK_TFLOPS, 4), "traffic": rf.get("traffic"), "traffic_static": rf.get("traffic_static"),
            "traffic_source": rf.get("traffic_source"), "algorithmic_flops_per_launch": flop_per_launch, "avg_launch_us": us,
            "entries_per_launch": rf.get("entries_per_launch"),
            "note": "fp32-fma-chain contract: the contraction runs on v_mfma_f32_32x32x2_f32 (bit-exact fmaf chain, 1/16 of the fp16 matrix rate) and "
                    "shares the SIMD's FP32 lanes with the softmax's vector work: per 32k layer 14-16 us of matrix issue + ~10 us of vector issue "
                    "against 8.4 us of K streaming at 8 TB/s",
            "hbm_view": {k_: rf[k_] for k_ in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch") if k_ in rf}}


DRIVER_EXTRA_KEYS_CAP = 20          # the driver's record keeps the contract keys + the NAMES of at most this many others (BENCH_r05.json)


def nest_extras(out):
    """Keeps the line's top-level key count inside what the driver's record lists (`ttft_hotpath_ms` fell off the end
    of its 20 extra key names; `step_ms_by_contract` would have been next): the other SCHEDULES of the same step go under `schedules`,
    the other launches' rooflines under `roofline_other_launches`, the compaction note into `compact`.  What BASELINE.json names
    (`ttft_ms`, `kv_compact_GBps` / `_frac`) and what the verdicts read (`step_ms_by_contract`, `kernels`, `placement_*`) stay on top."""
    sched = {k_: out.pop(k_) for k_ in ("layer_by_layer", "hold_2", "deferred_all_layers", "kv_order_index", "published_recipe") if k_ in out}
    if sched:
        out["schedules"] = sched
    other = {nk: out.pop(k_) for k_, nk in (("roofline_one_layer_launch", "one_layer"), ("roofline_pair_launch", "pair"),
                                              ("fp32_pipe_view", "matrix_pipe_view")) if k_ in out}
    if other:
        out["roofline_other_launches"] = other
    if "kv_compact_note" in out and isinstance(out.get("compact"), dict):
        out["compact"]["note"] = out.pop("kv_compact_note")
    return out


def contract_line(world, steps, warmup, ms_per_step, contraction, ranks_seen, backend, violations, defer, defer_hold):
    """The keys the bench contract names (and the few this path adds), the same for every N -- tests/test_bench_contract.py holds an
    N = 1 and an N = 8 line to it.  `value` = prompt tokens of ALL ranks / the slowest rank's time for `steps` steps."""
    value = world * CFG["S"] / (ms_per_step * 1e-3)
    out = {"metric": "prefill_hotpath_tokens_per_s", "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": steps,
           "warmup": warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f16", "data": "synthetic",
           "config": {"workload": "FastKV hot path (score+select+compact, 32 layers + TSP gather) of one Llama-3-8B prefill, "
                                  "32k context, TSP layer 15, budget 2048, window 8, kernel 7, maxpool",
                      "prompt_tokens_per_rank": CFG["S"], "parallelism": f"dp{world} (independent prompts)" if world > 1 else "single",
                      "contraction": contraction + (" (default: the fp32 fma chain, bit for bit the reference's matmul)" if contraction == "fmaf"
                                                    else " (opt-in: the gfx950 fp16 matrix instruction)")},
           "ttft_hotpath_ms": round(ms_per_step, 4),
           # the arithmetic contract of the contraction (utils.py:94) both sides run: "fmaf" (default since round 6) = the fp32 fma chain,
           # which IS the reference's fp16 matmul bit for bit; "mfma16" (FASTKV_CONTRACTION=mfma16, the opt-in fast mode) = the gfx950 fp16
           # matrix instruction on the fp16 operands, restated bit for bit by the oracle
           "contraction": contraction, "ranks_seen": ranks_seen, "backend": backend,
           # the fused launches' own check of where their workgroups ran (include/fastkv_hip.h: fastkv_placement_violations): 0 = every
           # pair of workgroups that shared a compute unit in a REGULAR launch of the warm-up and timed steps belonged to one head, as
           # those launches assume.  `placement_check` says which launches count at all (a 0 from launches that do not arm
           # the check would be vacuous): the rolling launches -- two of the step's three scoring launches -- never do
           "placement_violations": violations,
           "placement_check": {"regular_launches": "armed (fma chain: the placement policy acts on the count, fail safe by default; mfma16: counted only)",
                               "rolling_launches": "not armed: entries share compute units out of step by design (soaks: profiles/r06_soak_*.log)"}}
    if os.environ.get("FASTKV_FUSED") == "0":
        out["no_wait_kernels"] = "FASTKV_FUSED=0: staged scoring + wait-free selection (what ranks that share one GPU must run)"
    out["config"]["schedule"] = ("deferred, as baselines/fastkv/_wiring.py runs it by default: the 16 layers behind the TSP layer in ONE launch "
                                 f"sequence after the last layer, the layers in front of it in groups of {defer_hold} (FASTKV_DEFER_HOLD: a layer waits "
                                 "for its peers, q / k / v held meanwhile: 400 MiB per waiting layer); the library scores a group of three or more "
                                 "32k layers with ONE rolling launch (the entries follow each other over the chip two at a time and out of step; "
                                 "csrc/fused.hip launch_score_fused) and selects / copies it with one launch each; same rows, same order") if defer else "layer by layer (FASTKV_DEFER=0)"
    return out


def main():
    t_proc0 = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="N>1: skip the sequence-sharded / tensor-parallel legs")
    ap.add_argument("--leg", choices=LEGS, default=None, help="internal: run ONE multi-GPU leg in this (child) process")
    ap.add_argument("--cpu-baseline-child", action="store_true", help="internal: the CPU leg in its own process (cpu_baseline_child_main)")
    ap.add_argument("--no-extras", action="store_true", help="skip the instrumented replay / roofline-shape / CPU legs")
    ap.add_argument("--no-ttft", action="store_true", help="skip the whole-model TTFT leg (random-init Llama-3-8B, fastkv vs fullkv)")
    ap.add_argument("--rehearse", action="store_true", help="launcher rehearsal (tests): the ranks rendezvous, exchange one all-reduce and "
                    "rank 0 prints a line with value null -- no GPU work, nothing measured")
    a = ap.parse_args()
    if a.cpu_baseline_child:
        return cpu_baseline_child_main()

    one_rank_rccl = a.gpus == 1 and os.environ.get("BENCH_BACKEND") == "nccl"     # RCCL with the one rank a one-GPU box allows
    if (a.gpus > 1 or one_rank_rccl) and "RANK" not in os.environ and not a.leg:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it has not touched the GPU and never will)
        return self_launch(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if a.rehearse:
        return rehearse(a, rank, world)
    # The CPU leg first (N = 1 only), in its own process, while this one has neither touched the GPU nor started a thread pool: the host
    # cores are as idle as they get (the same code measured 84-816 ms per step from box to box when it ran last,
    # inside the bench process; what the child records about its host -- quota, load, candidates -- is in the line)
    cpu = None
    if world == 1 and rank == 0 and not a.leg and not a.no_extras and not a.no_cpu_baseline and torch.cuda.device_count() > 0:
        cpu = cpu_baseline()
    assert torch.cuda.is_available(), "bench.py needs the MI355X (there is no CPU fallback for the product path)"
    if world > max(1, torch.cuda.device_count()):
        # more ranks than GPUs (a rehearsal of the N-rank run on a one-GPU box, BENCH_BACKEND=gloo): the ranks' launches share compute
        # units, which the kernels with in-launch waits do not support (include/fastkv_hip.h "Residency") -- every rank runs the
        # no-wait kernels, as the header prescribes for processes that share a GPU
        os.environ.setdefault("FASTKV_FUSED", "0")
    if a.leg:
        return leg_child_main(a)
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or one_rank_rccl:
        import torch.distributed as dist
        backend = os.environ.get("BENCH_BACKEND", "nccl")       # "gloo": several ranks on one GPU (test boxes with a single device)
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)
            except (TypeError, RuntimeError, ValueError):                # a build without eager initialisation: lazy communicators
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group("nccl")
        else:
            dist.init_process_group(backend)

    from fastkv_amd._lib import load
    lib = load()
    work = HotPathPrefill(dev, seed=1000 + rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        work.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        work.step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    from fastkv_amd._lib import raise_if_aborted
    raise_if_aborted("bench")                                    # (behind the synchronisation: an abandoned launch is an error of the run)

    out = contract_line(world, a.steps, a.warmup, ms_per_step,
                        default_contraction(),
                        dist.get_world_size() if dist is not None else 1, dist.get_backend() if dist is not None else "none",
                        int(lib.fastkv_placement_violations(0)), work.defer, work.defer_hold)

    if world > 1:
        return finish_multi_rank(a, out, work, rank, world, dev, dist, t_proc0)

    if not a.no_extras:
        # instrumented replay of the same steps: per-kernel HIP-event durations on the launch stream
        profile_read(lib)
        lib.fastkv_profile_enable(1)
        for _ in range(a.steps):
            work.step()
        torch.cuda.synchronize()
        lib.fastkv_profile_enable(0)
        prof = profile_read(lib)
        kern = {n: {"launches_per_step": c / a.steps, "avg_us": round(ms / c * 1e3, 2), "us_per_step": round(ms / a.steps * 1e3, 1)}
                for n, (c, ms) in prof.items() if c}
        out["kernels"] = kern
        # dominant kernel: the scoring launch at S=32768 (the 16 post-TSP launches stream only 4 MiB each)
        S, Hkv, D, H, W = CFG["S"], CFG["Hkv"], CFG["D"], CFG["H"], CFG["window"]
        if rank == 0:
            lib.fastkv_profile_enable(1)
            from fastkv_amd import ops
            for i in range(8):
                ops.scores(*work.layers_in[i][:2], W, CFG["kernel"], CFG["pooling"], want_tsp=False)
            torch.cuda.synchronize()
            lib.fastkv_profile_enable(0)
            p2 = profile_read(lib)
            kname = "score_fused" if p2.get("score_fused", (0, 0))[0] else "score_logits"
            c, ms = p2[kname]
            us = ms / c * 1e3
            alg = Hkv * S * D * 2 + H * W * D * 2                # K once + the window queries
            flops = 2.0 * H * W * D * S                         # fp32 fma chain: H*W query rows x S keys x D
            traffic, tsrc = None, None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                traffic, tsrc = tj.get(kname + "_hbm_bytes_per_launch"), tj.get("source")
            # Headline = the HBM view SURVEY.md 8(d) names: algorithmic bytes of the dominant kernel / its average launch
            # duration (HIP events on the launch stream, measured in THIS run) against the 8 TB/s peak.
            out["roofline"] = {"kernel": kname + " (S=32768 launches)", "bound": "hbm",
                               "achieved": round(alg / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": traffic,
                               "traffic_static": True, "traffic_source": tsrc or "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / "
                               "WRITE_SIZE passes, corrected per guide; NOT measured in this run)",
                               "algorithmic_bytes_per_launch": alg, "avg_launch_us": round(us, 2),
                               "note": ("score_fused = logits (" + ("v_mfma_f32_32x32x16_f16 on the fp16 operands" if out["contraction"] == "mfma16" else "fp32 MFMA") +
                                        ") + softmax (2 in-kernel reductions over the head's workgroups) + window-row "
                                        "sum + pooling + head sum in one launch; logits stay in registers, row sums in LDS") if kname == "score_fused"
                                       else "matrix-pipe contraction; logits written as fp16"}
            # The matrix-pipe view of the same launch.  Contract "fmaf" (default): the contraction is the fp32 fma chain in ascending
            # head-dim order on v_mfma_f32_32x32x2_f32 (32 flop per K byte): the FP32 lanes saturate long before HBM does, and vector
            # work does not overlap the fp32 MFMAs of a SIMD (tools/probes/probe_overlap.hip: the two add up) -- this view becomes the
            # line's `roofline` below.  Contract "mfma16": the same flops on the fp16 matrix instruction are ~2 % of its peak.
            pipe_peak = FP32_MATRIX_PEAK_TFLOPS if out["contraction"] == "fmaf" else 2500.0    # dense fp16 MFMA peak (MI355X_MICROARCH.md)
            out["fp32_pipe_view"] = {"kernel": kname, "achieved_TFLOPs": round(flops / (us * 1e-6) / 1e12, 2),
                                     "peak_TFLOPs": pipe_peak, "frac": round(flops / (us * 1e-6) / 1e12 / pipe_peak, 4),
                                     "flop_per_launch": flops,
                                     "pipe": "fp32 matrix instruction" if out["contraction"] == "fmaf" else "fp16 matrix instruction (the contraction is ~2 % of its peak: the launch is bound by HBM and the vector work behind the contraction)"}
            if work.defer and kname == "score_fused":
                # In the default (deferred) schedule the launch that dominates the step scores TWO 32k layers (score_fused_kernel<128,4,2,1>
                # in the rocprofv3 trace): the headline prices that launch; the one-layer launch above stays beside it.
                try:
                    single = dict(out["roofline"])
                    pair = [work.layers_in[i] for i in (0, 1)]
                    qs2, ks2, vs2 = ([t[j] for t in pair] for j in range(3))
                    for _ in range(2):
                        ops.update_kv_entries(qs2, ks2, vs2, W, CFG["kernel"], CFG["pooling"], CFG["budget"], 0, "score")
                    torch.cuda.synchronize()
                    profile_read(lib)
                    lib.fastkv_profile_enable(1)
                    for i in range(8):
                        pair = [work.layers_in[(2 * i) % 14], work.layers_in[(2 * i + 1) % 14]]
                        qs2, ks2, vs2 = ([t[j] for t in pair] for j in range(3))
                        ops.update_kv_entries(qs2, ks2, vs2, W, CFG["kernel"], CFG["pooling"], CFG["budget"], 0, "score")
                    torch.cuda.synchronize()
                    lib.fastkv_profile_enable(0)
                    c2, ms2 = profile_read(lib)["score_fused"]
                    us2 = ms2 / c2 * 1e3
                    out["roofline"].update({"kernel": "score_fused (two S=32768 layers per launch: the dominant launch of the deferred schedule)",
                                            "achieved": round(2 * alg / (us2 * 1e-6) / 1e9, 1),
                                            "frac": round(2 * alg / (us2 * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                                            "traffic": tj.get("score_fused_pair_hbm_bytes_per_launch") if os.path.exists(tpath) else None,
                                            "traffic_note": "null unless profiles/traffic.json holds a pair-launch row: the PMC passes run the default "
                                                            "schedule (bench.py --no-extras), which scores groups with the rolling launch and has no pair launch",
                                            "algorithmic_bytes_per_launch": 2 * alg,
                                            "avg_launch_us": round(us2, 2)})
                    out["roofline_one_layer_launch"] = single
                    out["fp32_pipe_view"].update({"achieved_TFLOPs": round(2 * flops / (us2 * 1e-6) / 1e12, 2),
                                                  "frac": round(2 * flops / (us2 * 1e-6) / 1e12 / pipe_peak, 4),
                                                  "flop_per_launch": 2 * flops})
                    # Since round 4 a group of three or more 32k layers is scored by ONE rolling launch (csrc/fused.hip launch_score_fused:
                    # all entries in one grid, two on the chip at a time and out of step): that launch -- `defer_hold` layers, the
                    # grid 65536 x defer_hold of score_fused_kernel<128,4,2,1> in the rocprofv3 trace -- is what dominates the default
                    # step, and what the headline prices; the pair launch (the schedule of groups of two) stays beside it.
                    n_grp = int(work.defer_hold)
                    if n_grp >= 3:
                        nl = min(len(work.layers_in), 16)
                        grp = lambda i: [work.layers_in[(i * n_grp + j) % nl] for j in range(n_grp)]   # noqa: E731
                        for i in range(2):
                            qs3, ks3, vs3 = ([t[j] for t in grp(i)] for j in range(3))
                            ops.update_kv_entries(qs3, ks3, vs3, W, CFG["kernel"], CFG["pooling"], CFG["budget"], 0, "score")
                        torch.cuda.synchronize()
                        profile_read(lib)
                        lib.fastkv_profile_enable(1)
                        ncall = 6
                        for i in range(ncall):
                            qs3, ks3, vs3 = ([t[j] for t in grp(i)] for j in range(3))
                            ops.update_kv_entries(qs3, ks3, vs3, W, CFG["kernel"], CFG["pooling"], CFG["budget"], 0, "score")
                        torch.cuda.synchronize()
                        lib.fastkv_profile_enable(0)
                        c3, ms3 = profile_read(lib)["score_fused"]
                        if c3 == ncall:                          # one launch per group: the rolling launch is on
                            us3 = ms3 / c3 * 1e3
                            out["roofline_pair_launch"] = dict(out["roofline"])
                            out["roofline"].update({"kernel": f"score_fused ({n_grp} S=32768 layers in ONE rolling launch, two on the chip at a time and out of step: "
                                                              "the dominant launch of the default schedule)",
                                                    "achieved": round(n_grp * alg / (us3 * 1e-6) / 1e9, 1),
                                                    "frac": round(n_grp * alg / (us3 * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                                                    "traffic": tj.get("score_fused_group_hbm_bytes_per_launch") if os.path.exists(tpath) and
                                                    tj.get("score_fused_group_entries") == n_grp else None,
                                                    "algorithmic_bytes_per_launch": n_grp * alg, "avg_launch_us": round(us3, 2),
                                                    "entries_per_launch": n_grp})
                            out["fp32_pipe_view"].update({"achieved_TFLOPs": round(n_grp * flops / (us3 * 1e-6) / 1e12, 2),
                                                          "frac": round(n_grp * flops / (us3 * 1e-6) / 1e12 / pipe_peak, 4),
                                                          "flop_per_launch": n_grp * flops})
                except Exception as e:   # noqa: BLE001 -- the one-layer figures stay in place
                    out["roofline"]["pair_launch_error"] = repr(e)[:160]
            out["roofline"] = roofline_of_the_contract(out["roofline"], out["contraction"], out["fp32_pipe_view"]["flop_per_launch"])
            # the same step with every layer compressed inside its own attention forward, as the reference does it
            if work.defer:
                work.defer = False
                for _ in range(2):
                    work.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    work.step()
                torch.cuda.synchronize()
                ms_seq = (time.perf_counter() - t0) / a.steps * 1e3
                work.defer = True
                out["layer_by_layer"] = {"ms_per_step": round(ms_seq, 4), "tokens_per_s": round(CFG["S"] / (ms_seq * 1e-3), 1),
                                         "note": "32 sequential update_kv calls (FASTKV_DEFER=0): the call pattern of the reference"}
                # the round-2 default: long layers in pairs (one more layer's q / k / v held instead of seven)
                hold0, work.defer_hold = work.defer_hold, 2
                for _ in range(2):
                    work.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    work.step()
                torch.cuda.synchronize()
                ms_h2 = (time.perf_counter() - t0) / a.steps * 1e3
                work.defer_hold = hold0
                out["hold_2"] = {"ms_per_step": round(ms_h2, 4), "tokens_per_s": round(CFG["S"] / (ms_h2 * 1e-3), 1),
                                 "note": "FASTKV_DEFER_HOLD=2: the layers in front of the TSP layer in pairs (round 2's schedule)"}
                # ... and with the 15 layers in front of the TSP layer deferred as well (FASTKV_DEFER_MAX_LEN = prompt length: two
                # 32k layers per launch sequence; their full K/V stay alive until the end of the forward pass)
                work.defer_max_len = CFG["S"]
                for _ in range(2):
                    work.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    work.step()
                torch.cuda.synchronize()
                ms_all = (time.perf_counter() - t0) / a.steps * 1e3
                work.defer_max_len = int(os.environ.get("FASTKV_DEFER_MAX_LEN", "8192"))
                out["deferred_all_layers"] = {"ms_per_step": round(ms_all, 4), "tokens_per_s": round(CFG["S"] / (ms_all * 1e-3), 1),
                                              "note": "FASTKV_DEFER_MAX_LEN=32768: every layer but the TSP layer waits for the end of the "
                                                      "forward pass (+2 GB of K/V held); not the default"}
            # the same step with the K/V rows in ascending position (FASTKV_KV_ORDER=index; attention does not depend on the row
            # order): the 16 post-TSP layers keep every candidate and become single copy launches
            for c in work.clusters:
                c.kv_order = "index"
            for _ in range(2):
                work.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                work.step()
            torch.cuda.synchronize()
            ms_idx = (time.perf_counter() - t0) / a.steps * 1e3
            for c in work.clusters:
                c.kv_order = "score"
            out["kv_order_index"] = {"ms_per_step": round(ms_idx, 4), "tokens_per_s": round(CFG["S"] / (ms_idx * 1e-3), 1),
                                     "note": "rows in ascending position instead of the reference's score order"}
            cc, cms = prof["compact_kv"]
            out["compact"] = {"per_launch_avg_us": round(cms / cc * 1e3, 2), "per_layer_avg_us": round(cms / a.steps / CFG["layers"] * 1e3, 2),
                              "per_layer_algorithmic_bytes": 2 * (2 * Hkv * CFG["budget"] * D * 2) + Hkv * (CFG["budget"] - W) * 8,
                              "per_layer_order": "score (the reference's row order: the product default); average over the step's "
                                                 "compaction launches (deferred schedule: 8 two-layer launches at 32k + one for the 16 post-TSP layers)",
                              "roofline_shape": compact_roofline_shape(lib, dev, 10)}
            # the same (default, deferred) step under the OTHER contraction contract: the fp32 fma chain on v_mfma_f32_32x32x2_f32
            # (fmaf <-> mfma16: whichever the line's headline is not)
            other = "mfma" if out["contraction"] == "mfma16" else "mfma16"
            ops.set_score_engine(other)
            try:
                for _ in range(2):
                    work.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    work.step()
                torch.cuda.synchronize()
                ms_o = (time.perf_counter() - t0) / a.steps * 1e3
            finally:
                ops.set_score_engine("auto")
            out["other_contract"] = {"engine": other, "contraction": "fmaf" if other == "mfma" else "mfma16", "ms_per_step": round(ms_o, 4),
                                     "tokens_per_s": round(CFG["S"] / (ms_o * 1e-3), 1),
                                     "note": "same schedule, the other arithmetic contract of the contraction (both bit-exact against the "
                                             "oracle under the same contract)"}
            # the quantities BASELINE.json names, at the top level
            rs = out["compact"]["roofline_shape"]["score"]
            out["kv_compact_GBps"] = rs["achieved_GBps"]
            out["kv_compact_frac"] = rs["frac_of_8TBps"]
            out["kv_compact_note"] = ("KV gather/compact kernel at the 541 MB roofline shape, rows in the reference's score order (the product "
                                      "default; median over calls, HIP events); index order: compact.roofline_shape.index")
            # both arithmetic contracts of the step side by side at the top level (VERDICT r04 next #2c): the headline's and the other
            # one, GPU and CPU (the CPU leg's `value` is the fma chain -- the headline's contract since round 6 --, its `mfma16_contract` the restated matrix instruction)
            out["step_ms_by_contract"] = {out["contraction"]: out["ms_per_step"], out["other_contract"]["contraction"]: out["other_contract"]["ms_per_step"]}
            if cpu is not None:
                out["cpu_baseline"] = cpu
                if "mfma16_contract" in cpu:
                    out["cpu_ms_by_contract"] = {"fmaf": cpu["ms_per_step"], "mfma16": cpu["mfma16_contract"]["ms_per_step"]}
            try:
                out["published_recipe"] = published_recipe_leg(lib, dev, a.steps)
            except Exception as e:   # noqa: BLE001 -- an extra; the contract line does not depend on it
                out["published_recipe"] = {"error": repr(e)[:300]}
            if world == 1 and not a.no_ttft:
                out["ttft"] = whole_model_ttft(work)
                out["ttft_ms"] = out["ttft"].get("fastkv", {}).get("ttft_ms")
            if world == 1 and dist is None and not a.no_legs:
                # RCCL with the ONE rank this box allows: the sequence-sharded and the head-sharded operator in
                # child processes that initialise torch.distributed with backend "nccl", world size 1 -- every collective of
                # fastkv_amd/dist.py executes in RCCL on device tensors.  No scaling is claimed: one rank moves nothing over xGMI.
                del_env = {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                           "BENCH_BACKEND": "nccl"}
                saved = {k_: os.environ.get(k_) for k_ in del_env}
                os.environ.update(del_env)
                try:
                    out["rccl_one_rank"] = {name: spawn_leg(name, i, 3, 0, timeout_s=120) for i, name in enumerate(("seq_sharded_weak", "tp"))}
                finally:
                    for k_, v_ in saved.items():
                        if v_ is None:
                            os.environ.pop(k_, None)
                        else:
                            os.environ[k_] = v_
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if one_rank_rccl and not a.no_legs:
        # `BENCH_BACKEND=nccl python bench.py --gpus 1`: every leg with the ONE RCCL rank this box allows, in the line
        del work
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        budget = float(os.environ.get("FASTKV_BENCH_LEG_BUDGET_S", "240"))
        t_legs = time.perf_counter()
        for i, name in enumerate(LEGS):
            left = budget - (time.perf_counter() - t_legs)
            out[name] = {"skipped": "leg budget spent"} if left < 30 else spawn_leg(name, i, max(3, a.steps // 4), rank, timeout_s=min(150, left))
    if rank == 0:
        print(json.dumps(nest_extras(out)), flush=True)


def quick_group_roofline(lib, work):
    """The N > 1 line's `roofline` object (the N = 1 line measures it among its extras): rank 0 scores a group of `defer_hold` 32k layers six
    times with the library's per-kernel HIP events on -- ~3 ms of GPU time while the other ranks wait at the next barrier."""
    from fastkv_amd import ops
    S, Hkv, D, H, W = CFG["S"], CFG["Hkv"], CFG["D"], CFG["H"], CFG["window"]
    n_grp = int(work.defer_hold)
    nl = min(len(work.layers_in), 16)
    if n_grp < 3 or nl < n_grp:
        return None
    grp = lambda i: [work.layers_in[(i * n_grp + j) % nl] for j in range(n_grp)]   # noqa: E731
    def call(i):
        qs3, ks3, vs3 = ([t[j] for t in grp(i)] for j in range(3))
        ops.update_kv_entries(qs3, ks3, vs3, W, CFG["kernel"], CFG["pooling"], CFG["budget"], 0, "score")
    call(0)
    torch.cuda.synchronize()
    profile_read(lib)
    lib.fastkv_profile_enable(1)
    for i in range(6):
        call(i)
    torch.cuda.synchronize()
    lib.fastkv_profile_enable(0)
    c3, ms3 = profile_read(lib).get("score_fused", (0, 0.0))
    if not c3:
        return None
    us = ms3 / c3 * 1e3
    alg = (Hkv * S * D * 2 + H * W * D * 2) * n_grp * 6 // c3           # algorithmic bytes per scoring launch (6 groups in c3 launches)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and c3 == 6:
        tj = json.load(open(tpath))
        traffic = tj.get("score_fused_group_hbm_bytes_per_launch") if tj.get("score_fused_group_entries") == n_grp else None
    rf = {"kernel": f"score_fused ({n_grp} S=32768 layers per call, {c3 // 6} scoring launch(es) per call; rank 0)", "bound": "hbm",
          "achieved": round(alg / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
          "traffic": traffic, "traffic_static": True, "algorithmic_bytes_per_launch": alg, "avg_launch_us": round(us, 2)}
    return roofline_of_the_contract(rf, default_contraction(), 2.0 * H * W * D * S * n_grp * 6 // c3)


def finish_multi_rank(a, out, work, rank, world, dev, dist, t_proc0) -> int:
    """N > 1 ranks behind their timed run.  The contract line leaves EARLY: within FASTKV_BENCH_LINE_DEADLINE_S (default 75 s) of this
    rank's start, whatever the legs do (it used to wait for up to 420 s of legs; a driver whose limit is shorter
    would have got no line at all from the first multi-GPU run this code ever sees).  Order:
      1. rank 0 looks at its clock and decides how many of the two short legs (`seq_sharded_weak`, `tp`: the sequence-sharded and the
         head-sharded operator, 30 s each at most) still fit in front of the deadline; the decision is BROADCAST, so every rank spawns
         the same children;
      2. the process group is released, the short legs run in child processes (own rendezvous ports: a rank that dies in a collective
         costs that leg only) and land in the line;
      3. rank 0 prints the ONE line, flushed;
      4. the long legs (`seq_sharded_128k`, `sp_ttft_128k`) run afterwards within FASTKV_BENCH_LEG_BUDGET_S (default 120 s) and are
         reported on STDERR (`LEGS_JSON {...}`) and, where the directory exists, in gpurun_out/bench_legs_N<world>.json -- stdout
         carries the one line and nothing else."""
    deadline = float(os.environ.get("FASTKV_BENCH_LINE_DEADLINE_S", "75"))
    if rank == 0 and not a.no_extras:
        # the N > 1 line ALWAYS carries a `roofline` object (the driver keeps it among the parsed keys; a line
        # without one would lose the kernel's figure to the extra-key truncation).  Ranks that share one GPU (the rehearsal:
        # FASTKV_FUSED=0, no grouped scoring launch to price) say so in it instead of leaving it out.
        out["roofline"] = {"kernel": "score_fused", "bound": "mfma" if out["contraction"] == "fmaf" else "hbm", "achieved": None,
                           "peak": FP32_MATRIX_PEAK_TFLOPS if out["contraction"] == "fmaf" else HBM_PEAK_GBPS,
                           "unit": "TFLOP/s" if out["contraction"] == "fmaf" else "GB/s", "frac": None, "traffic": None,
                           "note": "not measured: the no-wait kernels (FASTKV_FUSED=0, ranks sharing a GPU) have no grouped scoring launch"}
        if os.environ.get("FASTKV_FUSED") != "0":
            try:
                from fastkv_amd._lib import load
                rf = quick_group_roofline(load(), work)
                if rf is not None:
                    out["roofline"] = rf
                else:
                    out["roofline"]["note"] = "not measured: the grouped scoring launch did not run on rank 0"
            except Exception as e:   # noqa: BLE001 -- an extra of the line
                out["roofline"]["note"] = "not measured: " + repr(e)[:200]
    inline = [("seq_sharded_weak", 30.0), ("tp", 30.0)]
    n_inline = 0
    if not a.no_legs:
        left = deadline - (time.perf_counter() - t_proc0)
        for _, need in inline:
            if left >= need + 5.0:
                n_inline += 1
                left -= need
    t = torch.tensor([n_inline], device=dev, dtype=torch.int64)
    dist.broadcast(t, src=0)                                     # rank 0's clock decides for everybody
    n_inline = int(t.item())
    dist.barrier()
    dist.destroy_process_group()
    del work.layers_in[:]                                        # (the caller still holds the object: free what it owns)
    work.hidden = None
    del work
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    steps = max(3, a.steps // 4)
    for i, (name, need) in enumerate(inline[:n_inline]):
        res = spawn_leg(name, i, steps, rank, timeout_s=need)
        if rank == 0:
            out[name] = res
    if rank == 0:
        out["legs_after_the_line"] = [] if a.no_legs else [n for n in LEGS if n not in [x[0] for x in inline[:n_inline]]]
        out["line_after_s"] = round(time.perf_counter() - t_proc0, 1)
        print(json.dumps(out), flush=True)
    if a.no_legs:
        return 0
    budget = float(os.environ.get("FASTKV_BENCH_LEG_BUDGET_S", "120"))
    t_legs = time.perf_counter()
    late = {}
    for i, name in enumerate(LEGS):
        if name in [x[0] for x in inline[:n_inline]]:
            continue
        left = budget - (time.perf_counter() - t_legs)
        # (every rank applies the same rule to its own clock, started at the same barrier give or take: a leg some ranks skip simply
        # times out on the others -- after the line, that costs nothing but the leg)
        res = {"skipped": "leg budget spent"} if left < 30 else spawn_leg(name, 10 + i, steps, rank, timeout_s=min(90.0, left))
        if rank == 0:
            late[name] = res
    if rank == 0:
        print("LEGS_JSON " + json.dumps(late), file=sys.stderr, flush=True)
        try:
            d = os.path.join(ROOT, "gpurun_out")
            if os.path.isdir(d):
                with open(os.path.join(d, f"bench_legs_N{world}.json"), "w") as f:
                    json.dump(late, f)
        except OSError:
            pass
    return 0


def published_recipe_leg(lib, dev, steps):
    """The hot path of one Llama-3-8B 32k prefill under the reference's PUBLISHED recipe (RECIPE above; the only setting its scripts ship)
    instead of the constant budget of BASELINE.json configs[1]: 16 layers of 32,768 tokens compressed to 3276 rows, the TSP gather of
    6553 hidden rows, 16 layers of 6553 tokens compressed to 3276 -- those are LONG layers for the default schedule (groups of eight,
    like the ones in front of the TSP layer), which the constant budget never produces."""
    import gc
    work = HotPathPrefill(dev, seed=7000, recipe=True)
    res = {"recipe": "--tsp_idx 15 --tsp_rate 0.2 --retain_rate 0.1 --eviction_mode proportional (scripts/eval_prefill.sh:4-12); maxpool, window 8, kernel 7",
           "shapes": f"layers 0-15: S = {CFG['S']} -> {int(CFG['S'] * RECIPE['retain_rate'])} rows; TSP length {work.tsp_len}; layers 16-31: "
                     f"S = {work.tsp_len} -> {int(work.tsp_len * (RECIPE['retain_rate'] / RECIPE['tsp_rate']))} rows"}

    def timed():
        for _ in range(2):
            work.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            work.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    ms = timed()
    res.update({"ms_per_step": round(ms, 4), "tokens_per_s": round(CFG["S"] / (ms * 1e-3), 1)})
    profile_read(lib)
    lib.fastkv_profile_enable(1)
    for _ in range(steps):
        work.step()
    torch.cuda.synchronize()
    lib.fastkv_profile_enable(0)
    res["kernels"] = {n: {"launches_per_step": c / steps, "avg_us": round(t / c * 1e3, 2), "us_per_step": round(t / steps * 1e3, 1)}
                      for n, (c, t) in profile_read(lib).items() if c}
    work.defer = False
    ms_seq = timed()
    work.defer = True
    res["layer_by_layer"] = {"ms_per_step": round(ms_seq, 4), "tokens_per_s": round(CFG["S"] / (ms_seq * 1e-3), 1)}
    caps = sorted({c.max_capacity_prompt for c in work.clusters})
    res["state_after"] = {"max_capacity_prompt": caps, "tsp_length": work.clusters[CFG["tsp_idx"]].tsp_length}
    from fastkv_amd._lib import raise_if_aborted
    raise_if_aborted("bench.published_recipe")
    del work
    gc.collect()
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    sys.exit(main() or 0)
