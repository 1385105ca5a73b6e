"""The committed bench lines (profiles/*_bench.json: bench.py's ONE JSON line as the GPU box printed it) carry what the
measurement contract asks for -- guards the schema against drift; the numbers themselves are judged from the files."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_lines_follow_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench.json")))
    assert files
    latest = files[-1]
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in d, (f, key)
        assert d["metric"] == "prefill_hotpath_tokens_per_s" and d["unit"] == "tokens/s" and d["higher_is_better"] is True
        assert d["dtype"] == "f16" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
        assert abs(d["value"] - d["n_gpus"] * 32768 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
        r = d["roofline"]
        for key in ("bound", "achieved", "peak", "unit", "frac"):
            assert key in r, (f, key)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        c = d["cpu_baseline"]
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in c, (f, key)
        assert c["kind"] in ("port", "reference")
    d = json.loads(open(latest).read().strip().splitlines()[-1])
    r = d["roofline"]
    if d.get("contraction") == "fmaf" and latest >= os.path.join(ROOT, "profiles", "r06"):
        # the fma-chain contract (default since round 6): the FP32 lanes bound the dominant launch, the HBM view stays beside it
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and "traffic" in r
        assert r["hbm_view"]["bound"] == "hbm" and r["hbm_view"]["unit"] == "GB/s" and 0 < r["hbm_view"]["frac"] < 1
    else:
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and "traffic" in r
    assert "score" in d["compact"]["roofline_shape"] and "index" in d["compact"]["roofline_shape"]


def _run_bench(args, env_extra=None, timeout=300):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_plain_command_with_several_gpus_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver starts the N=1 run): the process
    becomes the launcher, the ranks rendezvous over gloo and ONE line comes back on stdout, rc 0."""
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"], {"BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["steps"] == 3 and d["warmup"] == 1
    assert d["rehearsal"] is True and d["value"] is None                     # a rehearsal line is not a measurement


def test_the_contract_launcher_command_still_works():
    """The command the contract names: python -m torch.distributed.run ... bench.py --gpus N (RANK set: no self-launch)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2


def test_a_failing_rank_fails_the_launcher():
    """Without a GPU the ranks refuse to run (no CPU fallback): the launcher reports that as a non-zero exit code and no line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras"])
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_contract_keys_are_the_same_for_every_n_and_match_the_driver_record():
    """The N = 1 and the N = 8 line are built by ONE function (bench.contract_line): same keys, same config keys, value = all ranks'
    tokens over the slowest rank's time; and the driver's latest record of an N = 1 run (BENCH_rNN.json, when the round has one)
    carries every one of them."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    one = bench.contract_line(1, 20, 3, 0.6, "mfma16", 1, "none", 0, True, 8)
    eight = bench.contract_line(8, 20, 3, 0.7, "mfma16", 8, "nccl", 0, True, 8)
    assert set(one) == set(eight) and set(one["config"]) == set(eight["config"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config"):
        assert key in one, key
    assert one["scaling"] == "weak" and eight["n_gpus"] == 8 and "workload" in one["config"] and "model" not in one["config"]
    assert abs(one["value"] - 32768 / 0.6e-3) < 1 and abs(eight["value"] - 8 * 32768 / 0.7e-3) < 1
    assert "rolling launch" in one["config"]["schedule"]
    recs = sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")))
    if recs:
        _check_driver_record(json.load(open(recs[-1])), one, recs[-1])
    # a synthetic record the way the driver writes them, with 40 non-contract keys on the line: whatever the driver cuts, the check holds
    line = dict(one, roofline={}, cpu_baseline={}, **{"zz_extra_%02d" % i: i for i in range(40)})
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline")
    extra = sorted(k for k in line if k not in contract)
    for cap in (20, 5, len(extra)):
        rec = {"parsed": dict({k: line[k] for k in contract}, extra_keys=extra[:cap])}
        rec["parsed"]["config"] = dict(line["config"], workload=line["config"]["workload"][:120])
        _check_driver_record(rec, one, "synthetic, %d of %d extra key names kept" % (cap, len(extra)))


NEW_KEYS = {5: ("placement_check",)}      # keys of the line that did not exist when the record of round <key> was written
DRIVER_EXTRA_KEYS_CAP = 20        # the driver's record lists at most this many non-contract key NAMES, in sorted order (BENCH_r05.json)


def _check_driver_record(rec, one, what):
    """A driver-written record (BENCH_rNN.json) against the line bench.contract_line builds.  The driver keeps the contract keys in
    `parsed` and only the NAMES of the others, sorted and cut at DRIVER_EXTRA_KEYS_CAP: a full list is a truncated list, and a key that
    sorts behind its last element is not missing, it is out of the record's sight (round 5's red test: 24 extra keys, `ttft_hotpath_ms`
    fell off the end)."""
    parsed = rec.get("parsed") or {}
    if not parsed:
        return
    extra = parsed.get("extra_keys") or []                                    # (the driver lists the non-contract keys by name only)
    extra = sorted(extra) if isinstance(extra, (list, tuple, dict)) else []
    truncated = len(extra) >= DRIVER_EXTRA_KEYS_CAP or (extra and len(extra) < len([k for k in one if k not in parsed]))
    visible = (lambda k: k <= extra[-1]) if (truncated and extra) else (lambda k: True)
    # (a record is of the round that WROTE it: a key this round added to the line -- NEW_KEYS -- cannot be in an older record)
    missing = [k for k in one if k not in parsed and k not in extra and visible(k) and k not in NEW_KEYS.get(rec.get("n"), ())]
    assert not missing, (what, missing)
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config")
    assert all(k in parsed for k in contract), (what, [k for k in contract if k not in parsed])
    assert parsed["metric"] == one["metric"] and parsed["unit"] == one["unit"]
    w = parsed["config"]["workload"]                                          # (the driver keeps the first ~120 characters of a string)
    assert one["config"]["workload"].startswith(w[:100]) and len(w) >= 100


def test_the_eight_rank_rehearsal_line_survives_the_drivers_truncation():
    """the N = 8 line of the round's rehearsal (8 ranks sharing the one GPU of a test box over gloo:
    profiles/r06_bench8_gloo_rehearsal.json, produced by tools/r06/rehearse8.sh) carries `roofline` -- a key the driver keeps among the
    parsed ones -- and few enough other keys that `ranks_seen` (and every other extra key) fits the driver's list of 20 names."""
    path = os.path.join(ROOT, "profiles", "r06_bench8_gloo_rehearsal.json")
    if not os.path.exists(path):
        import pytest
        pytest.skip("no rehearsal profile of this round yet")
    line = json.load(open(path))["line"]
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline")
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["scaling"] == "weak"
    assert isinstance(line.get("roofline"), dict) and {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    extra = sorted(k for k in line if k not in contract)
    assert len(extra) <= DRIVER_EXTRA_KEYS_CAP and "ranks_seen" in extra, extra
    assert abs(line["value"] - 8 * 32768 / (line["ms_per_step"] * 1e-3)) <= 1e-3 * line["value"]


def test_the_full_line_fits_the_drivers_key_list():
    """bench.nest_extras: the N = 1 line with every extra the default run adds keeps its top-level key count inside the contract keys +
    the driver's 20 names, and what BASELINE.json and the verdicts read stays on top."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    line = bench.contract_line(1, 20, 5, 0.77, "fmaf", 1, "none", 0, True, 8)
    for k in ("roofline", "cpu_baseline", "kernels", "fp32_pipe_view", "roofline_pair_launch", "roofline_one_layer_launch", "layer_by_layer", "hold_2",
              "deferred_all_layers", "kv_order_index", "compact", "other_contract", "kv_compact_GBps", "kv_compact_frac", "kv_compact_note",
              "step_ms_by_contract", "cpu_ms_by_contract", "published_recipe", "ttft", "ttft_ms", "rccl_one_rank", "no_wait_kernels"):
        line[k] = {"x": 1}
    line = bench.nest_extras(line)
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline")
    extra = sorted(k for k in line if k not in contract)
    assert len(extra) <= bench.DRIVER_EXTRA_KEYS_CAP, (len(extra), extra)
    for k in ("step_ms_by_contract", "ttft_ms", "kv_compact_GBps", "kv_compact_frac", "kernels", "placement_violations", "placement_check", "contraction"):
        assert k in line, k
    assert set(line["schedules"]) == {"layer_by_layer", "hold_2", "deferred_all_layers", "kv_order_index", "published_recipe"}
    assert set(line["roofline_other_launches"]) == {"one_layer", "pair", "matrix_pipe_view"} and "note" in line["compact"]
    assert "contraction" in line["config"] and line["config"]["contraction"].startswith("fmaf")


def test_roofline_object_follows_the_contract_of_the_line():
    """bench.roofline_of_the_contract: under the fma chain (the default) the dominant launch is priced against the fp32 matrix peak --
    algorithmic flops / the SAME HIP-event duration -- with the HBM view (and the PMC traffic) beside it; under mfma16 the HBM view is
    the line's roofline unchanged.  Pure arithmetic: checked here so that the N > 1 path, which no box has run, cannot get it wrong."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    alg, us, n = 8 * 67174400, 265.0, 8
    hbm = {"kernel": "k", "bound": "hbm", "achieved": round(alg / (us * 1e-6) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
           "frac": round(alg / (us * 1e-6) / 1e9 / 8000.0, 4), "traffic": 573000000, "traffic_static": True, "algorithmic_bytes_per_launch": alg,
           "avg_launch_us": us, "entries_per_launch": n}
    flops = 2.0 * 32 * 8 * 128 * 32768 * n
    assert bench.roofline_of_the_contract(dict(hbm), "mfma16", flops) == hbm
    r = bench.roofline_of_the_contract(dict(hbm), "fmaf", flops)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and r["avg_launch_us"] == us and r["traffic"] == 573000000
    assert abs(r["achieved"] - flops / (us * 1e-6) / 1e12) < 0.01 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["hbm_view"]["bound"] == "hbm" and r["hbm_view"]["frac"] == hbm["frac"] and r["hbm_view"]["algorithmic_bytes_per_launch"] == alg
    assert bench.roofline_of_the_contract(None, "fmaf", flops) is None
    assert bench.default_contraction() in ("fmaf", "mfma16")


def test_cpu_leg_reads_the_cgroup_quota(monkeypatch, tmp_path):
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setenv("FASTKV_BENCH_NCPU", "256")
    n, quota = bench._effective_cpus()
    assert n == 256 and (quota is None or quota > 0)
