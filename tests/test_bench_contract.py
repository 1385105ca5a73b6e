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
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["unit"] == "GB/s" and "traffic" in d["roofline"]
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
