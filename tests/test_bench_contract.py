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
