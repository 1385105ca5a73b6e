#!/bin/bash
# usage (GPU box): tools/bench_hold.sh 8 16 ... -- the bench step under several FASTKV_DEFER_HOLD settings (layers a group of the deferred schedule waits for)
for h in "$@"; do
  FASTKV_DEFER_HOLD=$h timeout 250 python bench.py --steps 30 --warmup 5 --no-extras 2>/dev/null > /tmp/bh_$h.json
  python - <<PY
import json; d=json.load(open("/tmp/bh_$h.json")); print("FASTKV_DEFER_HOLD=$h: %.4f ms per step, %.1f M tokens/s" % (d["ms_per_step"], d["value"] / 1e6))
PY
done
