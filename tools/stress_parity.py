"""One-off stress run (not part of the test suite): N random geometries of the operator on the GPU against the CPU oracle,
bit for bit -- scores, indices, K/V rows, TSP index -- biased towards the fused path's shapes (W = 8, G in {4, 8}), long and
ragged prompts, both row orders, peaked inputs, special values sprinkled in.  usage: python tools/stress_parity.py [N] [seed]
  STRESS_ENTRIES_P=1     the separately-allocated-entries call on every eligible case (default: a quarter of them)
  STRESS_ALL_ENTRIES=1   compare every entry after a mismatch (default: stop at the first)
  STRESS_ONLY=i STRESS_REPEAT=r   replay case i of the sequence r times (the generator is consumed exactly as in a full run)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from stress_cases import run_stress

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
st = run_stress(N, int(sys.argv[2]) if len(sys.argv) > 2 else 12345, entries_p=float(os.environ.get("STRESS_ENTRIES_P", "0.25")),
                all_entries=os.environ.get("STRESS_ALL_ENTRIES", "0") == "1", only=int(os.environ.get("STRESS_ONLY", "-1")),
                repeat=int(os.environ.get("STRESS_REPEAT", "1")), log=lambda m: print(m, flush=True))
print("placement violations counted by the fused launches:", st["violations"])
print(f"{N} cases, {st['mismatches']} mismatches, {st['seconds']:.0f} s (entries calls: {st['entries_runs']} run, {st['entries_refused']} refused; "
      f"special-value cases {st['special']}, engines {sorted(st['engines'])})")
sys.exit(1 if st["mismatches"] else 0)
