"""Summarise one tools/profile_round.sh run (gpurun_out/<tag>_*) into the files kept under profiles/.

    python tools/profile_summarise.py r01c          # reads gpurun_out/, writes profiles/

Outputs: <tag>_bench.json, <tag>_bench_under_rocprofv3.json, <tag>_rocprofv3_kernel_stats.csv (rocprofv3's own --stats
table, all kernels of the run), <tag>_fk_kernels_by_grid.csv (our kernels from the kernel trace, split by grid size so
that the S=32768 and the post-TSP S=2048 launches are not averaged together), <tag>_pmc_fetch_write_summary.json and
traffic.json (per-launch HBM bytes read by bench.py).
"""
import collections, csv, glob, json, os, shutil, statistics as st, sys

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def one(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)      # (a tag that was run again: the newest files)
    return f[-1] if f else None


shutil.copy(os.path.join(G, f"{tag}_bench.json"), os.path.join(P, f"{tag}_bench.json"))
shutil.copy(os.path.join(G, f"{tag}_trace_bench.json"), os.path.join(P, f"{tag}_bench_under_rocprofv3.json"))
shutil.copy(one(f"{tag}_trace/*/*_kernel_stats.csv"), os.path.join(P, f"{tag}_rocprofv3_kernel_stats.csv"))
shutil.copy(os.path.join(G, f"{tag}_pmc_summary.json"), os.path.join(P, f"{tag}_pmc_fetch_write_summary.json"))
if os.path.exists(os.path.join(G, f"{tag}_pmc_mfma_summary.json")):
    shutil.copy(os.path.join(G, f"{tag}_pmc_mfma_summary.json"), os.path.join(P, f"{tag}_pmc_mfma_summary.json"))

agg = collections.defaultdict(list)
for r in csv.DictReader(open(one(f"{tag}_trace/*/*_kernel_trace.csv"))):
    if "fk::" in r["Kernel_Name"]:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        # keyed on the WHOLE grid (x, y, z in work-items, as rocprofv3 reports them): the roofline-shape launch of compact_kv
        # (grid 128 x 256 workgroups) must not be averaged with the 8-head launches that share its grid.x
        agg[(name, int(r["Grid_Size_X"]), int(r.get("Grid_Size_Y", 1)), int(r.get("Grid_Size_Z", 1)), int(r["Workgroup_Size_X"]))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(P, f"{tag}_fk_kernels_by_grid.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "GridSizeX", "GridSizeY", "GridSizeZ", "WorkgroupSizeX", "Calls", "AverageNs", "MedianNs", "MinNs", "MaxNs"])
    for (name, gx, gy, gz, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, gx, gy, gz, wg, len(v), round(st.mean(v), 1), st.median(v), min(v), max(v)])

pmc = json.load(open(os.path.join(G, f"{tag}_pmc_summary.json")))


def kib(counter, prefix):
    hits = {k: v for k, v in pmc.get(counter, {}).items() if k.startswith(prefix)}
    k = max(hits, key=lambda k: hits[k]["median"])          # the S=32768 launches are the largest of a kernel
    return hits[k]["median"]


fused = any(k.startswith("fk::score_fused") for k in pmc.get("FETCH_SIZE", {}))
dom = "fk::score_fused" if fused else "fk::score_logits"
# one 32k layer per launch: score_fused_kernel<128, 2, 2, 1>; two (the deferred schedule's pairs): <128, 4, 2, 1>
def head():
    import subprocess
    try:
        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True, timeout=10).stdout.strip() or "unknown"
    except Exception:   # noqa: BLE001
        return "unknown"


def first_key(prefix, contains=""):
    ks = [k for k in pmc.get("FETCH_SIZE", {}) if k.startswith(prefix) and contains in k]
    return max(ks, key=lambda k: pmc["FETCH_SIZE"][k]["median"]) if ks else prefix


one = first_key("fk::score_fused_kernel<128, 2, 2, 1", "grid=131072") if fused else dom
lf, lw = kib("FETCH_SIZE", one), kib("WRITE_SIZE", one)
pair = {}
# the rolling launch of a group (round 4): the same kernel with grid 65536 x entries (Grid_Size = 65536 * entries threads)
grp_keys = [k for k in pmc.get("FETCH_SIZE", {}) if k.startswith("fk::score_fused_kernel<128, 4, 2, 1") and "grid=" in k and int(k.split("grid=")[1]) > 131072]
group = {}
if grp_keys:
    gk = max(grp_keys, key=lambda k: pmc["FETCH_SIZE"][k]["n"])
    gf = pmc["FETCH_SIZE"][gk]["median"]
    gw = pmc.get("WRITE_SIZE", {}).get(gk, {"median": 0.0})["median"]
    group = {"score_fused_group_entries": int(gk.split("grid=")[1]) // 65536, "score_fused_group_hbm_bytes_per_launch": int((2 * gf + gw) * 1024),
             "score_fused_group_fetch_kib_raw": gf, "score_fused_group_write_kib": gw,
             "group_note": "the rolling launch: all 32k layers of a group in one grid (score_fused_kernel<128,4,2,1>, grid 65536 x entries), algorithmic 67.17 MB per entry"}
if any(k.startswith("fk::score_fused_kernel<128, 4, 2, 1") and k.endswith("grid=131072") for k in pmc.get("FETCH_SIZE", {})):
    pk = [k for k in pmc["FETCH_SIZE"] if k.startswith("fk::score_fused_kernel<128, 4, 2, 1") and k.endswith("grid=131072")][0]
    pf, pw = pmc["FETCH_SIZE"][pk]["median"], pmc.get("WRITE_SIZE", {}).get(pk, {"median": 0.0})["median"]
    pair = {"score_fused_pair_hbm_bytes_per_launch": int((2 * pf + pw) * 1024), "score_fused_pair_fetch_kib_raw": pf,
            "score_fused_pair_write_kib": pw,
            "pair_note": "two 32k layers per launch (score_fused_kernel<128,4,2,1>): algorithmic 134.35 MB; no scratch traffic to speak of "
                         "(12 B per lane: four values parked across phase B); the exponentials of two of a wave's four tiles wait for phase C in LDS"}
# the copy launches of the bench step: the one-layer launch (grid 524288: the layer-by-layer leg) when the run has it, else the largest
ckey = "fk::compact_kv_kernel<16> grid=524288" if any(k.startswith("fk::compact_kv_kernel<16> grid=524288") for k in pmc.get("FETCH_SIZE", {})) \
    else "fk::compact_kv_kernel<16>"
cf, cw = kib("FETCH_SIZE", ckey), kib("WRITE_SIZE", ckey)
json.dump({**pair, **group, **{
    "git_head_of_the_pmc_pass": head(),
    "source": f"{tag} (git {head()}): rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 3 --warmup 1 "
              "--no-extras`, medians over launches; counters are KiB; FETCH_SIZE doubled (gfx950 reports 1/2 of wide 16-B/lane "
              "reads, guides/MI355X_MICROARCH.md HBM section; the compact kernel calibrates it: 2*FETCH = its 8.39 MB of row reads)",
    dom[4:] + "_hbm_bytes_per_launch": int((2 * lf + lw) * 1024),
    dom[4:] + "_fetch_kib_raw": lf, dom[4:] + "_write_kib": lw,
    "compact_kv_hbm_bytes_per_launch": int((2 * cf + cw) * 1024),
    "compact_kv_fetch_kib_raw": cf, "compact_kv_write_kib": cw,
    "note": "dominant kernel: 2*FETCH = K once + Q window (algorithmic 67.17 MB) + hand-off records; WRITE = the fp16 window-row "
            "sums hs (score_fused) or the fp16 logits (score_logits)",
}}, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(open(os.path.join(P, f"{tag}_fk_kernels_by_grid.csv")).read())
print(open(os.path.join(P, "traffic.json")).read())
