"""Hunt: ONE layer (B = 1) through ops.update_kv again and again while a foreign kernel holds some compute units now and then -- does a
squeezed launch (workgroups of different heads sharing compute units: the placement check counts it) ever give a different result than
the idle chip?  Which output differs, and did the iteration count placement violations?  usage: repro_squeeze.py [seconds] [S] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastkv_amd import ops
from fastkv_amd._lib import load
budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
S = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
rng = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 5)
dev = torch.device("cuda:0")
L = load()
H, Hkv, D, W = 32, 8, 128, 8
g = torch.Generator(device=dev).manual_seed(11)
q = torch.randn(1, S, H, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
k = torch.randn(1, S, Hkv, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
v = torch.randn(1, S, Hkv, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
args = (W, 13, "avgpool", 2048, 0, "score")
ref = ops.update_kv(q, k, v, *args, return_indices=True, return_scores=True)
torch.cuda.synchronize()
names = ("k_out", "v_out", "tsp_idx", "idx", "scores")
side = torch.cuda.Stream()
t0, it, bad, squeezed = time.time(), 0, 0, 0
viol0 = L.fastkv_placement_violations(0)
while time.time() - t0 < budget_s:
    held = rng.random() < 0.6
    if held:
        assert L.fastkv_debug_occupy(rng.choice([16, 48, 96, 128, 200]), 128 * 1024, rng.choice([100, 300, 800, 1500]), side.cuda_stream) == 0
        if rng.random() < 0.5:
            time.sleep(rng.random() * 3e-4)
    got = ops.update_kv(q, k, v, *args, return_indices=True, return_scores=True)
    torch.cuda.current_stream().synchronize()
    v1 = L.fastkv_placement_violations(0)
    sq = v1 != viol0
    squeezed += sq
    viol0 = v1
    for nm, a, b in zip(names, got, ref):
        if a is None:
            continue
        aa, bb = (a.view(torch.int16), b.view(torch.int16)) if a.dtype == torch.float16 else (a, b)
        if not torch.equal(aa, bb):
            bad += 1
            ne = (aa != bb)
            print(f"it {it}: {nm} differs in {int(ne.sum())} elements (held={held}, violations counted in this call={sq}); first at {ne.nonzero()[0].tolist()}", flush=True)
    try:
        ops.raise_if_aborted("hunt")
    except Exception as ex:   # noqa: BLE001
        print("REPORTED", it, repr(ex)[:160], flush=True)
    it += 1
torch.cuda.synchronize()
print(f"{it} calls in {time.time() - t0:.0f} s, {squeezed} with placement violations, {bad} differing outputs; contraction {os.environ.get('FASTKV_CONTRACTION', 'fmaf')}")
