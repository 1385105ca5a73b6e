"""Round 5 A/B of the rolling launch's two sizes (set per process): FASTKV_FUSED_OCC3=0 -- entries of four tiles per wave, two workgroups
per compute unit (round 4) -- against 1.. -- entries of two tiles per wave, THREE workgroups per compute unit (csrc/fused.hip).
ops.scores of B 32k layers, rotating over K sets larger than the Infinity Cache: us per call / per layer, and a digest of the scores
(the two sizes must give the same bits)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastkv_amd import ops
dev = torch.device("cuda:0")
H, Hkv, D, W, S = 32, 8, 128, 8, int(os.environ.get("EXP_S", "32768"))
tag = f"OCC3={os.environ.get('FASTKV_FUSED_OCC3', '0')} CONV={os.environ.get('FASTKV_FUSED_CONVEYOR', '0')} TUNE={os.environ.get('FASTKV_FUSED_TUNE', '1')} SPLIT1={os.environ.get('FASTKV_FUSED_SPLIT1', '0')}"
for B in ([int(os.environ["EXP_B"])] if os.environ.get("EXP_B") else (3, 4, 5, 8, 16)):
    nset = max(2, 16 // B)
    g = torch.Generator(device=dev).manual_seed(17 + B)
    sets = [(torch.randn(B, S, H, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2),
             torch.randn(B, S, Hkv, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)) for _ in range(nset)]
    dig = hashlib.sha256()
    for i in range(nset):
        c, t = ops.scores(*sets[i], W, 7, "maxpool", want_tsp=True)
        torch.cuda.synchronize()
        dig.update(c.cpu().numpy().tobytes()); dig.update(t.cpu().numpy().tobytes())
    for i in range(4):
        ops.scores(*sets[i % nset], W, 7, "maxpool", want_tsp=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 40
    e0.record()
    for i in range(n):
        ops.scores(*sets[i % nset], W, 7, "maxpool", want_tsp=False)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    print(f"{tag} B={B}: {us:7.1f} us per call, {us / B:6.1f} us per layer, scores sha {dig.hexdigest()[:16]}", flush=True)
    del sets
from fastkv_amd._lib import raise_if_aborted
raise_if_aborted()
