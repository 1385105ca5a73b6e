"""Experiment (measurement build: FASTKV_BUILD_DIR=build_x_stag FASTKV_CXXFLAGS="-DFK_HUNT -DFK_OLD_NUMBERING -DFK_DBG_DELAY=8 -DFK_DBG_WHO=(yb==1)"):
the pair launch of score_fused with the two layers' workgroups sharing compute units (old numbering: workgroup (x, layer 0) beside
(x, layer 1)) and layer 1 started DELAY_TICKS late -- does one layer's memory-bound phase A overlap the other's vector phases?
Inputs rotate over six K sets (0.8 GB) so that nothing is served by the 256 MB Infinity Cache."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastkv_amd import ops, _lib
dev = torch.device("cuda:0")
H, Hkv, D, W, S, B = 32, 8, 128, 8, 32768, 2
lib = _lib.load()
sets = [(torch.randn(B, S, H, D, device=dev, dtype=torch.float16).transpose(1, 2), torch.randn(B, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)) for _ in range(6)]
for ticks in [int(a) for a in sys.argv[1:]] or [0, 500, 1000, 1250, 1500, 2000, 2500]:
    if hasattr(lib, "fastkv_debug_set_delay"):
        assert lib.fastkv_debug_set_delay(ticks) == 0
    for i in range(12):
        ops.scores(*sets[i % 6], W, 7, "maxpool", want_tsp=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 60
    e0.record()
    for i in range(n):
        ops.scores(*sets[i % 6], W, 7, "maxpool", want_tsp=False)
    e1.record()
    torch.cuda.synchronize()
    print(f"layer 1 delayed by {ticks / 100:5.1f} us: {e0.elapsed_time(e1) * 1000 / n:6.1f} us per call (ops.scores of two 32k layers)", flush=True)
