"""A NaN in one batch row's query window corrupts the scores of the batch rows whose workgroups share compute units with that row's
(fused scoring kernel).  Batched call, scores compared element by element."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from fastkv_amd import ops
from gen_inputs import make_qkv
from oracle import fastkv_oracle as O
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
H, Hkv, S, D, W, ks = int(os.environ.get("HQ", "8")), 1, int(os.environ.get("SLEN", "14695")), 128, 8, int(os.environ.get("KS", "13"))
B = int(os.environ.get("NE", "16"))
cap, tsp_len = 8316, 10400
ins = [make_qkv(9197 + 100000 * j, 1, H, Hkv, S, D, W, peaked=50) for j in range(B)]
q, k, v = (torch.cat([t[j] for t in ins], dim=0) for j in range(3))
kind = os.environ.get("SPECIAL", "Q")
if kind == "Q": q[0, 3, S - 2, 9] = float("nan")
if kind == "K": k[0, 0, 5000, 17] = float("nan")
if kind == "KALL": k[0, 0, :, 17] = float("nan")
want = O.update_kv(q, k, v, W, ks, "avgpool", cap, tsp_len, "index", return_scores=True)
qd, kd, vd = (t.transpose(1, 2).contiguous().to(dev).transpose(1, 2) for t in (q, k, v))
wsc = want[4].view(torch.int16)
if os.environ.get("DELAY_TICKS"):                        # measurement builds with -DFK_HUNT -DFK_DBG_DELAY (csrc/fk_hunt.h): the length of a delay in 100 MHz ticks
    from fastkv_amd._lib import load
    assert load().fastkv_debug_set_delay(int(os.environ["DELAY_TICKS"])) == 0
nbad = 0
for it in range(N):
    got = ops.update_kv(qd, kd, vd, W, ks, "avgpool", cap, tsp_len, "index", return_indices=True, return_scores=True)
    torch.cuda.synchronize()
    gsc = got[4].cpu().view(torch.int16)
    d = (gsc != wsc).nonzero()
    if len(d):
        nbad += 1
        rows = sorted(set(d[:, 0].tolist()))
        print("round", it, "rows", rows, "elements", len(d))
        for r in rows[:2]:
            pos = d[d[:, 0] == r][:, 2].tolist()
            print("   row", r, "positions", pos[:40])
            print("   gpu   ", [float(got[4][r, 0, p].cpu()) for p in pos[:12]])
            print("   oracle", [float(want[4][r, 0, p]) for p in pos[:12]])
print(N, "rounds,", nbad, "bad")
