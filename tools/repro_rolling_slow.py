"""Round 6: the known trigger of round 3's co-residency damage (docs/HISTORY.md: one entry of a multi-entry launch whose query window holds a
NaN -- all its tiles take the slow vector-ALU redo of phase A -- beside entries that run on) in the geometry of the ROLLING launch under the
fp32-fma-chain contract: 16 entries of G = 4 heads (H 32 / Hkv 8), 14,695 / 24,001 / 32,768 tokens, the NaN in entry 0 or in two entries.
The rolling launch puts the slow entry's workgroups beside OTHER entries' workgroups on its compute units, out of step, for several
entry lifetimes -- exactly the pairing round 3's second fence (adjacent spans of one head, in step) was built to avoid, now relying on the
first fence alone (no packed-fp32 instructions in the library).  Every launch is compared with the same call through the regular
launches (in step) bit for bit.  Usage: repro_rolling_slow.py [launches per shape = 100]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastkv_amd import ops
from fastkv_amd._lib import raise_if_aborted
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
H, Hkv, D, W = 32, 8, 128, 8
bad = tot = 0
t0 = time.time()
for S, B, slow in ((14695, 16, (0,)), (24001, 12, (0, 5)), (32768, 10, (1,))):
    g = torch.Generator(device=dev).manual_seed(S)
    q = torch.randn(B, S, H, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
    k = torch.randn(B, S, Hkv, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
    v = torch.randn(B, S, Hkv, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
    for e in slow:
        q[e, :, S - 3, 17] = float("nan")
    ops.set_fused_rolling(False)
    ref = ops.update_kv(q, k, v, W, 7, "avgpool", 2048, 2048, "score", return_indices=True, return_scores=True)
    torch.cuda.synchronize()
    ops.set_fused_rolling(True)
    for it in range(N):
        got = ops.update_kv(q, k, v, W, 7, "avgpool", 2048, 2048, "score", return_indices=True, return_scores=True)
        torch.cuda.synchronize()
        raise_if_aborted()
        same = all(torch.equal(a.view(torch.int16) if a.dtype == torch.float16 else a, b.view(torch.int16) if b.dtype == torch.float16 else b) for a, b in zip(got, ref))
        tot += 1
        if not same:
            bad += 1
            d = [(got[4][e] != ref[4][e]).sum().item() if not torch.isnan(ref[4][e]).any() else -1 for e in range(B)]
            print("MISMATCH", dict(S=S, B=B, slow=slow, it=it), "differing scores per entry (-1: NaN entry)", d, flush=True)
print(f"{tot} rolling launches ({os.environ.get('FASTKV_CONTRACTION', 'fmaf (default)')}), {bad} differ from the in-step launches; {time.time() - t0:.0f} s; no-wait mode {ops.no_wait_mode()}")
sys.exit(1 if bad else 0)
