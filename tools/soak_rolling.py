"""Soak of the rolling launch (csrc/fused.hip launch_score_fused): random groups of 3-20 entries of 8k-32k tokens (two to eight entries on the chip at a time), ragged lengths, both poolings, NaN / Inf
sprinkles, now and then a foreign kernel holding some compute units -- every output of `ops.update_kv` with the rolling launch ON must equal the
run with it OFF (launches of two entries, in step) bit for bit, and nothing may be reported.  Usage: soak_rolling.py [seconds] [seed]."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastkv_amd import ops
from fastkv_amd._lib import load, raise_if_aborted

budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
L = load()
H, Hkv, D, W = 32, 8, 128, 8
SMAX, BMAX, SMIN = int(os.environ.get("SOAK_SMAX", "32768")), int(os.environ.get("SOAK_BMAX", "20")), int(os.environ.get("SOAK_SMIN", "8192"))     # e.g. SOAK_SMAX=131072 SOAK_BMAX=6 SOAK_SMIN=40000: long prompts, rows taken in parts
g = torch.Generator(device=dev).manual_seed(7)
Q = torch.randn(BMAX, SMAX, H, D, generator=g, device=dev, dtype=torch.float16)
K = torch.randn(BMAX, SMAX, Hkv, D, generator=g, device=dev, dtype=torch.float16)
V = torch.randn(BMAX, SMAX, Hkv, D, generator=g, device=dev, dtype=torch.float16)
side = torch.cuda.Stream()
t0, it, bad, held = time.time(), 0, 0, 0
history = []
prev = ops.set_fused_rolling(True)
while time.time() - t0 < budget_s:
    B = rng.randint(1 if SMIN > 32768 else 3, BMAX)
    S = rng.choice([SMAX, rng.randint(max(SMIN, min(22000, SMAX)), SMAX), rng.randint(SMIN, SMAX), rng.randint(SMIN, SMAX) // 64 * 64, max(SMIN, 16384), SMIN])
    ks, pooling = rng.choice([1, 3, 5, 7, 13]), rng.choice(["avgpool", "maxpool"])
    cap, tsp, order = rng.choice([512, 2048, 3000]), rng.choice([0, 2048]), rng.choice(["score", "index"])
    b0 = rng.randint(0, BMAX - B)
    q, k, v = Q[b0:b0 + B, :S].transpose(1, 2), K[b0:b0 + B, :S].transpose(1, 2), V[b0:b0 + B, :S].transpose(1, 2)
    saved = []
    if rng.random() < 0.3:                                       # poison a few keys / a query row of one entry for this iteration
        e = rng.randrange(B)
        for _ in range(rng.randint(1, 4)):
            h, j = rng.randrange(Hkv), rng.randrange(S)
            saved.append((k, (e, h, j), k[e, h, j].clone()))
            k[e, h, j, rng.randrange(D)] = rng.choice([float("nan"), float("inf"), float("-inf")])
        if rng.random() < 0.5:
            h, r = rng.randrange(H), S - 1 - rng.randrange(W)
            saved.append((q, (e, h, r), q[e, h, r].clone()))
            q[e, h, r, 0] = float("nan")
    outs = {}
    history.append(dict(B=B, S=S, ks=ks, pooling=pooling, cap=cap, tsp=tsp, order=order, poisoned=bool(saved), b0=b0))
    del history[:-4]
    for rolling in (True, False):
        ops.set_fused_rolling(rolling and os.environ.get("SOAK_NO_ROLLING", "0") != "1")      # (SOAK_NO_ROLLING=1: the regular launches against themselves)
        if rolling and rng.random() < 0.25:                      # a foreign kernel takes 16-96 compute units for a few hundred microseconds
            assert L.fastkv_debug_occupy(rng.choice([16, 48, 96]), 128 * 1024, rng.choice([200, 600, 1500]), side.cuda_stream) == 0
            held += 1
        try:
            outs[rolling] = ops.update_kv(q, k, v, W, ks, pooling, cap, tsp, order, return_indices=True, return_scores=True)
            torch.cuda.current_stream().synchronize()
            raise_if_aborted()
        except Exception as ex:   # noqa: BLE001 -- a REPORTED launch (FASTKV_EABORTED / FASTKV_EPLACEMENT): say which group, go on with the fused kernels
            bad += 1
            print("REPORTED", dict(it=it, rolling=rolling), repr(ex)[:160], "groups:", history[-3:], flush=True)
            torch.cuda.synchronize()
            ops.set_no_wait_mode(False)
            outs[rolling] = ops.update_kv(q, k, v, W, ks, pooling, cap, tsp, order, return_indices=True, return_scores=True)
            torch.cuda.current_stream().synchronize()
    for t, ix, val in saved:
        t[ix] = val
    same = True
    for a, b in zip(outs[True], outs[False]):
        if (a is None) != (b is None):
            same = False
        elif a is not None:
            same = same and bool(torch.equal(a.view(torch.int16) if a.dtype == torch.float16 else a, b.view(torch.int16) if b.dtype == torch.float16 else b))
    if not same:
        bad += 1
        which = [nm for nm, a, b in zip(("k_out", "v_out", "tsp_idx", "idx", "scores"), outs[True], outs[False])
                 if a is not None and not torch.equal(a.view(torch.int16) if a.dtype == torch.float16 else a, b.view(torch.int16) if b.dtype == torch.float16 else b)]
        print("MISMATCH", dict(it=it, B=B, S=S, ks=ks, pooling=pooling, cap=cap, tsp=tsp, order=order, poisoned=bool(saved), b0=b0, differs=which), "previous groups:", history[-3:], flush=True)
        # which of the two was wrong?  a third run on the idle chip, regular launches
        torch.cuda.synchronize()
        ops.set_fused_rolling(False)
        third = ops.update_kv(q, k, v, W, ks, pooling, cap, tsp, order, return_indices=True, return_scores=True)
        torch.cuda.synchronize()
        agree = [all(a is None or torch.equal(a.view(torch.int16) if a.dtype == torch.float16 else a, b.view(torch.int16) if b.dtype == torch.float16 else b) for a, b in zip(outs[r], third)) for r in (True, False)]
        print("   a third run (regular, idle) agrees with: first run", agree[0], ", second run", agree[1], flush=True)
    try:
        raise_if_aborted()
    except Exception as ex:   # noqa: BLE001
        bad += 1
        print("REPORTED", it, repr(ex)[:200], flush=True)
    it += 1
torch.cuda.synchronize()
ops.set_fused_rolling(prev)
print(f"{it} groups in {time.time() - t0:.0f} s ({held} beside a foreign kernel): {bad} mismatches / reports; placement violations {L.fastkv_placement_violations(0)}")
sys.exit(1 if bad else 0)
