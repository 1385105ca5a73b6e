"""Debug helper for tools/stress_parity.py: re-generates case `it` of a stress run and prints where GPU and oracle scores differ."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from fastkv_amd import ops
from gen_inputs import make_qkv
from oracle import fastkv_oracle as O
target = [int(x) for x in sys.argv[1].split(",")]
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
dev = torch.device("cuda:0")
for it in range(max(target) + 1):
    fusedish = rng.random() < 0.7
    W = 8 if fusedish else rng.choice([1, 4, 8, 16])
    G = rng.choice([4, 4, 8]) if fusedish else rng.choice([1, 2, 3, 4, 8])
    Hkv = rng.choice([1, 2, 4, 8]); D = rng.choice([64, 128, 128, 128, 256]); B = rng.choice([1, 1, 1, 2]); ks = rng.choice([1, 3, 5, 7, 7, 13])
    S = rng.choice([rng.randint(W + 2 + ks, 3000), rng.randint(3000, 20000), rng.choice([2048, 4096, 8192, 16384, 32768])])
    if B * Hkv * S * D > 40e6:
        B, Hkv = 1, min(Hkv, 4)
    cap = rng.choice([rng.randint(W + 1, S), min(S, rng.choice([256, 512, 2048])), S]) if S > W + 2 else S
    cap = max(W + 1, min(cap, S))
    tsp_len = rng.choice([0, rng.randint(W + 1, S - 1)]) if S - 1 > W + 1 else 0
    pooling = rng.choice(["avgpool", "maxpool"]); order = rng.choice(["index", "score"]); peaked = rng.choice([0, 0, 50])
    special = rng.random() < 0.1
    inj = []
    if special:
        for _ in range(3):                                   # (an assignment evaluates its right-hand side first)
            val = rng.choice([float("inf"), float("-inf"), float("nan"), 60000.0, -60000.0])
            inj.append(("k", rng.randrange(B), rng.randrange(Hkv), rng.randrange(S), rng.randrange(D), val))
        if rng.random() < 0.5:
            val = rng.choice([float("inf"), float("nan"), 30000.0])
            inj.append(("q", 0, rng.randrange(Hkv * G), S - 1 - rng.randrange(W), rng.randrange(D), val))
    if it not in target:
        continue
    q, k, v = make_qkv(9000 + it, B, Hkv * G, Hkv, S, D, W, peaked=peaked)
    k = k.clone(); q = q.clone()
    for t, b, h, s_, d, val in inj:
        (k if t == "k" else q)[b, h, s_, d] = val
    print("case", it, dict(B=B, H=Hkv * G, Hkv=Hkv, S=S, D=D, W=W, ks=ks, cap=cap, tsp=tsp_len, pooling=pooling, order=order), "injected", inj)
    want = O.update_kv(q, k, v, W, ks, pooling, cap, tsp_len, order, return_scores=True)
    qd, kd, vd = (t.transpose(1, 2).contiguous().to(dev).transpose(1, 2) for t in (q, k, v))
    got = ops.update_kv(qd, kd, vd, W, ks, pooling, cap, tsp_len, order, return_indices=True, return_scores=True)
    g, w = got[4].cpu().view(torch.int16), want[4].view(torch.int16)
    nz = (g != w).nonzero()
    print(" differing", len(nz))
    for idx in nz[:8].tolist():
        print("  at", idx, "gpu %04x" % (int(g[tuple(idx)]) & 0xffff), "oracle %04x" % (int(w[tuple(idx)]) & 0xffff))
