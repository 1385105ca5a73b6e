"""One-off stress run (not part of the test suite): N random geometries of the operator on the GPU against the CPU oracle,
bit for bit -- scores, indices, K/V rows, TSP index -- biased towards the fused path's shapes (W = 8, G in {4, 8}), long and
ragged prompts, both row orders, peaked inputs, special values sprinkled in.  usage: python tools/stress_parity.py [N] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from fastkv_amd import ops
from gen_inputs import make_qkv
from oracle import fastkv_oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
dev = torch.device("cuda:0")
t0 = time.time()
fails = 0
n_entries_runs = n_entries_refused = 0
main_stream = torch.cuda.current_stream()
side_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for it in range(N):
    fusedish = rng.random() < 0.7
    W = 8 if fusedish else rng.choice([1, 4, 8, 16])
    G = rng.choice([4, 4, 8]) if fusedish else rng.choice([1, 2, 3, 4, 8])
    Hkv = rng.choice([1, 2, 4, 8])
    D = rng.choice([64, 128, 128, 128, 256])
    B = rng.choice([1, 1, 1, 2])
    ks = rng.choice([1, 3, 5, 7, 7, 13])
    S = rng.choice([rng.randint(W + 2 + ks, 3000), rng.randint(3000, 20000), rng.choice([2048, 4096, 8192, 16384, 32768])])
    if rng.random() < 0.12:                                  # many (batch x KV head) rows: the grouped-ranking compaction
        B, Hkv = rng.choice([8, 16, 32]), rng.choice([4, 8])
        S = rng.randint(W + 2 + ks, 1500)
    if B * Hkv * S * D > 40e6:
        B, Hkv = 1, min(Hkv, 4)
    cap = rng.choice([rng.randint(W + 1, S), min(S, rng.choice([256, 512, 2048])), S]) if S > W + 2 else S
    cap = max(W + 1, min(cap, S))
    tsp_len = rng.choice([0, rng.randint(W + 1, S - 1)]) if S - 1 > W + 1 else 0
    pooling = rng.choice(["avgpool", "maxpool"])
    order = rng.choice(["index", "score"])
    peaked = rng.choice([0, 0, 50])
    q, k, v = make_qkv(9000 + it, B, Hkv * G, Hkv, S, D, W, peaked=peaked)
    special = rng.random() < 0.1
    if special:                                              # special values: inf / nan / huge in a few K rows and one Q row
        k = k.clone(); q = q.clone()
        for _ in range(3):
            k[rng.randrange(B), rng.randrange(Hkv), rng.randrange(S), rng.randrange(D)] = rng.choice([float("inf"), float("-inf"), float("nan"), 60000.0, -60000.0])
        if rng.random() < 0.5:
            q[0, rng.randrange(Hkv * G), S - 1 - rng.randrange(W), rng.randrange(D)] = rng.choice([float("inf"), float("nan"), 30000.0])
    tag = dict(it=it, B=B, H=Hkv * G, Hkv=Hkv, S=S, D=D, W=W, ks=ks, cap=cap, tsp_len=tsp_len, pooling=pooling, order=order, peaked=peaked, special=special)
    want = O.update_kv(q, k, v, W, ks, pooling, cap, tsp_len, order, return_scores=True)
    qd, kd, vd = (t.transpose(1, 2).contiguous().to(dev).transpose(1, 2) for t in (q, k, v))
    # the call sequence is part of the test: engines, the scores-only entry point and the strided (cache slab) variant are mixed
    # in at random, all sharing one workspace whose hand-off areas move with the shape
    engine = rng.choice(["auto", "auto", "auto", "valu", "mfma"])
    ops.set_score_engine(engine)
    stream = side_streams[rng.randrange(len(side_streams))] if rng.random() < 0.25 else torch.cuda.current_stream()
    stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(stream)                            # every stream has its own workspace (and epoch); launches are chained
    pre = rng.random() < 0.2
    if pre:
        c_only, t_only = ops.scores(qd, kd, W, ks, pooling)
    slab = rng.random() < 0.15
    if slab:
        rows = cap + rng.randint(0, 64)
        ks_, vs_ = (torch.zeros(B, Hkv, rows, D, dtype=torch.float16, device=dev) for _ in range(2))
        got = list(ops.update_kv(qd, kd, vd, W, ks, pooling, cap, tsp_len, order, return_indices=True, return_scores=True,
                                 out=(ks_[:, :, :cap], vs_[:, :, :cap])))
        got[0], got[1] = ks_[:, :, :cap], vs_[:, :, :cap]
    else:
        got = ops.update_kv(qd, kd, vd, W, ks, pooling, cap, tsp_len, order, return_indices=True, return_scores=True)
    torch.cuda.synchronize()
    torch.cuda.set_stream(main_stream)
    ops.set_score_engine("auto")
    if fusedish and B == 1 and rng.random() < 0.25:
        # the same geometry as several SEPARATELY ALLOCATED entries in one launch sequence (fastkv_update_kv_ptrs_f16); refused
        # (nothing launched) when that many do not fit the fused kernel's residency
        ne = rng.choice([2, 2, 3, 5, 16])
        ins = [(q, k, v)] + [make_qkv(9000 + it + 100000 * j, B, Hkv * G, Hkv, S, D, W, peaked=peaked) for j in range(1, ne)]
        if S * ne * Hkv * D < 150e6:
            try:
                dq, dk, dv = ([t[j].transpose(1, 2).contiguous().to(dev).transpose(1, 2) for t in ins] for j in range(3))
                qwin = rng.random() < 0.4                        # only the window rows of q kept (a waiting layer of DeferredCompression)
                if qwin:
                    dq = [ops.window_rows(t, W) for t in dq]
                # (more entries than one fused launch holds are scored by several launches and selected / copied once)
                ge = ops.update_kv_entries(dq, dk, dv, W, ks, pooling, cap, tsp_len, order, return_indices=True, q_window=qwin)
                torch.cuda.synchronize()
                n_entries_runs += 1
                for j, (qj, kj, vj) in enumerate(ins):
                    wj = want if j == 0 else O.update_kv(qj, kj, vj, W, ks, pooling, cap, tsp_len, order)
                    okj = torch.equal(ge[0][j].cpu().view(torch.int16), wj[0].view(torch.int16)) and torch.equal(ge[3][j:j + 1].cpu(), wj[2]) and \
                        torch.equal(ge[1][j].cpu().view(torch.int16), wj[1].view(torch.int16)) and \
                        ((ge[2] is None and wj[3] is None) or torch.equal(ge[2][j:j + 1].cpu(), wj[3]))
                    if not okj:
                        fails += 1
                        print("MISMATCH (entries)", tag, dict(entries=ne, entry=j), flush=True)
                        break
            except ops.FastKVNativeError if hasattr(ops, "FastKVNativeError") else Exception as e:
                if "unsupported" not in str(e).lower():
                    raise
                n_entries_refused += 1
    if pre and not (torch.equal(c_only.cpu().view(torch.int16), want[4].view(torch.int16))):
        print("MISMATCH (scores-only entry point)", dict(it=it, engine=engine), flush=True)
        fails += 1
    ok = torch.equal(got[4].cpu().view(torch.int16), want[4].view(torch.int16)) and torch.equal(got[3].cpu(), want[2]) and \
        torch.equal(got[0].cpu().view(torch.int16), want[0].view(torch.int16)) and torch.equal(got[1].cpu().view(torch.int16), want[1].view(torch.int16)) and \
        ((got[2] is None and want[3] is None) or torch.equal(got[2].cpu(), want[3]))
    if not ok:
        fails += 1
        what = dict(scores=torch.equal(got[4].cpu().view(torch.int16), want[4].view(torch.int16)), idx=torch.equal(got[3].cpu(), want[2]),
                    k=torch.equal(got[0].cpu().view(torch.int16), want[0].view(torch.int16)), v=torch.equal(got[1].cpu().view(torch.int16), want[1].view(torch.int16)),
                    tsp=(got[2] is None and want[3] is None) or torch.equal(got[2].cpu(), want[3]))
        nd = int((got[4].cpu().view(torch.int16) != want[4].view(torch.int16)).sum())
        print("MISMATCH", tag, what, "score elements differing", nd, "nan in oracle scores", int(torch.isnan(want[4]).sum()), flush=True)
        gsc, wsc = got[4].cpu().view(torch.int16), want[4].view(torch.int16)
        for ix in (gsc != wsc).nonzero()[:16].tolist():
            print("   at", ix, "gpu %04x oracle %04x" % (int(gsc[tuple(ix)]) & 0xffff, int(wsc[tuple(ix)]) & 0xffff), flush=True)
print(f"{N} cases, {fails} mismatches, {time.time() - t0:.0f} s (entries calls: {n_entries_runs} run, {n_entries_refused} refused)")
sys.exit(1 if fails else 0)
