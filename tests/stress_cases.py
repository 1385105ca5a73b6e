"""Randomised parity stress of the operator (shared by tools/stress_parity.py and tests/test_stress_gpu.py): N random geometries on
the GPU against the CPU oracle, bit for bit -- scores, indices, K/V rows, TSP index -- biased towards the fused path's shapes
(W = 8, G in {4, 8}), long and ragged prompts, both row orders, peaked inputs, special values sprinkled in, the call sequence
(engines, streams, the scores-only entry point, cache-slab outputs, separately allocated entries) part of the test.
Every random decision of a case is drawn before any work, so `only=i` replays case i of a sequence exactly."""
import random
import time

import torch


def run_stress(N=200, seed=12345, entries_p=0.25, all_entries=False, only=-1, repeat=1, log=print):
    """Returns a dict: cases, mismatches, seconds, entries_runs, entries_refused, max_entry_rows (largest entries x KV heads of an
    entries call that ran), special (cases with non-finite / huge values), engines (set of engines used), violations."""
    from fastkv_amd import ops
    from fastkv_amd._lib import load as _load
    from gen_inputs import make_qkv
    from oracle import fastkv_oracle as O

    rng = random.Random(seed)
    dev = torch.device("cuda:0")
    violations0 = int(_load().fastkv_placement_violations(0))          # (a running count of the process)
    import os
    from helpers import default_contraction as _dc
    default_contraction = _dc()                                         # (what "auto" means on both sides of THIS process)
    t0 = time.time()
    st = dict(cases=0, mismatches=0, entries_runs=0, entries_refused=0, max_entry_rows=0, special=0, engines=set())
    main_stream = torch.cuda.current_stream()
    side_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ONLY, REPEAT, ENTRIES_P, ALL_ENTRIES = only, repeat, entries_p, all_entries

    def print(*a, **k):   # noqa: A001 -- the body below was written against print
        k.pop("flush", None)
        log(" ".join(str(x) for x in a))

    def same16(a, b):
        return torch.equal(a.cpu().view(torch.int16), b.view(torch.int16))

    def run_case(c, rep):
        """One pass of case `c` (a dict of its decisions, inputs and oracle results) on the GPU; returns the number of mismatches."""
        bad = 0
        B, Hkv, D, W, ks, cap, tsp_len, pooling, order = (c[x] for x in ("B", "Hkv", "D", "W", "ks", "cap", "tsp_len", "pooling", "order"))
        q, k, v = c["ins"][0]
        want = c["wants"][0]
        qd, kd, vd = (t.transpose(1, 2).contiguous().to(dev).transpose(1, 2) for t in (q, k, v))
        # the call sequence is part of the test: engines, the scores-only entry point and the strided (cache slab) variant are mixed
        # in at random, all sharing one workspace whose hand-off areas move with the shape
        ops.set_score_engine(c["engine"])
        stream = side_streams[c["stream_pick"]] if c["stream_pick"] >= 0 else torch.cuda.current_stream()
        stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(stream)                            # every stream has its own workspace (and epoch); launches are chained
        if c["pre"]:
            c_only, t_only = ops.scores(qd, kd, W, ks, pooling)
        if c["slab"]:
            rows = cap + c["slab_extra"]
            ks_, vs_ = (torch.zeros(B, Hkv, rows, D, dtype=torch.float16, device=dev) for _ in range(2))
            got = list(ops.update_kv(qd, kd, vd, W, ks, pooling, cap, tsp_len, order, return_indices=True, return_scores=True,
                                     out=(ks_[:, :, :cap], vs_[:, :, :cap])))
            got[0], got[1] = ks_[:, :, :cap], vs_[:, :, :cap]
        else:
            got = ops.update_kv(qd, kd, vd, W, ks, pooling, cap, tsp_len, order, return_indices=True, return_scores=True)
        torch.cuda.synchronize()
        torch.cuda.set_stream(main_stream)
        ne, qwin = c["ne"], c["qwin"]
        if ne:
            # the same geometry as several SEPARATELY ALLOCATED entries in one launch sequence (fastkv_update_kv_ptrs_f16); refused
            # (nothing launched) when that many do not fit the fused kernel's residency
            try:
                dq, dk, dv = ([t[j].transpose(1, 2).contiguous().to(dev).transpose(1, 2) for t in c["ins"]] for j in range(3))
                if qwin:                                         # only the window rows of q kept (a waiting layer of DeferredCompression)
                    dq = [ops.window_rows(t, W) for t in dq]
                # (more entries than one fused launch holds are scored by several launches and selected / copied once)
                ge = ops.update_kv_entries(dq, dk, dv, W, ks, pooling, cap, tsp_len, order, return_indices=True, q_window=qwin)
                torch.cuda.synchronize()
                st["entries_runs"] += 1; st["max_entry_rows"] = max(st["max_entry_rows"], ne * Hkv)
                for j, wj in enumerate(c["wants"]):
                    what = dict(k=not same16(ge[0][j], wj[0]), v=not same16(ge[1][j], wj[1]), idx=int((ge[3][j:j + 1].cpu() != wj[2]).sum()),
                                tsp=0 if ge[2] is None and wj[3] is None else int((ge[2][j:j + 1].cpu() != wj[3]).sum()))
                    if what["k"] or what["v"] or what["idx"] or what["tsp"]:
                        bad += 1
                        print("MISMATCH (entries)", c["tag"], dict(entries=ne, entry=j, q_window=qwin, engine=c["engine"], rep=rep), what, flush=True)
                        if what["idx"]:
                            print("   first differing (row, head, slot):", (ge[3][j:j + 1].cpu() != wj[2]).nonzero()[:6].tolist(), flush=True)
                            for h in range(Hkv):
                                a, b2 = set(ge[3][j, h].cpu().tolist()), set(wj[2][0, h].tolist())
                                print(f"   head {h}: positions only the GPU kept {sorted(a - b2)[:24]}, only the oracle kept {sorted(b2 - a)[:24]}", flush=True)
                        if what["tsp"]:
                            a, b2 = set(ge[2][j].cpu().tolist()), set(wj[3][0].tolist())
                            print(f"   TSP: positions only the GPU kept {sorted(a - b2)[:24]}, only the oracle kept {sorted(b2 - a)[:24]}", flush=True)
                        if not ALL_ENTRIES:
                            break
            except ops.FastKVNativeError as e:
                if "unsupported" not in str(e).lower():
                    raise
                st["entries_refused"] += 1
        ops.set_score_engine("auto")
        if c["pre"] and not same16(c_only, want[4]):
            print("MISMATCH (scores-only entry point)", dict(it=c["tag"]["it"], engine=c["engine"], rep=rep), flush=True)
            bad += 1
        what = dict(scores=same16(got[4], want[4]), idx=torch.equal(got[3].cpu(), want[2]), k=same16(got[0], want[0]), v=same16(got[1], want[1]),
                    tsp=(got[2] is None and want[3] is None) or torch.equal(got[2].cpu(), want[3]))
        if not all(what.values()):
            bad += 1
            gsc, wsc = got[4].cpu().view(torch.int16), want[4].view(torch.int16)
            print("MISMATCH", c["tag"], dict(rep=rep), what, "score elements differing", int((gsc != wsc).sum()), "nan in oracle scores",
                  int(torch.isnan(want[4]).sum()), flush=True)
            for ix in (gsc != wsc).nonzero()[:16].tolist():
                print("   at", ix, "gpu %04x oracle %04x" % (int(gsc[tuple(ix)]) & 0xffff, int(wsc[tuple(ix)]) & 0xffff), flush=True)
        return bad


    for it in range(N):
        live = ONLY < 0 or it == ONLY
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
        if live:
            q, k, v = make_qkv(9000 + it, B, Hkv * G, Hkv, S, D, W, peaked=peaked)
        special = rng.random() < 0.1
        if special:                                              # special values: inf / nan / huge in a few K rows and one Q row
            if live:
                k, q = k.clone(), q.clone()
            for _ in range(3):
                val = rng.choice([float("inf"), float("-inf"), float("nan"), 60000.0, -60000.0])
                at = (rng.randrange(B), rng.randrange(Hkv), rng.randrange(S), rng.randrange(D))
                if live:
                    k[at] = val
            if rng.random() < 0.5:
                val = rng.choice([float("inf"), float("nan"), 30000.0])
                at = (0, rng.randrange(Hkv * G), S - 1 - rng.randrange(W), rng.randrange(D))
                if live:
                    q[at] = val
        tag = dict(it=it, B=B, H=Hkv * G, Hkv=Hkv, S=S, D=D, W=W, ks=ks, cap=cap, tsp_len=tsp_len, pooling=pooling, order=order, peaked=peaked, special=special)
        # every random decision of the case before any work (a replay of one case consumes the generator exactly as the full run)
        case = dict(B=B, Hkv=Hkv, D=D, W=W, ks=ks, cap=cap, tsp_len=tsp_len, pooling=pooling, order=order, tag=tag)
        case["engine"] = rng.choice(["auto", "auto", "auto", "valu", "mfma"])
        case["stream_pick"] = rng.randrange(len(side_streams)) if rng.random() < 0.25 else -1
        case["pre"] = rng.random() < 0.2
        case["slab"] = rng.random() < 0.15
        case["slab_extra"] = rng.randint(0, 64) if case["slab"] else 0
        ne, qwin = 0, False
        if fusedish and B == 1 and rng.random() < ENTRIES_P:
            ne = rng.choice([2, 2, 3, 5, 16])
            if S * ne * Hkv * D < 150e6:
                qwin = rng.random() < 0.4
            else:
                ne = 0
        case["ne"], case["qwin"] = ne, qwin
        if not live:
            continue
        st["cases"] += 1
        st["special"] += int(special)
        st["engines"].add(case["engine"])
        case["ins"] = [(q, k, v)] + [make_qkv(9000 + it + 100000 * j, B, Hkv * G, Hkv, S, D, W, peaked=peaked) for j in range(1, ne)]
        # the oracle under the contract the case's engine computes: "valu" / "mfma" = the fp32 fma chain, "auto" = the library default
        # (the fp16 matrix instruction); the entries call of the case runs on the same engine
        O.set_contraction("fmaf" if case["engine"] in ("valu", "mfma") else default_contraction)
        case["wants"] = [O.update_kv(q, k, v, W, ks, pooling, cap, tsp_len, order, return_scores=True)] + \
            [O.update_kv(qj, kj, vj, W, ks, pooling, cap, tsp_len, order) for qj, kj, vj in case["ins"][1:]]
        if ONLY >= 0:
            print("replaying", tag, {x: case[x] for x in ("engine", "stream_pick", "pre", "slab", "ne", "qwin")}, flush=True)
        for rep in range(REPEAT if ONLY >= 0 else 1):
            st["mismatches"] += run_case(case, rep)
    O.set_contraction(default_contraction)
    st["violations"] = int(_load().fastkv_placement_violations(0)) - violations0
    st["seconds"] = time.time() - t0
    return st
