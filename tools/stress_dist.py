"""One-off stress run for the sequence-sharded operator on the GPU (not part of the suite): random geometries, 2-4 ranks sharing
the box's one GPU (gloo rendezvous, FASTKV_FUSED=0), random RAGGED shard splits, both orders / poolings, owned-rows and
replicated output, discovered or given shard lengths -- every case against the CPU oracle on the whole prompt (the harness of
tests/test_dist_gpu.py).  usage: python tools/stress_dist.py [N] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

if __name__ == "__main__":
    import test_dist_gpu as T
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
    fails, t0 = 0, time.time()
    for it in range(N):
        world = rng.choice([2, 2, 3, 4])
        W = rng.choice([8, 8, 4, 16]); ks = rng.choice([1, 3, 5, 7, 7])
        Hkv = rng.choice([1, 2, 4, 8]); G = rng.choice([1, 2, 4, 4, 8]); D = rng.choice([64, 128, 128])
        B = rng.choice([1, 1, 2])
        S = rng.randint(world * (W + ks) + 50, 9000)
        # ragged split: random cut points, every shard >= kernel//2 (and > window on the last one)
        while True:
            cuts = sorted(rng.sample(range(1, S), world - 1))
            lens = [b - a for a, b in zip([0] + cuts, cuts + [S])]
            if min(lens) >= max(ks // 2, 1) + 1 and lens[-1] >= W + ks // 2 + 1:
                break
        cap = rng.choice([rng.randint(W + 1, S), min(S, rng.choice([64, 256, 2048])), S])
        cap = max(W + 1, min(cap, S))
        tsp_len = rng.choice([0, rng.randint(W + 1, S - 1)])
        case = dict(seed=7000 + it, B=B, H=Hkv * G, Hkv=Hkv, S=S, D=D, W=W, ks=ks, pooling=rng.choice(["avgpool", "maxpool"]), cap=cap,
                    tsp_len=tsp_len, order=rng.choice(["index", "score"]), replicate=rng.random() < 0.3, discover=rng.random() < 0.3)
        try:
            if rng.random() < 0.3:                               # heads over ranks instead of the sequence
                world = rng.choice([2, 4])
                case["Hkv"] = world * rng.choice([1, 2]); case["H"] = case["Hkv"] * G
                case.pop("replicate"); case.pop("discover")
                lens = "heads"
                T._run(T._tp_worker, world, case, (), timeout=600)
            else:
                T._run(T._sp_worker, world, case, (lens,), timeout=600)
            print(f"it {it}: world {world} lens {lens} {case} OK", flush=True)
        except AssertionError as e:
            fails += 1
            print(f"it {it}: world {world} lens {lens} {case} MISMATCH {str(e)[:600]}", flush=True)
    print(f"{N} cases, {fails} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if fails else 0)
