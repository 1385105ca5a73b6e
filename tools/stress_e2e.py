"""One-off stress run for the model-level decode path (not part of the suite): ONE two-layer model of the Llama-3-8B geometry,
several prompts of random lengths and budgets one after the other; for each, greedy decode (a) eagerly over DynamicCache with
SDPA and (b) over the slab cache with the HIP step kernels captured in a graph -- compared token by token / to fp16 tolerance.
Exercises the reuse of workspaces, arrival counters and graph pools across prompts.  usage: python tools/stress_e2e.py [N] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from baselines.monkeypatch import replace_llama, set_model
from benchmark import e2e, prefill

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
fails = 0


def build(slab, cap, tsp_len, S):
    os.environ["FASTKV_SLAB_CACHE"] = slab
    a = prefill.parse_args(["--model_path", "llama3-8b", "--num_layers", "2", "--device", "cuda", "--save_txt", "", "--method", "fastkv",
                            "--max_capacity_prompts", str(cap), "--tsp_len", str(tsp_len), "--tsp_idx", "0"])
    a.save_txt = False
    a.context_lengths = [S]
    replace_llama("fastkv")
    torch.manual_seed(3)
    m = prefill.build_model(a, "cuda")
    set_model(m, a)
    return m


for it in range(N):
    S = rng.randint(300, 6000)
    cap = rng.choice([64, 128, 512, min(2048, S - 8)])
    cap = max(16, min(cap, S - 8))
    tsp_len = max(cap, min(S - 8, rng.choice([256, 1024, 2048])))
    steps = rng.randint(3, 12)
    ids = torch.randint(0, 1000, (1, S), generator=torch.Generator().manual_seed(100 + it)).cuda()
    toks, logs = {}, {}
    for mode in ("0", "1"):
        model = build(mode, cap, tsp_len, S + 64)
        with torch.no_grad():
            out = model(ids, attention_mask=torch.ones_like(ids))
            pkv = out.past_key_values
            first = out.logits[:, -1].argmax(-1, keepdim=True)
            if mode == "0":
                tok, ts, lg = first, [], []
                for _ in range(steps):
                    o = model(input_ids=tok, past_key_values=pkv)
                    lg.append(o.logits[:, -1].float().cpu())
                    tok = o.logits[:, -1].argmax(-1, keepdim=True)
                    ts.append(int(tok[0, 0]))
                toks[mode], logs[mode], first0 = ts, lg, first
            else:
                def timed(fn):
                    fn(); torch.cuda.synchronize(); return 0.0, None
                _, ts = e2e.graph_decode(model, pkv, first, steps, timed)
                toks[mode] = ts
                ok_first = bool(torch.equal(first, first0))
        del model, pkv
        torch.cuda.empty_cache()
    agree = sum(int(x == y) for x, y in zip(toks["0"], toks["1"]))
    ok = ok_first and toks["1"][0] == toks["0"][0] and agree >= steps - 2
    print(f"it {it}: S={S} cap={cap} tsp_len={tsp_len} steps={steps} first token equal {ok_first} tokens agree {agree}/{steps}", "OK" if ok else "MISMATCH", flush=True)
    fails += 0 if ok else 1
print(f"{N} prompts, {fails} mismatches")
sys.exit(1 if fails else 0)
