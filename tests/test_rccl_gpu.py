"""RCCL itself (`torch.distributed` backend "nccl" on ROCm) under every collective of the sharded paths, with the ONE rank a
one-GPU box allows (VERDICT r03 missing #2 / next #4: three rounds of multi-GPU code had only ever met gloo's host
implementation).

World size 1 cannot show scaling and moves no bytes over xGMI; what it does prove on hardware: every `nccl` branch of
fastkv_amd/dist.py and fastkv_amd/sp_model.py executes on DEVICE tensors (no host staging: `_host_staged` is false), RCCL accepts
every dtype / shape / split list those branches hand it -- fp32 MAX over row maxima + NaN flags, int64 SUM of the fixed-point row
sums, int64 candidate all-gather, the fp16 packet all-gather, int32-word SUM of fp16 rows (`replicate`, `tsp_assemble`),
`all_to_all_single` with explicit split lists, K/V `all_gather_into_tensor`, broadcast -- stream-ordered against the HIP kernels
around them, and the results equal the ORACLE's.

NaN under ReduceOp.MAX (the audit VERDICT asked for): no NaN ever reaches the MAX reduction.  `fastkv_sp_rowmax_f16` takes the row
maximum with `fmaxf`, which drops NaNs, and reports a NaN in the row as a separate 0 / 1 float behind the maxima (csrc/score.hip
`row_stats_kernel` mode 1); MAX over {0, 1} flags and over finite-or-infinite maxima is the same in every implementation.  The
case with NaN / Inf in K and in a window row of Q below goes through that path."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q_out):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    msg = []
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        assert dist.get_backend() == "nccl"
        import fastkv_amd.dist as D
        from fastkv_amd import sp_model
        from gen_inputs import make_qkv
        from oracle import fastkv_oracle as O
        dev = torch.device("cuda:0")

        # what RCCL is handed: every collective of the two modules, with dtype and element count
        seen = []
        for name in ("all_reduce", "all_gather_into_tensor", "all_to_all_single", "broadcast"):
            real = getattr(dist, name)

            def spy(*a, _real=real, _name=name, **kw):
                t = a[1] if _name == "all_gather_into_tensor" else a[0]
                assert t.is_cuda, f"{_name} was handed a host tensor under the nccl backend"
                seen.append((_name, str(t.dtype).replace("torch.", ""), str(kw.get("op", ""))))
                return _real(*a, **kw)

            setattr(dist, name, spy)

        def on_dev(t):
            return t.transpose(1, 2).contiguous().to(dev).transpose(1, 2)

        # ---- (1) the sequence-sharded operator: 4 collectives (5 with replicate), lengths given or discovered, batch 2, special values
        cases = [
            dict(seed=51, B=1, H=32, Hkv=8, S=4096, D=128, W=8, ks=7, pooling="maxpool", cap=512, tsp_len=2048, order="score"),
            dict(seed=52, B=2, H=8, Hkv=2, S=1000, D=128, W=8, ks=5, pooling="avgpool", cap=128, tsp_len=0, order="index", discover=True),
            dict(seed=53, B=1, H=4, Hkv=4, S=777, D=64, W=8, ks=7, pooling="avgpool", cap=100, tsp_len=300, order="score", replicate=True),
            dict(seed=54, B=1, H=16, Hkv=4, S=3000, D=128, W=8, ks=7, pooling="maxpool", cap=256, tsp_len=1000, order="score", special=True),
        ]
        for c in cases:
            q, k, v = make_qkv(c["seed"], c["B"], c["H"], c["Hkv"], c["S"], c["D"], c["W"], full_q=True)
            if c.get("special"):
                q, k = q.clone(), k.clone()
                k[0, 1, 100, 5] = float("nan")                              # a NaN key: head 1's rows of that group turn NaN
                k[0, 2, 200, 7] = float("inf")
                q[0, 0, c["S"] - 3, 9] = float("nan")                        # a NaN in a window row of query head 0
                k[0, 3, 17, 1] = 60000.0
            want = O.update_kv(q, k, v, c["W"], c["ks"], c["pooling"], c["cap"], c["tsp_len"], c["order"])
            before = sum(D.COLLECTIVES.values())
            out = D.sp_update_kv(on_dev(k), on_dev(q), on_dev(v), window_size=c["W"], kernel_size=c["ks"], pooling=c["pooling"],
                                 capacity=c["cap"], tsp_len=c["tsp_len"], order=c["order"],
                                 shard_lengths=None if c.get("discover") else [c["S"]], replicate=c.get("replicate", False))
            torch.cuda.synchronize()
            ncoll = sum(D.COLLECTIVES.values()) - before
            ok = torch.equal(out[0].cpu().view(torch.int16), want[0].view(torch.int16)) and \
                torch.equal(out[1].cpu().view(torch.int16), want[1].view(torch.int16)) and torch.equal(out[3].cpu(), want[2])
            ok = ok and ((out[2] is None and want[3] is None) or torch.equal(out[2].cpu(), want[3]))
            if not ok:
                msg.append(f"sp_update_kv over nccl differs from the oracle: {c}")
            if ncoll != (5 if c.get("replicate") else 4):
                msg.append(f"sp_update_kv issued {ncoll} collectives: {c}")

        # ---- (2) the head-sharded operator on the TSP layer: one all-gather of fp16 score rows
        c = dict(seed=61, B=1, H=32, Hkv=8, S=4096, D=128, W=8, ks=7, pooling="maxpool", cap=512, tsp_len=2048, order="score")
        q, k, v = make_qkv(c["seed"], c["B"], c["H"], c["Hkv"], c["S"], c["D"], c["W"], full_q=True)
        want = O.update_kv(q, k, v, c["W"], c["ks"], c["pooling"], c["cap"], c["tsp_len"], c["order"])
        out = D.tp_update_kv(on_dev(k), on_dev(q), on_dev(v), window_size=c["W"], kernel_size=c["ks"], pooling=c["pooling"],
                             capacity=c["cap"], tsp_len=c["tsp_len"], order=c["order"])
        torch.cuda.synchronize()
        if not (torch.equal(out[0].cpu(), want[0]) and torch.equal(out[1].cpu(), want[1]) and torch.equal(out[3].cpu(), want[2])
                and torch.equal(out[2].cpu(), want[3])):
            msg.append("tp_update_kv over nccl differs from the oracle")

        # ---- (3) one sharded layer (+ one replicated layer behind the TSP layer) of the sequence-parallel model, both layouts
        from baselines.monkeypatch import replace_llama, set_model
        from benchmark import prefill
        S = 4096
        a = prefill.parse_args(["--model_path", "llama3-8b", "--num_layers", "2", "--device", "cuda", "--save_txt", "", "--method",
                                "fastkv", "--max_capacity_prompts", "512", "--tsp_len", "2048", "--tsp_idx", "0", "--pooling", "maxpool"])
        a.save_txt = False
        a.context_lengths = [S]
        replace_llama("fastkv")
        torch.manual_seed(41)
        model = prefill.build_model(a, "cuda")
        set_model(model, a)
        ids = torch.randint(0, 1000, (1, S), generator=torch.Generator().manual_seed(43)).cuda()
        with torch.no_grad():
            ref = model(ids, attention_mask=torch.ones_like(ids))
        ref_tsp = model.model.layers[0].self_attn.tsp_idx.cpu()
        ref_logits = ref.logits.float().cpu()
        ref_cache = [(l.keys.cpu(), l.values.cpu()) for l in ref.past_key_values.layers]
        del ref
        for mode, opname in (("heads", "tp_update_kv"), ("gather", "sp_update_kv")):
            captured = []
            real = getattr(D, opname)

            def spy_op(k_, q_, v_, _real=real, **kw):
                o = _real(k_, q_, v_, **kw)
                captured.append(((k_.cpu(), q_.cpu(), v_.cpu()), kw, tuple(None if t is None else t.cpu() for t in o)))
                return o

            setattr(D, opname, spy_op)
            try:
                ctx = sp_model.SPContext(shard_lengths=[S], replicate=True, mode=mode)
                a2a0 = sp_model.COLLECTIVES["all_to_all"]
                with torch.no_grad():
                    out = sp_model.sp_prefill(model, ids, ctx)
                torch.cuda.synchronize()
            finally:
                setattr(D, opname, real)
            if len(captured) != 1:
                msg.append(f"{mode}: {len(captured)} sharded operator calls (1 expected)")
                continue
            if mode == "heads" and sp_model.COLLECTIVES["all_to_all"] - a2a0 != 2:
                msg.append(f"heads: {sp_model.COLLECTIVES['all_to_all'] - a2a0} all-to-alls for the one sharded layer")
            (k_, q_, v_), kw, (ko, vo, tsp, kv_idx) = captured[0]
            want = O.update_kv(q_, k_, v_, kw["window_size"], kw["kernel_size"], kw["pooling"], kw["capacity"], kw["tsp_len"], kw["order"])
            if not (torch.equal(ko, want[0]) and torch.equal(vo, want[1]) and torch.equal(kv_idx, want[2]) and torch.equal(tsp, want[3])):
                msg.append(f"{mode}: the sharded operator inside the model differs from the oracle on the captured inputs")
            c0 = out.past_key_values.layers[0]
            if not (torch.equal(c0.keys.cpu(), want[0]) and torch.equal(c0.values.cpu(), want[1])):
                msg.append(f"{mode}: layer-0 cache rows differ from the oracle's")
            # one rank holds the whole prompt: same GEMM shapes as the single-process model, so the comparison can be strict
            overlap = len(set(tsp[0].tolist()) & set(ref_tsp[0].tolist())) / ref_tsp.shape[1]
            lg = out.logits.float().cpu()
            scale = float(ref_logits.abs().max())
            if overlap < 0.98 or not bool(torch.isfinite(lg).all()) or float((lg - ref_logits).abs().max()) > 2e-2 * scale:
                msg.append(f"{mode}: TSP overlap {overlap:.3f}, logits differ by {float((lg - ref_logits).abs().max()):.3e} (scale {scale:.3e})")
            if out.past_key_values.layers[1].keys.shape != ref_cache[1][0].shape:
                msg.append(f"{mode}: layer-1 (replicated) cache has the wrong shape")
        # ---- (4) no TSP reduction (short prompt): the last-token broadcast branch of the model loop
        ctx = sp_model.SPContext(shard_lengths=[300], replicate=True, mode="gather")
        with torch.no_grad():
            out = sp_model.sp_prefill(model, ids[:, :300], ctx)
        torch.cuda.synchronize()
        if out.logits.shape[:2] != (1, 1) or not bool(torch.isfinite(out.logits).all()):
            msg.append("short prompt (no TSP reduction): bad logits")

        kinds = {(n, d, o.split(".")[-1]) for n, d, o in seen}
        need = {("all_reduce", "float32", "MAX"), ("all_reduce", "int64", "SUM"), ("all_reduce", "int32", "SUM"),
                ("all_gather_into_tensor", "float16", ""), ("all_gather_into_tensor", "int64", ""), ("all_to_all_single", "float16", ""),
                ("broadcast", "float16", "")}
        if not need <= kinds:
            msg.append(f"collectives that never reached RCCL: {sorted(need - kinds)} (seen {sorted(kinds)})")
        q_out.put(("ok" if not msg else "; ".join(msg), sorted(kinds), len(seen)))
    except Exception as e:   # noqa: BLE001
        import traceback
        q_out.put(("EXC " + repr(e) + traceback.format_exc(), [], 0))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:   # noqa: BLE001
            pass


def test_every_collective_of_the_sharded_paths_runs_through_rccl():
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q_out))
    p.start()
    res, kinds, n = q_out.get(timeout=900)
    p.join(timeout=120)
    print(f"RCCL (1 rank): {n} collectives, kinds {kinds}")
    assert res == "ok", res
