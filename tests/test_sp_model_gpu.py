"""Sequence-parallel prefill of the patched model on the MI355X (fastkv_amd/sp_model.py): two layers of the
Llama-3-8B geometry, ONE 4096-token prompt over ranks that share the test box's GPU (gloo rendezvous, collectives staged
through the host, FASTKV_FUSED=0 because the ranks share the device).

Checked exactly: what the sharded attention module fed to the sequence-sharded operator (the ranks' q/k/v shards, captured and
concatenated) goes through the CPU ORACLE, and the operator's outputs -- per-head indices, TSP index, the rows that went into
the cache -- must be identical.  Checked to fp16 tolerance: the last-token logits against the single-process patched model
(the ranks' GEMMs run at other shapes than the single-process ones, so their fp16 outputs differ in the last bit here and
there; the selection may then differ in a few positions -- the overlap of the TSP sets is asserted >= 98 %)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lens, q_out):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FASTKV_FUSED="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fastkv_amd.dist as D
        from baselines.monkeypatch import replace_llama, set_model
        from benchmark import prefill
        from fastkv_amd.sp_model import SPContext, sp_prefill
        from oracle import fastkv_oracle as O
        S = sum(lens)

        def build():
            a = prefill.parse_args(["--model_path", "llama3-8b", "--num_layers", "2", "--device", "cuda", "--save_txt", "", "--method",
                                    "fastkv", "--max_capacity_prompts", "512", "--tsp_len", "2048", "--tsp_idx", "0", "--pooling", "maxpool"])
            a.save_txt = False
            a.context_lengths = [S]
            replace_llama("fastkv")
            torch.manual_seed(41)
            m = prefill.build_model(a, "cuda")
            set_model(m, a)
            return m

        ids = torch.randint(0, 1000, (1, S), generator=torch.Generator().manual_seed(43)).cuda()
        model = build()
        with torch.no_grad():
            ref = model(ids, attention_mask=torch.ones_like(ids))
        ref_tsp = model.model.layers[0].self_attn.tsp_idx.cpu()
        ref_logits = ref.logits.float().cpu()
        del ref
        # spy on the sequence-sharded operator inside the attention module
        captured = []
        real = D.sp_update_kv

        def spy(k, q, v, **kw):
            out = real(k, q, v, **kw)
            captured.append(((k.cpu(), q.cpu(), v.cpu()), kw, tuple(None if t is None else t.cpu() for t in out)))
            return out

        D.sp_update_kv = spy
        lo, hi = sum(lens[:rank]), sum(lens[:rank + 1])
        ctx = SPContext(shard_lengths=lens, replicate=True)
        with torch.no_grad():
            out = sp_prefill(model, ids[:, lo:hi], ctx)
        torch.cuda.synchronize()
        D.sp_update_kv = real
        msg = []
        assert len(captured) == 1                                       # layer 0 is the only sharded layer (it is the TSP layer)
        (k, q, v), kw, (ko, vo, tsp, kv_idx) = captured[0]
        # concatenate the ranks' shards (gloo, CPU) and replay the whole prompt through the oracle
        def cat(t):                                                     # equal shards in this test
            t = t.contiguous()
            parts = [torch.empty_like(t) for _ in lens]
            dist.all_gather(parts, t)
            return torch.cat(parts, dim=2)

        kf, qf, vf = cat(k), cat(q), cat(v)
        want = O.update_kv(qf, kf, vf, kw["window_size"], kw["kernel_size"], kw["pooling"], kw["capacity"], kw["tsp_len"], kw["order"])
        if not (torch.equal(ko, want[0]) and torch.equal(vo, want[1]) and torch.equal(kv_idx, want[2]) and torch.equal(tsp, want[3])):
            msg.append("sequence-sharded operator inside the model differs from the oracle on the captured inputs")
        cache0 = out.past_key_values.layers[0]
        if not (torch.equal(cache0.keys.cpu(), want[0]) and torch.equal(cache0.values.cpu(), want[1])):
            msg.append("layer-0 cache rows differ from the oracle's")
        overlap = len(set(tsp[0].tolist()) & set(ref_tsp[0].tolist())) / ref_tsp.shape[1]
        if overlap < 0.98:
            msg.append(f"TSP sets of the sharded and the single-process run overlap only {overlap:.3f}")
        lg = out.logits.float().cpu()
        scale = float(ref_logits.abs().max())
        if not bool(torch.isfinite(lg).all()) or float((lg - ref_logits).abs().max()) > 0.15 * scale:
            msg.append(f"logits differ by {float((lg - ref_logits).abs().max()):.3e} (scale {scale:.3e})")
        if out.past_key_values.layers[1].keys.shape != (1, 8, 512, 128):
            msg.append("layer-1 (replicated) cache has the wrong shape")
        q_out.put((rank, True if not msg else "; ".join(msg)))
    except Exception as e:   # noqa: BLE001
        import traceback
        q_out.put((rank, "EXC " + repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lens", [[2048, 2048], [1024, 1024, 1024, 1024]])
def test_sequence_parallel_prefill_on_gpu(lens):
    world = len(lens)
    ctx = mp.get_context("spawn")
    for attempt in range(2):
        q_out = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, lens, q_out)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q_out.get(timeout=900) for _ in range(world)]
        for p in procs:
            p.join(timeout=120)
        # one more try on a fresh port when a rank could not rendezvous (test rig); a wrong result is never retried
        rig = [r for r in res if isinstance(r[1], str) and r[1].startswith("EXC") and
               any(t in r[1] for t in ("Address already in use", "Connection refused", "Connection reset", "connectFullMesh", "Broken pipe"))]
        if not rig or attempt == 1:
            break
    assert all(r[1] is True for r in res), res
