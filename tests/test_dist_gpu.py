"""Sequence-sharded operator on the MI355X: two ranks share the one GPU of the test box (gloo rendezvous, collectives
staged through the host), every local stage runs through the C ABI (`fastkv_sp_*`).  Result must be bit-identical to the
fused single-GPU operator."""
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


def _worker(rank, world, port, case, lens, q_out):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    # two ranks share ONE GPU here, which include/fastkv_hip.h rules out for the fused scoring kernel (all its workgroups
    # must be resident): the ranks take the staged kernels
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FASTKV_FUSED="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastkv_amd import ops
        from fastkv_amd.dist import sp_update_kv
        from gen_inputs import make_qkv
        dev = torch.device("cuda:0")
        q, k, v = make_qkv(case["seed"], case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"], full_q=True)
        qd, kd, vd = (t.transpose(1, 2).contiguous().to(dev).transpose(1, 2) for t in (q, k, v))
        lo, hi = sum(lens[:rank]), sum(lens[:rank + 1])
        out = sp_update_kv(kd[:, :, lo:hi], qd[:, :, lo:hi], vd[:, :, lo:hi], window_size=case["W"], kernel_size=case["ks"],
                           pooling=case["pooling"], capacity=case["cap"], tsp_len=case["tsp_len"], order=case["order"],
                           shard_lengths=lens)
        want = ops.update_kv(qd, kd, vd, case["W"], case["ks"], case["pooling"], case["cap"], case["tsp_len"], case["order"],
                             return_indices=True)
        torch.cuda.synchronize()
        ok = torch.equal(out[0], want[0]) and torch.equal(out[1], want[1]) and torch.equal(out[3], want[3])
        ok = ok and ((out[2] is None and want[2] is None) or torch.equal(out[2], want[2]))
        q_out.put((rank, bool(ok)))
    except Exception as e:   # noqa: BLE001
        import traceback
        q_out.put((rank, "EXC " + repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


CASES = [
    (dict(seed=51, B=1, H=32, Hkv=8, S=4096, D=128, W=8, ks=7, pooling="maxpool", cap=512, tsp_len=2048, order="score"), [2048, 2048]),
    (dict(seed=52, B=2, H=8, Hkv=2, S=1000, D=128, W=8, ks=5, pooling="avgpool", cap=128, tsp_len=0, order="index"), [333, 667]),
    (dict(seed=53, B=1, H=4, Hkv=4, S=777, D=64, W=8, ks=7, pooling="avgpool", cap=100, tsp_len=300, order="score"), [700, 77]),
]


@pytest.mark.parametrize("case,lens", CASES)
def test_two_ranks_on_one_gpu_match_fused_operator(case, lens):
    world = len(lens)
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, lens, q_out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q_out.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] is True for r in res), res


def _tp_worker(rank, world, port, case, q_out):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FASTKV_FUSED="0")      # see _worker
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastkv_amd import ops
        from fastkv_amd.dist import tp_update_kv
        from gen_inputs import make_qkv
        dev = torch.device("cuda:0")
        q, k, v = make_qkv(case["seed"], case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"])
        qd, kd, vd = (t.transpose(1, 2).contiguous().to(dev).transpose(1, 2) for t in (q, k, v))
        hl, G = case["Hkv"] // world, case["H"] // case["Hkv"]
        ks, qs = slice(rank * hl, (rank + 1) * hl), slice(rank * hl * G, (rank + 1) * hl * G)
        out = tp_update_kv(kd[:, ks], qd[:, qs], vd[:, ks], window_size=case["W"], kernel_size=case["ks"], pooling=case["pooling"],
                           capacity=case["cap"], tsp_len=case["tsp_len"], order=case["order"])
        want = ops.update_kv(qd, kd, vd, case["W"], case["ks"], case["pooling"], case["cap"], case["tsp_len"], case["order"],
                             return_indices=True)
        torch.cuda.synchronize()
        ok = torch.equal(out[0], want[0][:, ks]) and torch.equal(out[1], want[1][:, ks]) and torch.equal(out[3], want[3][:, ks])
        ok = ok and ((out[2] is None and want[2] is None) or torch.equal(out[2], want[2]))
        q_out.put((rank, bool(ok)))
    except Exception as e:   # noqa: BLE001
        import traceback
        q_out.put((rank, "EXC " + repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_head_sharded_on_gpu_matches_whole_operator():
    """tp_update_kv (KV heads split over 2 ranks that share the test GPU; the score-row all-gather is staged through the
    host): local K/V/indices equal the whole operator's head slices, tsp_idx equals its TSP index."""
    case = dict(seed=61, B=1, H=32, Hkv=8, S=4096, D=128, W=8, ks=7, pooling="maxpool", cap=512, tsp_len=2048, order="score")
    port = _free_port()
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, case, q_out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q_out.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] is True for r in res), res
