"""World-size-2/3 gloo tests (CPU) of the sequence-sharded operator: the distributed control flow of fastkv_amd/dist.py
with oracle-backed local stages must reproduce the single-process oracle BIT FOR BIT (scores are position-local, the
softmax statistics are exact integer/float reductions, the candidate all-gather preserves the canonical tie rule)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fastkv_amd.dist as D
        from gen_inputs import make_qkv
        from oracle import fastkv_oracle as O
        from sp_oracle_ops import OracleLocalOps
        O.set_threads(2)
        q, k, v = make_qkv(case["seed"], case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"], full_q=True)
        lo, hi = sum(lens[:rank]), sum(lens[:rank + 1])
        replicate = case.get("replicate", False)
        before = dict(D.COLLECTIVES)
        try:
            out = D.sp_update_kv(k[:, :, lo:hi], q[:, :, lo:hi], v[:, :, lo:hi], window_size=case["W"], kernel_size=case["ks"],
                                 pooling=case["pooling"], capacity=case["cap"], tsp_len=case["tsp_len"], order=case["order"],
                                 local_ops=OracleLocalOps(), shard_lengths=None if case.get("discover") else lens,
                                 replicate=replicate)
        except ValueError as e:
            q_out.put((rank, "ValueError" if case.get("expect_error") else "EXC " + repr(e)))
            return
        ncoll = sum(D.COLLECTIVES.values()) - sum(before.values())
        want = O.update_kv(q, k, v, case["W"], case["ks"], case["pooling"], case["cap"], case["tsp_len"], case["order"])
        ko, vo = out[0], out[1]
        if not replicate:
            # every rank holds only the rows it owns: the ranks' tensors add up (as bit patterns) to the single-process result
            both = torch.stack([ko, vo]).view(torch.int32)
            dist.all_reduce(both, op=dist.ReduceOp.SUM)
            ko, vo = both.view(torch.float16).view(2, *out[0].shape)
        ok = torch.equal(ko, want[0]) and torch.equal(vo, want[1]) and torch.equal(out[3], want[2])
        ok = ok and ((out[2] is None and want[3] is None) or torch.equal(out[2], want[3]))
        ok = ok and ncoll == (5 if replicate else 4)                 # the link budget of fastkv_amd/dist.py, TSP layer or not
        q_out.put((rank, bool(ok) if ok else f"mismatch (collectives {ncoll})"))
    except Exception as e:   # noqa: BLE001
        import traceback
        q_out.put((rank, "EXC " + repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


CASES = [
    # two even shards, avgpool, TSP on, reference row order, replicated output
    (dict(seed=41, B=1, H=8, Hkv=2, S=512, D=128, W=8, ks=7, pooling="avgpool", cap=96, tsp_len=160, order="score", replicate=True),
     [256, 256]),
    # three ragged shards, maxpool (plateaus straddle shard borders), B=2, index order, lengths discovered by the first all-gather
    (dict(seed=42, B=2, H=4, Hkv=2, S=700, D=64, W=8, ks=5, pooling="maxpool", cap=128, tsp_len=0, order="index", discover=True),
     [300, 150, 250]),
    # budget larger than a shard: the last rank contributes fewer candidates than k; rows stay where they are owned
    (dict(seed=43, B=1, H=8, Hkv=1, S=400, D=128, W=8, ks=7, pooling="maxpool", cap=300, tsp_len=350, order="score"), [350, 50]),
    # a middle shard shorter than the window (but not than the halo): no window rows there, nothing special
    (dict(seed=44, B=1, H=8, Hkv=2, S=600, D=128, W=8, ks=7, pooling="avgpool", cap=64, tsp_len=100, order="score"), [290, 5, 305]),
]


def _run(case, lens, timeout=300):
    world = len(lens)
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, lens, q_out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q_out.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    return res


@pytest.mark.parametrize("case,lens", CASES)
def test_sequence_sharded_matches_single_process_oracle(case, lens):
    res = _run(case, lens)
    assert all(r[1] is True for r in res), res


@pytest.mark.parametrize("discover", [False, True])
def test_unusable_split_fails_on_every_rank_together(discover):
    """A shard shorter than the pooling halo (or a last shard without the whole window): every rank raises the same
    ValueError -- before any collective when the lengths are given, right after the first one when they are discovered --
    instead of one rank failing late and its peers waiting in a collective for ever."""
    case = dict(seed=45, B=1, H=8, Hkv=2, S=400, D=128, W=8, ks=7, pooling="avgpool", cap=64, tsp_len=0, order="score",
                discover=discover, expect_error=True)
    res = _run(case, [200, 2, 198], timeout=120)
    assert all(r[1] == "ValueError" for r in res), res
    res = _run(case, [391, 9], timeout=120)
    assert all(r[1] == "ValueError" for r in res), res


def _tp_worker(rank, world, port, case, q_out):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastkv_amd.dist import tp_update_kv
        from gen_inputs import make_qkv
        from oracle import fastkv_oracle as O
        from sp_oracle_ops import OracleTPOps
        O.set_threads(2)
        q, k, v = make_qkv(case["seed"], case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"])
        hl, G = case["Hkv"] // world, case["H"] // case["Hkv"]
        ks, qs = slice(rank * hl, (rank + 1) * hl), slice(rank * hl * G, (rank + 1) * hl * G)
        out = tp_update_kv(k[:, ks], q[:, qs], v[:, ks], window_size=case["W"], kernel_size=case["ks"], pooling=case["pooling"],
                           capacity=case["cap"], tsp_len=case["tsp_len"], order=case["order"], local_ops=OracleTPOps())
        want = O.update_kv(q, k, v, case["W"], case["ks"], case["pooling"], case["cap"], case["tsp_len"], case["order"])
        ok = torch.equal(out[0], want[0][:, ks]) and torch.equal(out[1], want[1][:, ks]) and torch.equal(out[3], want[2][:, ks])
        ok = ok and ((out[2] is None and want[3] is None) or torch.equal(out[2], want[3]))
        q_out.put((rank, bool(ok)))
    except Exception as e:   # noqa: BLE001
        import traceback
        q_out.put((rank, "EXC " + repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [
    (dict(seed=51, B=1, H=8, Hkv=2, S=600, D=128, W=8, ks=7, pooling="maxpool", cap=96, tsp_len=200, order="score"), 2),
    (dict(seed=52, B=2, H=12, Hkv=3, S=400, D=64, W=8, ks=5, pooling="avgpool", cap=64, tsp_len=0, order="index"), 3),
])
def test_head_sharded_matches_single_process_oracle(case, world):
    """Tensor-parallel operator (one slice of the KV heads per rank): local K/V and indices equal the single-process
    oracle's slices; the TSP index (sum over ALL heads) is identical on every rank and equal to the oracle's."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_tp_worker, args=(r, world, port, case, q_out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q_out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] is True for r in res), res
