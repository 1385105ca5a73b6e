"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs, and
against the golden vectors captured from the reference.  Bit-exact: scores, indices, K/V rows."""
import os

import numpy as np
import pytest
import torch

from gen_inputs import make_qkv
from golden_cases import CASES
from helpers import CONTRACTION_OF_ENGINE, assert_score_parity, default_contraction, expected_kv, f16_from_bits, load_golden, ulp_diff

pytestmark = pytest.mark.gpu

SMALL = [c for c in CASES if CASES[c]["S"] <= 4096]
BIG = [c for c in CASES if CASES[c]["S"] > 4096]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _to_dev(t, dev):
    # keep the [B,S,H,D]-physical / [B,H,S,D]-logical layout (llama_model.py:117-122)
    return t.transpose(1, 2).contiguous().to(dev).transpose(1, 2)


def _run_both(name, dev, order, engine="auto"):
    """Oracle and HIP operator on golden case `name`, both under the contraction contract `engine` computes ("auto": the default
    of both sides)."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    case = CASES[name]
    O.set_contraction(CONTRACTION_OF_ENGINE.get(engine) or default_contraction())
    ops.set_score_engine(engine)
    q, k, v = make_qkv(case["seed"], case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"],
                       peaked=case.get("peaked", 0))
    ref = O.update_kv(q, k, v, case["W"], case["ks"], case["pooling"], case["cap"], case["tsp_len"], order, return_scores=True)
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
    assert kd.stride() == k.stride()
    got = ops.update_kv(qd, kd, vd, case["W"], case["ks"], case["pooling"], case["cap"], case["tsp_len"], order,
                        return_indices=True, return_scores=True)
    torch.cuda.synchronize()
    return case, (q, k, v), ref, got


@pytest.mark.parametrize("order", ["index", "score"])
@pytest.mark.parametrize("name", SMALL + BIG)
def test_update_kv_bit_exact_vs_oracle(name, order, dev):
    case, (q, k, v), (ko, vo, idx, tsp, c, t), (gko, gvo, gtsp, gidx, gc) = _run_both(name, dev, order)
    assert torch.equal(gc.cpu().view(torch.int16), c.view(torch.int16)), "scores differ from the oracle (bit patterns)"
    assert torch.equal(gidx.cpu(), idx), "per-head indices differ from the oracle"
    assert torch.equal(gko.cpu(), ko) and torch.equal(gvo.cpu(), vo), "compacted K/V differ"
    assert gko.is_contiguous() and gko.dtype == torch.float16 and list(gko.shape) == list(ko.shape)
    if case["tsp_len"]:
        assert gtsp.dtype == torch.int64 and torch.equal(gtsp.cpu(), tsp)
    else:
        assert gtsp is None


@pytest.mark.parametrize("name", SMALL + BIG)
def test_update_kv_vs_reference_golden(name, dev):
    """Directly against what the reference produced: canonical top-k of ITS scores, its TSP index, score ulps."""
    case, (q, k, v), _, (gko, gvo, gtsp, gidx, gc) = _run_both(name, dev, "index")
    g = load_golden(name)
    assert torch.equal(gidx.cpu(), torch.from_numpy(g["idx_canonical"].astype(np.int64)))
    if case["tsp_len"]:
        assert torch.equal(gtsp.cpu(), torch.from_numpy(g["tsp_canonical"].astype(np.int64)))
    if "c_ref" in g:
        assert_score_parity(gc.cpu(), f16_from_bits(g["c_ref"]), default_contraction(), name)
    # compressed KV values vs the reference rows, canonical order: exact (tolerance 1e-3 in the north star is slack)
    want = expected_kv(k, torch.from_numpy(g["idx_canonical"].astype(np.int64)), case["W"])
    assert torch.allclose(gko.cpu().float(), want.float(), atol=1e-3, rtol=0) and torch.equal(gko.cpu(), want)


def test_stage_apis_and_properties(dev):
    """score / select / compact separately, plus size-independent properties of the selection."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    q, k, v = make_qkv(21, 1, 8, 2, 3000, 128, 8)
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
    c, t = ops.scores(qd, kd, 8, 7, "maxpool")
    co, to, _ = O.scores(q, k, 8, 7, "maxpool")
    assert torch.equal(c.cpu().view(torch.int16), co.view(torch.int16)) and torch.equal(t.cpu().view(torch.int16), to.view(torch.int16))
    n = c.shape[-1]
    for kk in (1, 5, 1024, 1025, n - 1, n):
        for order in ("index", "score"):
            idx = ops.select(c[0], kk, order).cpu()
            for g in range(2):
                assert torch.equal(idx[g], O.canonical_topk(co[0, g].contiguous(), kk, order)), (kk, order)
    # k == n is a permutation; index order is sorted; idempotence on the selected scores
    full = ops.select(c[0], n, "score").cpu()
    assert torch.equal(torch.sort(full, dim=1).values, torch.arange(n).expand(2, -1))
    idx = ops.select(c[0], 500, "index")
    assert bool((idx[:, 1:] > idx[:, :-1]).all())
    sub = torch.gather(c[0], 1, idx)
    assert torch.equal(torch.sort(ops.select(sub, 500, "index"), dim=1).values.cpu(), torch.arange(500).expand(2, -1))
    # append = window union of the TSP index (utils.py:128-129)
    ti = ops.select(t, 100, "index", append=8).cpu()
    assert torch.equal(ti[0, -8:], torch.arange(n, n + 8))
    # compact with explicit indices
    idx3 = ops.select(c[0], 200, "score")[None]
    ko, vo = ops.compact(kd, vd, idx3.contiguous(), 8)
    assert torch.equal(ko.cpu(), expected_kv(k, idx3.cpu(), 8)) and torch.equal(vo.cpu(), expected_kv(v, idx3.cpu(), 8))
    # stage 3 in the reference's row order from the ASCENDING winners + their score rows (fastkv_compact_ranked_f16): equals the
    # explicit score-ordered gather, ties by position included (maxpool plateaus)
    for kk in (1, 200, 1000):
        asc = ops.select(c[0], kk, "index")[None].contiguous()
        srt = ops.select(c[0], kk, "score")[None].contiguous()
        ko2, vo2, got = ops.compact(kd, vd, asc, 8, scores=c, return_sorted=True)
        assert torch.equal(got.cpu(), srt.cpu()), kk
        assert torch.equal(ko2.cpu(), expected_kv(k, srt.cpu(), 8)) and torch.equal(vo2.cpu(), expected_kv(v, srt.cpu(), 8))


def test_degenerate_rows(dev):
    """All-equal scores (the reference benchmark's all-ones prompt) and heavy ties: canonical = lowest positions."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    z = torch.zeros(3, 40000, dtype=torch.float16, device=dev)
    idx = ops.select(z, 2040, "score").cpu()
    assert torch.equal(idx, torch.arange(2040).expand(3, -1))
    r = (torch.arange(40000) % 7).to(torch.float16)[None].to(dev)
    for order in ("index", "score"):
        got = ops.select(r, 9000, order).cpu()[0]
        assert torch.equal(got, O.canonical_topk(r[0].cpu().contiguous(), 9000, order))
    # identical K rows (all-ones prompt): every key scores the same -> first k positions + window
    q, k, v = make_qkv(5, 1, 8, 2, 2048, 128, 8)
    k = k.clone(); k[:] = k[:, :, :1]
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
    ko, vo, tsp, kvi = ops.update_kv(qd, kd, vd, 8, 7, "avgpool", 256, 512, "score", return_indices=True)
    want = O.update_kv(q, k, v, 8, 7, "avgpool", 256, 512, "score")
    assert torch.equal(kvi.cpu(), want[2]) and torch.equal(tsp.cpu(), want[3])
    # the same on rows long enough for the chunked selection (several workgroups per row): ties in every chunk, the
    # quota of equal values runs out in the middle of one; K rows periodic in the position -> a handful of distinct scores
    for period, cap, tsp_len in ((1, 2048, 1024), (5, 3000, 4100), (64, 1500, 0)):
        q, k, v = make_qkv(6, 1, 8, 2, 10000, 128, 8)
        k = k.clone(); k[:, :, :9992] = k[:, :, :period].repeat(1, 1, 9992 // period + 1, 1)[:, :, :9992]
        qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
        for order in ("index", "score"):
            ko, vo, tsp, kvi = ops.update_kv(qd, kd, vd, 8, 7, "maxpool", cap, tsp_len, order, return_indices=True)
            want = O.update_kv(q, k, v, 8, 7, "maxpool", cap, tsp_len, order)
            assert torch.equal(kvi.cpu(), want[2]), (period, order)
            assert torch.equal(ko.cpu(), want[0]) and torch.equal(vo.cpu(), want[1])
            if tsp_len:
                assert torch.equal(tsp.cpu(), want[3])


def test_gather_rows_tsp_propagation(dev):
    """hidden.gather(1, tsp_idx[...,None].expand(..)) of llama_model.py:255-257."""
    from fastkv_amd import ops
    h = torch.randn(2, 700, 4096, dtype=torch.float16, device=dev)
    idx = torch.stack([torch.sort(torch.randperm(700, device=dev)[:333]).values for _ in range(2)])
    out = ops.gather_rows(h, idx)
    assert torch.equal(out, torch.gather(h, 1, idx[..., None].expand(-1, -1, 4096)))


def test_tsp_propagate_is_both_gathers_of_the_decoder_layer(dev):
    """fastkv_tsp_propagate = `position_ids.gather(1, tsp_idx)` + `hidden.gather(1, tsp_idx[..., None].expand(...))` of
    llama_model.py:254-257 in one launch: both outputs against torch, batched and broadcast position ids, odd hidden sizes; and an
    index outside the prompt reads the clamped row instead of faulting (what the index tensor of a REPORTED call may hold)."""
    from fastkv_amd import ops
    for B, S, hid, k in ((2, 700, 4096, 333), (1, 32768, 4096, 2048), (3, 100, 64, 100), (1, 9, 8, 4)):
        h = torch.randn(B, S, hid, dtype=torch.float16, device=dev)
        idx = torch.stack([torch.sort(torch.randperm(S, device=dev)[:k]).values for _ in range(B)])
        for pos in (torch.arange(S, device=dev)[None].expand(B, -1), (torch.arange(S, device=dev) * 3 + 5)[None].repeat(B, 1), torch.arange(S, device=dev)[None]):
            if pos.shape[0] != B and B > 1:
                continue                                                   # (torch.gather itself needs the batch to match)
            out, npos = ops.tsp_propagate(h, pos, idx)
            assert torch.equal(out, torch.gather(h, 1, idx[..., None].expand(-1, -1, hid)))
            assert torch.equal(npos, torch.gather(pos, 1, idx)) and npos.is_contiguous() and npos.shape == idx.shape
    h = torch.randn(1, 64, 4096, dtype=torch.float16, device=dev)
    out, npos = ops.tsp_propagate(h, torch.arange(64, device=dev)[None], torch.tensor([[3, 64, -5, 10 ** 12]], device=dev))
    assert torch.equal(out[0, 0], h[0, 3]) and torch.equal(out[0, 1], h[0, 63]) and torch.equal(out[0, 2], h[0, 0]) and torch.equal(out[0, 3], h[0, 63])
    assert npos.tolist() == [[3, 63, 0, 63]]


def test_cluster_dropin_matches_oracle_cluster(dev):
    """FastKVCluster.update_kv end to end incl. host logic: early-out identity, proportional mode, TSP guard."""
    from fastkv_amd import FastKVCluster
    from oracle.fastkv_oracle import OracleFastKVCluster
    q, k, v = make_qkv(9, 1, 8, 2, 1000, 128, 8)
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
    for kw in (dict(max_capacity_prompt=128, tsp_layer=True, tsp_length=256, pooling="maxpool"),
               dict(max_capacity_prompt=128, tsp_layer=False, pooling="avgpool"),
               dict(max_capacity_prompt=512, tsp_layer=True, tsp_length=2048, eviction_mode="proportional", retain_rate=0.1, tsp_rate=0.2),
               dict(max_capacity_prompt=1000, tsp_layer=True, tsp_length=999),
               dict(max_capacity_prompt=128, tsp_layer=True, tsp_length=1000)):
        a, b = FastKVCluster(**kw), OracleFastKVCluster(**kw)
        ka, va, ta = a.update_kv(kd, qd, vd, None, 4, 0)
        kb, vb, tb = b.update_kv(k, q, v, None, 4, 0)
        assert torch.equal(ka.cpu(), kb) and torch.equal(va.cpu(), vb)
        assert (ta is None) == (tb is None) and (ta is None or torch.equal(ta.cpu(), tb))
        assert a.max_capacity_prompt == b.max_capacity_prompt and a.tsp_length == b.tsp_length
    c = FastKVCluster(max_capacity_prompt=2048)
    ka, va, ta = c.update_kv(kd, qd, vd, None, 4, 0)
    assert ka is kd and va is vd and ta is None                     # utils.py:89-91 returns the input objects


def test_determinism(dev):
    from fastkv_amd import ops
    q, k, v = make_qkv(33, 1, 32, 8, 8192, 128, 8)
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
    outs = [ops.update_kv(qd, kd, vd, 8, 7, "maxpool", 1024, 2048, "score", return_indices=True, return_scores=True) for _ in range(3)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


def test_bad_arguments_raise(dev):
    from fastkv_amd import FastKVCluster, ops
    from fastkv_amd._lib import FastKVNativeError
    q, k, v = make_qkv(1, 1, 4, 2, 600, 128, 8)
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
    with pytest.raises(ValueError, match="Pooling method not supported"):
        FastKVCluster(max_capacity_prompt=64, pooling="l2pool").update_kv(kd, qd, vd, None, 2, 0)
    with pytest.raises(FastKVNativeError):
        ops.update_kv(qd, kd, vd, 8, 6, "avgpool", 64)            # even kernel size
    with pytest.raises(RuntimeError):
        ops.update_kv(q, k, v, 8, 7, "avgpool", 64)               # CPU tensors: no fallback


@pytest.mark.parametrize("engine", ["valu", "mfma", "mfma16"])
@pytest.mark.parametrize("name", ["tiny_avg", "ragged_mha_d64", "gqa8_w16", "cfg1", "cfg2_max"])
def test_both_contraction_engines_bit_exact(name, engine, dev):
    """Every engine against the oracle under ITS contract: the vector-ALU and the FP32 matrix-pipe engines are the same fp32 fma
    chain (oracle "fmaf"), "mfma16" is the fp16 matrix instruction on the fp16 operands (oracle "mfma16")."""
    from fastkv_amd import ops
    try:
        case, (q, k, v), (ko, vo, idx, tsp, c, t), (gko, gvo, gtsp, gidx, gc) = _run_both(name, dev, "score", engine)
    finally:
        ops.set_score_engine("auto")
    assert torch.equal(gc.cpu().view(torch.int16), c.view(torch.int16))
    assert torch.equal(gidx.cpu(), idx) and torch.equal(gko.cpu(), ko) and torch.equal(gvo.cpu(), vo)


def test_matrix_instruction_matches_its_restatement_live(dev):
    """The instruction the default contract leans on, on THIS GPU, against the oracle's restatement of it (fastkv_debug_mfma16 =
    the raw v_mfma_f32_32x32x16_f16 chain; oracle mfma16_tiles): fresh random tiles every run would hide a failure's inputs, so the
    seeds are fixed -- 3 x 1500 tiles of single instructions with an accumulator and of head_dim 64 / 128 / 256 chains, operands
    N(0, s) for s from 1e-3 to 200, exponent-uniform operands over the whole fp16 range, special values sprinkled in."""
    from fastkv_amd._lib import load
    from oracle import fastkv_oracle as O
    L = load()
    g = torch.Generator().manual_seed(20251003)
    T = 1500

    def rand_tiles(dd, kind):
        if kind == "normal":
            sc = 10.0 ** (torch.rand(T, 1, 1, generator=g) * 5.3 - 3.0)
            a, b = ((torch.randn(T, 32, dd, generator=g) * sc).half() for _ in range(2))
        else:                                          # every exponent of fp16 equally likely (subnormals included), random sign / mantissa
            a, b = (torch.randint(0, 0x7c00, (T, 32, dd), generator=g).to(torch.int16).view(torch.float16) *
                    (torch.randint(0, 2, (T, 32, dd), generator=g) * 2 - 1).half() for _ in range(2))
        if kind == "special":
            for t_, v_ in ((a, float("inf")), (b, float("-inf")), (a, float("nan")), (b, 65504.0), (a, -0.0)):
                m = torch.rand(T, 32, dd, generator=g) < 2e-4
                t_[m] = v_
        return a.contiguous(), b.contiguous()

    total = 0
    for dd, kind, with_c in ((16, "normal", True), (16, "wide", True), (128, "normal", False), (128, "special", False), (64, "wide", False),
                             (256, "normal", True)):
        a, b = rand_tiles(dd, kind)
        c = (torch.randn(T, 32, 32, generator=g) * 10.0 ** (torch.rand(T, 1, 1, generator=g) * 6 - 3)).contiguous() if with_c else None
        out = torch.empty(T, 32, 32, dtype=torch.float32, device=dev)
        ad, bd = a.to(dev), b.to(dev)
        cd = c.to(dev) if c is not None else None
        assert L.fastkv_debug_mfma16(ad.data_ptr(), bd.data_ptr(), cd.data_ptr() if cd is not None else None, out.data_ptr(), T, dd, None) == 0
        torch.cuda.synchronize()
        want, got = O.mfma16_tiles(a, b, c), out.cpu()
        ok = (got.view(torch.int32) == want.view(torch.int32)) | (torch.isnan(got) & torch.isnan(want))
        assert bool(ok.all()), (dd, kind, int((~ok).sum()))
        total += ok.numel()
    assert total == 6 * T * 1024
    # Controlled exponent distances: an accumulator in [2^e, 2^(e+1)) -- mantissas at both ends of the binade included -- on top of 16
    # products that all have the unnormalised exponent e - d, for d = -12 .. 48.  d = 28 is the distance at which this chip drops the
    # products (oracle/fastkv_oracle.c step 2b, found exactly this way); the sweep is here so that another such distance -- or a
    # different chip revision -- cannot go unnoticed.
    Ts = 60
    for e_acc in (1, 8, 15):
        for d in range(-12, 49):
            s_ = e_acc - d + 30                                   # ea + eb of every product
            lo, hi = max(1, s_ - 30), min(30, s_ - 1)
            if lo > hi:
                continue
            eak = torch.randint(lo, hi + 1, (Ts, 1, 16), generator=g).expand(Ts, 32, 16)
            ebk = s_ - eak
            ma, mb = (torch.randint(0, 1024, (Ts, 32, 16), generator=g) for _ in range(2))
            sa, sb = (torch.randint(0, 2, (Ts, 32, 16), generator=g) for _ in range(2))
            a = ((sa << 15) | (eak << 10) | ma).to(torch.int16).view(torch.float16).contiguous()
            b = ((sb << 15) | (ebk << 10) | mb).to(torch.int16).view(torch.float16).contiguous()
            mant = torch.randint(0, 1 << 23, (Ts, 32, 32), generator=g)
            edge = torch.rand(Ts, 32, 32, generator=g)
            mant = torch.where(edge < 0.05, torch.zeros_like(mant), torch.where(edge > 0.95, torch.full_like(mant, (1 << 23) - 1), mant))
            cbits = (torch.randint(0, 2, (Ts, 32, 32), generator=g) << 31) | ((e_acc + 127) << 23) | mant
            c = cbits.to(torch.int32).view(torch.float32).contiguous()
            out = torch.empty(Ts, 32, 32, dtype=torch.float32, device=dev)
            ad, bd, cd = a.to(dev), b.to(dev), c.to(dev)
            assert L.fastkv_debug_mfma16(ad.data_ptr(), bd.data_ptr(), cd.data_ptr(), out.data_ptr(), Ts, 16, None) == 0
            torch.cuda.synchronize()
            want, got = O.mfma16_tiles(a, b, c), out.cpu()
            nbad = int((got.view(torch.int32) != want.view(torch.int32)).sum())
            assert nbad == 0, (e_acc, d, nbad)


def test_arithmetic_contract_on_gpu(dev):
    """Every primitive of csrc/fk_device.h against its twin in oracle/fastkv_oracle.c, bit for bit."""
    import ctypes
    import math
    from fastkv_amd._lib import load
    from oracle import fastkv_oracle as O
    L, Lo = load(), O.lib()

    def run(op, a, b=None):
        ad = a.contiguous().to(dev)
        bd = b.contiguous().to(dev) if b is not None else None
        out = torch.zeros_like(ad)
        o64 = torch.zeros(a.numel(), dtype=torch.int64, device=dev)
        rc = L.fastkv_debug_contract(op, ad.data_ptr(), bd.data_ptr() if bd is not None else None, out.data_ptr(), o64.data_ptr(),
                                     a.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        return out.cpu(), o64.cpu()

    bits = lambda t: t.view(torch.int32)
    gen = torch.Generator().manual_seed(7)
    # deterministic exp over the softmax range (+ the cut-off and -inf)
    d = torch.cat([-torch.rand(100000, generator=gen) * 90, torch.tensor([0.0, -87.0, -87.0001, -1e9, float("-inf")])])
    g, _ = run(0, d)
    want = torch.tensor([Lo.fastkv_oracle_det_expf(float(x)) for x in d.tolist()])
    assert torch.equal(bits(g), bits(want))
    g2, _ = run(9, d)                                                     # the two-per-instruction twin (v_pk_fma_f32)
    assert torch.equal(bits(g2), bits(want))
    # logit scaling: scale_div == IEEE division for EVERY fp16 value, D = 64 / 128 / 256
    allh = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.float16).float()
    allh = allh[~torch.isnan(allh)]
    for D in (64, 128, 256):
        c = torch.full_like(allh, float(torch.tensor(math.sqrt(D), dtype=torch.float32)))
        g, _ = run(8, allh, c)
        assert torch.equal(bits(g), bits(allh / c)), D
        g, _ = run(11, allh, c)
        assert torch.equal(bits(g), bits(allh / c)), D
        # the two-operation quotient of the fused kernel's epilogue (q = fma(x, rc, x * rc_lo)): every FINITE fp16 value, zeros with their signs
        fin = allh[torch.isfinite(allh)]
        g, _ = run(13, fin, c[:fin.numel()])
        assert torch.equal(bits(g), bits(fin / c[:fin.numel()])), D
    # IEEE division / reciprocal as used for 1/sum and /kernel_size
    y = torch.rand(200000, generator=gen) * 4000 + 1e-3
    assert torch.equal(bits(run(1, torch.ones_like(y), y)[0]), bits(1.0 / y))
    x = torch.rand(200000, generator=gen)
    assert torch.equal(bits(run(1, x, torch.full_like(x, 7.0))[0]), bits(x / 7.0))
    # fp32 -> fp16 -> fp32 incl. the subnormal range; plain mul / add
    z = torch.cat([torch.rand(100000, generator=gen) * 2e-4, torch.rand(100000, generator=gen) * 70000])
    assert torch.equal(bits(run(2, z)[0]), bits(z.half().float()))
    # the packed conversion (v_cvt_pk_f16_f32): same bits as the scalar one on random values and on EVERY fp16 rounding
    # boundary (the midpoint of two neighbouring fp16 values, one fp32 ulp below it and one above it), both signs, overflow
    assert torch.equal(bits(run(12, z)[0]), bits(z.half().float()))
    h = torch.arange(0, 0x7C00, dtype=torch.int32).to(torch.int16).view(torch.float16).float()      # all finite non-negative fp16
    mid = (h[:-1] + h[1:]) / 2                                                                        # exact in fp32
    mb = mid.view(torch.int32)
    pts = torch.cat([mid, (mb - 1).view(torch.float32), (mb + 1).view(torch.float32), h, torch.tensor([65519.0, 65520.0, 65536.0, 1e30])])
    pts = torch.cat([pts, -pts])
    if pts.numel() % 2:
        pts = torch.cat([pts, pts[:1]])
    assert torch.equal(bits(run(12, pts)[0]), bits(pts.half().float())) and torch.equal(bits(run(2, pts)[0]), bits(pts.half().float()))
    a, b = torch.rand(100000, generator=gen) * 1e-3, torch.rand(100000, generator=gen)
    assert torch.equal(bits(run(6, a, b)[0]), bits(a * b)) and torch.equal(bits(run(7, a, b)[0]), bits(a + b))
    # fixed-point softmax sum: conversion both ways
    e = torch.cat([torch.rand(50000, generator=gen), torch.rand(50000, generator=gen) * 1e-9, torch.tensor([0.0, 1.0, 2.0 ** -41, 2.0 ** -40])])
    g, g64 = run(3, e)
    w64 = torch.tensor([Lo.fastkv_oracle_exp_to_fix(float(v)) for v in e.tolist()], dtype=torch.int64)
    assert torch.equal(g64, w64)
    wf = torch.tensor([Lo.fastkv_oracle_fix_to_f32(int(v)) for v in w64.tolist()])
    assert torch.equal(bits(g), bits(wf))
    assert torch.equal(run(10, e)[1], w64)
    # ... and the four-operation conversion of the fused kernel's phase B (round 6: hi = RNE(e * 2^20), signed lo, both by the fp32
    # adder against a magic constant): the same 64-bit value RNE(e * 2^40) -- random values, the exponentials themselves, and every
    # kind of tie (k + 1/2) * 2^-40 and (k + 1/2) * 2^-20 the fp32 grid holds near the ranges where the roundings happen
    ties = torch.cat([(torch.arange(0, 4096, dtype=torch.float64) + 0.5) * 2.0 ** -40, (torch.arange(0, 4096, dtype=torch.float64) + 0.5) * 2.0 ** -20,
                      (torch.arange(0, 4096, dtype=torch.float64) * 2 + 1) * 2.0 ** -41 + 2.0 ** -20,
                      1.0 - (torch.arange(0, 4096, dtype=torch.float64) + 0.5) * 2.0 ** -24]).float()
    ee = torch.cat([e, ties, want[torch.isfinite(want)][:60000], torch.tensor([1.6e-38, 2.0 ** -42, 3 * 2.0 ** -42, 0.5, 0.9999999])])
    if ee.numel() % 2:
        ee = torch.cat([ee, ee[:1]])
    w64b = torch.tensor([Lo.fastkv_oracle_exp_to_fix(float(v)) for v in ee.tolist()], dtype=torch.int64)
    assert torch.equal(run(14, ee)[1], w64b)
    big = torch.randint(0, 2 ** 62, (50000,), generator=gen, dtype=torch.int64)
    hi, lo = (big >> 32).to(torch.int32).view(torch.float32), (big & 0xFFFFFFFF).to(torch.int64).to(torch.int32).view(torch.float32)
    # NaN bit patterns survive the device copy, so op 5 sees exactly `big`
    g, _ = run(5, hi, lo)
    wf = torch.tensor([Lo.fastkv_oracle_fix_to_f32(int(v)) for v in big.tolist()])
    assert torch.equal(bits(g), bits(wf))


@pytest.mark.parametrize("shape", [
    # (B, H, Hkv, S, D, W, ks, pooling, cap, tsp_len): long prompt, published proportional recipe at 128k (streaming select path)
    dict(B=1, H=32, Hkv=8, S=131072, D=128, W=8, ks=7, pooling="avgpool", cap=13107, tsp_len=26214),
    # head_dim 256 (Gemma-style geometry), G=2
    dict(B=1, H=8, Hkv=4, S=5000, D=256, W=8, ks=7, pooling="maxpool", cap=600, tsp_len=1200),
    # budget above 16384 winners per head: the ranking reads its key list from L2 instead of LDS
    dict(B=1, H=4, Hkv=2, S=40000, D=64, W=8, ks=5, pooling="avgpool", cap=20000, tsp_len=0),
    # wide window (W=32, G=2 -> 64 query rows = two MFMA row blocks), batch 3
    dict(B=3, H=4, Hkv=2, S=2100, D=128, W=32, ks=13, pooling="maxpool", cap=300, tsp_len=700),
    # the fused scoring kernel's other instantiations (G=4, W=8): head_dim 64 and 256, four tiles per wave (S=65536 on 8 KV
    # heads), batch 2 with a ragged last tile, a wide pooling kernel (halo of 15 positions from each neighbour)
    dict(B=1, H=16, Hkv=4, S=3000, D=64, W=8, ks=7, pooling="avgpool", cap=400, tsp_len=900),
    dict(B=1, H=8, Hkv=2, S=2500, D=256, W=8, ks=5, pooling="maxpool", cap=300, tsp_len=0),
    dict(B=1, H=32, Hkv=8, S=65536, D=128, W=8, ks=7, pooling="maxpool", cap=4096, tsp_len=8192),
    dict(B=2, H=16, Hkv=4, S=5003, D=128, W=8, ks=31, pooling="avgpool", cap=700, tsp_len=1500),
    # 8 and 12 query heads per KV head (Llama-3-70B geometry; one tensor-parallel rank of it at 32k): the fused kernel runs
    # two / three virtual heads per KV head and chains their head sums
    dict(B=1, H=16, Hkv=2, S=3100, D=128, W=8, ks=7, pooling="maxpool", cap=500, tsp_len=1000),
    dict(B=2, H=12, Hkv=1, S=1500, D=64, W=8, ks=5, pooling="avgpool", cap=1500, tsp_len=0),
    dict(B=1, H=8, Hkv=1, S=32768, D=128, W=8, ks=7, pooling="maxpool", cap=2048, tsp_len=2048),
])
def test_large_and_unusual_shapes_bit_exact(shape, dev):
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    s = shape
    q, k, v = make_qkv(77, s["B"], s["H"], s["Hkv"], s["S"], s["D"], s["W"])
    want = O.update_kv(q, k, v, s["W"], s["ks"], s["pooling"], s["cap"], s["tsp_len"], "score", return_scores=True)
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
    got = ops.update_kv(qd, kd, vd, s["W"], s["ks"], s["pooling"], s["cap"], s["tsp_len"], "score", return_indices=True,
                        return_scores=True)
    torch.cuda.synchronize()
    assert torch.equal(got[4].cpu().view(torch.int16), want[4].view(torch.int16))          # scores
    assert torch.equal(got[3].cpu(), want[2])                                              # per-head indices, reference order
    assert torch.equal(got[0].cpu(), want[0]) and torch.equal(got[1].cpu(), want[1])       # compacted K / V
    if s["tsp_len"]:
        assert torch.equal(got[2].cpu(), want[3])


def _child(code, env_extra):
    import os
    import subprocess
    import sys
    env = dict(os.environ, **env_extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)


def test_three_kernel_path_is_bit_exact_too(dev):
    """FASTKV_FUSED=0 (read once per process) takes score_logits + row_stats + score_finalize instead of the fused kernel:
    same bits.  A child process runs the oracle comparison of this file on three cases."""
    r = _child("import sys, pytest; sys.exit(pytest.main(['tests/test_hip_parity.py', '-q', '-x', '-m', 'gpu', '-k', "
               "'bit_exact_vs_oracle and (tiny_avg or cfg1 or cfg2_max)']))", {"FASTKV_FUSED": "0"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_uninitialised_workspace_fails_loudly(dev):
    """include/fastkv_hip.h: a workspace that never saw fastkv_workspace_init makes the scoring kernel leave at once and
    report FASTKV_EABORTED through fastkv_last_status (no trap: the context stays usable -- the same process then runs the
    call correctly); with the initialisation the same code returns the oracle's result."""
    code = """
import ctypes, sys, torch
sys.path.insert(0, 'tests')
from gen_inputs import make_qkv
from fastkv_amd._lib import Problem, load
L = load(); dev = torch.device('cuda:0')
q, k, v = make_qkv(3, 1, 32, 8, 4096, 128, 8)
qd, kd, vd = (t.transpose(1, 2).contiguous().to(dev).transpose(1, 2) for t in (q, k, v))
p = Problem(B=1, H=32, Hkv=8, S=4096, D=128, window=8, kernel=7, pooling=1, capacity=512, tsp_len=0, order=1, reserved=0)
ws = torch.full((L.fastkv_workspace_bytes(ctypes.byref(p)),), 0xAB, dtype=torch.uint8, device=dev)
if INIT:
    assert L.fastkv_workspace_init(ws.data_ptr(), ws.numel(), None) == 0
ko = torch.empty(1, 8, 512, 128, dtype=torch.float16, device=dev); vo = torch.empty_like(ko)
S4 = ctypes.c_int64 * 4
rc = L.fastkv_update_kv_f16(ctypes.byref(p), qd.data_ptr(), S4(*qd.stride()), kd.data_ptr(), S4(*kd.stride()), vd.data_ptr(),
                            S4(*vd.stride()), ko.data_ptr(), vo.data_ptr(), None, None, None, ws.data_ptr(), ws.numel(), None)
assert rc == 0
torch.cuda.synchronize()
st = L.fastkv_last_status()
if not INIT:
    assert st == -5, st                                    # FASTKV_EABORTED, reported once
    assert L.fastkv_last_status() == 0
    print('reported')
    # the context is alive: initialise and repeat
    assert L.fastkv_workspace_init(ws.data_ptr(), ws.numel(), None) == 0
    rc = L.fastkv_update_kv_f16(ctypes.byref(p), qd.data_ptr(), S4(*qd.stride()), kd.data_ptr(), S4(*kd.stride()), vd.data_ptr(),
                                S4(*vd.stride()), ko.data_ptr(), vo.data_ptr(), None, None, None, ws.data_ptr(), ws.numel(), None)
    assert rc == 0
    torch.cuda.synchronize()
    st = L.fastkv_last_status()
assert st == 0
from oracle import fastkv_oracle as O
want = O.update_kv(q, k, v, 8, 7, 'maxpool', 512, 0, 'score')
assert torch.equal(ko.cpu(), want[0]) and torch.equal(vo.cpu(), want[1])
print('child ok')
"""
    good = _child("INIT = True\n" + code, {})
    assert good.returncode == 0 and "child ok" in good.stdout and "reported" not in good.stdout, good.stdout[-1500:] + good.stderr[-1500:]
    bad = _child("INIT = False\n" + code, {"FASTKV_SPIN_LIMIT_MS": "50"})
    assert bad.returncode == 0 and "reported" in bad.stdout and "child ok" in bad.stdout, bad.stdout[-1500:] + bad.stderr[-1500:]


def test_graph_replay_with_changing_inputs(dev):
    """The operator captured in a HIP graph (workspace initialisation included in the capture: no warm-up on the capture
    stream) and replayed on new data every time: the hand-off tokens of the fused scoring kernel come from the epoch in the
    workspace, not from a launch argument, so every replay equals the eager result."""
    from fastkv_amd import ops
    case = CASES["cfg1"]
    shapes = (case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"])
    q0, k0, v0 = make_qkv(900, *shapes)
    qs, ks, vs = (_to_dev(t, dev) for t in (q0, k0, v0))
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            out = ops.update_kv(qs, ks, vs, case["W"], case["ks"], case["pooling"], case["cap"], case["tsp_len"], "score",
                                return_indices=True)
    for seed in (901, 902, 903, 901):
        q, k, v = make_qkv(seed, *shapes)
        qs.copy_(_to_dev(q, dev)); ks.copy_(_to_dev(k, dev)); vs.copy_(_to_dev(v, dev))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        want = ops.update_kv(_to_dev(q, dev), _to_dev(k, dev), _to_dev(v, dev), case["W"], case["ks"], case["pooling"], case["cap"],
                             case["tsp_len"], "score", return_indices=True)
        torch.cuda.synchronize()
        assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1]) and torch.equal(out[3], want[3]), seed
        assert torch.equal(out[2], want[2])


@pytest.mark.parametrize("contraction", ["mfma16", "fmaf"])
def test_fused_path_on_special_values(contraction, dev):
    """Subnormal / signed-zero / huge K values must give the oracle's bits; with Inf or NaN in K or in a window row of Q the
    scores must agree with the oracle AND with a second kernel family (score_logits + row_stats + score_finalize): NaN in the
    same places, every other value bit for bit.  Under the fp32 contract the fused kernel converts K on the matrix pipe (fp16 MFMA
    x permutation matrix: exact for finite values) and falls back to the vector-ALU conversion when a result is NaN; the second
    family is the vector-ALU engine.  Under the default contract the fp16 matrix instruction follows IEEE on Inf / NaN operands
    element by element, as the oracle's restatement of it does; the second family is the staged matrix kernels (no-wait mode)."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    O.set_contraction(contraction)
    first = "mfma" if contraction == "fmaf" else "mfma16"

    class second:                                  # context: the other kernel family of the same contract
        def __enter__(self):
            if contraction == "fmaf":
                ops.set_score_engine("valu")
            else:
                ops.set_score_engine("mfma16")
                self.was = ops.set_no_wait_mode(True)

        def __exit__(self, *a):
            ops.set_score_engine(first)
            if contraction != "fmaf":
                ops.set_no_wait_mode(self.was)

    ops.set_score_engine(first)
    B, H, Hkv, S, D, W = 1, 16, 4, 2200, 128, 8
    q, k, v = make_qkv(321, B, H, Hkv, S, D, W)
    k = k.clone()
    k[:, :, 100:400] *= 1e-5                      # fp16 subnormals
    k[:, :, 500:520] = 0.0
    k[:, 1, 500:520:2] = -0.0
    k[:, :, 700:760] *= 200.0                     # large finite values
    want = O.update_kv(q, k, v, W, 7, "avgpool", 300, 600, "score", return_scores=True)
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
    got = ops.update_kv(qd, kd, vd, W, 7, "avgpool", 300, 600, "score", return_indices=True, return_scores=True)
    assert torch.equal(got[4].cpu().view(torch.int16), want[4].view(torch.int16)) and torch.equal(got[3].cpu(), want[2])
    assert torch.equal(got[0].cpu(), want[0]) and torch.equal(got[2].cpu(), want[3])
    # non-finite K: the fused kernel against the second kernel family, scores bit for bit
    k2 = k.clone()
    k2[0, 0, 1000, 5] = float("inf")
    k2[0, 2, 1500, 77] = float("-inf")
    k2[0, 3, 64, 0] = float("nan")
    k2d = _to_dev(k2, dev)
    c_f, t_f = ops.scores(qd, k2d, W, 7, "maxpool")
    with second():
        c_v, t_v = ops.scores(qd, k2d, W, 7, "maxpool")
    torch.cuda.synchronize()

    def bits(x):
        return x.view(torch.int16)

    assert torch.equal(bits(c_f), bits(c_v)) and torch.equal(bits(t_f), bits(t_v))
    assert torch.isnan(c_f[0, 0]).any() and not torch.isnan(c_f[0, 1]).any()
    # ... and against the ORACLE, selection included.  A generated NaN has no portable sign (x86: negative default NaN, this
    # GPU: positive), so both sides store every NaN score as 0x7e00 (f2h_score): it then ranks first on both, as in torch.topk.
    nan_bits = bits(c_f)[torch.isnan(c_f)]
    assert bool((nan_bits == 0x7e00).all()) and bool((bits(t_f)[torch.isnan(t_f)] == 0x7e00).all())
    q2 = q.clone()
    q2[0, 5, S - 3, 17] = float("inf")               # a window row with +inf: inf - inf in the softmax of that row
    q2d = _to_dev(q2, dev)
    for qq, qqd, pooling, order in ((q, qd, "maxpool", "score"), (q2, q2d, "avgpool", "index"), (q2, q2d, "maxpool", "score")):
        want = O.update_kv(qq, k2, v, W, 7, pooling, 300, 600, order, return_scores=True)
        for family in ("first", "second"):
            if family == "first":
                got = ops.update_kv(qqd, k2d, vd, W, 7, pooling, 300, 600, order, return_indices=True, return_scores=True)
            else:
                with second():
                    got = ops.update_kv(qqd, k2d, vd, W, 7, pooling, 300, 600, order, return_indices=True, return_scores=True)
            assert torch.equal(bits(got[4].cpu()), bits(want[4])), (pooling, order, family)
            assert torch.equal(got[3].cpu(), want[2]) and torch.equal(got[2].cpu(), want[3]), (pooling, order, family)
            assert torch.equal(bits(got[0].cpu()), bits(want[0])) and torch.equal(bits(got[1].cpu()), bits(want[1]))


def test_keep_all_in_index_order_is_a_copy(dev):
    """capacity == S with ascending-position order and no TSP: the output is the input's candidates followed by the window
    rows -- the operator must still return exactly the oracle's K/V/indices (it skips the scoring kernels)."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    q, k, v = make_qkv(77, 2, 8, 2, 1536, 128, 8)
    want = O.update_kv(q, k, v, 8, 7, "avgpool", 1536, 0, "index")
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
    got = ops.update_kv(qd, kd, vd, 8, 7, "avgpool", 1536, 0, "index", return_indices=True)
    assert torch.equal(got[0].cpu(), want[0]) and torch.equal(got[1].cpu(), want[1]) and torch.equal(got[3].cpu(), want[2])
    assert got[2] is None and torch.equal(got[0].cpu(), k)


def test_randomised_shapes_bit_exact(dev):
    """40 seeded random geometries (both scoring paths get their share: G*W == 32 with W == 8 takes the fused kernel, the
    rest the staged kernels or the vector-ALU engine), short prompts, every knob of the operator."""
    import random
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    rng = random.Random(20240917)
    for it in range(40):
        fusedish = it % 2 == 0
        W = 8 if fusedish else rng.choice([1, 2, 4, 8, 16])
        Hkv = rng.choice([1, 2, 3, 4])
        G = 4 if fusedish else rng.choice([1, 2, 4, 8])
        D = rng.choice([64, 128, 128, 256])
        B = rng.choice([1, 1, 2])
        ks = rng.choice([1, 3, 5, 7, 7, 13])
        S = rng.randint(W + 2 + ks, 1500)
        cap = rng.randint(W + 1, S)
        tsp_len = rng.choice([0, 0, rng.randint(W + 1, S - 1)]) if S - 1 > W + 1 else 0
        pooling = rng.choice(["avgpool", "maxpool"])
        order = rng.choice(["index", "score"])
        q, k, v = make_qkv(5000 + it, B, Hkv * G, Hkv, S, D, W)
        want = O.update_kv(q, k, v, W, ks, pooling, cap, tsp_len, order, return_scores=True)
        qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
        got = ops.update_kv(qd, kd, vd, W, ks, pooling, cap, tsp_len, order, return_indices=True, return_scores=True)
        torch.cuda.synchronize()
        tag = dict(it=it, B=B, H=Hkv * G, Hkv=Hkv, S=S, D=D, W=W, ks=ks, cap=cap, tsp_len=tsp_len, pooling=pooling, order=order)
        assert torch.equal(got[4].cpu().view(torch.int16), want[4].view(torch.int16)), tag
        assert torch.equal(got[3].cpu(), want[2]), tag
        assert torch.equal(got[0].cpu(), want[0]) and torch.equal(got[1].cpu(), want[1]), tag
        assert (got[2] is None and want[3] is None) or torch.equal(got[2].cpu(), want[3]), tag


def test_two_streams_are_serialised_by_the_library(dev):
    """The fused scoring kernel of one call must not overlap with another one's on the GPU; calls issued on two streams of
    this process without any synchronisation between them still give the oracle's result (an event chain in the library)."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    case = CASES["cfg2_max"]
    shapes = (case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"])
    data = [make_qkv(910 + i, *shapes) for i in range(2)]
    devd = [tuple(_to_dev(t, dev) for t in d) for d in data]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    outs = [None, None]
    for rep in range(3):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                outs[i] = ops.update_kv(*devd[i], case["W"], case["ks"], case["pooling"], case["cap"], 0, "score", return_indices=True)
    torch.cuda.synchronize()
    for i in (0, 1):
        want = O.update_kv(*data[i], case["W"], case["ks"], case["pooling"], case["cap"], 0, "score")
        assert torch.equal(outs[i][0].cpu(), want[0]) and torch.equal(outs[i][3].cpu(), want[2])


def test_compaction_into_a_cache_slab(dev):
    """out=(k_view, v_view): the compacted rows land directly in the first `capacity` rows of a pre-sized cache slab
    [B,Hkv,capacity+64,D] (the compaction is the cache write); the rest of the slab is untouched."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    case = CASES["cfg1"]
    q, k, v = make_qkv(case["seed"], case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"])
    want = O.update_kv(q, k, v, case["W"], case["ks"], case["pooling"], case["cap"], case["tsp_len"], "score")
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
    cap = case["cap"]
    kslab = torch.full((case["B"], case["Hkv"], cap + 64, case["D"]), 7.0, dtype=torch.float16, device=dev)
    vslab = torch.full_like(kslab, -3.0)
    got = ops.update_kv(qd, kd, vd, case["W"], case["ks"], case["pooling"], cap, case["tsp_len"], "score",
                        out=(kslab[:, :, :cap], vslab[:, :, :cap]))
    torch.cuda.synchronize()
    assert got[0].data_ptr() == kslab.data_ptr()
    assert torch.equal(kslab[:, :, :cap].cpu(), want[0]) and torch.equal(vslab[:, :, :cap].cpu(), want[1])
    assert torch.all(kslab[:, :, cap:] == 7.0) and torch.all(vslab[:, :, cap:] == -3.0)
    assert torch.equal(got[2].cpu(), want[3])


@pytest.mark.parametrize("shape", [
    dict(B=1, H=32, Hkv=8, S=2048, D=128, ks=7, pooling="maxpool"),          # the post-TSP layers of the 32k configuration
    dict(B=2, H=16, Hkv=4, S=1111, D=64, ks=5, pooling="avgpool"),
    dict(B=1, H=16, Hkv=2, S=3000, D=128, ks=7, pooling="maxpool"),          # 8 query heads per KV head (two virtual heads)
    dict(B=1, H=8, Hkv=2, S=700, D=256, ks=3, pooling="avgpool"),
])
def test_keep_all_layers_bit_exact(shape, dev):
    """capacity == S without a TSP index (the post-TSP layers): same K/V, same index order as the oracle, in both row orders,
    also into a strided cache slab."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    s = shape
    q, k, v = make_qkv(4242, s["B"], s["H"], s["Hkv"], s["S"], s["D"], 8)
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
    for order in ("score", "index"):
        want = O.update_kv(q, k, v, 8, s["ks"], s["pooling"], s["S"], 0, order, return_scores=True)
        got = ops.update_kv(qd, kd, vd, 8, s["ks"], s["pooling"], s["S"], 0, order, return_indices=True, return_scores=True)
        torch.cuda.synchronize()
        assert torch.equal(got[4].cpu().view(torch.int16), want[4].view(torch.int16)), order
        assert torch.equal(got[3].cpu(), want[2]), order
        assert torch.equal(got[0].cpu(), want[0]) and torch.equal(got[1].cpu(), want[1]), order
    kslab = torch.zeros(s["B"], s["Hkv"], s["S"] + 40, s["D"], dtype=torch.float16, device=dev)
    vslab = torch.zeros_like(kslab)
    ops.update_kv(qd, kd, vd, 8, s["ks"], s["pooling"], s["S"], 0, "score", out=(kslab[:, :, :s["S"]], vslab[:, :, :s["S"]]))
    want = O.update_kv(q, k, v, 8, s["ks"], s["pooling"], s["S"], 0, "score")
    assert torch.equal(kslab[:, :, :s["S"]].cpu(), want[0]) and torch.equal(vslab[:, :, :s["S"]].cpu(), want[1])
    assert not kslab[:, :, s["S"]:].any()


# ------------------------------------------------------------------------------------------------ residency of the in-launch hand-offs
_RESIDENCY_CHILD = """
import ctypes, sys, time, torch
sys.path.insert(0, 'tests')
from gen_inputs import make_qkv
from fastkv_amd import ops
from fastkv_amd._lib import load, FastKVNativeError
from oracle import fastkv_oracle as O
from helpers import default_contraction
O.set_contraction(default_contraction())                                    # (children may run under FASTKV_CONTRACTION=fmaf)
L = load(); dev = torch.device('cuda:0')
q, k, v = make_qkv(77, 1, 32, 8, 32768, 128, 8)
qd, kd, vd = (t.transpose(1, 2).contiguous().to(dev).transpose(1, 2) for t in (q, k, v))
want = O.update_kv(q, k, v, 8, 7, 'maxpool', 2048, 2048, 'score')
def run():
    out = ops.update_kv(qd, kd, vd, 8, 7, 'maxpool', 2048, 2048, 'score', return_indices=True)
    torch.cuda.current_stream().synchronize()                               # this stream only: not the holding kernel's
    return out
def same(out):
    return torch.equal(out[0].cpu(), want[0]) and torch.equal(out[1].cpu(), want[1]) and torch.equal(out[3].cpu(), want[2]) \\
        and torch.equal(out[2].cpu(), want[3])
assert same(run()) and L.fastkv_last_status() == 0                      # idle GPU
side = torch.cuda.Stream()
# "another kernel holds compute units": 128 workgroups with 128 KiB of LDS each sit on half of the chip for HOLD_MS
assert L.fastkv_debug_occupy(128, 128 * 1024, HOLD_MS * 1000, side.cuda_stream) == 0
time.sleep(0.02)
t0 = time.perf_counter()
out = run()
dt = (time.perf_counter() - t0) * 1e3
print('held call took %.1f ms' % dt)
torch.cuda.current_stream().synchronize()
viol = L.fastkv_placement_violations(RESET)
print('placement violations', viol)
"""


def test_operator_next_to_a_kernel_that_holds_half_the_chip(dev):
    """A long-running kernel on another stream holds half of the compute units while the operator runs (32k shape: the fused
    scoring kernel's 512 workgroups and the split selection wait for partners that cannot become resident until the other
    kernel ends).  The launch is delayed, not broken: same bits as the oracle, no report."""
    # The placement policy belongs to the fp32-fma-chain contract (the hazard the pairing fences off needs that contract's matrix phase:
    # include/fastkv_hip.h fastkv_placement_violations): what those launches count is acted upon.  Launches of the mfma16 contract
    # COUNT as well (round 6, ADVICE r05: the check is armed, so a 0 is a measurement) and nothing ever acts on their count: same
    # delay, same bits, violations counted, nothing reported and no switch of kernels under the DEFAULT policy.
    r = _child("HOLD_MS = 300\nRESET = 1\n" + _RESIDENCY_CHILD +
               "assert same(out) and L.fastkv_last_status() == 0\nassert dt > 150, dt\nassert viol > 0, viol\nprint('child ok')\n",
               {"FASTKV_STRICT_PLACEMENT": "0", "FASTKV_CONTRACTION": "fmaf"})
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    r = _child("HOLD_MS = 300\nRESET = 0\n" + _RESIDENCY_CHILD +
               "assert same(out) and L.fastkv_last_status() == 0\nassert dt > 150, dt\nassert viol > 0, viol\n"
               "assert same(run()) and L.fastkv_last_status() == 0 and not ops.no_wait_mode()\nprint('child ok')\n",
               {"FASTKV_CONTRACTION": "mfma16"})
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    # (viol > 0: squeezed onto half of the chip, workgroups of different heads shared compute units -- the launch's own placement check
    # counts that; the result was right all the same.  FASTKV_STRICT_PLACEMENT=0 above = count only.)
    # The DEFAULT policy fails safe: the next call reports FASTKV_EPLACEMENT once -- the caller redoes the affected call --
    # and the process has switched to the no-wait kernels, so the redo cannot be exposed again whatever else runs on the GPU;
    # FASTKV_STRICT_PLACEMENT=1 reports and keeps the fused kernels.
    for env, want_no_wait in (({"FASTKV_CONTRACTION": "fmaf"}, True), ({"FASTKV_CONTRACTION": "fmaf", "FASTKV_STRICT_PLACEMENT": "1"}, False)):
        r = _child("HOLD_MS = 300\nRESET = 0\n" + _RESIDENCY_CHILD + f"""
from fastkv_amd._lib import FASTKV_EPLACEMENT
assert viol > 0 and not ops.no_wait_mode()
try:
    run()
    raise SystemExit('no report')
except FastKVNativeError as e:
    assert e.code == FASTKV_EPLACEMENT, str(e)
assert ops.no_wait_mode() == {want_no_wait}
assert L.fastkv_placement_violations(0) == viol                          # (reported violations stay in the running count)
import warnings
from fastkv_amd._lib import raise_if_aborted
with warnings.catch_warnings():
    warnings.simplefilter('error')                                       # ... and are not warned about a second time (ADVICE r04)
    raise_if_aborted('after the report')
torch.cuda.synchronize()
L.fastkv_profile_enable(1)
assert same(run()) and L.fastkv_last_status() == 0                       # the redo
names = [L.fastkv_profile_kernel_name(i).decode() for i in range(L.fastkv_profile_kernels())]
import numpy as np
cnt = np.zeros(len(names), dtype=np.int64); ms = np.zeros(len(names), dtype=np.float64)
L.fastkv_profile_read(cnt.ctypes.data, ms.ctypes.data)
ran = {{n for n, c in zip(names, cnt) if c}}
print('kernels of the redo:', sorted(ran))
assert ('score_fused' in ran) == (not {want_no_wait}), ran
print('child ok')
""", env)
        assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_abandoned_launch_is_reported_not_trapped(dev):
    """The same with a wait limit (FASTKV_SPIN_LIMIT_MS=40) shorter than the other kernel's hold: the waiting workgroups give
    up, the launch ends (no trap, no hang, no fault in the stages behind it), the process learns about it through
    FASTKV_EABORTED at the next call -- and the context is alive: the call after that is bit-exact again."""
    code = "HOLD_MS = 600\nRESET = 1\n" + _RESIDENCY_CHILD + """
assert dt < 550, dt                                                         # it did not wait for the other kernel
try:
    run()                                                                   # the NEXT call reports the abandoned one
    raise SystemExit('no report')
except FastKVNativeError as e:
    assert 'gave up' in str(e), str(e)
torch.cuda.synchronize()                                                    # the holding kernel has ended by now or soon
time.sleep(0.7)
assert L.fastkv_last_status() == 0
assert same(run()) and L.fastkv_last_status() == 0
print('child ok')
"""
    # (FASTKV_STRICT_PLACEMENT=0: this test is about the abandoned wait; what a launch that was squeezed next to a foreign kernel
    # reports about its placement is the test above)
    r = _child(code, {"FASTKV_SPIN_LIMIT_MS": "40", "FASTKV_STRICT_PLACEMENT": "0"})
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_a_launch_of_at_most_one_workgroup_per_unit_gets_through_a_held_gpu(dev):
    """Workgroups are numbered unit by unit (round 3): a fused launch with no more workgroups than the chip has compute units (here
    8k: 8 heads x 32 spans = 256) is dispatched head after head, so a head's workgroups become resident together even when another
    kernel holds half of the chip -- the launch completes while the other kernel is still running, same bits, nothing reported,
    although the wait limit (40 ms) is far below the other kernel's hold (600 ms).  (Launches with two workgroups per unit still
    need all of them resident: the two tests above.)"""
    code = "HOLD_MS = 600\nRESET = 1\n" + _RESIDENCY_CHILD.replace("32768", "8192") + """
assert dt < 300, dt                                                         # it did not wait for the other kernel
assert same(out) and L.fastkv_last_status() == 0
torch.cuda.synchronize()
print('child ok')
"""
    r = _child(code, {"FASTKV_SPIN_LIMIT_MS": "40", "FASTKV_STRICT_PLACEMENT": "0"})
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_seed_sweep_32k_on_gpu(dev):
    """All 24 sweep cases at S = 32768 (tests/golden/sweep32k.npz: canonical top-k of the REFERENCE's scores, 12 seeds x the
    constant budget and 12 x the published proportional recipe) replayed through the operator: identical index sets on every
    row except the ones the fixture lists (oracle and GPU agree there too -- same arithmetic)."""
    from fastkv_amd import ops
    from golden_cases import SWEEP_CASES
    from test_oracle_golden import _sweep, check_against_sweep
    z, meta = _sweep()
    for name, case in SWEEP_CASES.items():
        q, k, v = make_qkv(case["seed"], case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"])
        qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
        ko, vo, tsp, idx = ops.update_kv(qd, kd, vd, case["W"], case["ks"], case["pooling"], case["cap"], case["tsp_len"], "index",
                                         return_indices=True)
        check_against_sweep(name, idx.cpu(), tsp.cpu(), z, meta)
        assert torch.equal(ko.cpu(), expected_kv(k, idx.cpu(), case["W"]))


def test_wide_sweep_32k_on_gpu(dev):
    """Every second case of the wide sweep (tests/golden/sweep_wide.npz: 24 seeds x {constant, recipe} x {maxpool, avgpool} + 24 peaked
    cases, canonical top-k of the REFERENCE's scores as row digests) through the operator under the default contract: every row
    matches the reference's digest except the rows the fixture lists, which differ by exactly the listed positions."""
    from fastkv_amd import ops
    from golden_cases import SWEEP_WIDE_CASES
    from test_oracle_golden import _sweep_wide, check_against_wide_sweep
    z, meta = _sweep_wide()
    for i, (name, case) in enumerate(SWEEP_WIDE_CASES.items()):
        if i % 2:
            continue
        q, k, v = make_qkv(case["seed"], case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"], peaked=case.get("peaked", 0))
        qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
        ko, vo, tsp, idx = ops.update_kv(qd, kd, vd, case["W"], case["ks"], case["pooling"], case["cap"], case["tsp_len"], "index",
                                         return_indices=True)
        check_against_wide_sweep(name, idx.cpu(), tsp.cpu(), z, meta)


def test_per_query_head_selection_snapkv_rule(dev):
    """ops.update_kv_per_query_head: the SnapKV baseline's rule (selection per QUERY head, /root/reference/baselines/snapkv/
    utils.py:57-102) through the ordinary operator -- with K/V already repeated to H heads (what the SnapKV attention module
    passes) and with the un-repeated [B,Hkv,S,D] tensors read through a zero head stride (no repeat_kv copy).  Both equal the
    oracle bit for bit and canonical_topk of the REFERENCE's scores (tests/golden/snapkv.npz)."""
    import os
    from fastkv_amd import ops
    from golden_cases import SNAPKV_CASES
    from helpers import GOLDEN
    from oracle import fastkv_oracle as O
    z = np.load(os.path.join(GOLDEN, "snapkv.npz"))
    for name, c in SNAPKV_CASES.items():
        q, k, v = make_qkv(c["seed"], c["B"], c["H"], c["Hkv"], c["S"], c["D"], c["W"])
        G = c["H"] // c["Hkv"]
        kr, vr = (t.repeat_interleave(G, dim=1) for t in (k, v))
        qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
        krd, vrd = (_to_dev(t, dev) for t in (kr, vr))
        for order in ("index", "score"):
            want = O.update_kv(q, kr, vr, c["W"], c["ks"], c["pooling"], c["cap"], 0, order)
            for kk_, vv_ in ((kd, vd), (krd, vrd)):
                ko, vo, idx = ops.update_kv_per_query_head(qd, kk_, vv_, c["W"], c["ks"], c["pooling"], c["cap"], order, return_indices=True)
                assert torch.equal(idx.cpu(), want[2]) and torch.equal(ko.cpu(), want[0]) and torch.equal(vo.cpu(), want[1]), (name, order)
            if order == "index":
                assert torch.equal(idx.cpu(), torch.from_numpy(z[name + ".idx"].astype(np.int64))), name


def test_gemfilter_rule_on_gpu(dev):
    """fastkv_amd.variants.standard_dis_index (window-1 logits -> head sum -> pool -> select through the C ABI) against the
    oracle's restatement of /root/reference/baselines/gemfilter/utils.py:25-38 and the vectors captured from the reference:
    ranked tensor, indices (in order) and distances bit for bit; unrepeated keys give what repeated keys give."""
    import numpy as np
    from fastkv_amd import variants
    from golden_cases import GEMFILTER_CASES
    from helpers import GOLDEN, f16_from_bits
    from oracle import fastkv_oracle as O
    z = np.load(os.path.join(GOLDEN, "gemfilter.npz"))
    from fastkv_amd import ops
    for contraction, engine in (("mfma16", "mfma16"), ("fmaf", "mfma")):
      O.set_contraction(contraction)
      ops.set_score_engine(engine)
      for name, c in GEMFILTER_CASES.items():
        q, k, _ = make_qkv(c["seed"], c["B"], c["H"], c["Hkv"], c["S"], c["D"], 8)
        ql = q[:, :, -1:, :]
        wd, wi = O.standard_dis_index(k, ql, c["k"], pool=c["pool"], kernel_size=c["ks"], sum_over_heads=c["sum_over_heads"])
        qd, kd = _to_dev(ql.contiguous(), dev), _to_dev(k, dev)
        gd, gi = variants.standard_dis_index(kd, qd, c["k"], pool=c["pool"], kernel_size=c["ks"], sum_over_heads=c["sum_over_heads"])
        assert torch.equal(gi.cpu(), wi) and torch.equal(gd.cpu().view(torch.int16), wd.view(torch.int16)), (name, contraction)
        if contraction == "fmaf":
            # the reference's vectors, bit for bit: a property of the fp32 fma chain (its accumulation order resembles torch's CPU
            # kernel; tests/test_oracle_golden.py has the tolerance protocol of the other contract)
            assert torch.equal(gi.cpu(), torch.from_numpy(z[name + ".idx"].astype(np.int64))), name
            assert torch.equal(gd.cpu().view(torch.int16), f16_from_bits(z[name + ".ref_dist"]).view(torch.int16)), name
        kr = kd.repeat_interleave(c["H"] // c["Hkv"], dim=1)                    # as find_context passes them (repeat_kv)
        gd2, gi2 = variants.standard_dis_index(kr, qd, c["k"], pool=c["pool"], kernel_size=c["ks"], sum_over_heads=c["sum_over_heads"])
        assert torch.equal(gi2, gi) and torch.equal(gd2, gd), name
    # ragged length (n % 8 != 0), max-free, norm
    q, k, _ = make_qkv(7, 1, 8, 2, 1003, 128, 8)
    wd, wi = O.standard_dis_index(k, q[:, :, -1:, :], 77, norm=4, pool=True, kernel_size=5, sum_over_heads=True)
    gd, gi = variants.standard_dis_index(_to_dev(k, dev), _to_dev(q[:, :, -1:, :].contiguous(), dev), 77, norm=4, pool=True, kernel_size=5,
                                         sum_over_heads=True)
    assert torch.equal(gi.cpu(), wi) and torch.equal(gd.cpu(), wd)


def test_score_order_with_many_heads_groups_instead_of_counting(dev):
    """With >= 64 (batch x KV head) rows the ORDER_SCORE slots come from one grouping pass per head (rank_group_kernel) instead of the
    comparison counting inside the copy kernel.  Same result: the stand-alone ranked compaction equals the explicit
    score-ordered gather (heavy ties: quantised scores), and the whole operator at B * Hkv = 64 equals the oracle."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    B, Hkv, S, D, W = 16, 8, 640, 64, 8
    g = torch.Generator().manual_seed(5)
    k = torch.randn(B, Hkv, S, D, generator=g).half()
    v = torch.randn(B, Hkv, S, D, generator=g).half()
    sc = (torch.rand(B * Hkv, S - W, generator=g) * 8).floor().half() / 8          # 64 distinct values: long tie runs
    kd, vd, scd = _to_dev(k, dev), _to_dev(v, dev), sc.to(dev)
    for kk in (1, 2, 3, 100, 257, S - W):
        asc = ops.select(scd, kk, "index").view(B, Hkv, kk).contiguous()
        srt = ops.select(scd, kk, "score").view(B, Hkv, kk).contiguous()
        ko, vo, got = ops.compact(kd, vd, asc, W, scores=scd.view(B, Hkv, S - W), return_sorted=True)
        assert torch.equal(got, srt), kk
        assert torch.equal(ko.cpu(), expected_kv(k, srt.cpu(), W)) and torch.equal(vo.cpu(), expected_kv(v, srt.cpu(), W)), kk
    q, k2, v2 = make_qkv(91, 8, 32, 8, 1100, 128, 8)
    want = O.update_kv(q, k2, v2, 8, 7, "maxpool", 300, 500, "score")
    got = ops.update_kv(_to_dev(q, dev), _to_dev(k2, dev), _to_dev(v2, dev), 8, 7, "maxpool", 300, 500, "score", return_indices=True)
    assert torch.equal(got[0].cpu(), want[0]) and torch.equal(got[1].cpu(), want[1])
    assert torch.equal(got[3].cpu(), want[2]) and torch.equal(got[2].cpu(), want[3])
    want = O.update_kv(q, k2, v2, 8, 7, "avgpool", 1100, 0, "score")               # keep-all layers: every candidate ranked
    got = ops.update_kv(_to_dev(q, dev), _to_dev(k2, dev), _to_dev(v2, dev), 8, 7, "avgpool", 1100, 0, "score", return_indices=True)
    assert torch.equal(got[0].cpu(), want[0]) and torch.equal(got[3].cpu(), want[2])
    # longer winner lists (round 5: the published recipe keeps 3276 rows per head at 32k, 13,107 at 128k; the grouping pass used to stop at
    # 2688 winners and such calls fell back to the counting): 2048 bins (<= 6656 winners), 1024 bins (<= 24,000, beyond 8704 with more
    # than 64 KiB of LDS), and the counting fallback beyond
    B2, S2 = 8, 24100
    k3 = torch.randn(B2, Hkv, S2, D, generator=g).half()
    v3 = torch.randn(B2, Hkv, S2, D, generator=g).half()
    sc3 = (torch.rand(B2 * Hkv, S2 - W, generator=g) * 512).floor().half() / 512      # ties everywhere
    k3d, v3d, sc3d = _to_dev(k3, dev), _to_dev(v3, dev), sc3.to(dev)
    for kk in (2689, 3268, 6656, 6657, 8704, 8705, 13099, 24000, 24001, S2 - W):
        asc = ops.select(sc3d, kk, "index").view(B2, Hkv, kk).contiguous()
        srt = ops.select(sc3d, kk, "score").view(B2, Hkv, kk).contiguous()
        ko, vo, got = ops.compact(k3d, v3d, asc, W, scores=sc3d.view(B2, Hkv, S2 - W), return_sorted=True)
        assert torch.equal(got, srt), kk
        assert torch.equal(ko.cpu(), expected_kv(k3, srt.cpu(), W)) and torch.equal(vo.cpu(), expected_kv(v3, srt.cpu(), W)), kk


def test_every_operator_call_spends_its_handoff_token(dev):
    """Found by tools/stress_parity.py (1 call in 800, deterministic in its sequence): a call on the STAGED scoring path (window
    4: no fused launch) still runs the split selection, whose counters are granules tagged with the workspace's current token,
    but it used to leave the epoch where it was -- the next call (fused path, hand-off areas at other, shape-dependent
    offsets) then found granules that already carried ITS token and took a pooling halo from them.  Now every call advances
    the epoch.  Checked directly (the epoch word of the control block moves by one call's share of tokens -- EPOCH_STRIDE = 64: the
    s-th scoring launch of a call uses epoch + s, csrc/fk_host.h -- per call, whatever the path) and by the original sequence (cases 48 -> 49 of `stress_parity.py 60 777`), repeated."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    c48 = dict(B=1, H=16, Hkv=8, S=16384, D=128, W=4, ks=3, cap=256, tsp_len=10882, pooling="maxpool", order="score", seed=9048)
    c49 = dict(B=1, H=8, Hkv=2, S=12499, D=128, W=8, ks=7, cap=512, tsp_len=6295, pooling="maxpool", order="index", seed=9049)

    def run(c, want=None):
        q, k, v = make_qkv(c["seed"], c["B"], c["H"], c["Hkv"], c["S"], c["D"], c["W"])
        if want is None:
            want = O.update_kv(q, k, v, c["W"], c["ks"], c["pooling"], c["cap"], c["tsp_len"], c["order"], return_scores=True)
        got = ops.update_kv(_to_dev(q, dev), _to_dev(k, dev), _to_dev(v, dev), c["W"], c["ks"], c["pooling"], c["cap"], c["tsp_len"],
                            c["order"], return_indices=True, return_scores=True)
        torch.cuda.synchronize()
        assert torch.equal(got[4].cpu().view(torch.int16), want[4].view(torch.int16)), c
        assert torch.equal(got[3].cpu(), want[2]) and torch.equal(got[2].cpu(), want[3]) and torch.equal(got[0].cpu(), want[0]), c
        return want

    def epoch():
        ws = ops._ws_cache[(dev.index, ops._stream(), "op")]
        return int(ws[:16].view(torch.int32)[2].item())

    w48 = run(c48)
    e0 = epoch()
    run(c48, w48)
    assert epoch() == e0 + 64                                     # staged scoring + split selection: the token is spent all the same
    w49 = run(c49)
    assert epoch() == e0 + 128
    for _ in range(6):
        run(c48, w48)
        run(c49, w49)




def test_workgroups_sharing_a_compute_unit_are_neighbours_of_one_unit(dev):
    """The pairing of `score_fused` rests on the GPU's dispatch order (workgroup p and p + #CUs land on one compute unit when the
    launch has more workgroups than units).  Checked HERE, on the machine the tests run on, through the placement hook: every
    compute unit that hosted two workgroups of a 512-workgroup launch hosted spans 2j and 2j + 1 of ONE unit; launches of up to 256
    workgroups put one workgroup on a unit and number them unit by unit."""
    import ctypes
    from fastkv_amd import ops
    from fastkv_amd._lib import load
    L = load()
    def placement(n_wgs):
        buf = (ctypes.c_uint32 * (4 * n_wgs))()
        assert L.fastkv_debug_fused_placement(1, ctypes.cast(buf, ctypes.c_void_p), 4 * n_wgs) == 0
        rows = [tuple(buf[4 * i:4 * i + 4]) for i in range(n_wgs)]
        # HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13; XCC_ID 3:0
        return [dict(where=(x & 15, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15), unit=u, span=sp) for h, x, u, sp in rows]
    try:
        assert L.fastkv_debug_fused_placement(1, None, 0) == 0
        L.fastkv_placement_violations(1)
        for S, Hkv, H, B, n_wgs in ((32768, 8, 32, 1, 512), (14695, 1, 8, 16, 512), (8192, 8, 32, 1, 256)):
            q, k, v = make_qkv(321, B, H, Hkv, S, 128, 8)
            qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
            ops.update_kv(qd, kd, vd, 8, 7, "maxpool", 2048, 0, "index")
            torch.cuda.synchronize()
            pl = placement(n_wgs)
            nspan = max(w["span"] for w in pl) + 1
            assert sorted((w["unit"], w["span"]) for w in pl) == [(u, sp) for u in range(n_wgs // nspan) for sp in range(nspan)], S   # each once
            by_cu = {}
            for w in pl:
                by_cu.setdefault(w["where"], []).append(w)
            if n_wgs > 256:
                assert len(by_cu) == 256 and all(len(ws) == 2 for ws in by_cu.values()), (S, len(by_cu))
                for ws in by_cu.values():
                    a, b2 = sorted(ws, key=lambda w: w["span"])
                    assert a["unit"] == b2["unit"] and a["span"] % 2 == 0 and b2["span"] == a["span"] + 1, (S, ws)
            else:
                assert len(by_cu) == n_wgs, (S, len(by_cu))          # one workgroup per compute unit
                assert [(w["unit"], w["span"]) for w in pl] == sorted((w["unit"], w["span"]) for w in pl)   # unit by unit in launch order
        # ... and the kernel's own check (every launch, every workgroup) agrees: nobody met another unit on its compute unit
        assert L.fastkv_placement_violations(0) == 0
    finally:
        L.fastkv_debug_fused_placement(0, None, 0)


def test_a_slow_entry_does_not_disturb_the_entries_beside_it(dev):
    """Regression (round 3): more workgroups than compute units = two workgroups per unit.  With the launch's linear order the
    partner of a workgroup belonged to ANOTHER entry; when that entry was slow in phase A (a NaN in its query window sends every
    tile through the vector-ALU redo) the partner ran its later phases beside it and produced wrong row sums / window-row sums in
    10-20 % of the launches (tools/repro_nan_mate.py; docs/HISTORY.md).  The kernel now pairs adjacent spans of ONE unit, which the
    hand-offs keep in step.  16 entries x 2 virtual heads x 15 (-> 16) spans = 512 workgroups; the NaN entry itself is NaN all over
    (as in the oracle), every other entry must be bit-exact, launch after launch."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    H, Hkv, S, D, W, ks, cap, tsp_len, n = 8, 1, 14695, 128, 8, 13, 8316, 10400, 16
    ins = [make_qkv(9197 + 100000 * j, 1, H, Hkv, S, D, W, peaked=50) for j in range(n)]
    q0 = ins[5][0].clone()
    q0[0, 3, S - 2, 9] = float("nan")
    ins[5] = (q0, ins[5][1], ins[5][2])
    wants = [O.update_kv(q, k, v, W, ks, "avgpool", cap, tsp_len, "index") for q, k, v in ins]
    qs, kks, vs = ([_to_dev(t[j], dev) for t in ins] for j in range(3))
    for rnd in range(40):
        got = ops.update_kv_entries(qs, kks, vs, W, ks, "avgpool", cap, tsp_len, "index", return_indices=True)
        torch.cuda.synchronize()
        for i, want in enumerate(wants):
            assert torch.equal(got[3][i:i + 1].cpu(), want[2]) and torch.equal(got[2][i:i + 1].cpu(), want[3]), (rnd, i)
            assert torch.equal(got[0][i].cpu().view(torch.int16), want[0].view(torch.int16)), (rnd, i)
    # the same through the batched entry point, scores compared element by element (NaN rows by their bit patterns)
    q, k, v = (torch.cat([t[j] for t in ins], dim=0) for j in range(3))
    want = O.update_kv(q, k, v, W, ks, "avgpool", cap, tsp_len, "index", return_scores=True)
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
    for rnd in range(20):
        got = ops.update_kv(qd, kd, vd, W, ks, "avgpool", cap, tsp_len, "index", return_indices=True, return_scores=True)
        assert torch.equal(got[4].cpu().view(torch.int16), want[4].view(torch.int16)), rnd
    from fastkv_amd._lib import load
    assert load().fastkv_placement_violations(0) == 0                     # (the placement check of every launch so far in this process)


def test_load_time_selftest_of_the_co_residency_fixes(dev):
    """fastkv_amd/selftest.py (FASTKV_SELFTEST=1 runs it behind the first workspace initialisation): 16 entries with one slow
    (NaN-window) entry, compressed together 100 times on the fp32 contract, against the same entries compressed one by one -- the
    product checking itself for the round-3 damage without the oracle.  0 differing launches; and the env-gated hook runs it."""
    from fastkv_amd import selftest
    assert selftest.co_residency(100, dev) == 0
    r = _child("""
import torch, time
from fastkv_amd import ops, selftest
q = torch.randn(1, 600, 8, 128, device='cuda').half().transpose(1, 2); k = torch.randn(1, 600, 2, 128, device='cuda').half().transpose(1, 2)
t0 = time.time(); ops.update_kv(q, k, k, 8, 7, 'avgpool', 64); torch.cuda.synchronize()
assert selftest._ran, 'the self-test did not run at the first workspace initialisation'
print('child ok', round(time.time() - t0, 2))
""", {"FASTKV_SELFTEST": "1"})
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_operator_over_separately_allocated_entries(dev):
    """fastkv_update_kv_ptrs_f16 / ops.update_kv_entries: the batch entries are separate tensors (the layers of a model whose
    compression was deferred), addressed through device-side pointer tables, processed in ONE launch sequence.  Every entry's
    result equals the oracle's for that entry alone: the 16 keep-all layers behind the TSP layer in score order (the case the
    call exists for), a selecting geometry with a TSP index, and output written into views of larger slabs."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    for n, (H, Hkv, S, D, W, ks, pooling, cap, tsp_len, order, slab) in ((16, (32, 8, 2048, 128, 8, 7, "maxpool", 2048, 0, "score", False)),
                                                                      (4, (32, 8, 4096, 128, 8, 7, "avgpool", 512, 1024, "score", True)),
                                                                      (3, (16, 2, 3001, 64, 8, 5, "maxpool", 300, 0, "index", False)),
                                                                      (4, (32, 8, 2048, 128, 8, 7, "maxpool", 2048, 0, "index", True))):
        ins = [make_qkv(500 + 10 * n + i, 1, H, Hkv, S, D, W) for i in range(n)]
        qs, kks, vs = ([_to_dev(t[j], dev) for t in ins] for j in range(3))
        outs = None
        if slab:
            slabs = [torch.zeros(2, 1, Hkv, cap + 40, D, dtype=torch.float16, device=dev) for _ in range(n)]
            outs = ([s_[0, :, :, :cap] for s_ in slabs], [s_[1, :, :, :cap] for s_ in slabs])
        got = ops.update_kv_entries(qs, kks, vs, W, ks, pooling, cap, tsp_len, order, outs=outs, return_indices=True)
        torch.cuda.synchronize()
        for i, (q, k, v) in enumerate(ins):
            want = O.update_kv(q, k, v, W, ks, pooling, cap, tsp_len, order)
            assert torch.equal(got[0][i].cpu(), want[0]) and torch.equal(got[1][i].cpu(), want[1]), (n, i)
            assert torch.equal(got[3][i:i + 1].cpu(), want[2]), (n, i)
            assert (got[2] is None and want[3] is None) or torch.equal(got[2][i:i + 1].cpu(), want[3]), (n, i)
        if slab:
            assert all(float(s_[:, :, :, cap:].abs().max()) == 0.0 for s_ in slabs)         # nothing written past the views
    # entries with a batch dimension: every batch row of every tensor is an entry of the library call (deferred compression of a
    # batched prompt); rows i*B .. (i+1)*B-1 of the index tensors belong to tensor i
    ins = [make_qkv(650 + i, 2, 32, 8, 2048, 128, 8) for i in range(3)]
    qs, kks, vs = ([_to_dev(t[j], dev) for t in ins] for j in range(3))
    got = ops.update_kv_entries(qs, kks, vs, 8, 7, "maxpool", 300, 600, "score", return_indices=True)
    torch.cuda.synchronize()
    for i, (q, k, v) in enumerate(ins):
        want = O.update_kv(q, k, v, 8, 7, "maxpool", 300, 600, "score")
        assert got[0][i].shape == (2, 8, 300, 128)
        assert torch.equal(got[0][i].cpu(), want[0]) and torch.equal(got[1][i].cpu(), want[1]), i
        assert torch.equal(got[3][2 * i:2 * i + 2].cpu(), want[2]) and torch.equal(got[2][2 * i:2 * i + 2].cpu(), want[3]), i
    # two 32k layers per fused scoring launch (four tiles per wave) is what the residency limit allows: the library scores a longer
    # list in several launches (2 + 2 for four entries, 1 + 1 + 1 for three) and selects / copies all entries with one launch each.
    # The same with only the WINDOW rows of q kept (what a waiting layer of DeferredCompression holds).
    assert ops.fused_entries(32, 8, 32768, 128, 8, 7) == 2 and ops.fused_entries(32, 8, 2048, 128, 8, 7) >= 16
    assert ops.fused_entries(32, 8, 32768, 128, 4, 7) == 0                                             # window 4: the staged path
    ins = [make_qkv(700 + i, 1, 32, 8, 32768, 128, 8) for i in range(4)]
    qs, kks, vs = ([_to_dev(t[j], dev) for t in ins] for j in range(3))
    wants = [O.update_kv(*ins[i], 8, 7, "maxpool", 2048, 2048, "score") for i in range(4)]
    for n, qwin in ((2, False), (3, False), (4, False), (4, True)):
        qq = [ops.window_rows(t, 8) for t in qs[:n]] if qwin else qs[:n]
        got = ops.update_kv_entries(qq, kks[:n], vs[:n], 8, 7, "maxpool", 2048, 2048, "score", return_indices=True, q_window=qwin)
        torch.cuda.synchronize()
        for i in range(n):
            want = wants[i]
            assert torch.equal(got[0][i].cpu(), want[0]) and torch.equal(got[1][i].cpu(), want[1]), (n, qwin, i)
            assert torch.equal(got[3][i:i + 1].cpu(), want[2]) and torch.equal(got[2][i:i + 1].cpu(), want[3]), (n, qwin, i)
    # a geometry off the fused path is refused before anything is launched (window 4): the caller goes entry by entry
    ins = [make_qkv(900 + i, 1, 8, 2, 1000, 128, 4) for i in range(2)]
    qs, kks, vs = ([_to_dev(t[j], dev) for t in ins] for j in range(3))
    with pytest.raises(Exception) as ei:
        ops.update_kv_entries(qs, kks, vs, 4, 7, "maxpool", 128, 0, "score")
    assert "unsupported" in str(ei.value).lower() or "-3" in str(ei.value) or "UNSUPPORTED" in str(ei.value)


@pytest.mark.parametrize("shape", [
    dict(B=2, H=32, Hkv=8, S=32768, D=128, ks=7, pooling="avgpool", cap=2048, tsp=2048),     # two 32k layers of Llama-3-8B per launch
    dict(B=1, H=32, Hkv=8, S=60001, D=64, ks=5, pooling="maxpool", cap=3000, tsp=0),          # one long ragged prompt, head_dim 64
    dict(B=4, H=8, Hkv=2, S=40000, D=256, ks=7, pooling="avgpool", cap=700, tsp=900),         # head_dim 256, a batch
])
def test_four_tiles_per_wave_bit_exact(shape, dev):
    """score_fused_kernel<D,4,NB,1>: a wave owns four tiles; the exponentials of two of them wait for phase C in LDS (the other two
    in registers), the window-row sums are kept as fp16 bits.  Scores, indices, rows and the TSP index equal the oracle's, also with
    non-finite values in K and in a window row of Q (the general softmax path of such tiles)."""
    from fastkv_amd import ops
    from oracle import fastkv_oracle as O
    s = shape
    q, k, v = make_qkv(8100 + s["B"], s["B"], s["H"], s["Hkv"], s["S"], s["D"], 8)
    for special in (False, True):
        if special:
            k, q = k.clone(), q.clone()
            k[0, 0, 1000, 5] = float("inf")
            k[-1, 1, s["S"] - 3000, 1] = float("nan")
            k[0, -1, 64:90] *= 1e-5
            q[-1, 3, s["S"] - 2, 7] = float("inf")
        qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))
        for order in ("score", "index"):
            want = O.update_kv(q, k, v, 8, s["ks"], s["pooling"], s["cap"], s["tsp"], order, return_scores=True)
            got = ops.update_kv(qd, kd, vd, 8, s["ks"], s["pooling"], s["cap"], s["tsp"], order, return_indices=True, return_scores=True)
            torch.cuda.synchronize()
            assert torch.equal(got[4].cpu().view(torch.int16), want[4].view(torch.int16)), (special, order)
            assert torch.equal(got[3].cpu(), want[2]), (special, order)
            assert torch.equal(got[0].cpu().view(torch.int16), want[0].view(torch.int16)), (special, order)
            assert torch.equal(got[1].cpu().view(torch.int16), want[1].view(torch.int16)), (special, order)
            if s["tsp"]:
                assert torch.equal(got[2].cpu(), want[3]), (special, order)


def test_index_bounds_debug_mode(dev):
    """The index-bounds debug mode SURVEY.md 5 plans in place of a GPU sanitizer (FASTKV_DEBUG_BOUNDS=1, or a build with
    -DFK_DEBUG_BOUNDS): every gather -- compact_kv, gather_rows -- REPORTS an index outside [0, S) (FASTKV_EBOUNDS from
    fastkv_last_status) besides clamping it.  The randomised-shapes case, the entries / keep-all cases and the slab-cache wiring run
    in that mode in a child process: no index the selection hands to a gather is ever out of range; a deliberately bad index is
    reported in that mode and silently clamped in the product mode."""
    code = """
import sys, pytest, torch
rc = pytest.main(['tests/test_hip_parity.py', 'tests/test_wiring_gpu.py', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider', '-k',
                  'test_randomised_shapes_bit_exact or test_operator_over_separately_allocated_entries or test_keep_all_layers_bit_exact or test_slab_cache_in_place'])
from fastkv_amd import ops
from fastkv_amd._lib import load
torch.cuda.synchronize()
st = load().fastkv_last_status()
print('status after the cases:', st)
x = torch.arange(64 * 128, dtype=torch.float16, device='cuda').view(1, 64, 128)
y = ops.gather_rows(x, torch.tensor([[3, 64]], device='cuda'))         # 64 is out of range: reads the clamped row 63
torch.cuda.synchronize()
print('status after the bad index:', load().fastkv_last_status(), 'clamped:', bool(torch.equal(y[0, 1], x[0, 63])))
sys.exit(int(rc) or (0 if st == 0 else 9))
"""
    r = _child(code, {"FASTKV_DEBUG_BOUNDS": "1"})
    assert r.returncode == 0 and " passed" in r.stdout and "status after the cases: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "status after the bad index: -7 clamped: True" in r.stdout, r.stdout[-500:]
    r = _child(code.replace("rc = pytest.main", "rc = 0 and pytest.main"), {"FASTKV_DEBUG_BOUNDS": "0"})       # product mode: clamped, not reported
    assert r.returncode == 0 and "status after the bad index: 0 clamped: True" in r.stdout, r.stdout[-1000:] + r.stderr[-1000:]


@pytest.mark.parametrize("shape", [
    dict(B=1, H=16, Hkv=8, S=32768, D=128, ks=7, pooling="maxpool", cap=2048, tsp=2048),      # G = 2 (Llama-3.2-1B-like grouping)
    dict(B=1, H=4, Hkv=4, S=777, D=64, ks=7, pooling="avgpool", cap=100, tsp=300),             # MHA, ragged, head_dim 64
    dict(B=2, H=6, Hkv=2, S=3000, D=128, ks=5, pooling="avgpool", cap=300, tsp=0),             # G = 3
    dict(B=1, H=8, Hkv=8, S=2048, D=128, ks=7, pooling="maxpool", cap=2048, tsp=0),            # MHA, every candidate kept
])
def test_small_groups_take_the_fused_kernel(shape, dev):
    """1-3 query heads per KV head (MHA, G = 2 models, the per-query-head rule's views): ONE zero-padded 32-row block per KV head on
    the fused scoring kernel (round 2: the staged three-kernel path with its logits round trip).  Scores, indices, rows and the TSP
    index equal the oracle's; the library's profiler confirms which kernel ran."""
    import ctypes
    from fastkv_amd import ops
    from fastkv_amd._lib import load
    from oracle import fastkv_oracle as O
    s = shape
    L = load()
    q, k, v = make_qkv(9100 + s["H"], s["B"], s["H"], s["Hkv"], s["S"], s["D"], 8)
    qd, kd, vd = (_to_dev(t, dev) for t in (q, k, v))

    def read():
        n = L.fastkv_profile_kernels()
        cnt, ms = (ctypes.c_int64 * n)(), (ctypes.c_double * n)()
        assert L.fastkv_profile_read(cnt, ms) == 0
        return {L.fastkv_profile_kernel_name(i).decode(): int(cnt[i]) for i in range(n)}

    for order in ("score", "index"):
        want = O.update_kv(q, k, v, 8, s["ks"], s["pooling"], s["cap"], s["tsp"], order, return_scores=True)
        read()
        L.fastkv_profile_enable(1)
        got = ops.update_kv(qd, kd, vd, 8, s["ks"], s["pooling"], s["cap"], s["tsp"], order, return_indices=True, return_scores=True)
        torch.cuda.synchronize()
        L.fastkv_profile_enable(0)
        ran = read()
        assert ran["score_fused"] == 1 and ran["score_logits"] == 0 and ran["row_stats"] == 0, ran
        assert torch.equal(got[4].cpu().view(torch.int16), want[4].view(torch.int16)), order
        assert torch.equal(got[3].cpu(), want[2]), order
        assert torch.equal(got[0].cpu(), want[0]) and torch.equal(got[1].cpu(), want[1]), order
        if s["tsp"]:
            assert torch.equal(got[2].cpu(), want[3]), order
