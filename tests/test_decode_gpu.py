"""Decode over the compressed slab cache on the MI355X: the HIP append / GQA decode-attention kernels
against a plain PyTorch fp32 reference of the same step, and the graph-replayed greedy decode of benchmark/e2e.py against
the eager DynamicCache path of the same model (/root/reference/benchmark/e2e.py:72-93 is the loop being reproduced).

Tolerance, not bit equality: the kernel sums the softmax in another order than SDPA (slices per KV head, online softmax per
wave); outputs are fp16, so |diff| <= 2e-3 * max|ref| + 1e-3 is a few fp16 ulps."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_step(q, kslab, vslab, L, scaling):
    """fp32 attention of q [B,H,1,D] over rows 0..L-1 of the slabs, GQA by head repetition."""
    B, H, _, D = q.shape
    Hkv = kslab.shape[1]
    k = kslab[:, :, :L].float().repeat_interleave(H // Hkv, dim=1)
    v = vslab[:, :, :L].float().repeat_interleave(H // Hkv, dim=1)
    s = torch.einsum("bhqd,bhkd->bhqk", q.float(), k) * scaling
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("bhqk,bhkd->bhqd", p, v)                      # [B,H,1,D]
    return o.transpose(1, 2).reshape(B, 1, H * D)


@pytest.mark.parametrize("B,H,Hkv,D,L0,rows", [
    (1, 32, 8, 128, 2048, 2304),       # Llama-3-8B layer after a budget-2048 prefill
    (1, 32, 8, 128, 1, 64),            # a cache of one row
    (2, 8, 8, 64, 63, 200),            # MHA, head_dim 64, batch 2, ragged slice ends
    (1, 8, 1, 128, 3276, 3500),        # Llama-3-70B TP rank (G = 8), the proportional recipe's capacity
    (1, 16, 8, 256, 130, 256),         # G = 2, head_dim 256
])
def test_decode_kernels_match_fp32_reference(B, H, Hkv, D, L0, rows):
    from fastkv_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(B * 1000 + L0)
    kslab = torch.randn(B, Hkv, rows, D, generator=g, device=dev, dtype=torch.float16)
    vslab = torch.randn(B, Hkv, rows, D, generator=g, device=dev, dtype=torch.float16)
    untouched_k = kslab.clone()
    len_dev = torch.tensor([L0], dtype=torch.int32, device=dev)
    scaling = D ** -0.5
    for step in range(3):
        # q / k_new / v_new exactly as the attention module hands them over: [B,1,h,D] storage viewed as [B,h,1,D]
        q = torch.randn(B, 1, H, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
        kn = torch.randn(B, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
        vn = torch.randn(B, 1, Hkv, D, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
        for nsplit in (0, 5):
            if nsplit:                                              # second pass over the same step: rewind the length
                len_dev.fill_(L0 + step)
            ops.decode_append(kslab, vslab, kn, vn, len_dev)
            out = ops.decode_attention(q, kslab, vslab, len_dev, scaling, nsplit=nsplit)
            torch.cuda.synchronize()
            L = L0 + step + 1
            assert int(len_dev.item()) == L
            assert torch.equal(kslab[:, :, L - 1], kn[:, :, 0]) and torch.equal(vslab[:, :, L - 1], vn[:, :, 0])
            ref = _ref_step(q, kslab, vslab, L, scaling)
            tol = 2e-3 * float(ref.abs().max()) + 1e-3
            assert float((out.float() - ref).abs().max()) <= tol, (step, nsplit, float((out.float() - ref).abs().max()), tol)
    assert torch.equal(kslab[:, :, :L0], untouched_k[:, :, :L0]) and torch.equal(kslab[:, :, L0 + 3:], untouched_k[:, :, L0 + 3:])


def test_graph_replayed_decode_matches_eager_dynamic_cache(monkeypatch):
    """Two layers of the Llama-3-8B geometry, 700-token prompt compressed to 128 rows per layer, then 6 greedy decode steps:
    (a) eager over DynamicCache with SDPA (the reference's loop), (b) benchmark.e2e.graph_decode over the slab cache (HIP
    decode kernels, ONE captured step replayed).  Same tokens are fed to both (teacher forcing with (a)'s tokens would hide
    nothing here: the check is on the logits of every step), logits agree to fp16 tolerance, the caches hold the same rows."""
    from baselines.monkeypatch import replace_llama, set_model
    from benchmark import e2e, prefill

    def build(slab):
        monkeypatch.setenv("FASTKV_SLAB_CACHE", slab)
        a = prefill.parse_args(["--model_path", "llama3-8b", "--num_layers", "2", "--device", "cuda", "--save_txt", "", "--method",
                                "fastkv", "--max_capacity_prompts", "128", "--tsp_len", "256", "--tsp_idx", "0"])
        a.save_txt = False
        a.context_lengths = [700]
        replace_llama("fastkv")
        torch.manual_seed(3)
        model = prefill.build_model(a, "cuda")
        set_model(model, a)
        return model

    ids = torch.randint(0, 1000, (1, 700), generator=torch.Generator().manual_seed(5)).cuda()
    steps = 6
    # (a) eager, DynamicCache
    model = build("0")
    logits_a, toks_a = [], []
    with torch.no_grad():
        out = model(ids, attention_mask=torch.ones_like(ids))
        pkv = out.past_key_values
        tok = out.logits[:, -1].argmax(-1, keepdim=True)
        first = tok.clone()
        for _ in range(steps):
            out = model(input_ids=tok, past_key_values=pkv)          # positions restart at the compressed length (e2e.py:82-90)
            logits_a.append(out.logits[:, -1].float().cpu())
            tok = out.logits[:, -1].argmax(-1, keepdim=True)
            toks_a.append(int(tok[0, 0]))
    keys_a = [pkv.layers[i].keys.clone() for i in range(2)]
    del model, pkv
    # (b) slab cache + graph replay; capture the logits of every step through a hook on lm_head
    model = build("1")
    logits_b = []
    hook = model.lm_head.register_forward_hook(lambda m, i, o: logits_b.append(o[:, -1].float().clone()))
    with torch.no_grad():
        out = model(ids, attention_mask=torch.ones_like(ids))
        pkv = out.past_key_values
        assert torch.equal(out.logits[:, -1].argmax(-1, keepdim=True), first)
        logits_b.clear()

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            return 0.0, None

        _, toks_b = e2e.graph_decode(model, pkv, first, steps, timed)
    hook.remove()
    # the hook fired once for the eager step, once during the capture (garbage: nothing runs then) and the graph's output
    # buffer holds the LAST replayed step afterwards; the tokens are recorded on the device for every step
    assert len(toks_b) == steps
    scale = max(float(l.abs().max()) for l in logits_a)
    assert float((logits_b[0].cpu() - logits_a[0]).abs().max()) <= 2e-2 * scale + 2e-3          # step 1 (eager, HIP kernels)
    assert float((logits_b[-1].cpu() - logits_a[-1]).abs().max()) <= 3e-2 * scale + 2e-3         # last replayed step
    # greedy tokens: identical unless two logits of a random-initialised model are within the tolerance of each other
    agree = sum(int(x == y) for x, y in zip(toks_a, toks_b))
    assert toks_b[0] == toks_a[0] and agree >= steps - 2, (toks_a, toks_b)
    for i in range(2):
        kb = pkv.layers[i].keys
        assert kb.shape == keys_a[i].shape == (1, 8, 128 + steps, 128)
        assert torch.equal(kb[:, :, :128], keys_a[i][:, :, :128])                                # the compacted prefill rows
        if agree == steps:
            assert float((kb[:, :, 128:].float() - keys_a[i][:, :, 128:].float()).abs().max()) <= 5e-2


def test_single_launch_step_operators_match_the_stock_modules():
    """decode_rmsnorm / decode_rope_ / decode_silu_mul against the modules they stand in for during a static decode step
    (transformers' LlamaRMSNorm, apply_rotary_pos_emb, LlamaMLP's act_fn(gate) * up), fp16 on the GPU.  RoPE repeats the stock
    fp16 rounding sequence (bit-exact); the other two differ at most by an fp16 ulp where an fp32 intermediate rounds the
    other way."""
    from transformers.models.llama import modeling_llama as ML
    from fastkv_amd import ops
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(2, 1, 4096, generator=g, device=dev, dtype=torch.float16) * 3
    norm = ML.LlamaRMSNorm(4096, eps=1e-5).to(dev).half()
    norm.weight.data = torch.randn(4096, generator=g, device=dev, dtype=torch.float16)
    want, got = norm(x), ops.decode_rmsnorm(x, norm.weight, norm.variance_epsilon)
    assert float((got.float() - want.float()).abs().max()) <= 2e-3 * float(want.abs().max())
    assert float((got != want).float().mean()) < 0.02
    # rope
    q = torch.randn(2, 1, 32, 128, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
    k = torch.randn(2, 1, 8, 128, generator=g, device=dev, dtype=torch.float16).transpose(1, 2)
    ang = torch.rand(2, 1, 64, generator=g, device=dev) * 6.28
    cos, sin = torch.cat([ang.cos(), ang.cos()], -1).half(), torch.cat([ang.sin(), ang.sin()], -1).half()
    wq, wk = ML.apply_rotary_pos_emb(q, k, cos, sin)
    q2, k2 = q.clone(), k.clone()
    ops.decode_rope_(q2, k2, cos, sin)
    assert torch.equal(q2, wq) and torch.equal(k2, wk)
    # silu * up
    a = torch.randn(2, 1, 14336, generator=g, device=dev, dtype=torch.float16) * 2
    b = torch.randn(2, 1, 14336, generator=g, device=dev, dtype=torch.float16)
    want = torch.nn.functional.silu(a) * b
    got = ops.decode_silu_mul(a, b)
    assert float((got.float() - want.float()).abs().max()) <= 2e-3 * float(want.abs().max())
    assert float((got != want).float().mean()) < 0.02
    torch.set_grad_enabled(True)


@pytest.mark.parametrize("B,K,rows,norm,glu,res", [
    (1, 4096, [4096, 1024, 1024], True, False, False),     # RMSNorm + q/k/v of Llama-3-8B
    (1, 4096, [4096], False, False, True),                 # o_proj + residual
    (1, 4096, [14336, 14336], True, True, False),          # RMSNorm + gate/up + SiLU*up
    (1, 14336, [4096], False, False, True),                # down_proj + residual (28 segments: the one-segment tail loop runs)
    (2, 1536, [1000, 24], True, False, True),              # ragged row counts, three segments, batch 2
    (4, 1024, [77], False, False, False),                  # batch 4
    (1, 4096, [128256], False, False, False),              # lm_head
    (2, 512, [6, 6], False, True, False),                  # a single segment, fewer columns than one wave handles
])
def test_decode_gemv_matches_stock_modules(B, K, rows, norm, glu, res):
    """ops.decode_gemv against the stock fp16 modules it stands in for (nn.Linear / LlamaRMSNorm / silu * up / residual add),
    evaluated by PyTorch on the GPU.  Same rounding points, another accumulation order: the fp32 accumulators differ by
    ~1e-3 relative at K = 4096, so outputs agree to a few fp16 ulps of the row's scale -- and exactly with an fp64
    evaluation of the same rounding sequence to half that."""
    from transformers.models.llama import modeling_llama as ML
    from fastkv_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(K + B)
    x = torch.randn(B, 1, K, generator=g, device=dev, dtype=torch.float16) * 2
    ws = [(torch.randn(n, K, generator=g, device=dev, dtype=torch.float16) * K ** -0.5).contiguous() for n in rows]
    nw = (torch.randn(K, generator=g, device=dev, dtype=torch.float16) * 0.5 + 1).contiguous() if norm else None
    n_out = rows[0] if glu else sum(rows)
    r = torch.randn(B, 1, n_out, generator=g, device=dev, dtype=torch.float16) if res else None
    with torch.no_grad():
        xin = x
        if norm:
            m = ML.LlamaRMSNorm(K, eps=1e-5).to(dev).half()
            m.weight.data = nw
            xin = m(x)
        ys = [torch.nn.functional.linear(xin, w) for w in ws]
        want = torch.nn.functional.silu(ys[0]) * ys[1] if glu else torch.cat(ys, dim=-1)
        if res:
            want = r + want
        # the same rounding sequence with exact (fp64) dot products
        y64 = [(xin.double() @ w.double().t()).half() for w in ws]
        want64 = (torch.nn.functional.silu(y64[0].float()).half().float() * y64[1].float()).half() if glu else torch.cat(y64, dim=-1)
        if res:
            want64 = (r.float() + want64.float()).half()
    got = ops.decode_gemv(x, ws, norm_weight=nw, eps=1e-5, glu=glu, residual=r)
    torch.cuda.synchronize()
    assert got.shape == want.shape
    scale = float(want64.float().abs().max())
    assert float((got.float() - want64.float()).abs().max()) <= 2e-3 * scale + 1e-3
    assert float((got.float() - want.float()).abs().max()) <= 4e-3 * scale + 1e-3
    assert float((got != want64).float().mean()) < 0.05


@pytest.mark.parametrize("B,H,Hkv,D,L0,rows", [
    (1, 32, 8, 128, 2048, 2304),       # Llama-3-8B layer after a budget-2048 prefill
    (1, 32, 8, 128, 0, 64),            # an empty cache: the step's own row is all there is
    (2, 8, 8, 64, 63, 200),            # MHA, head_dim 64, batch 2 (the new row opens a new 64-row tile)
    (1, 8, 1, 128, 3276, 3500),        # G = 8
    (1, 16, 8, 256, 130, 256),         # G = 2, head_dim 256
])
def test_fused_step_attention_matches_the_separate_kernels_and_fp32(B, H, Hkv, D, L0, rows):
    """ops.decode_step_attention (RoPE + append + attention + merge in one launch, arrival counters) over 4 steps, two slice
    counts: the slab rows it writes are bit-identical to apply_rotary_pos_emb's K row / the V row, the length advances, the
    counters are zero again, and the output matches an fp32 reference over the rotated tensors to fp16 tolerance."""
    from transformers.models.llama import modeling_llama as ML
    from fastkv_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(B * 77 + L0)
    kslab = torch.randn(B, Hkv, rows, D, generator=g, device=dev, dtype=torch.float16)
    vslab = torch.randn(B, Hkv, rows, D, generator=g, device=dev, dtype=torch.float16)
    # rows the cache does not hold yet are whatever the allocation held: NaN and Inf patterns must not reach a result (0 * NaN)
    kslab[:, :, L0 + 4:] = float("nan")
    vslab[:, :, L0 + 4:] = float("nan")
    vslab[:, :, L0 + 4::3] = float("inf")
    len_dev = torch.tensor([L0], dtype=torch.int32, device=dev)
    scaling = D ** -0.5
    for step in range(4):
        qkv = torch.randn(B, 1, (H + 2 * Hkv) * D, generator=g, device=dev, dtype=torch.float16)     # as the fused projection leaves it
        q = qkv[..., :H * D].view(B, 1, H, D).transpose(1, 2)
        k = qkv[..., H * D:(H + Hkv) * D].view(B, 1, Hkv, D).transpose(1, 2)
        v = qkv[..., (H + Hkv) * D:].view(B, 1, Hkv, D).transpose(1, 2)
        ang = torch.rand(B, 1, D // 2, generator=g, device=dev) * 6.28
        cos, sin = torch.cat([ang.cos(), ang.cos()], -1).half(), torch.cat([ang.sin(), ang.sin()], -1).half()
        wq, wk = ML.apply_rotary_pos_emb(q, k, cos, sin)
        nsplit = (0, 5, 1, 64)[step]
        out = ops.decode_step_attention(q, k, v, cos, sin, kslab, vslab, len_dev, scaling, nsplit=nsplit)
        torch.cuda.synchronize()
        L = L0 + step + 1
        assert int(len_dev.item()) == L
        assert torch.equal(kslab[:, :, L - 1], wk[:, :, 0]) and torch.equal(vslab[:, :, L - 1], v[:, :, 0])
        ref = _ref_step(wq, kslab, vslab, L, scaling)
        tol = 2e-3 * float(ref.abs().max()) + 1e-3
        assert float((out.float() - ref).abs().max()) <= tol, (step, float((out.float() - ref).abs().max()), tol)
    cnt = ops._step_counters[(dev.index, ops._stream())]
    assert int(cnt[0]) == 0 and int(cnt[2:].abs().sum()) == 0          # arrivals back at zero; word 1 is the launch epoch


@pytest.mark.parametrize("B,V", [(1, 128256), (2, 32000), (4, 4096), (1, 8)])
def test_greedy_step_tail_matches_torch_argmax(B, V):
    """ops.decode_greedy (argmax + next token + positions + token log in one launch) against torch.argmax on the same logits,
    incl. ties (the first maximal value wins), -inf rows, NaN (counts as maximal, as in torch); scratch words back at zero; several
    steps through one scratch / log."""
    from fastkv_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(V + B)
    scratch = ops.new_greedy_scratch(dev, B)
    tok = torch.zeros(B, 1, dtype=torch.int64, device=dev)
    pos = torch.full((B, 1), 100, dtype=torch.int64, device=dev)
    log = torch.full((8,), -1, dtype=torch.int64, device=dev)
    log_index = torch.zeros(1, dtype=torch.int64, device=dev)
    want_log = []
    for step in range(6):
        logits = torch.randn(B, 3, V, generator=g, device=dev, dtype=torch.float16)
        if step == 1:
            logits[:, -1, :] = logits[:, -1, :].round()                       # many ties
        if step == 2:
            logits[0, -1, :] = float("-inf")
        if step == 3 and V > 8:
            logits[-1, -1, V // 2] = float("nan")
            logits[-1, -1, V // 2 + 5] = float("nan")
        if step == 4:
            logits[0, -1, V - 1] = 100.0                                      # the last element
        if step == 5 and V > 16:
            # a row whose maximum is zero, a -0.0 in front of a +0.0: equal for torch.argmax, the first one wins
            logits[-1, -1, :] = -logits[-1, -1, :].abs() - 1.0
            logits[-1, -1, 11] = -0.0
            logits[-1, -1, 14] = 0.0
        ref = logits[:, -1, :].argmax(dim=-1, keepdim=True)
        ops.decode_greedy(logits, scratch, tok, pos, log, log_index)
        torch.cuda.synchronize()
        assert torch.equal(tok, ref), (step, tok.tolist(), ref.tolist())
        want_log.append(int(ref[0, 0]))
        assert int(pos[0, 0]) == 101 + step and int(log_index) == step + 1 and int(scratch.abs().sum()) == 0
    assert log[:6].tolist() == want_log and int(log[6]) == -1


@pytest.mark.parametrize("B,D,theta,rope", [(1, 128, 500000.0, "llama3"), (2, 64, 10000.0, "default"), (4, 128, 1000000.0, "default")])
def test_step_rotary_tables_match_the_stock_module(B, D, theta, rope):
    """ops.decode_rotary (one launch) against transformers' LlamaRotaryEmbedding.forward for one position per batch entry, bit for
    bit, at small and very large positions (llama3 rope scaling changes inv_freq only; the tables' arithmetic is the same)."""
    from transformers import LlamaConfig
    from transformers.models.llama import modeling_llama as ML
    from fastkv_amd import ops
    dev = torch.device("cuda:0")
    cfg = LlamaConfig(hidden_size=D * 4, num_attention_heads=4, num_key_value_heads=2, head_dim=D, max_position_embeddings=131072, rope_theta=theta)
    if rope == "llama3":
        cfg.rope_parameters = {"rope_type": "llama3", "rope_theta": theta, "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                               "original_max_position_embeddings": 8192}
    mod = ML.LlamaRotaryEmbedding(cfg).to(dev)
    x = torch.zeros(B, 1, D * 4, dtype=torch.float16, device=dev)
    for base in (0, 1, 2047, 32768, 131071, 1048575):
        pos = (torch.arange(B, device=dev, dtype=torch.int64) * 7 + base).view(B, 1).contiguous()
        cos, sin = mod(x, pos)
        gc, gs = ops.decode_rotary(mod.inv_freq, pos, float(mod.attention_scaling), D)
        torch.cuda.synchronize()
        assert torch.equal(gc.view(torch.int16), cos.view(torch.int16)) and torch.equal(gs.view(torch.int16), sin.view(torch.int16)), base
