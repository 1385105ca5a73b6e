"""One-off stress run for the decode-step kernels (not part of the suite): random shapes of ops.decode_step_attention (against an
fp32 reference over apply_rotary_pos_emb's tensors; the slab rows it writes bit for bit) and of ops.decode_gemv (against an fp64
evaluation of the stock modules' rounding sequence), in one process so that the arrival counters and workspaces are reused
across shapes.  usage: python tools/stress_decode.py [N] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from transformers.models.llama import modeling_llama as ML
from fastkv_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
fails, t0 = 0, time.time()
for it in range(N):
    if rng.random() < 0.5:
        # ---- attention step
        D = rng.choice([64, 128, 128, 256]); G = rng.choice([1, 2, 4, 8]); Hkv = rng.choice([1, 2, 8]); B = rng.choice([1, 1, 2])
        H = Hkv * G
        L0 = rng.choice([0, 1, 63, 64, rng.randint(2, 5000)]); rows = L0 + rng.randint(1, 70)
        nsplit = rng.choice([0, 1, 3, 16, 33, 64])
        kslab = torch.randn(B, Hkv, rows, D, generator=g, device=dev, dtype=torch.float16)
        vslab = torch.randn(B, Hkv, rows, D, generator=g, device=dev, dtype=torch.float16)
        len_dev = torch.tensor([L0], dtype=torch.int32, device=dev)
        steps = min(3, rows - L0)
        for step in range(steps):
            qkv = torch.randn(B, 1, (H + 2 * Hkv) * D, generator=g, device=dev, dtype=torch.float16)
            q = qkv[..., :H * D].view(B, 1, H, D).transpose(1, 2)
            k = qkv[..., H * D:(H + Hkv) * D].view(B, 1, Hkv, D).transpose(1, 2)
            v = qkv[..., (H + Hkv) * D:].view(B, 1, Hkv, D).transpose(1, 2)
            ang = torch.rand(B, 1, D // 2, generator=g, device=dev) * 6.28
            cos, sin = torch.cat([ang.cos(), ang.cos()], -1).half(), torch.cat([ang.sin(), ang.sin()], -1).half()
            wq, wk = ML.apply_rotary_pos_emb(q, k, cos, sin)
            out = ops.decode_step_attention(q, k, v, cos, sin, kslab, vslab, len_dev, D ** -0.5, nsplit=nsplit)
            torch.cuda.synchronize()
            L = L0 + step + 1
            kk = kslab[:, :, :L].float().repeat_interleave(G, dim=1); vv = vslab[:, :, :L].float().repeat_interleave(G, dim=1)
            p = torch.softmax(torch.einsum("bhqd,bhkd->bhqk", wq.float(), kk) * D ** -0.5, dim=-1)
            ref = torch.einsum("bhqk,bhkd->bhqd", p, vv).transpose(1, 2).reshape(B, 1, H * D)
            err = float((out.float() - ref).abs().max()); tol = 2e-3 * float(ref.abs().max()) + 1e-3
            ok = int(len_dev.item()) == L and torch.equal(kslab[:, :, L - 1], wk[:, :, 0]) and torch.equal(vslab[:, :, L - 1], v[:, :, 0]) and err <= tol
            if not ok:
                fails += 1
                print("MISMATCH attention", dict(it=it, B=B, H=H, Hkv=Hkv, D=D, L0=L0, rows=rows, nsplit=nsplit, step=step, err=err, tol=tol, len=int(len_dev.item())), flush=True)
                break
    else:
        # ---- GEMV
        B = rng.choice([1, 1, 2, 4]); K = 512 * rng.choice([1, 2, 3, 7, 8, 8, 16, 28])
        if B * K * 2 > 65536 - 256:                           # the input rows live in LDS beside a few static words
            B = 1
        glu = rng.random() < 0.3
        nm = 2 if glu else rng.choice([1, 1, 2, 3])
        n0 = rng.choice([1, 5, 64, 1000, 4096, rng.randint(1, 20000)])
        rows_ = [n0, n0] if glu else [n0] + [rng.randint(1, 3000) for _ in range(nm - 1)]
        norm = rng.random() < 0.4; res = (not glu) and rng.random() < 0.4
        x = torch.randn(B, 1, K, generator=g, device=dev, dtype=torch.float16) * 2
        ws = [(torch.randn(n, K, generator=g, device=dev, dtype=torch.float16) * K ** -0.5) for n in rows_]
        nw = (torch.randn(K, generator=g, device=dev, dtype=torch.float16) * 0.5 + 1) if norm else None
        n_out = rows_[0] if glu else sum(rows_)
        r = torch.randn(B, 1, n_out, generator=g, device=dev, dtype=torch.float16) if res else None
        xin = x
        if norm:
            xf = x.float(); xin = nw * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).half()
        y64 = [(xin.double() @ w.double().t()).half() for w in ws]
        want = (torch.nn.functional.silu(y64[0].float()).half().float() * y64[1].float()).half() if glu else torch.cat(y64, dim=-1)
        if res:
            want = (r.float() + want.float()).half()
        got = ops.decode_gemv(x, ws, norm_weight=nw, eps=1e-5, glu=glu, residual=r)
        torch.cuda.synchronize()
        scale = float(want.float().abs().max())
        err = float((got.float() - want.float()).abs().max())
        if not (got.shape == want.shape and err <= 3e-3 * scale + 1e-3):
            fails += 1
            print("MISMATCH gemv", dict(it=it, B=B, K=K, rows=rows_, norm=norm, glu=glu, res=res, err=err, scale=scale), flush=True)
cnt = ops._step_counters.get((dev.index, ops._stream()))
if cnt is not None and int(cnt[0]) != 0:
    fails += 1
    print("arrival counters not back at zero", flush=True)
print(f"{N} cases, {fails} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
