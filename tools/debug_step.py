"""Debug helper: ops.decode_step_attention against an fp32 reference for one shape at several slice counts."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fastkv_amd import ops
dev = torch.device("cuda:0")
B, H, Hkv, D, L0, rows = [int(a) for a in sys.argv[1:7]] if len(sys.argv) > 6 else (1, 32, 8, 128, 2048, 2304)
g = torch.Generator(device=dev).manual_seed(5)
for nsplit in [int(a) for a in sys.argv[7:]] or [1, 0, 5]:
    kslab = torch.randn(B, Hkv, rows, D, generator=g, device=dev, dtype=torch.float16)
    vslab = torch.randn(B, Hkv, rows, D, generator=g, device=dev, dtype=torch.float16)
    len_dev = torch.tensor([L0], dtype=torch.int32, device=dev)
    qkv = torch.randn(B, 1, (H + 2 * Hkv) * D, generator=g, device=dev, dtype=torch.float16)
    q = qkv[..., :H * D].view(B, 1, H, D).transpose(1, 2)
    k = qkv[..., H * D:(H + Hkv) * D].view(B, 1, Hkv, D).transpose(1, 2)
    v = qkv[..., (H + Hkv) * D:].view(B, 1, Hkv, D).transpose(1, 2)
    cos = torch.ones(B, 1, D, device=dev, dtype=torch.float16); sin = torch.zeros_like(cos)
    out = ops.decode_step_attention(q, k, v, cos, sin, kslab, vslab, len_dev, D ** -0.5, nsplit=nsplit)
    torch.cuda.synchronize()
    L = L0 + 1
    kk = kslab[:, :, :L].float().repeat_interleave(H // Hkv, dim=1); vv = vslab[:, :, :L].float().repeat_interleave(H // Hkv, dim=1)
    p = torch.softmax(torch.einsum("bhqd,bhkd->bhqk", q.float(), kk) * D ** -0.5, dim=-1)
    ref = torch.einsum("bhqk,bhkd->bhqd", p, vv).transpose(1, 2).reshape(B, 1, H * D)
    err = (out.float() - ref).abs().reshape(B, H, D)
    print(f"nsplit {nsplit}: max err {float(err.max()):.4g}; per head max:", [round(float(e), 3) for e in err.amax(dim=(0, 2))][:16], "row appended ok:", bool(torch.equal(kslab[:, :, L0], k[:, :, 0])))
    if float(err.max()) > 1e-2:
        h = int(err.amax(dim=(0, 2)).argmax()); print("   head", h, "err by dim (first 16):", [round(float(e), 3) for e in err[0, h, :16]], " got/ref ratio:", [round(float(a / b), 3) for a, b in zip(out.float().reshape(B, H, D)[0, h, :8], ref.reshape(B, H, D)[0, h, :8])])
from fastkv_amd._lib import raise_if_aborted
raise_if_aborted()
