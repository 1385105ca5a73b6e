"""Per-layer time of ops.decode_step_attention (csrc/decode_step.hip) at the shape of benchmark/e2e.py: Llama-3-8B geometry, 32 layers with
their own slabs (budget 2048 + room), one captured step = 32 dependent launches, replayed.  Prints us per launch."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fastkv_amd import ops

dev = torch.device("cuda:0")
B, H, Hkv, D = 1, 32, 8, 128
L0, rows, layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 2304, 32
nsplit = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = torch.Generator(device=dev).manual_seed(1)
slabs = [(torch.randn(B, Hkv, rows, D, generator=g, device=dev, dtype=torch.float16), torch.randn(B, Hkv, rows, D, generator=g, device=dev, dtype=torch.float16))
         for _ in range(layers)]
lens = [torch.tensor([L0], dtype=torch.int32, device=dev) for _ in range(layers)]
qkv = torch.randn(B, 1, (H + 2 * Hkv) * D, generator=g, device=dev, dtype=torch.float16)
q = qkv[..., :H * D].view(B, 1, H, D).transpose(1, 2)
k = qkv[..., H * D:(H + Hkv) * D].view(B, 1, Hkv, D).transpose(1, 2)
v = qkv[..., (H + Hkv) * D:].view(B, 1, Hkv, D).transpose(1, 2)
ang = torch.rand(B, 1, D // 2, generator=g, device=dev) * 6.28
cos, sin = torch.cat([ang.cos(), ang.cos()], -1).half(), torch.cat([ang.sin(), ang.sin()], -1).half()
cnt, ws = ops.new_step_counters(dev), ops.new_decode_workspace(dev, B, H, D)


def step():
    outs = []
    for (ks, vs), ld in zip(slabs, lens):
        outs.append(ops.decode_step_attention(q, k, v, cos, sin, ks, vs, ld, D ** -0.5, nsplit=nsplit, counters=cnt, workspace=ws))
    return outs


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step()
    torch.cuda.synchronize()
    for ld in lens:
        ld.fill_(L0)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        step()
    torch.cuda.synchronize()
    for ld in lens:
        ld.fill_(L0)
    steps = 64
    for _ in range(8):
        gr.replay()
    for ld in lens:
        ld.fill_(L0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(steps):
        gr.replay()
    e1.record(s)
    torch.cuda.synchronize()
print(f"L0 {L0} rows {rows} nsplit {nsplit}: {e0.elapsed_time(e1) * 1000 / steps / layers:.2f} us per launch (graph replay, {layers} dependent launches per step), "
      f"len after = {int(lens[0].item())}")
from fastkv_amd._lib import raise_if_aborted
raise_if_aborted()
