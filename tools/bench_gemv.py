"""GEMV of the decode step on the MI355X: ops.decode_gemv (csrc/gemv.hip) against torch.nn.functional.linear (hipBLASLt) at the
Llama-3-8B shapes, HIP-event timing of graph-free back-to-back launches over ROTATING weight copies (so that nothing is
served from the L2 / Infinity Cache).  usage: python tools/bench_gemv.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastkv_amd import ops

dev = torch.device("cuda:0")
SHAPES = [("qkv+norm", 4096, [4096, 1024, 1024], True, False, False), ("o+res", 4096, [4096], False, False, True),
          ("gate/up+norm+silu", 4096, [14336, 14336], True, True, False), ("down+res", 14336, [4096], False, False, True),
          ("lm_head", 4096, [128256], False, False, False)]
COPIES = 12


def timeit(fn, n):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, K, rows, norm, glu, res in SHAPES:
    copies = COPIES if sum(rows) < 100000 else 3
    ws = [[(torch.randn(n, K, device=dev, dtype=torch.float16) * K ** -0.5) for n in rows] for _ in range(copies)]
    x = torch.randn(1, 1, K, device=dev, dtype=torch.float16)
    nw = torch.ones(K, device=dev, dtype=torch.float16) if norm else None
    n_out = rows[0] if glu else sum(rows)
    r = torch.randn(1, 1, n_out, device=dev, dtype=torch.float16) if res else None
    nbytes = sum(rows) * K * 2
    t_new = timeit(lambda i: ops.decode_gemv(x, ws[i % copies], norm_weight=nw, eps=1e-5, glu=glu, residual=r), 60)

    def stock(i):
        w = ws[i % copies]
        xin = ops.decode_rmsnorm(x, nw, 1e-5) if norm else x
        ys = [torch.nn.functional.linear(xin, m) for m in w]
        y = ops.decode_silu_mul(ys[0], ys[1]) if glu else ys[0]
        return r + y if res else y

    t_old = timeit(stock, 60)
    print(f"{name:20s} {nbytes / 1e6:8.1f} MB  decode_gemv {t_new:8.1f} us = {nbytes / t_new / 1e6:5.2f} TB/s   stock modules {t_old:8.1f} us = {nbytes / t_old / 1e6:5.2f} TB/s", flush=True)
