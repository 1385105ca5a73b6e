"""Deterministic synthetic Q/K/V for the parity tests and bench.py (build-owned, integer-only).

splitmix64 -> 16-bit uniforms -> Irwin-Hall(12) approximate N(0,1) -> fp16.  Only integer
arithmetic and one exact int->float64->fp16 conversion are involved, so the same seed gives the
same bits on every machine and numpy version (torch.randn / numpy Generators promise neither).

Layout follows /root/reference/baselines/fastkv/llama_model.py:117-122: tensors are allocated
`[B,S,H,D]` contiguous and handed over as the `[B,H,S,D]` transposed view.
"""
from __future__ import annotations

import numpy as np
import torch

_M = (1 << 64) - 1


def _splitmix64(start: int, count: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (np.arange(1, count + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(start & _M)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


_fast = None


def _fast_lib():
    """tests/gen_fast.c compiled on first use (gcc, a second); None if that is not possible -> the numpy version below."""
    global _fast
    if _fast is None:
        import ctypes
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        src, lib = os.path.join(here, "gen_fast.c"), os.path.join(here, "libgen_fast.so")
        try:
            if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
                tmp = lib + ".tmp.%d" % os.getpid()
                subprocess.check_call(["gcc", "-O3", "-mf16c", "-fopenmp", "-shared", "-fPIC", "-o", tmp, src])
                os.replace(tmp, lib)
            L = ctypes.CDLL(lib)
            L.gen_normal_f16.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_void_p]
            L.gen_normal_f16.restype = None
            _fast = L
        except Exception:   # noqa: BLE001 - no compiler: the numpy version is the definition anyway
            _fast = False
    return _fast or None


def normal_f16(seed: int, stream: int, count: int, scale: float = 1.0, use_c: bool = True) -> np.ndarray:
    """`count` approximately-N(0, scale^2) values as fp16, from (seed, stream)."""
    out = np.empty(count, dtype=np.float16)
    CH = 1 << 20
    base = ((seed * 0x100000001B3) ^ (stream * 0xD6E8FEB86659FD93)) & _M
    if use_c and scale == 1.0 and count > 0 and _fast_lib() is not None:
        _fast_lib().gen_normal_f16(base, count, out.ctypes.data)          # bit-identical twin of the loop below
        return out
    for lo in range(0, count, CH):
        m = min(CH, count - lo)
        r = _splitmix64(base + 3 * lo * 0x9E3779B97F4A7C15, 3 * m).reshape(m, 3)
        acc = np.zeros(m, dtype=np.int64)
        for c in range(3):
            w = r[:, c]
            for s in (0, 16, 32, 48):
                acc += ((w >> np.uint64(s)) & np.uint64(0xFFFF)).astype(np.int64)
        x = (acc - 6 * 65536).astype(np.float64) / 65536.0          # variance 1 (12 uniforms)
        out[lo:lo + m] = (x * scale).astype(np.float16)
    return out


def make_qkv(seed: int, B: int, H: int, Hkv: int, S: int, D: int, window: int = 8, peaked: int = 0,
             alpha: float = 0.35, device: str = "cpu", full_q: bool = False):
    """Q [B,H,S,D], K,V [B,Hkv,S,D] fp16 as transposed views of `[B,S,H,D]` storage.

    Only the last `window` rows of Q are read by the path (utils.py:94); unless `full_q`, the
    other rows are zeros (saves generation time, same results).  `peaked > 0` plants that many
    heavy-hitter keys per KV head (K[p] += alpha * sum of the group's window queries) so that the
    top-k has a clear margin.
    """
    q_phys = torch.zeros(B, S, H, D, dtype=torch.float16)
    nq = B * (S if full_q else window) * H * D
    qv = torch.from_numpy(normal_f16(seed, 1, nq)).view(B, -1, H, D)
    q_phys[:, -qv.shape[1]:] = qv
    k_phys = torch.from_numpy(normal_f16(seed, 2, B * S * Hkv * D)).view(B, S, Hkv, D).clone()
    v_phys = torch.from_numpy(normal_f16(seed, 3, B * S * Hkv * D)).view(B, S, Hkv, D).clone()
    if peaked:
        G = H // Hkv
        n = S - window
        pos = (_splitmix64(seed * 7919 + 17, B * Hkv * peaked) % np.uint64(n)).astype(np.int64).reshape(B, Hkv, peaked)
        qw = q_phys[:, -window:].float().view(B, window, Hkv, G, D).sum(dim=(1, 3))        # [B,Hkv,D]
        for b in range(B):
            for g in range(Hkv):
                p = torch.from_numpy(np.unique(pos[b, g]))
                k_phys[b, p, g] = (k_phys[b, p, g].float() + alpha * qw[b, g]).half()
    q, k, v = (t.transpose(1, 2) for t in (q_phys, k_phys, v_phys))
    if device != "cpu":
        q, k, v = (t.transpose(1, 2).to(device).transpose(1, 2) for t in (q, k, v))
    return q, k, v
