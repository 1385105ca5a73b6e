"""Shared helpers of the parity tests."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name: str):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def load_meta():
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)


def f16_from_bits(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).view(torch.float16)


def ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """|a - b| in fp16 ulps for same-sign finite values (scores are non-negative)."""
    return (a.contiguous().view(torch.int16).to(torch.int32) - b.contiguous().view(torch.int16).to(torch.int32)).abs()


def expected_kv(k: torch.Tensor, idx: torch.Tensor, window: int) -> torch.Tensor:
    """Rows `idx` of every head followed by the window rows (utils.py:114-121), on k's device."""
    B, Hkv, S, D = k.shape
    n = S - window
    sel = torch.gather(k[:, :, :n], 2, idx[..., None].expand(-1, -1, -1, D))
    return torch.cat([sel, k[:, :, n:]], dim=2)
