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


# ---------------------------------------------------------------------------------------------- the two contraction contracts
# oracle/fastkv_oracle.c "the contraction": "fmaf" = the fp32 fma chain (HIP engines "valu" / "mfma"; the default of both sides since
# round 6), "mfma16" = the gfx950 fp16 matrix instruction on the fp16 operands (HIP engine "mfma16"; FASTKV_CONTRACTION=mfma16).
# Against the REFERENCE (round 6, stage by stage: tests/test_oracle_golden.py): the fma chain reproduces the reference's fp16 LOGITS bit
# for bit (0 of 1.0e9 over the wide sweep) and what is left at the SCORES -- 3.3e-4 of the elements, 2 ulp at most -- is the softmax
# denominator's summation order alone (the oracle's reference-order mode removes all of it); the matrix instruction moves 1e-3 of the
# logits by an ulp before the softmax starts: 8.8e-4 of the scores, up to 6 ulp on PEAKED inputs.  The gates below hold on the GOLDEN
# cases (fmaf is SURVEY 8(c)'s own gate, mfma16 the wider one that contract needs); the wide sweep has its own bounds
# (tests/test_oracle_golden.py WIDE_BOUNDS).  The index-level protocol is the same for both.
CONTRACTIONS = ("fmaf", "mfma16")
ENGINES_OF = {"fmaf": ("mfma", "valu"), "mfma16": ("mfma16",)}
CONTRACTION_OF_ENGINE = {"valu": "fmaf", "mfma": "fmaf", "mfma16": "mfma16"}
SCORE_GATES = {"fmaf": dict(max_ulp=1, frac=0.001, floor=1), "mfma16": dict(max_ulp=2, frac=0.002, floor=3)}


def default_contraction() -> str:
    """What "auto" means on the HIP side of THIS process (FASTKV_CONTRACTION; fmaf unless set to mfma16) -- and the oracle's default."""
    return "mfma16" if os.environ.get("FASTKV_CONTRACTION", "fmaf")[:1] in ("m", "M") else "fmaf"


def assert_score_parity(got: torch.Tensor, ref: torch.Tensor, contraction: str, what: str = "") -> None:
    """Scores of an implementation under `contraction` vs the REFERENCE's scores: within the contract's measured gate."""
    g = SCORE_GATES[contraction]
    d = ulp_diff(got, ref)
    nbad = int((d > 0).sum())
    assert int(d.max()) <= g["max_ulp"], (what, contraction, int(d.max()))
    assert nbad <= max(g["floor"], int(g["frac"] * d.numel())), (what, contraction, nbad, d.numel())
