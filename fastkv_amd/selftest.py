"""Load-time self-test of the co-residency fixes (docs/HISTORY.md, tools/probes/README.md "Erratum note").

Round 3 found a silent wrong answer when a workgroup of the fused scoring kernel ran its vector phases beside ANOTHER head's workgroup
that was still in the fp32-contract matrix phase on the same compute unit; the trigger in the wild was one entry of a multi-entry
launch whose query window held a NaN (its tiles take the slow vector-ALU redo).  Two fixes closed it empirically (no packed-fp32
instructions; partners on a compute unit are adjacent spans of one head).  A compiler, driver or firmware change could re-open it
without any test noticing, so the product can check ITSELF -- without the oracle, which it must never import: the same entries are
compressed once together (one launch sequence: co-resident workgroups) and once one by one (nobody shares a compute unit with
another entry), on the fp32 contract, and the two must agree bit for bit.

    FASTKV_SELFTEST=1      run `co_residency()` once per process, behind the first workspace initialisation (~50 ms); a difference
                           raises FastKVNativeError.  CI runs and tests/test_hip_parity.py call it directly.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import FastKVNativeError


def co_residency(launches: int = 100, device=None, seed: int = 197) -> int:
    """Returns the number of launches (of `launches`) in which an entry's compressed rows differed between the multi-entry launch
    sequence and the entry-by-entry reference.  0 on a healthy build."""
    dev = torch.device(device if device is not None else "cuda:0")
    n, H, Hkv, S, D, W, ks, cap = 16, 16, 2, 14695, 128, 8, 7, 512          # the stress case that exposed it: 16 entries, G = 8
    g = torch.Generator(device="cpu").manual_seed(seed)
    qs, ks_, vs = [], [], []
    for i in range(n):
        qw = torch.randn(1, W, H, D, generator=g).half()
        if i == 0:
            qw[0, 3, 5, 17] = float("nan")                                  # the slow entry: every tile of its heads takes the redo path
        qs.append(qw.to(dev).transpose(1, 2).contiguous())                  # [1,H,W,D] window rows (q_window entries)
        ks_.append(torch.randn(1, S, Hkv, D, generator=g).half().to(dev).transpose(1, 2))
        vs.append(torch.randn(1, S, Hkv, D, generator=g).half().to(dev).transpose(1, 2))
    saved = ops._engine
    ops.set_score_engine("mfma")                                            # the contract that has the matrix phase in question
    try:
        ref = []
        for i in range(n):
            ko, vo, _, idx = ops.update_kv(qs[i], ks_[i], vs[i], W, ks, "maxpool", cap, 0, "score", return_indices=True, q_window=True)
            ref.append((ko.clone(), vo.clone(), idx.clone()))
        torch.cuda.synchronize(dev)
        bad = 0
        for _ in range(launches):
            k_outs, v_outs, _, idx = ops.update_kv_entries(qs, ks_, vs, W, ks, "maxpool", cap, 0, "score", return_indices=True, q_window=True)
            same = all(torch.equal(k_outs[i].view(torch.int16), ref[i][0].view(torch.int16)) and
                       torch.equal(v_outs[i].view(torch.int16), ref[i][1].view(torch.int16)) and torch.equal(idx[i:i + 1], ref[i][2])
                       for i in range(1, n))                                # (entry 0 is NaN everywhere: compared by its indices only)
            same = same and torch.equal(idx[0:1], ref[0][2])
            bad += 0 if same else 1
        torch.cuda.synchronize(dev)
    finally:
        ops._engine = saved
    return bad


_ran = False


def maybe_run_at_load(device) -> None:
    """Called behind the first operator-workspace initialisation when FASTKV_SELFTEST=1."""
    global _ran
    if _ran:
        return
    _ran = True
    bad = co_residency(100, device)
    if bad:
        raise FastKVNativeError(f"fastkv_amd self-test: {bad} of 100 multi-entry launches differ from the entry-by-entry results "
                                "(co-residency damage, tools/probes/README.md \"Erratum note\"): do not trust this build / driver")
