"""Seed sweep at the graded length: how often does the oracle's <= 1-ulp score noise move a selected index?

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_sweep.py            # the 24 cases of rounds 3-4 -> sweep32k.npz, sweep_meta.json
    python tests/golden/make_sweep.py wide       # round 5: 120 cases (24 seeds x {constant, recipe} x {maxpool, avgpool} + 24 peaked)
                                                 # -> sweep_wide.npz (row digests; full rows only where a row flips), sweep_wide_meta.json
                                                 # (per-contraction rates with 95 % Wilson intervals, overall and per family)

For every case of tests/golden_cases.py:SWEEP_CASES (Llama-3-8B geometry, S = 32768; both poolings; the constant budget of
BASELINE.json configs[1] and the published proportional recipe) the REFERENCE (baselines/fastkv/utils.py:80-134, imported,
not copied) runs with the `Tensor.topk` spy of make_golden.py, which exposes its score tensors; the canonical top-k (value
descending, position ascending) of those scores is the index set an implementation must produce.  The oracle
(oracle/fastkv_oracle.c) runs on the same inputs, and the script records, per case and per row,
  * how many score elements differ between oracle and reference (always by 1 fp16 ulp),
  * whether one of them lies within 1 ulp of the row's k-th value (a "straddle": the only way an index can move),
  * how many indices actually differ (`flips`).
Fixtures: tests/golden/sweep32k.npz (canonical indices as uint16, tie metadata, every 64th reference score) and
tests/golden/sweep_meta.json (the statistics; rows that flip are LISTED there, never dropped -- the replay tests check that
exactly those rows differ and all others are identical).
"""
from __future__ import annotations

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]

import numpy as np
import torch

from gen_inputs import make_qkv
from golden_cases import SWEEP_CASES, SWEEP_WIDE_CASES
from baselines.fastkv.utils import FastKVCluster      # the reference

sys.path.append(ROOT)                                 # after the reference: only `oracle` is taken from the repository
from oracle import fastkv_oracle as O                 # noqa: E402

torch.set_num_threads(8)
O.set_threads(8)


STAGE_DIGESTS = {}          # case name -> (digest of the reference's logits, of its probabilities): the wide sweep's stage-level pin


def tensor_digest(t) -> int:
    """64-bit digest of a tensor's bytes (fp16 bit patterns)."""
    import hashlib
    return int.from_bytes(hashlib.blake2b(t.contiguous().view(torch.int16).numpy().tobytes(), digest_size=8).digest(), "little")


def run_reference(q, k, v, case, name=None):
    cl = FastKVCluster(window_size=case["W"], max_capacity_prompt=case["cap"], kernel_size=case["ks"],
                       pooling=case["pooling"], tsp_layer=True, tsp_length=case["tsp_len"])
    spied = []
    orig = torch.Tensor.topk
    orig_softmax = torch.nn.functional.softmax

    def spy(self, *a, **kw):
        spied.append(self.detach().clone())
        return orig(self, *a, **kw)

    def spy_softmax(x, *a, **kw):                               # utils.py:103: input = the logits after utils.py:94-101
        y = orig_softmax(x, *a, **kw)
        if name is not None:
            STAGE_DIGESTS[name] = (tensor_digest(x), tensor_digest(y.to(x.dtype)))
        return y

    torch.Tensor.topk = spy
    torch.nn.functional.softmax = spy_softmax
    try:
        cl.update_kv(k, q, v, None, q.shape[1] // k.shape[1], 0)
    finally:
        torch.Tensor.topk = orig
        torch.nn.functional.softmax = orig_softmax
    return spied[0], spied[1]


def canonical(row: torch.Tensor, k: int) -> torch.Tensor:
    srt = torch.sort(row.float(), descending=True, stable=True)
    return torch.sort(srt.indices[:k]).values, srt.values[k - 1]


def bits(t):
    return t.contiguous().view(torch.int16).to(torch.int32)


def row_digest(idx) -> int:
    """64-bit digest of an ascending index row (the wide fixture stores these instead of the rows)."""
    import hashlib
    return int.from_bytes(hashlib.blake2b(np.ascontiguousarray(np.asarray(idx, dtype=np.int32)).tobytes(), digest_size=8).digest(), "little")


def wilson(k, n, z=1.96):
    """95 % Wilson score interval of a proportion k / n."""
    if n == 0:
        return [0.0, 0.0]
    ph = k / n
    den = 1 + z * z / n
    mid = (ph + z * z / (2 * n)) / den
    half = z * ((ph * (1 - ph) / n + z * z / (4 * n * n)) ** 0.5) / den
    return [max(0.0, mid - half), min(1.0, mid + half)]


def sweep_one(contraction, refs, arrays, cases=None, compact=False, softmax="contract"):
    """The oracle under one contraction contract against the reference runs in `refs`; fills `arrays` (reference-only data: the same
    for every contraction) and returns {"cases": ..., "summary": ...}.  `compact` (the wide sweep): `arrays` gets a digest per row, the
    tie metadata, and the full canonical row only where the row flips under this contraction."""
    cases = SWEEP_CASES if cases is None else cases
    O.set_contraction(contraction)
    O.set_softmax(softmax)
    meta = {"cases": {}}
    tot_rows = tot_straddle = tot_flip_rows = tot_flips = tot_mism = tot_el = tot_invalid = max_ulp = 0
    tot_logit_mism = tot_logit_el = tot_prob_cases_equal = 0
    fam = {}
    for name, case in cases.items():
        q, k, v = make_qkv(case["seed"], case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"], peaked=case.get("peaked", 0))
        if name not in refs:
            c_, t_ = run_reference(q, k, v, case, name if compact else None)
            refs[name] = (c_, t_) if not compact else (c_.clone(), t_.clone())
        c_ref, t_ref = refs[name]
        _, _, idx_or, tsp_or, c_or, t_or = O.update_kv(q, k, v, case["W"], case["ks"], case["pooling"], case["cap"], case["tsp_len"],
                                                       "index", return_scores=True)
        stage = None
        if compact:
            # the stage-level pin: the oracle's logits / probabilities against the reference's (digests; the element count of the
            # logits that differ needs the reference's tensor, which is recomputed here once per contraction and dropped again)
            _, _, lg_or, pr_or = O.stages(q, k, case["W"], case["ks"], case["pooling"])
            stage = {"logits_equal": tensor_digest(lg_or) == STAGE_DIGESTS[name][0], "probs_equal": tensor_digest(pr_or) == STAGE_DIGESTS[name][1]}
            if not stage["logits_equal"]:
                grab = {}
                orig_softmax = torch.nn.functional.softmax

                def spy_softmax(x, *a, **kw):
                    grab["x"] = x.detach().clone()
                    return orig_softmax(x, *a, **kw)

                torch.nn.functional.softmax = spy_softmax
                try:
                    FastKVCluster(window_size=case["W"], max_capacity_prompt=case["cap"], kernel_size=case["ks"], pooling=case["pooling"],
                                  tsp_layer=True, tsp_length=case["tsp_len"]).update_kv(k, q, v, None, q.shape[1] // k.shape[1], 0)
                finally:
                    torch.nn.functional.softmax = orig_softmax
                stage["logits_that_differ"] = int((bits(lg_or) != bits(grab["x"])).sum())
            else:
                stage["logits_that_differ"] = 0
            tot_logit_mism += stage["logits_that_differ"]
            tot_logit_el += lg_or.numel()
            tot_prob_cases_equal += int(stage["probs_equal"])
        B, Hkv, n = c_ref.shape
        kk, tk = case["cap"] - case["W"], case["tsp_len"] - case["W"]
        d = (bits(c_or) - bits(c_ref)).abs()
        rows = []
        can = torch.empty(B, Hkv, kk, dtype=torch.int64)
        ties = np.zeros((B, Hkv, 3), dtype=np.int64)
        for b in range(B):
            for g in range(Hkv):
                sel, vk = canonical(c_ref[b, g], kk)
                can[b, g] = sel
                vk_bits = int(bits(vk.to(torch.float16)))
                ties[b, g] = (vk_bits, int((c_ref[b, g].float() > vk).sum()), int((c_ref[b, g].float() == vk).sum()))
                mm = d[b, g] > 0
                near = mm & (((bits(c_ref[b, g]) - vk_bits).abs() <= 1) | ((bits(c_or[b, g]) - vk_bits).abs() <= 1))
                got = set(idx_or[b, g].tolist())
                flips = sorted(got - set(sel.tolist()))
                # is the oracle's set still an answer `topk` could have given on the REFERENCE's scores (every element above the
                # k-th value taken, the rest from the tie plateau, where torch's choice is arbitrary: SURVEY A.2)?
                rf = c_ref[b, g].float()
                valid = set(torch.nonzero(rf > vk).flatten().tolist()) <= got <= set(torch.nonzero(rf >= vk).flatten().tolist())
                rows.append({"row": [b, g], "mismatching_scores": int(mm.sum()), "straddling": int(near.sum()), "flips": len(flips),
                             "flipped_positions": flips, "valid_topk_of_reference_scores": bool(valid)})
        tsel, tvk = canonical(t_ref[0], tk)
        tcan = torch.cat([tsel, torch.arange(n, case["S"])])
        tvk_bits = int(bits(tvk.to(torch.float16)))
        td = (bits(t_or) - bits(t_ref)).abs()
        tmm = td[0] > 0
        tnear = tmm & (((bits(t_ref[0]) - tvk_bits).abs() <= 1) | ((bits(t_or[0]) - tvk_bits).abs() <= 1))
        tgot = set(tsp_or[0, :tk].tolist())
        tflips = sorted(tgot - set(tsel.tolist()))
        trf = t_ref[0].float()
        tvalid = set(torch.nonzero(trf > tvk).flatten().tolist()) <= tgot <= set(torch.nonzero(trf >= tvk).flatten().tolist())
        if compact:
            arrays[name + ".stage_dig"] = np.array(STAGE_DIGESTS[name], dtype=np.uint64)          # reference logits, probabilities
            arrays[name + ".c_dig"] = np.array([[tensor_digest(c_ref[b, g]) for g in range(Hkv)] for b in range(B)], dtype=np.uint64)
            arrays[name + ".t_dig"] = np.array([tensor_digest(t_ref[0])], dtype=np.uint64)
            arrays[name + ".dig"] = np.array([[row_digest(can[b, g].numpy()) for g in range(Hkv)] for b in range(B)], dtype=np.uint64)
            arrays[name + ".tsp_dig"] = np.array([row_digest(tcan.numpy())], dtype=np.uint64)
            arrays[name + ".ties"] = ties
            for r in rows:
                if r["flips"]:
                    arrays[name + ".row%d_%d" % tuple(r["row"])] = can[r["row"][0], r["row"][1]].numpy().astype(np.uint16)
            if tflips:
                arrays[name + ".tsp"] = tcan.numpy().astype(np.uint16)
            case = {k_: v_ for k_, v_ in case.items()}
        else:
            arrays[name + ".idx"] = can.numpy().astype(np.uint16)
            arrays[name + ".tsp"] = tcan.numpy().astype(np.uint16)
            arrays[name + ".ties"] = ties
            arrays[name + ".c_sampled"] = c_ref[..., ::64].contiguous().view(torch.int16).numpy()
        m = {"case": case, "score_elements": int(d.numel()), "mismatching_scores": int((d > 0).sum()), "max_ulp": int(d.max()),
             "rows": rows, "tsp": {"mismatching_scores": int(tmm.sum()), "max_ulp": int(td.max()), "straddling": int(tnear.sum()),
                                   "flips": len(tflips), "flipped_positions": tflips,
                                   "valid_topk_of_reference_scores": bool(tvalid)}}
        if compact:                                               # (keep the meta small: per row only what is not the default)
            m["rows"] = [r for r in rows if r["flips"] or r["straddling"] or not r["valid_topk_of_reference_scores"]]
            m["rows_total"] = len(rows)
            m["stage"] = stage
        meta["cases"][name] = m
        f = fam.setdefault(case.get("family", "all"), dict(cases=0, rows=0, flip_rows=0, flips=0, invalid=0, mism=0, el=0))
        f["cases"] += 1
        f["rows"] += len(rows) + 1
        f["flip_rows"] += sum(1 for r in rows if r["flips"]) + (1 if tflips else 0)
        f["flips"] += sum(r["flips"] for r in rows) + len(tflips)
        f["invalid"] += sum(1 for r in rows if not r["valid_topk_of_reference_scores"]) + (0 if tvalid else 1)
        f["mism"] += int((d > 0).sum()) + int(tmm.sum())
        f["el"] += int(d.numel()) + int(td.numel())
        max_ulp = max(max_ulp, int(d.max()), int(td.max()))
        tot_rows += len(rows) + 1
        tot_straddle += sum(1 for r in rows if r["straddling"]) + (1 if tnear.any() else 0)
        tot_flip_rows += sum(1 for r in rows if r["flips"]) + (1 if tflips else 0)
        tot_flips += sum(r["flips"] for r in rows) + len(tflips)
        tot_invalid += sum(1 for r in rows if not r["valid_topk_of_reference_scores"]) + (0 if tvalid else 1)
        tot_mism += int((d > 0).sum()) + int(tmm.sum())
        tot_el += int(d.numel()) + int(td.numel())
        print(contraction, name, "mismatching", m["mismatching_scores"], "max ulp", m["max_ulp"], "straddling rows",
              sum(1 for r in rows if r["straddling"]), "flipping rows", sum(1 for r in rows if r["flips"]), "tsp flips", len(tflips), flush=True)
    meta["summary"] = {"cases": len(cases), "rows": tot_rows, "score_elements": tot_el, "mismatching_scores": tot_mism,
                       "mismatch_rate": tot_mism / tot_el, "max_ulp": max_ulp, "rows_with_a_straddling_mismatch": tot_straddle,
                       "rows_that_flip": tot_flip_rows, "indices_flipped": tot_flips,
                       "row_flip_rate": tot_flip_rows / tot_rows,
                       "rows_whose_set_is_not_a_valid_topk_of_the_reference_scores": tot_invalid,
                       "row_flip_rate_ci95": wilson(tot_flip_rows, tot_rows), "invalid_row_rate": tot_invalid / tot_rows,
                       "invalid_row_rate_ci95": wilson(tot_invalid, tot_rows)}
    if compact:
        meta["summary"].update({"logit_elements": tot_logit_el, "logits_that_differ_from_the_reference": tot_logit_mism,
                                "logit_mismatch_rate": tot_logit_mism / max(1, tot_logit_el),
                                "cases_whose_probabilities_equal_the_reference_bit_for_bit": tot_prob_cases_equal})
    O.set_softmax("contract")
    meta["families"] = {k_: dict(v_, mismatch_rate=v_["mism"] / v_["el"], row_flip_rate=v_["flip_rows"] / v_["rows"],
                                 row_flip_rate_ci95=wilson(v_["flip_rows"], v_["rows"]), invalid_row_rate=v_["invalid"] / v_["rows"],
                                 invalid_row_rate_ci95=wilson(v_["invalid"], v_["rows"])) for k_, v_ in fam.items()}
    return meta


def main_wide():
    arrays, refs = {}, {}
    meta = {"note": "the WIDE sweep (round 5): 24 seeds x {constant budget 2048 / 2048, published recipe 3276 / 6553} x {maxpool, avgpool} of randn "
                    "inputs + 24 peaked cases (3000 planted heavy hitters per KV head), S = 32768, Llama-3-8B geometry; per contraction "
                    "contract of the oracle: how far its scores / index sets are from the REFERENCE's (baselines/fastkv/utils.py:80-134, "
                    "imported), overall and per family, with 95 % Wilson intervals on the per-row rates.  Rows = 8 KV heads + the TSP row per case.",
            "contractions": {}}
    for contraction in ("fmaf", "mfma16"):
        meta["contractions"][contraction] = sweep_one(contraction, refs, arrays, SWEEP_WIDE_CASES, compact=True)
        print(contraction, json.dumps(meta["contractions"][contraction]["summary"], indent=1))
        print(contraction, json.dumps(meta["contractions"][contraction]["families"], indent=1))
    # round 6: the oracle with the fma chain AND the reference-order softmax (torch's AVX-512 kernel restated; its AVX2 twin beside it):
    # what is left between the contract and the reference is exactly the denominator's summation order (+ the exp polynomial)
    meta["reference_order"] = {}
    for sm in ("torch_avx512", "torch_avx2"):
        r = sweep_one("fmaf", refs, {}, SWEEP_WIDE_CASES, compact=True, softmax=sm)
        meta["reference_order"][sm] = {"summary": r["summary"], "families": r["families"]}
        print("fmaf +", sm, json.dumps(r["summary"], indent=1))
    np.savez_compressed(os.path.join(HERE, "sweep_wide.npz"), **arrays)
    with open(os.path.join(HERE, "sweep_wide_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "wide":
        return main_wide()
    arrays, refs = {}, {}
    meta = {"note": "flips = indices of canonical_topk(oracle scores) that are not in canonical_topk(reference scores); one entry per "
                    "contraction contract of the oracle (oracle/fastkv_oracle.c: the fp32 fma chain / the gfx950 fp16 matrix instruction)",
            "contractions": {}}
    for contraction in ("fmaf", "mfma16"):
        meta["contractions"][contraction] = sweep_one(contraction, refs, arrays)
        print(contraction, json.dumps(meta["contractions"][contraction]["summary"], indent=1))
    np.savez_compressed(os.path.join(HERE, "sweep32k.npz"), **arrays)
    with open(os.path.join(HERE, "sweep_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
