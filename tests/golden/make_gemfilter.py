"""Golden vectors for the GemFilter selection rule, captured from the REFERENCE
(/root/reference/baselines/gemfilter/utils.py:25-38 `standard_dis_index`, imported, not copied; build container only):

    python tests/golden/make_gemfilter.py

Called as `find_context` calls it (utils.py:46-52): keys repeated to H heads, the last query row.  A `torch.topk` spy exposes
the tensor the reference ranks.  Stored per case: that tensor (fp16 bits), the reference's own indices and distances, and
canonical_topk (value descending, lowest position first) of the reference's tensor."""
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
from golden_cases import GEMFILTER_CASES
from baselines.gemfilter.utils import repeat_kv, standard_dis_index      # the reference

torch.set_num_threads(8)


def main():
    arrays = {}
    for name, c in GEMFILTER_CASES.items():
        q, k, _ = make_qkv(c["seed"], c["B"], c["H"], c["Hkv"], c["S"], c["D"], 8)
        G = c["H"] // c["Hkv"]
        spied = []
        orig = torch.topk

        def spy(x, *a, **kw):
            spied.append(x.detach().clone())
            return orig(x, *a, **kw)

        torch.topk = spy
        try:
            dist, idx = standard_dis_index(repeat_kv(k, G), q[:, :, -1:, :], c["k"], pool=c["pool"], kernel_size=c["ks"],
                                           sum_over_heads=c["sum_over_heads"])
        finally:
            torch.topk = orig
        sc = spied[0]                                             # [B, 1 or H, n]
        can = torch.empty(sc.shape[0], sc.shape[1], c["k"], dtype=torch.int64)
        for b in range(sc.shape[0]):
            for r in range(sc.shape[1]):
                can[b, r] = torch.sort(sc[b, r].float(), descending=True, stable=True).indices[:c["k"]]
        assert idx.shape == can.shape
        arrays[name + ".scores"] = sc.view(torch.int16).numpy()
        arrays[name + ".ref_idx"] = idx.numpy().astype(np.int32)
        arrays[name + ".ref_dist"] = dist.view(torch.int16).numpy()
        arrays[name + ".idx"] = can.numpy().astype(np.int32)
        print(name, tuple(sc.shape), "reference pick == canonical:", bool(torch.equal(idx, can)))
    np.savez_compressed(os.path.join(HERE, "gemfilter.npz"), **arrays)


if __name__ == "__main__":
    main()
