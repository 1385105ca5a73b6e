"""Golden vectors for the per-query-head selection rule (the SnapKV baseline), captured from the REFERENCE
(/root/reference/baselines/snapkv/utils.py:57-102, imported, not copied; build container only):

    python tests/golden/make_snapkv.py

`SnapKVCluster.update_kv` receives K/V already repeated to H heads (snapkv/llama_model.py:161-170); a `Tensor.topk` spy exposes
its score tensor [B,H,n].  Stored per case: the reference's scores (fp16 bits) and canonical_topk of them per query head."""
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
from golden_cases import SNAPKV_CASES
from baselines.snapkv.utils import SnapKVCluster, repeat_kv      # the reference

torch.set_num_threads(8)


def main():
    arrays = {}
    for name, c in SNAPKV_CASES.items():
        q, k, v = make_qkv(c["seed"], c["B"], c["H"], c["Hkv"], c["S"], c["D"], c["W"])
        G = c["H"] // c["Hkv"]
        cl = SnapKVCluster(window_size=c["W"], max_capacity_prompt=c["cap"], kernel_size=c["ks"], pooling=c["pooling"])
        spied = []
        orig = torch.Tensor.topk

        def spy(self, *a, **kw):
            spied.append(self.detach().clone())
            return orig(self, *a, **kw)

        torch.Tensor.topk = spy
        try:
            ko, vo = cl.update_kv(repeat_kv(k, G), q, repeat_kv(v, G), None, G)
        finally:
            torch.Tensor.topk = orig
        sc = spied[0]                                             # [B,H,n]
        kk = c["cap"] - c["W"]
        can = torch.empty(c["B"], c["H"], kk, dtype=torch.int64)
        for b in range(c["B"]):
            for h in range(c["H"]):
                srt = torch.sort(sc[b, h].float(), descending=True, stable=True)
                can[b, h] = torch.sort(srt.indices[:kk]).values
        assert ko.shape == (c["B"], c["H"], c["cap"], c["D"])
        arrays[name + ".scores"] = sc.view(torch.int16).numpy()
        arrays[name + ".idx"] = can.numpy().astype(np.int32)
        print(name, tuple(sc.shape))
    np.savez_compressed(os.path.join(HERE, "snapkv.npz"), **arrays)


if __name__ == "__main__":
    main()
