"""Generate the golden vectors under tests/golden/ by running the REFERENCE implementation.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py

This is the only file in the repository that imports the reference
(/root/reference/baselines/fastkv/utils.py).  Nothing of the reference is copied: the fixtures
hold outputs (score tensors, indices, digests, host-logic results) for inputs that are
regenerated from seeds by tests/gen_inputs.py.

How the internals are captured (the reference never returns them):
  * scores   -- `torch.Tensor.topk` is wrapped by a recording spy for the duration of the call:
                call #1 sees `attn_cache` [B,Hkv,n] (utils.py:113), call #2 the TSP row (utils.py:127).
  * indices  -- V is replaced by a position code (V[...,0]=pos//256, V[...,1]=pos%256, exact in
                fp16), so `value_states_out[...,0]*256+[...,1]` is the per-head index order.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
# the reference's `baselines` package must win over this repository's drop-in package of the same name
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]

import numpy as np
import torch

from gen_inputs import make_qkv
from golden_cases import CASES, HOST_CASES
from baselines.fastkv.utils import FastKVCluster, compress_fastkv      # the reference

torch.set_num_threads(8)


def run_reference(q, k, v, case):
    cl = FastKVCluster(window_size=case["W"], max_capacity_prompt=case["cap"], kernel_size=case["ks"],
                       pooling=case["pooling"], tsp_layer=case["tsp_len"] > 0, tsp_length=case["tsp_len"] or 2048)
    spied = []
    orig = torch.Tensor.topk

    def spy(self, *a, **kw):
        spied.append(self.detach().clone())
        return orig(self, *a, **kw)

    torch.Tensor.topk = spy
    try:
        ko, vo, tsp = cl.update_kv(k, q, v, None, q.shape[1] // k.shape[1], 0)
    finally:
        torch.Tensor.topk = orig
    return ko, vo, tsp, spied


def position_code(v):
    B, Hkv, S, D = v.shape
    pos = torch.arange(S)
    vc = torch.zeros(B, S, Hkv, D, dtype=torch.float16)
    vc[..., 0] = (pos // 256).to(torch.float16)[None, :, None]
    vc[..., 1] = (pos % 256).to(torch.float16)[None, :, None]
    return vc.transpose(1, 2)


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()


def main():
    meta = {}
    for name, case in CASES.items():
        q, k, v = make_qkv(case["seed"], case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"],
                           peaked=case.get("peaked", 0))
        ko, vo, tsp, spied = run_reference(q, k, v, case)
        c_ref = spied[0]
        t_ref = spied[1] if len(spied) > 1 else None
        # second run with position-coded V to expose the reference's index order
        _, vcode, _, _ = run_reference(q, k, position_code(v), case)
        kk = case["cap"] - case["W"]
        idx_ref = (vcode[..., 0].float() * 256 + vcode[..., 1].float()).to(torch.int64)[:, :, :kk]
        B, Hkv, S, D = k.shape
        n = S - case["W"]
        # self-checks on the capture
        exp_k = torch.gather(k[:, :, :n], 2, idx_ref[..., None].expand(-1, -1, -1, D))
        assert torch.equal(exp_k, ko[:, :, :kk]), name
        assert torch.equal(ko[:, :, kk:], k[:, :, n:]) and torch.equal(vo[:, :, kk:], v[:, :, n:]), name
        arrays = {"idx_ref": idx_ref.to(torch.int32).numpy()}
        if case.get("store_scores", "full") == "full":
            arrays["c_ref"] = c_ref.view(torch.int16).numpy()
            if t_ref is not None:
                arrays["t_ref"] = t_ref.view(torch.int16).numpy()
        else:
            st = int(case["store_scores"])
            arrays["c_ref_sampled"] = c_ref[..., ::st].contiguous().view(torch.int16).numpy()
            if t_ref is not None:
                arrays["t_ref_sampled"] = t_ref[..., ::st].contiguous().view(torch.int16).numpy()
        # canonical top-k of the reference's own scores (value desc, index asc), index-ascending order
        can = torch.empty(B, Hkv, kk, dtype=torch.int64)
        ties = np.zeros((B, Hkv, 3), dtype=np.int64)      # k-th value bits, #strictly greater, #equal
        for b in range(B):
            for g in range(Hkv):
                row = c_ref[b, g].float()
                srt = torch.sort(row, descending=True, stable=True)
                sel = torch.sort(srt.indices[:kk]).values
                can[b, g] = sel
                vk = srt.values[kk - 1]
                ties[b, g] = (int(c_ref[b, g][srt.indices[kk - 1]].view(torch.int16)), int((row > vk).sum()), int((row == vk).sum()))
                got = set(idx_ref[b, g].tolist())
                assert set(torch.nonzero(row > vk).flatten().tolist()) <= got <= set(torch.nonzero(row >= vk).flatten().tolist())
        arrays["idx_canonical"] = can.to(torch.int32).numpy()
        arrays["ties"] = ties
        if tsp is not None:
            arrays["tsp_ref"] = tsp.to(torch.int32).numpy()
            tk = case["tsp_len"] - case["W"]
            tcan = torch.empty(B, case["tsp_len"], dtype=torch.int64)
            for b in range(B):
                srt = torch.sort(t_ref[b].float(), descending=True, stable=True)
                tcan[b] = torch.cat([torch.sort(srt.indices[:tk]).values, torch.arange(n, S)])
            arrays["tsp_canonical"] = tcan.to(torch.int32).numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
        meta[name] = {"case": case, "sha256_c_ref": sha(c_ref), "sha256_t_ref": sha(t_ref) if t_ref is not None else None,
                      "sha256_k_out": sha(ko), "sha256_v_out": sha(vo), "tsp_is_none": tsp is None,
                      "out_strides_k": list(ko.stride()), "out_contiguous": bool(ko.is_contiguous())}
        print(name, "ok", {k_: a.shape for k_, a in arrays.items()})

    # ---- host logic goldens (utils.py:25-46, :82-91, :123-132) -------------------------------
    host = {}
    for name, hc in HOST_CASES["update_kv_host"].items():
        q, k, v = make_qkv(11, 1, 4, 2, hc["S"], 128, 8)
        cl = FastKVCluster(window_size=8, max_capacity_prompt=hc["cap"], kernel_size=7, pooling="avgpool",
                           tsp_layer=hc["tsp_layer"], tsp_length=hc["tsp_len"], tsp_rate=hc.get("tsp_rate", 0.25),
                           retain_rate=hc.get("retain_rate", 0.25), eviction_mode=hc.get("mode", "constant"))
        ko, vo, tsp = cl.update_kv(k, q, v, None, 2, 0)
        host[name] = {"same_k_object": ko is k, "same_v_object": vo is v, "tsp_is_none": tsp is None,
                      "k_shape": list(ko.shape), "tsp_shape": None if tsp is None else list(tsp.shape),
                      "max_capacity_prompt_after": cl.max_capacity_prompt, "tsp_length_after": cl.tsp_length}
    # compress_fastkv attribute push on a duck-typed 32-layer model
    for name, a in HOST_CASES["compress_fastkv"].items():
        layers = [types.SimpleNamespace(self_attn=types.SimpleNamespace(kv_cluster=FastKVCluster())) for _ in range(a["layers"])]
        model = types.SimpleNamespace(model=types.SimpleNamespace(layers=layers))
        args = types.SimpleNamespace(window_size=[a["window_size"]] * a["layers"], kernel_size=[a["kernel_size"]] * a["layers"],
                                     pooling=a["pooling"], max_capacity_prompts=a["max_capacity_prompts"], tsp_len=a["tsp_len"],
                                     tsp_rate=a["tsp_rate"], eviction_mode=a["eviction_mode"], tsp_idx=a["tsp_idx"],
                                     retain_rate=a["retain_rate"])
        compress_fastkv(model, args)
        host["compress_" + name] = [dict(vars(l.self_attn.kv_cluster)) for l in layers]
    try:
        FastKVCluster(pooling="l2pool").update_kv(*make_qkv(1, 1, 2, 2, 600, 128)[1::-1], make_qkv(1, 1, 2, 2, 600, 128)[2], None, 1, 0)
    except ValueError as e:
        host["bad_pooling_error"] = ["ValueError", str(e)]
    try:
        FastKVCluster(window_size=8, max_capacity_prompt=8)
    except AssertionError:
        host["cap_le_window_error"] = ["AssertionError"]
    meta["host"] = host

    # ---- decoder-layer TSP propagation (llama_model.py:252-259), captured from the reference function itself.
    # Import-time shims only (flash_attn is not installed, the class layout is transformers-4.45): nothing is executed from them.
    from transformers.models.llama import modeling_llama
    fa = types.ModuleType("flash_attn")
    fa.flash_attn_func = fa.flash_attn_varlen_func = None
    sys.modules["flash_attn"] = fa
    modeling_llama.LlamaFlashAttention2 = modeling_llama.LlamaAttention
    import transformers.utils as tu
    tu.is_flash_attn_greater_or_equal_2_10 = lambda: True
    from baselines.fastkv.llama_model import llama_decoderlayer_forward_fastkv as ref_layer_forward

    class _Attn:                                    # duck-typed attention: scales its input, exposes tsp_idx / kv_cluster
        def __init__(self, tsp_idx, tsp_layer):
            self.tsp_idx, self.kv_cluster = tsp_idx, types.SimpleNamespace(tsp_layer=tsp_layer)

        def __call__(self, hidden_states=None, **kw):
            return hidden_states * 0.5, None, None

    Bh, Sh, Hh = 2, 48, 64
    hid = ((torch.arange(Bh * Sh * Hh, dtype=torch.float32).view(Bh, Sh, Hh) % 97) / 8.0).half()
    tsp = torch.stack([torch.sort(torch.randperm(Sh, generator=torch.Generator().manual_seed(5 + b))[:17]).values for b in range(Bh)])
    pos = torch.arange(Sh)[None].expand(Bh, -1)
    layer_g = {"hidden_in": hid.view(torch.int16).numpy(), "tsp_idx": tsp.numpy(), "position_ids": pos.numpy()}
    for tag, tsp_layer, idx in (("tsp", True, tsp), ("not_tsp_layer", False, tsp), ("tsp_none", True, None)):
        self_ = types.SimpleNamespace(input_layernorm=lambda x: x, post_attention_layernorm=lambda x: x, mlp=lambda x: x * 0.25,
                                      self_attn=_Attn(idx, tsp_layer))
        out = ref_layer_forward(self_, hid, position_ids=pos)
        layer_g["hidden_out_" + tag] = out[0].view(torch.int16).numpy()
        layer_g["has_new_pos_" + tag] = np.array(self_.new_position_ids is not None)
        if self_.new_position_ids is not None:
            layer_g["new_pos_" + tag] = self_.new_position_ids.numpy()
    np.savez_compressed(os.path.join(HERE, "decoder_layer_tsp.npz"), **layer_g)
    print("decoder_layer_tsp ok")
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote meta.json")


if __name__ == "__main__":
    main()
