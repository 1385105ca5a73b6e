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
  * logits / probabilities (round 6: the stage-level pin) -- `torch.nn.functional.softmax` is wrapped the same way: its INPUT is the
                fp16 tensor [B,H,W,S] after matmul, division and window mask (utils.py:94-101), its output rounded to fp16 is what
                utils.py:103 hands on.  Stored in full for the cases up to 4k tokens, as sha256 + every 64th column for the 32k cases.
  * `softmax_probe.npz` -- what the installed torch's CPU softmax (fp32) computes for a few fp16-valued rows, and the values of its
                internal exponential (rows whose sum is exactly 2^14, so that p = exp(x - max) * 2^-14, one exact scaling down to the subnormals): the pin of the oracle's
                restatement of that kernel (oracle/fastkv_oracle.c "the softmax", FK_SOFTMAX_TORCH_AVX512 / _AVX2).
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
    orig_softmax = torch.nn.functional.softmax
    stage = {}

    def spy(self, *a, **kw):
        spied.append(self.detach().clone())
        return orig(self, *a, **kw)

    def spy_softmax(x, *a, **kw):                               # utils.py:103 (`nn.functional.softmax`, looked up at call time)
        stage["logits"] = x.detach().clone()                     # fp16 [B,H,W,S]: after utils.py:94-101
        y = orig_softmax(x, *a, **kw)
        stage["probs"] = y.detach().to(x.dtype)                  # what `.to(query_states.dtype)` makes of it
        return y

    torch.Tensor.topk = spy
    torch.nn.functional.softmax = spy_softmax
    try:
        ko, vo, tsp = cl.update_kv(k, q, v, None, q.shape[1] // k.shape[1], 0)
    finally:
        torch.Tensor.topk = orig
        torch.nn.functional.softmax = orig_softmax
    run_reference.stage = stage
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


def softmax_probe():
    """What torch's CPU softmax computes (this container: torch 2.10, AVX-512), for the restatement in oracle/fastkv_oracle.c:
    (1) fp32 outputs for fp16-valued rows of several lengths (ragged tails included); (2) its exponential alone -- rows of 16384
    zeros followed by values below -10: every lane sum is exactly 1024 and stays there (the tail's exponentials are below half an
    ulp of it), the sum is 2^14, so p is the exponential the kernel computed times 2^-14 (exact above the subnormals)."""
    from gen_inputs import normal_f16
    arrays = {}
    for i, (S, scale) in enumerate(((777, 3.0), (4096, 2.0), (32768, 3.0), (32768, 1.0), (100003, 3.0), (24, 2.0), (9, 1.0))):
        x = (torch.from_numpy(normal_f16(900 + i, 1, S).astype(np.float32)) * scale).to(torch.float16)
        y = torch.nn.functional.softmax(x[None].float(), dim=-1)[0]
        arrays["row%d_x" % i] = x.view(torch.int16).numpy()
        arrays["row%d_p" % i] = y.view(torch.int32).numpy()
    u = torch.from_numpy(normal_f16(950, 1, 16384).astype(np.float32))
    t = (-(u.abs() * 12.0 + 10.0)).to(torch.float16)             # -10 .. about -60 and beyond
    t[:8] = torch.tensor([-10.0, -87.0, -88.0, -103.5, -104.0, -104.5, -200.0, -60000.0], dtype=torch.float16)
    x = torch.cat([torch.zeros(16384, dtype=torch.float16), t])
    y = torch.nn.functional.softmax(x[None].float(), dim=-1)[0]
    assert float(y[0]) == 2.0 ** -14
    arrays["exp_x"] = t.view(torch.int16).numpy()
    arrays["exp_p"] = y[16384:].contiguous().view(torch.int32).numpy()      # = exp(x) * 2^-14 (one fp32 product; subnormal below x = -77.6)
    arrays["torch_config"] = np.array([torch.__version__, "AVX512" if "AVX512" in torch.__config__.show() else "other"])
    np.savez_compressed(os.path.join(HERE, "softmax_probe.npz"), **arrays)
    print("softmax_probe ok", torch.__version__)


def main():
    meta = {}
    for name, case in CASES.items():
        q, k, v = make_qkv(case["seed"], case["B"], case["H"], case["Hkv"], case["S"], case["D"], case["W"],
                           peaked=case.get("peaked", 0))
        ko, vo, tsp, spied = run_reference(q, k, v, case)
        c_ref = spied[0]
        t_ref = spied[1] if len(spied) > 1 else None
        lg_ref, pr_ref = run_reference.stage["logits"], run_reference.stage["probs"]
        assert lg_ref.shape == pr_ref.shape == (case["B"], case["H"], case["W"], case["S"]) and lg_ref.dtype == pr_ref.dtype == torch.float16
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
        # the two internal stages: full for the small cases, every 64th column (+ sha256 in meta.json) for the 32k cases
        if case["S"] <= 4096:
            arrays["logits_ref"] = lg_ref.view(torch.int16).numpy()
            arrays["probs_ref"] = pr_ref.view(torch.int16).numpy()
        else:
            arrays["logits_ref_sampled"] = lg_ref[..., ::64].contiguous().view(torch.int16).numpy()
            arrays["probs_ref_sampled"] = pr_ref[..., ::64].contiguous().view(torch.int16).numpy()
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
                      "sha256_logits_ref": sha(lg_ref), "sha256_probs_ref": sha(pr_ref),
                      "sha256_k_out": sha(ko), "sha256_v_out": sha(vo), "tsp_is_none": tsp is None,
                      "out_strides_k": list(ko.stride()), "out_contiguous": bool(ko.is_contiguous())}
        print(name, "ok", {k_: a.shape for k_, a in arrays.items()})

    softmax_probe()

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
