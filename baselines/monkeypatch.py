"""Plugin boundary with the reference's names and semantics (/root/reference/baselines/monkeypatch.py:12,59,104):
`replace_llama(method)`, `replace_mistral(method)` swap classes / forwards of the installed `transformers` BEFORE the
model is constructed; `set_model(model, args)` pushes the per-layer configuration.  Only the methods on the path this
repository implements exist: "fastkv" and the "fullkv" comparison arm (the other baselines are out of scope)."""
from __future__ import annotations

from transformers.models.llama import modeling_llama
from transformers.models.mistral import modeling_mistral

SUPPORTED = ("fastkv", "fullkv")

_stock = {}


def _remember(mod, names):
    for n in names:
        _stock.setdefault((mod.__name__, n), getattr(mod, n))


def _stock_of(mod, name):
    return _stock[(mod.__name__, name)]


def _check(method):
    if method not in SUPPORTED:
        raise NotImplementedError(f"method {method!r}: only {SUPPORTED} are implemented by the MI355X hot path")


def replace_llama(method):
    _check(method)
    _remember(modeling_llama, ["LlamaAttention"])
    _stock.setdefault("llama_model_forward", modeling_llama.LlamaModel.forward)
    _stock.setdefault("llama_layer_forward", modeling_llama.LlamaDecoderLayer.forward)
    if method == "fastkv":
        from baselines.fastkv.llama_model import (LlamaFastKVAttention, llama_decoderlayer_forward_fastkv,
                                                  llama_model_forward_fastkv)
        modeling_llama.LlamaAttention = LlamaFastKVAttention          # picked up by LlamaDecoderLayer.__init__
        modeling_llama.LlamaDecoderLayer.forward = llama_decoderlayer_forward_fastkv
        modeling_llama.LlamaModel.forward = llama_model_forward_fastkv
    else:
        from baselines.fullkv.llama_model import make_model_forward_general
        modeling_llama.LlamaAttention = _stock_of(modeling_llama, "LlamaAttention")
        modeling_llama.LlamaDecoderLayer.forward = _stock["llama_layer_forward"]
        modeling_llama.LlamaModel.forward = make_model_forward_general(_stock["llama_model_forward"])
    # every compressing method decodes at the TRUE positions (monkeypatch.py:55-56); the full-KV arm keeps the stock preparation
    LlamaForCausalLM.prepare_inputs_for_generation = (prepare_inputs_for_generation_llama if method != "fullkv"
                                                      else _STOCK_PREPARE["llama"])


def replace_mistral(method):
    _check(method)
    _remember(modeling_mistral, ["MistralAttention"])
    _stock.setdefault("mistral_model_forward", modeling_mistral.MistralModel.forward)
    _stock.setdefault("mistral_layer_forward", modeling_mistral.MistralDecoderLayer.forward)
    if method == "fastkv":
        from baselines.fastkv.mistral_model import (MistralFastKVAttention, mistral_decoderlayer_forward_fastkv,
                                                    mistral_model_forward_fastkv)
        modeling_mistral.MistralAttention = MistralFastKVAttention
        modeling_mistral.MistralDecoderLayer.forward = mistral_decoderlayer_forward_fastkv
        modeling_mistral.MistralModel.forward = mistral_model_forward_fastkv
    else:
        from baselines.fullkv.llama_model import make_model_forward_general
        modeling_mistral.MistralAttention = _stock_of(modeling_mistral, "MistralAttention")
        modeling_mistral.MistralDecoderLayer.forward = _stock["mistral_layer_forward"]
        modeling_mistral.MistralModel.forward = make_model_forward_general(_stock["mistral_model_forward"])
    MistralForCausalLM.prepare_inputs_for_generation = (prepare_inputs_for_generation_mistral if method != "fullkv"
                                                        else _STOCK_PREPARE["mistral"])      # monkeypatch.py:101-102


def set_model(model, args):
    """Per-layer configuration push (/root/reference/baselines/monkeypatch.py:104-147): scalar window / kernel sizes
    become per-layer lists, then `compress_fastkv` writes the cluster attributes."""
    if args.method == "fullkv":
        return
    _check(args.method)
    layers = len(model.model.layers)
    for name in ("window_size", "kernel_size"):
        val = getattr(args, name)
        if not isinstance(val, (list, tuple)):
            setattr(args, name, [val] * layers)
    from baselines.fastkv.utils import compress_fastkv
    compress_fastkv(model, args)


# ---- generation-input preparation (/root/reference/baselines/monkeypatch.py:249-389) ---------------------------------
# The reference overrides `prepare_inputs_for_generation` of transformers 4.45 so that, after a COMPRESSED prefill, decode
# steps get their true positions (attention-mask cumsum, `:280-288`) instead of positions derived from the shorter cache.
# The installed transformers (5.x) already derives `position_ids` from the attention mask inside `generate()`; these two
# functions keep the reference's names and enforce exactly that rule on top of the stock implementation; `replace_llama` /
# `replace_mistral` install them for every compressing method, as the reference does (`:55-56`, `:101-102`).
def _prepare_inputs_true_positions(stock):
    def prepare(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, position_ids=None, **kwargs):
        model_inputs = stock(self, input_ids, past_key_values=past_key_values, attention_mask=attention_mask,
                             inputs_embeds=inputs_embeds, position_ids=position_ids, **kwargs)
        if attention_mask is not None and position_ids is None:
            pos = attention_mask.long().cumsum(-1) - 1                 # monkeypatch.py:280-283
            pos.masked_fill_(attention_mask == 0, 1)
            ids = model_inputs.get("input_ids")
            n_new = ids.shape[1] if ids is not None else model_inputs["inputs_embeds"].shape[1]
            model_inputs["position_ids"] = pos[:, -n_new:].clone(memory_format=torch.contiguous_format)
        return model_inputs
    return prepare


import torch  # noqa: E402  (only the two functions below need it)
from transformers import LlamaForCausalLM, MistralForCausalLM  # noqa: E402

_STOCK_PREPARE = {"llama": LlamaForCausalLM.prepare_inputs_for_generation, "mistral": MistralForCausalLM.prepare_inputs_for_generation}
prepare_inputs_for_generation_llama = _prepare_inputs_true_positions(_STOCK_PREPARE["llama"])
prepare_inputs_for_generation_mistral = _prepare_inputs_true_positions(_STOCK_PREPARE["mistral"])
