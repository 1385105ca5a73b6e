"""FastKV patches for Mistral / Ministral (names of /root/reference/baselines/fastkv/mistral_model.py:
MistralFastKVAttention :37, mistral_decoderlayer_forward_fastkv :163, mistral_model_forward_fastkv :236).  The only
difference from Llama is the sliding-window argument of the attention call and of the mask builder."""
from transformers.masking_utils import create_causal_mask, create_sliding_window_causal_mask
from transformers.models.mistral import modeling_mistral

from ._wiring import decoderlayer_forward_fastkv, make_attention_class, make_model_forward

_STOCK_ATTENTION = getattr(modeling_mistral, "_fastkv_stock_attention", modeling_mistral.MistralAttention)
modeling_mistral._fastkv_stock_attention = _STOCK_ATTENTION

MistralFastKVAttention = make_attention_class(_STOCK_ATTENTION, modeling_mistral,
                                              lambda self: {"sliding_window": getattr(self.config, "sliding_window", None)})
MistralFastKVAttention.__name__ = MistralFastKVAttention.__qualname__ = "MistralFastKVAttention"
mistral_decoderlayer_forward_fastkv = decoderlayer_forward_fastkv
mistral_model_forward_fastkv = make_model_forward(
    modeling_mistral, lambda config: create_causal_mask if config.sliding_window is None else create_sliding_window_causal_mask)
