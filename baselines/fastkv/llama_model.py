"""FastKV patches for Llama (names of /root/reference/baselines/fastkv/llama_model.py: LlamaFastKVAttention :88,
llama_decoderlayer_forward_fastkv :193, llama_model_forward_fastkv :273), built on the installed transformers."""
from transformers.masking_utils import create_causal_mask
from transformers.models.llama import modeling_llama

from ._wiring import decoderlayer_forward_fastkv, make_attention_class, make_model_forward

_STOCK_ATTENTION = getattr(modeling_llama, "_fastkv_stock_attention", modeling_llama.LlamaAttention)
modeling_llama._fastkv_stock_attention = _STOCK_ATTENTION

LlamaFastKVAttention = make_attention_class(_STOCK_ATTENTION, modeling_llama, lambda self: {})
LlamaFastKVAttention.__name__ = LlamaFastKVAttention.__qualname__ = "LlamaFastKVAttention"
llama_decoderlayer_forward_fastkv = decoderlayer_forward_fastkv
llama_model_forward_fastkv = make_model_forward(modeling_llama, lambda config: create_causal_mask)
