"""Full-KV comparison arm: the stock model forward plus the last-token cut before lm_head
(/root/reference/baselines/fullkv/llama_model.py:140-141), for Llama and Mistral alike."""
from transformers.modeling_outputs import BaseModelOutputWithPast


def make_model_forward_general(stock_forward):
    def model_forward_general(self, *args, **kwargs):
        out = stock_forward(self, *args, **kwargs)
        return BaseModelOutputWithPast(last_hidden_state=out.last_hidden_state[:, -1:, :], past_key_values=out.past_key_values)

    return model_forward_general
