"""Model wiring on the MI355X with the product cluster (HIP kernels): the slab cache -- compaction writes straight into the
layer's pre-sized cache buffer (fastkv_update_kv_strided_f16), decode appends in place -- against DynamicCache."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(slab, monkeypatch):
    from baselines.monkeypatch import replace_llama, set_model
    from benchmark import prefill
    monkeypatch.setenv("FASTKV_SLAB_CACHE", slab)
    # two layers of the Llama-3-8B geometry (head_dim 128: the HIP path has no head_dim-32 "tiny" variant)
    a = prefill.parse_args(["--model_path", "llama3-8b", "--num_layers", "2", "--device", "cuda", "--save_txt", "", "--method",
                            "fastkv", "--max_capacity_prompts", "128", "--tsp_len", "256", "--tsp_idx", "0"])
    a.save_txt = False
    a.context_lengths = [700]
    replace_llama("fastkv")
    torch.manual_seed(3)
    model = prefill.build_model(a, "cuda")
    set_model(model, a)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 1000, (1, 700), generator=g).cuda()
    logits = []
    with torch.no_grad():
        out = model(ids, attention_mask=torch.ones_like(ids))
        pkv = out.past_key_values
        logits.append(out.logits.float().cpu())
        for step in range(4):
            nxt = out.logits[:, -1].argmax(-1, keepdim=True)
            out = model(nxt, past_key_values=pkv, position_ids=torch.tensor([[700 + step]], device="cuda"))
            logits.append(out.logits.float().cpu())
    torch.cuda.synchronize()
    return logits, pkv


def test_slab_cache_in_place_compaction_matches_dynamic_cache(monkeypatch):
    from fastkv_amd.cache import FastKVSlabCache
    dyn, pkv_d = _run("0", monkeypatch)
    slab, pkv_s = _run("1", monkeypatch)
    assert isinstance(pkv_s, FastKVSlabCache) and not isinstance(pkv_d, FastKVSlabCache)
    layer = pkv_s.layers[0]
    assert layer.keys.data_ptr() == layer.kslab.data_ptr() and layer.len == 128 + 4          # rows were written in place
    for i in range(2):
        assert torch.equal(pkv_s.layers[i].keys, pkv_d.layers[i].keys) and torch.equal(pkv_s.layers[i].values, pkv_d.layers[i].values)
    for x, y in zip(dyn, slab):
        assert torch.equal(x, y)
