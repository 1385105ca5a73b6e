"""Operator-level retrieval check on the MI355X (a stand-in for the reference's RULER needle runs, SURVEY.md 8(f)#4, which
need real weights and are out of reach here): keys that the window queries attend to strongly are planted at random
positions of a 32k prompt; the operator -- CPU oracle and HIP path alike -- must keep every one of them in every KV head's
cache and in the TSP index, at the budgets of BASELINE.json configs[1], and drop them from neither after pooling (the
neighbours of a needle ride along: utils.py:105-108)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pooling", ["avgpool", "maxpool"])
def test_planted_needles_survive_compression(pooling):
    from fastkv_amd import ops
    from gen_inputs import make_qkv
    from oracle import fastkv_oracle as O
    B, H, Hkv, S, D, W, cap, tsp_len = 1, 32, 8, 32768, 128, 8, 2048, 2048
    G = H // Hkv
    q, k, v = make_qkv(2024, B, H, Hkv, S, D, W)
    k = k.clone()
    g = torch.Generator().manual_seed(7)
    needles = torch.randperm(S - W - 64, generator=g)[:24] + 32             # away from the edges and the window
    for hk in range(Hkv):
        # the direction the window queries of the group share most: their mean; a needle key points along it
        qbar = q[0, hk * G:(hk + 1) * G, S - W:].float().mean(dim=(0, 1))
        qbar = qbar / qbar.norm()
        k[0, hk, needles] = (qbar * 12.0).half() + (torch.randn(len(needles), D, generator=g) * 0.1).half()
    want = O.update_kv(q, k, v, W, 7, pooling, cap, tsp_len, "index")
    dev = torch.device("cuda:0")
    to = lambda t: t.transpose(1, 2).contiguous().to(dev).transpose(1, 2)   # noqa: E731
    got = ops.update_kv(to(q), to(k), to(v), W, 7, pooling, cap, tsp_len, "index", return_indices=True)
    assert torch.equal(got[3].cpu(), want[2]) and torch.equal(got[2].cpu(), want[3])
    need = set(needles.tolist())
    for hk in range(Hkv):
        kept = set(got[3][0, hk].tolist())
        assert need <= kept, (pooling, hk, sorted(need - kept))
    assert need <= set(got[2][0].tolist())
    # the kept rows ARE the planted keys
    pos = got[3][0, 0].cpu()
    row = {int(p): i for i, p in enumerate(pos.tolist())}
    for p in list(need)[:5]:
        assert torch.equal(got[0][0, 0, row[p]].cpu(), k[0, 0, p])
