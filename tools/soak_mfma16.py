"""Soak of the oracle's restatement of v_mfma_f32_32x32x16_f16 against the instruction on the GPU it runs on (not part of the suite:
tests/test_hip_parity.py::test_matrix_instruction_matches_its_restatement_live is the bounded version).  Structured random tiles: per
tile an accumulator exponent, an exponent distance d to the largest products, a spread of the other products below them, a number of
non-zero products per block, accumulator mantissas at both ends of the binade now and then; single instructions and chains.
usage: python tools/soak_mfma16.py [tiles per round] [rounds] [seed]; mismatching inputs go to gpurun_out/mfma16_soak_bad.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from fastkv_amd._lib import load
from oracle import fastkv_oracle as O
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = torch.Generator().manual_seed(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
L = load(); dev = torch.device("cuda:0")
bad_total = 0; n_total = 0; dump = {}
for rnd in range(ROUNDS):
    dd = (16, 16, 32, 128)[rnd % 4]
    e_acc = torch.randint(-20, 31, (T, 1, 1), generator=g)
    d = torch.randint(-15, 51, (T, 1, 1), generator=g)
    s_ = (e_acc - d + 30)                                           # ea + eb of the largest products of the tile
    spread = torch.randint(0, 12, (T, 1, dd), generator=g) * (torch.rand(T, 1, dd, generator=g) < 0.7)
    ssum = (s_ - spread).clamp(2, 60)
    eak = (torch.rand(T, 1, dd, generator=g) * (ssum - 1).clamp(max=29)).long().clamp(min=1)
    eak = torch.minimum(eak, ssum - 1).clamp(1, 30)
    ebk = (ssum - eak).clamp(1, 30)
    ma, mb = (torch.randint(0, 1024, (T, 32, dd), generator=g) for _ in range(2))
    sa, sb = (torch.randint(0, 2, (T, 32, dd), generator=g) for _ in range(2))
    a = ((sa << 15) | (eak.expand(T, 32, dd) << 10) | ma)
    b = ((sb << 15) | (ebk.expand(T, 32, dd) << 10) | mb)
    nz = torch.rand(T, 1, dd, generator=g) < (torch.rand(T, 1, 1, generator=g) * 1.2)       # tiles from sparse to full
    a = torch.where(nz.expand(T, 32, dd), a, torch.zeros_like(a)).to(torch.int16).view(torch.float16).contiguous()
    b = b.to(torch.int16).view(torch.float16).contiguous()
    mant = torch.randint(0, 1 << 23, (T, 32, 32), generator=g)
    edge = torch.rand(T, 32, 32, generator=g)
    mant = torch.where(edge < 0.04, torch.zeros_like(mant), torch.where(edge > 0.96, torch.full_like(mant, (1 << 23) - 1), mant))
    mant = torch.where((edge > 0.04) & (edge < 0.06), torch.ones_like(mant), mant)
    cbits = (torch.randint(0, 2, (T, 32, 32), generator=g) << 31) | ((e_acc + 127).expand(T, 32, 32) << 23) | mant
    c = cbits.to(torch.int32).view(torch.float32).contiguous()
    c = torch.where(torch.rand(T, 32, 32, generator=g) < 0.03, torch.zeros_like(c), c)
    out = torch.empty(T, 32, 32, dtype=torch.float32, device=dev)
    ad, bd, cd = a.to(dev), b.to(dev), c.to(dev)
    assert L.fastkv_debug_mfma16(ad.data_ptr(), bd.data_ptr(), cd.data_ptr(), out.data_ptr(), T, dd, None) == 0
    torch.cuda.synchronize()
    want, got = O.mfma16_tiles(a, b, c), out.cpu()
    bad = got.view(torch.int32) != want.view(torch.int32)
    nb = int(bad.sum()); bad_total += nb; n_total += bad.numel()
    print(f"round {rnd} dd {dd}: {nb} mismatches of {bad.numel()}", flush=True)
    if nb:
        tiles = bad.flatten(1).any(1).nonzero().flatten()[:50]
        dump[f"r{rnd}_a"], dump[f"r{rnd}_b"] = a[tiles].view(torch.int16).numpy(), b[tiles].view(torch.int16).numpy()
        dump[f"r{rnd}_c"], dump[f"r{rnd}_o"] = c[tiles].numpy(), got[tiles].numpy()
        dump[f"r{rnd}_e"], dump[f"r{rnd}_d"] = e_acc[tiles].flatten().numpy(), d[tiles].flatten().numpy()
if dump:
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "mfma16_soak_bad.npz"), **dump)
print(f"{bad_total} mismatches of {n_total}")
sys.exit(1 if bad_total else 0)
