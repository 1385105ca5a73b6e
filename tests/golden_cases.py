"""Case table shared by tests/golden/make_golden.py (capture) and the parity tests (replay).

Shapes follow BASELINE.json's configs: cfg1 = Llama-3-8B 4k / budget 512,
cfg2 = Llama-3-8B 32k / budget 2048 / TSP length 2048, cfg5 = one TP rank of Llama-3-70B.
`store_scores`: "full" keeps the reference's whole score tensors in the fixture; an integer N
keeps every N-th element (the big cases are pinned through their canonical indices instead).
"""

CASES = {
    # small, everything stored; B=2 exercises the per-batch-row TSP index
    "tiny_avg": dict(seed=3, B=2, H=8, Hkv=2, S=640, D=128, W=8, ks=7, pooling="avgpool", cap=96, tsp_len=160),
    "tiny_max": dict(seed=4, B=2, H=8, Hkv=2, S=640, D=128, W=8, ks=5, pooling="maxpool", cap=96, tsp_len=160),
    # ragged length (S not a multiple of any tile), MHA (G=1), D=64
    "ragged_mha_d64": dict(seed=5, B=1, H=4, Hkv=4, S=777, D=64, W=8, ks=7, pooling="avgpool", cap=100, tsp_len=300),
    # G=8 (cfg5 rank geometry: H=8, Hkv=1), wider window
    "gqa8_w16": dict(seed=6, B=1, H=8, Hkv=1, S=1500, D=128, W=16, ks=7, pooling="maxpool", cap=144, tsp_len=400),
    # k == n permutation case (post-TSP layers): S == cap, TSP still evaluated
    "k_eq_n": dict(seed=7, B=1, H=8, Hkv=2, S=256, D=128, W=8, ks=7, pooling="avgpool", cap=256, tsp_len=64),
    # non-TSP layer
    "no_tsp": dict(seed=8, B=1, H=8, Hkv=2, S=1024, D=128, W=8, ks=7, pooling="maxpool", cap=128, tsp_len=0),
    # BASELINE.json configs[0]
    "cfg1": dict(seed=0, B=1, H=32, Hkv=8, S=4096, D=128, W=8, ks=7, pooling="maxpool", cap=512, tsp_len=2048),
    # BASELINE.json configs[1]
    "cfg2_max": dict(seed=0, B=1, H=32, Hkv=8, S=32768, D=128, W=8, ks=7, pooling="maxpool", cap=2048, tsp_len=2048,
                     store_scores="full"),
    "cfg2_avg_peaked": dict(seed=2, B=1, H=32, Hkv=8, S=32768, D=128, W=8, ks=7, pooling="avgpool", cap=2048,
                            tsp_len=2048, peaked=3000, store_scores=16),
    # the published recipe at 32k: proportional retain 0.1 / tsp_rate 0.2 -> cap 3276, tsp_len 6553
    "cfg2_recipe": dict(seed=1, B=1, H=32, Hkv=8, S=32768, D=128, W=8, ks=7, pooling="avgpool", cap=3276,
                        tsp_len=6553, store_scores=16),
}

HOST_CASES = {
    "update_kv_host": {
        "early_out": dict(S=300, cap=512, tsp_layer=True, tsp_len=128),                       # utils.py:89-91
        "s_eq_cap": dict(S=512, cap=512, tsp_layer=True, tsp_len=128),                        # compress, k == n
        "s_eq_tsp": dict(S=600, cap=128, tsp_layer=True, tsp_len=600),                        # strict > at utils.py:126
        "not_tsp_layer": dict(S=600, cap=128, tsp_layer=False, tsp_len=128),
        "proportional": dict(S=1000, cap=512, tsp_layer=True, tsp_len=2048, mode="proportional", retain_rate=0.1, tsp_rate=0.2),
        "proportional_post_tsp": dict(S=655, cap=512, tsp_layer=False, tsp_len=2048, mode="proportional", retain_rate=0.5, tsp_rate=0.2),
    },
    "compress_fastkv": {
        "published": dict(layers=32, window_size=8, kernel_size=7, pooling="avgpool", max_capacity_prompts=512, tsp_len=2048,
                          tsp_rate=0.2, eviction_mode="proportional", tsp_idx=15, retain_rate=0.1),
        "constant": dict(layers=32, window_size=8, kernel_size=7, pooling="maxpool", max_capacity_prompts=2048, tsp_len=2048,
                         tsp_rate=0.2, eviction_mode="constant", tsp_idx=15, retain_rate=0.1),
    },
}

# Seed sweep at the graded length (tests/golden/make_sweep.py): 12 seeds x {BASELINE.json configs[1] (constant budget 2048, TSP
# length 2048, maxpool = the CLI default), the published recipe (proportional: retain 0.1 -> 3276, tsp_rate 0.2 -> 6553, avgpool)}
SWEEP_CASES = {}
for _s in range(12):
    SWEEP_CASES[f"sweep_max_{_s:02d}"] = dict(seed=100 + _s, B=1, H=32, Hkv=8, S=32768, D=128, W=8, ks=7, pooling="maxpool",
                                              cap=2048, tsp_len=2048)
    SWEEP_CASES[f"sweep_recipe_{_s:02d}"] = dict(seed=200 + _s, B=1, H=32, Hkv=8, S=32768, D=128, W=8, ks=7, pooling="avgpool",
                                                 cap=3276, tsp_len=6553)

# The WIDE sweep (round 5, VERDICT r04 next #2a; tests/golden/make_sweep.py wide -> sweep_wide.npz / sweep_wide_meta.json): 24 seeds x
# {constant budget, published recipe} x {maxpool, avgpool} = 96 randn cases + 24 "peaked" cases (3000 planted heavy-hitter keys per KV
# head, gen_inputs.make_qkv(peaked=...): attention with margins randn does not have), the four configurations taking turns
_WIDE_CFG = {"cmax": dict(pooling="maxpool", cap=2048, tsp_len=2048), "cavg": dict(pooling="avgpool", cap=2048, tsp_len=2048),
             "rmax": dict(pooling="maxpool", cap=3276, tsp_len=6553), "ravg": dict(pooling="avgpool", cap=3276, tsp_len=6553)}
SWEEP_WIDE_CASES = {}
for _f, (_fam, _cfg) in enumerate(_WIDE_CFG.items()):
    for _s in range(24):
        SWEEP_WIDE_CASES[f"wide_{_fam}_{_s:02d}"] = dict(seed=300 + 100 * _f + _s, B=1, H=32, Hkv=8, S=32768, D=128, W=8, ks=7, family=_fam, **_cfg)
for _s in range(24):
    _fam = list(_WIDE_CFG)[_s % 4]
    SWEEP_WIDE_CASES[f"wide_peak_{_s:02d}"] = dict(seed=700 + _s, B=1, H=32, Hkv=8, S=32768, D=128, W=8, ks=7, family="peak", peaked=3000,
                                                  **_WIDE_CFG[_fam])

# Per-query-head selection (the SnapKV baseline's rule, tests/golden/make_snapkv.py)
SNAPKV_CASES = {
    "snap_avg": dict(seed=31, B=1, H=8, Hkv=2, S=700, D=128, W=8, ks=5, pooling="avgpool", cap=96),
    "snap_max_b2": dict(seed=32, B=2, H=8, Hkv=4, S=1500, D=128, W=16, ks=7, pooling="maxpool", cap=200),
}


# The GemFilter rule (/root/reference/baselines/gemfilter/utils.py:25-38 `standard_dis_index`; find_context calls it with
# pool=True, sum_over_heads=True on the last query row): tests/golden/make_gemfilter.py -> gemfilter.npz
GEMFILTER_CASES = {
    "gem_ctx": dict(seed=41, B=1, H=8, Hkv=2, S=3000, D=128, k=256, pool=True, ks=5, sum_over_heads=True),
    "gem_heads": dict(seed=42, B=1, H=4, Hkv=4, S=1000, D=64, k=64, pool=False, ks=5, sum_over_heads=False),
    "gem_heads_pool": dict(seed=43, B=2, H=8, Hkv=4, S=1200, D=128, k=100, pool=True, ks=7, sum_over_heads=False),
    "gem_8b": dict(seed=44, B=1, H=32, Hkv=8, S=8192, D=128, k=1024, pool=True, ks=5, sum_over_heads=True),
}
