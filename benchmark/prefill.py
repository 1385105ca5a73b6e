"""Prefill-latency (TTFT) harness: this repository's counterpart of /root/reference/benchmark/prefill.py.

Same flags (`:183-219`), same structure: `replace_llama/replace_mistral(method)` before the model is built, all-ones
prompts of the requested context lengths (`:55-56`, `:252`), `set_model(model, args)`, `num_warmups` untimed +
`num_runs` timed `model(input_ids, attention_mask)` calls bracketed by device events (`:99-110`), mean / std / 95 % CI /
peak memory report (`:132-146`) and the appended txt line (`:148-176`).

Differences forced by the environment: there is no network and no checkpoint on the GPU box, so `--model_path` may
be a geometry name ("llama3-8b", "mistral-7b", "llama3-70b-tp8-rank", "tiny") and the model is random-initialised from
the config; `--context_lengths` overrides the 8192/32768/131072 loop; attention runs through PyTorch-ROCm SDPA
(`flash_attn` is not installed).  `--device cpu` (config 1, CPU plumbing; the reference cannot run there at all: it needs
`torch.cuda.Event`) times with the wall clock and works as it stands for `--method fullkv`; `--method fastkv` on the CPU
RAISES by design -- the product cluster has no CPU fallback -- unless the caller injects a cluster (`args.cluster_factory`,
which is what tests/test_wiring.py does with the oracle's cluster)."""
from __future__ import annotations

import argparse
import math
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

GEOMETRIES = {
    # name: (model_type, hidden, layers, heads, kv_heads, intermediate, vocab)
    "llama3-8b": ("llama", 4096, 32, 32, 8, 14336, 128256),
    "mistral-7b": ("mistral", 4096, 32, 32, 8, 14336, 32768),
    "ministral-8b": ("mistral", 4096, 36, 32, 8, 12288, 131072),
    "llama3-70b-tp8-rank": ("llama", 1024, 80, 8, 1, 3584, 128256),     # per-rank shapes of TP=8 (H=8, Hkv=1, G=8)
    "tiny": ("llama", 256, 4, 8, 2, 512, 1024),
}


def set_seed(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def build_model(args, device):
    from transformers import AutoModelForCausalLM
    if args.model_path in GEOMETRIES:
        mtype, hidden, layers, heads, kvh, inter, vocab = GEOMETRIES[args.model_path]
        if mtype == "llama":
            from transformers import LlamaConfig as Cfg
        else:
            from transformers import MistralConfig as Cfg
        kw = dict(hidden_size=hidden, num_hidden_layers=args.num_layers or layers, num_attention_heads=heads,
                  num_key_value_heads=kvh, intermediate_size=inter, vocab_size=vocab, head_dim=128 if hidden >= 1024 else 32,
                  max_position_embeddings=max(args.context_lengths) + 64, rope_theta=500000.0, use_cache=args.use_cache)
        if mtype == "mistral":
            # v0.2+ checkpoints ship `sliding_window: null`; v0.1's 4096 reaches the mask builder and the attention call
            # (/root/reference/baselines/fastkv/mistral_model.py:143-153) through `--sliding_window`
            kw["sliding_window"] = getattr(args, "sliding_window", None) or None
        cfg = Cfg(**kw)
        cfg._attn_implementation = args.attn_implementation
        with torch.device(device):
            model = AutoModelForCausalLM.from_config(cfg, dtype=args.dtype, attn_implementation=args.attn_implementation)
    else:
        model = AutoModelForCausalLM.from_pretrained(args.model_path, dtype=args.dtype, low_cpu_mem_usage=True,
                                                     device_map="auto" if device != "cpu" else None, use_cache=args.use_cache,
                                                     attn_implementation=args.attn_implementation)
    return model.eval()


def _raise_if_aborted():
    """After a synchronisation: did any launch of the run give up a bounded in-kernel wait (FASTKV_EABORTED)?  fullkv arms and
    CPU runs never load the library, so only ask when it is loaded."""
    import sys
    lib = sys.modules.get("fastkv_amd._lib")
    if lib is not None and lib._lib is not None:
        lib.raise_if_aborted("benchmark")


def main(model, args):
    from baselines.monkeypatch import set_model
    dev = next(model.parameters()).device
    input_id = torch.ones((args.eval_batch_size, args.context_length), dtype=torch.int64, device=dev)
    if args.random_tokens:                                    # all-ones ids make every key identical (degenerate selection)
        g = torch.Generator(device="cpu").manual_seed(args.seed)
        input_id = torch.randint(0, model.config.vocab_size, input_id.shape, generator=g).to(dev)
    attn_mask = torch.ones_like(input_id)
    set_model(model, args)
    if args.cluster_factory is not None:                      # CPU plumbing run: stand-in cluster objects
        for layer in model.model.layers:
            old = layer.self_attn.kv_cluster
            layer.self_attn.kv_cluster = args.cluster_factory(old)

    use_events = dev.type == "cuda"

    def one_run():
        if use_events:
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
        t0 = time.perf_counter()
        with torch.no_grad():
            out = model(input_id, attention_mask=attn_mask)
        if use_events:
            end.record()
            torch.cuda.synchronize()
            _raise_if_aborted()                               # everything of this run has completed: an abandoned launch is an error
            return start.elapsed_time(end) / 1000.0, out
        return time.perf_counter() - t0, out

    for _ in range(args.num_warmups):
        one_run()
    lat = []
    out = None
    for _ in range(args.num_runs):
        t, out = one_run()
        lat.append(t)
    lat = np.array(lat)
    mean, std = float(lat.mean()), float(lat.std(ddof=1)) if len(lat) > 1 else 0.0
    ci = 1.96 * std / math.sqrt(len(lat)) if len(lat) > 1 else 0.0
    mem = torch.cuda.max_memory_allocated() / 2 ** 30 if use_events else 0.0
    cache_lens = []
    pkv = out.past_key_values
    if pkv is not None:
        for i in range(len(model.model.layers)):
            try:
                cache_lens.append(int(pkv.layers[i].keys.shape[-2]))
            except Exception:   # noqa: BLE001
                cache_lens.append(int(pkv.get_seq_length(i)))
    res = {"method": args.method, "context_length": args.context_length, "ttft_s_mean": mean, "ttft_s_std": std, "ci95": ci,
           "tokens_per_s": args.eval_batch_size * args.context_length / mean, "max_mem_GiB": mem, "cache_lens": cache_lens,
           "logits_shape": list(out.logits.shape)}
    print(f"[prefill] {args.method} ctx={args.context_length} TTFT {mean * 1e3:.2f} ms +- {ci * 1e3:.2f} "
          f"({res['tokens_per_s']:.0f} tok/s) peak mem {mem:.2f} GiB cache lens {cache_lens[:2]}..{cache_lens[-2:]}")
    if args.save_txt:
        try:
            os.makedirs(args.save_dir, exist_ok=True)
            with open(os.path.join(args.save_dir, "prefill_summary.txt"), "a") as f:
                f.write(f"{args.model_path}\t{args.method}\t{args.context_length}\t{mean:.6f}\t{std:.6f}\t{ci:.6f}\t{mem:.3f}\n")
        except OSError as e:
            print(f"Failed to save summary txt: {e}")
    return res


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    # Base settings (reference flags)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--model_name", type=str, default=None)
    p.add_argument("--model_path", type=str, default="llama3-8b")
    p.add_argument("--use_fast_tokenizer", type=bool, default=True)
    p.add_argument("--output_attentions", type=bool, default=False)
    p.add_argument("--use_cache", type=bool, default=True)
    p.add_argument("--attn_implementation", type=str, default="sdpa", choices=["flash_attention_2", "sdpa", "eager"])
    # Benchmark settings
    p.add_argument("--genlen", type=int, default=128)
    p.add_argument("--decode_graph", type=int, default=1, help="e2e.py on the GPU with FASTKV_SLAB_CACHE=1: capture the decode "
                                                             "step in a HIP graph (HIP decode attention over the slab cache)")
    p.add_argument("--num_warmups", type=int, default=1)
    p.add_argument("--num_runs", type=int, default=1)
    p.add_argument("--eval_batch_size", type=int, default=1)
    # KV cache compression
    p.add_argument("--method", type=str, default="fastkv", choices=["fullkv", "fastkv"])
    p.add_argument("--eviction_mode", type=str, default="constant", choices=["constant", "proportional"])
    p.add_argument("--retain_rate", type=float, default=0.1)
    p.add_argument("--max_capacity_prompts", type=int, default=512)
    p.add_argument("--window_size", type=int, default=8)
    p.add_argument("--kernel_size", type=int, default=7)
    p.add_argument("--pooling", type=str, default="maxpool")
    p.add_argument("--merge", type=str, default=None)
    # FastKV
    p.add_argument("--tsp_len", type=int, default=2048)
    p.add_argument("--tsp_rate", type=float, default=0.2)
    p.add_argument("--tsp_idx", type=int, default=15)
    # Save results
    p.add_argument("--save_txt", type=bool, default=True)
    p.add_argument("--save_dir", type=str, default="outputs/benchmark")
    # additions of this harness
    p.add_argument("--context_lengths", type=int, nargs="+", default=[8192, 32768, 131072])
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--num_layers", type=int, default=0, help="override the depth of a geometry (0 = as published)")
    p.add_argument("--random_tokens", action="store_true")
    p.add_argument("--sliding_window", type=int, default=0, help="Mistral geometries: sliding-window attention of this many tokens "
                                                               "(0 = none, as the v0.2+ checkpoints)")
    a = p.parse_args(argv)
    a.dtype = torch.float16 if a.device != "cpu" else torch.float32
    a.cluster_factory = None
    return a


def run(args):
    set_seed(args.seed)
    from baselines.monkeypatch import replace_llama, replace_mistral
    replace_llama(args.method)
    replace_mistral(args.method)
    model = build_model(args, args.device)
    results = []
    for context_length in args.context_lengths:
        args.context_length = context_length
        print(f"Prefill latency benchmark ({args.method}) | Context length={context_length}")
        results.append(main(model, args))
    return results


if __name__ == "__main__":
    run(parse_args())
