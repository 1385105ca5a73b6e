"""End-to-end latency harness: counterpart of /root/reference/benchmark/e2e.py (`:53-176`): one prefill followed by
`genlen-1` greedy decode steps over the (compressed) cache, every forward bracketed by device events (`:72-93`),
throughput = (genlen-1) / total time.  Same flags as benchmark/prefill.py (+ `--genlen`); models are random-initialised
geometries (no checkpoints on the GPU box).  Positions restart at the compressed length when no `position_ids` are passed,
exactly as in the reference (`:82-90`).

Two decode paths:
  * eager (any device, any cache): `model(input_ids=tok, past_key_values=pkv)` per token, attention through PyTorch SDPA;
  * `--decode_graph` (default on the GPU with the slab cache, FASTKV_SLAB_CACHE=1): the cache goes into static-decode mode
    (fastkv_amd/cache.py: lengths in device memory), the attention module appends and attends through the HIP decode
    kernels (csrc/decode.hip), and the whole step -- model forward, argmax, token and position update -- is captured once
    in a HIP graph and replayed for the remaining tokens: no host work per layer, no shape changes, no reallocation."""
from __future__ import annotations

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from benchmark import prefill as P


def graph_decode(model, pkv, tok, steps, timed):
    """`steps` greedy decode steps over a slab cache in static-decode mode: the first one eagerly (it warms every library up on
    this stream and is a real, counted step), then ONE capture of the step and `steps - 1` replays.  Returns (ms, tokens)."""
    dev = tok.device
    pkv.enable_static_decode(steps + 8)
    pkv.reserve_steps(steps)                                      # replays advance device-side lengths only: room for all of them, checked now
    tok_buf = tok.clone()
    # positions restart at the compressed length (reference e2e.py:82-90 passes no position_ids); kept on the device
    pos_buf = torch.full((tok.shape[0], 1), pkv.get_seq_length(), dtype=torch.int64, device=dev)
    out_tok = torch.zeros(steps, dtype=torch.int64, device=dev)
    step_no = torch.zeros(1, dtype=torch.int64, device=dev)

    # greedy sampling + the step's bookkeeping (next token, positions, token log) in ONE launch (csrc/decode.hip decode_greedy_kernel)
    # instead of torch's argmax reduction and four small kernels; FASTKV_DECODE_GREEDY=0 keeps the torch sequence
    fused_tail = os.environ.get("FASTKV_DECODE_GREEDY", "1") != "0" and dev.type == "cuda"
    greedy_scratch = None
    if fused_tail:
        from fastkv_amd import ops as _ops
        greedy_scratch = _ops.new_greedy_scratch(dev, tok.shape[0])

    def step():
        out = model(input_ids=tok_buf, past_key_values=pkv, position_ids=pos_buf)
        if fused_tail and out.logits.dtype == torch.float16 and out.logits.shape[-1] % 8 == 0:
            _ops.decode_greedy(out.logits, greedy_scratch, tok_buf, pos_buf, out_tok, step_no)
            return
        nxt = out.logits[:, -1, :].argmax(dim=-1, keepdim=True)
        tok_buf.copy_(nxt)
        pos_buf.add_(1)
        out_tok.index_copy_(0, step_no, nxt[0])
        step_no.add_(1)

    # the vocabulary projection of the step (1 GB of weights for Llama-3) through the weight-streaming GEMV as well
    from baselines.fastkv._wiring import _gemv_ok
    from fastkv_amd import ops
    lm = getattr(model, "lm_head", None)
    patched = lm is not None and os.environ.get("FASTKV_DECODE_GEMV", "1") != "0" and _gemv_ok(lm) and tok.shape[0] in (1, 2, 4)
    if patched:
        stock = lm.forward
        lm.forward = lambda h: ops.decode_gemv(h, [lm.weight]) if (h.is_cuda and h.dim() == 3 and h.shape[1] == 1
                                                                   and h.dtype == torch.float16) else stock(h)
    try:
        total, _ = timed(step)                                    # eager: step 1
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):                # records step 2 (nothing runs during the capture)
                step()
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(steps - 1):
            dt, _ = timed(g.replay)
            total += dt
    finally:
        if patched:
            del lm.forward                                        # back to the class's forward
    pkv.finish_static_decode()                                    # host mirrors <- device lengths
    return total, [int(x) for x in out_tok.tolist()]


def main(model, args):
    from baselines.monkeypatch import set_model
    dev = next(model.parameters()).device
    input_id = torch.ones((args.eval_batch_size, args.context_length), dtype=torch.int64, device=dev)
    if args.random_tokens:
        g = torch.Generator(device="cpu").manual_seed(args.seed)
        input_id = torch.randint(0, model.config.vocab_size, input_id.shape, generator=g).to(dev)
    attn_mask = torch.ones_like(input_id)
    set_model(model, args)
    if args.cluster_factory is not None:
        for layer in model.model.layers:
            layer.self_attn.kv_cluster = args.cluster_factory(layer.self_attn.kv_cluster)
    use_events = dev.type == "cuda"

    def timed(fn):
        if use_events:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn()
            e.record()
            torch.cuda.synchronize()
            P._raise_if_aborted()                                 # an abandoned launch / an overrun slab is an error of the run
            return s.elapsed_time(e), out
        t0 = time.perf_counter()
        out = fn()
        return (time.perf_counter() - t0) * 1e3, out

    results = []
    graph_mode = False
    for it in range(args.num_warmups + args.num_runs):
        with torch.no_grad():
            t_prefill, out = timed(lambda: model(input_id, attention_mask=attn_mask))
            pkv = out.past_key_values
            tok = out.logits[:, -1, :].argmax(dim=-1, keepdim=True)
            generated = [int(tok[0, 0])]
            t_decode = 0.0
            graph_mode = bool(getattr(args, "decode_graph", True)) and use_events and hasattr(pkv, "enable_static_decode") \
                and args.genlen > 2
            if graph_mode:
                t_decode, toks = graph_decode(model, pkv, tok, args.genlen - 1, timed)
                generated += toks
            else:
                for _ in range(args.genlen - 1):
                    dt, out = timed(lambda: model(input_ids=tok, past_key_values=pkv))
                    t_decode += dt
                    pkv = out.past_key_values
                    tok = out.logits[:, -1, :].argmax(dim=-1, keepdim=True)
                    generated.append(int(tok[0, 0]))
        if it >= args.num_warmups:
            results.append((t_prefill, t_decode, len(generated)))
        cache_len = int(pkv.layers[0].keys.shape[-2])
        del out, pkv
    pre = np.array([r[0] for r in results])
    dec = np.array([r[1] for r in results])
    tot = pre + dec
    res = {"method": args.method, "context_length": args.context_length, "genlen": args.genlen, "prefill_ms": float(pre.mean()),
           "decode_ms_per_token": float(dec.mean() / max(1, args.genlen - 1)),
           "throughput_tok_s": float((args.genlen - 1) / (tot.mean() / 1e3)), "final_cache_len_layer0": cache_len,
           "decode_path": "hip graph replay over the slab cache (HIP decode attention)" if graph_mode else "eager (SDPA)"}
    print(f"[e2e] {args.method} ctx={args.context_length} gen={args.genlen}: prefill {res['prefill_ms']:.1f} ms, "
          f"decode {res['decode_ms_per_token']:.2f} ms/token, e2e throughput {res['throughput_tok_s']:.1f} tok/s, cache {cache_len}")
    return res


def run(args):
    P.set_seed(args.seed)
    from baselines.monkeypatch import replace_llama, replace_mistral
    replace_llama(args.method)
    replace_mistral(args.method)
    model = P.build_model(args, args.device)
    out = []
    for context_length in args.context_lengths:
        args.context_length = context_length
        print(f"E2E latency benchmark ({args.method}) | Context length={context_length}")
        out.append(main(model, args))
    return out


if __name__ == "__main__":
    run(P.parse_args())
