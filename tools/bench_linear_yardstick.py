#!/usr/bin/env python3
"""hipBLASLt yardstick for the decode-step GEMM phase: the four Linear layers of a transformer layer (qkv, wo, w13, w2) as a COLD
chain over L distinct weight sets (each layer's weights are read once per pass, > 256 MB between two uses of one set), torch.matmul
eager and graph-replayed, next to the in-tree fused GEMMs timed the same way through vlg_linear's sibling entry (the real decode step
is timed by bench.py; this isolates the GEMM phase).  GPU box only.

  python tools/bench_linear_yardstick.py            # C4: GPT-XL, 32 rows; C2: GPT-L, 16 rows; 4-row shard
Prints one JSON line per (model, rows)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

dev = torch.device("cuda")
SHAPES = {"GPT-XL": (1280, 3584, 36), "GPT-L": (1024, 2816, 24), "GPT-3B": (3200, 8704, 24)}


def chain(model, M, reps=5):
    D, F, L = SHAPES[model]
    g = torch.Generator(device="cpu").manual_seed(0)
    layers = []
    for _ in range(L):
        layers.append([(torch.randn(n, k, generator=g) * 0.02).to(dev, torch.bfloat16) for (n, k) in ((3 * D, D), (D, D), (2 * F, D), (D, F))])
    xs = [torch.randn(M, k, generator=g).to(dev, torch.bfloat16) for k in (D, D, D, F)]
    outs = [torch.empty(M, n, device=dev, dtype=torch.bfloat16) for n in (3 * D, D, 2 * F, D)]

    def one_pass():
        for ws in layers:
            for j in range(4):
                torch.matmul(xs[j], ws[j].t(), out=outs[j])

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps / L      # us per layer

    res = {"model": model, "rows": M, "layers": L, "weights_MB_per_layer": (4 * D * D + 3 * D * F) * 2 / 1e6}
    res["hipblaslt_eager_us_per_layer"] = timed(one_pass)
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        one_pass()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            one_pass()
    res["hipblaslt_graph_us_per_layer"] = timed(gr.replay)
    # per shape, cold (cycling through the L copies), eager
    per = {}
    for j, nm in enumerate(("qkv", "wo", "w13", "w2")):
        def f(j=j):
            for ws in layers:
                torch.matmul(xs[j], ws[j].t(), out=outs[j])
        per[nm] = timed(f)
    res["hipblaslt_eager_us_per_shape"] = per
    res["hbm_floor_us_at_6.29TBs"] = res["weights_MB_per_layer"] / 6.29
    return res


def main():
    which = sys.argv[1:] or ["GPT-XL:32", "GPT-XL:16", "GPT-XL:4", "GPT-L:16", "GPT-XL:8", "GPT-3B:64"]
    for w in which:
        m, r = w.split(":")
        print(json.dumps(chain(m, int(r))), flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
