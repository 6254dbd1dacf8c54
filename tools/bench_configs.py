#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json / SURVEY.md §8d on one MI355X (random-init weights, synthetic conditions):
  serve-readme : the reference's only published workload (autoregressive/serve/README.md:12-16): c2i 384x384 (576 tokens),
                 8 classes, cfg 4.0, top-k 2000, bf16, for GPT-B/L/XL/XXL/3B  -> sampling wall time (+ VQ decode time)
  c2           : GPT-L c2i 384x384 (576 tokens), 8 classes, cfg 4.0 (16 rows), top-k 2000
  c3           : GPT-XL t2i 512x512 (1024 tokens), 120 text tokens, cfg 7.5, top-k 1000, B = 4
  c5           : GPT-3B c2i 384x384, B = 32 per GPU, cfg 1.65 (B' = 64, head_dim 100)
Prints one JSON line per configuration."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd as V  # noqa: E402
from video_llamagen_amd.sample_common import synthetic_text  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def c2i(name, B, grid, cfg, top_k, vq):
    m = V.GPT_models[name](block_size=grid * grid, cls_token_num=1, model_type="c2i").to(dev, torch.bfloat16).init_random_weights(seed=1)
    cond = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(0)).to(dev)
    ids = [None]

    def run():
        ids[0] = V.generate(m, cond, grid * grid, cfg_scale=cfg, temperature=1.0, top_k=top_k, top_p=1.0, sample_logits=True, seed=7)
    ts = timed(run)
    td = timed(lambda: vq.decode_code(ids[0], [B, 8, grid, grid]))
    wb, kb, ob = m.algorithmic_bytes()
    return {"model": name, "B": B, "tokens": grid * grid, "cfg": cfg, "top_k": top_k, "sampling_s": ts, "tokens_per_s": B * grid * grid / ts,
            "vq_decode_s": td, "hbm_floor_s_at_8TBs": (wb + kb + ob) / 8e12}


def main():
    which = sys.argv[1:] or ["serve-readme", "c3", "c5"]
    vq = V.VQ_models["VQ-16"]().to(dev).init_random_weights(seed=2)
    if "serve-readme" in which:
        ref = {"GPT-B": (7.80, 2.39), "GPT-L": (13.72, 3.48), "GPT-XL": (19.76, 4.84), "GPT-XXL": (26.38, 6.36), "GPT-3B": (14.73, 6.26)}
        for name in ("GPT-B", "GPT-L", "GPT-XL", "GPT-XXL", "GPT-3B"):
            r = c2i(name, 8, 24, 4.0, 2000, vq)
            r.update(config="serve-readme", ref_a100_pytorch_s=ref[name][0], ref_a100_vllm_s=ref[name][1],
                     speedup_vs_a100_pytorch=ref[name][0] / r["sampling_s"], speedup_vs_a100_vllm=ref[name][1] / r["sampling_s"])
            print(json.dumps(r), flush=True)
    if "c2" in which:   # BASELINE config 2: GPT-L c2i 384 px, 8 classes, cfg 4.0 (16 rows), top-k 2000
        r = c2i("GPT-L", 8, 24, 4.0, 2000, vq)
        r["config"] = "c2"
        print(json.dumps(r), flush=True)
    if "c3" in which:
        m = V.GPT_models["GPT-XL"](block_size=1024, cls_token_num=120, model_type="t2i").to(dev, torch.bfloat16).init_random_weights(seed=1)
        cond, mask = synthetic_text(4, 120, 2048, 1, dev)
        ids = [None]

        def run():
            ids[0] = V.generate(m, cond, 1024, mask, cfg_scale=7.5, temperature=1.0, top_k=1000, top_p=1.0, sample_logits=True, seed=7)
        ts = timed(run)
        td = timed(lambda: vq.decode_code(ids[0], [4, 8, 32, 32]))
        wb, kb, ob = m.algorithmic_bytes()
        print(json.dumps({"config": "c3", "model": "GPT-XL t2i", "B": 4, "tokens": 1024, "sampling_s": ts, "tokens_per_s": 4096 / ts,
                          "vq_decode_s": td, "hbm_floor_s_at_8TBs": (wb + kb + ob) / 8e12}), flush=True)
        del m
    if "c5" in which:
        r = c2i("GPT-3B", 32, 24, 1.65, 2000, vq)
        r["config"] = "c5 (per-GPU shard of B=256 over 8 GPUs)"
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
