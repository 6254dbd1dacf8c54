#!/usr/bin/env python3
"""BASELINE config 5 shard (GPT-3B c2i 384x384, 32 images per GPU, cfg 1.65 -> 64 rows, head_dim 100): python tools/bench_c5.py [tokens]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd as V  # noqa: E402

dev = torch.device("cuda", 0)
m = V.GPT_models["GPT-3B"](block_size=576, cls_token_num=1, model_type="c2i").to(dev, torch.bfloat16).init_random_weights(seed=1)
cond = torch.randint(0, 1000, (32,), generator=torch.Generator().manual_seed(0)).to(dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 576
kw = dict(cfg_scale=1.65, temperature=1.0, top_k=0, top_p=1.0, sample_logits=True, seed=7)
V.generate(m, cond, N, **kw)
torch.cuda.synchronize()
t = time.perf_counter()
ids = V.generate(m, cond, N, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("c5 shard: %d tokens x 32 images in %.3f s = %.0f tokens/s; ids checksum %d" % (N, dt, 32 * N / dt, int(ids.sum())))
