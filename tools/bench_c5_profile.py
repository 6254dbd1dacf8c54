#!/usr/bin/env python3
"""BASELINE config 5 shard (GPT-3B c2i 384 px, 32 images, cfg 1.65 = 64 cache rows, head_dim 100) - two generate() calls for rocprofv3 (GPU box only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd as V  # noqa: E402

m = V.GPT_models["GPT-3B"](block_size=576, cls_token_num=1, model_type="c2i").to("cuda", torch.bfloat16).init_random_weights(seed=1)
cond = torch.randint(0, 1000, (32,), generator=torch.Generator().manual_seed(0)).cuda()
for _ in range(2):
    torch.cuda.synchronize()
    t = time.perf_counter()
    V.generate(m, cond, 576, cfg_scale=1.65, temperature=1.0, top_k=0, top_p=1.0, sample_logits=True, seed=7)
    torch.cuda.synchronize()
    print("C5 shard:", round(time.perf_counter() - t, 3), "s")
