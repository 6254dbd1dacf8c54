#!/usr/bin/env python3
"""GPT-3B (D 3200: wider than the fused GEMM's RMSNorm prologue) c2i 576 tokens at 4 / 16 / 32 cache rows (GPU box only).
VLG_FUSED_NOPRO_ROWS=0 forces the slab path for the A/B."""
import os
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd as V
m = V.GPT_models["GPT-3B"](block_size=576, cls_token_num=1, model_type="c2i").to("cuda", torch.bfloat16).init_random_weights(seed=1)
for B, cfg in ((8, 1.65), (16, 1.65), (16, 1.0), (4, 1.0)):
    cond = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(0)).cuda()
    V.generate(m, cond, 576, cfg_scale=cfg, temperature=1.0, top_k=0, top_p=1.0, sample_logits=True, seed=7)
    torch.cuda.synchronize(); t = time.perf_counter()
    V.generate(m, cond, 576, cfg_scale=cfg, temperature=1.0, top_k=0, top_p=1.0, sample_logits=True, seed=7)
    torch.cuda.synchronize(); print("B", B, "cfg", cfg, "rows", B * (2 if cfg > 1 else 1), round(time.perf_counter() - t, 3), "s")
