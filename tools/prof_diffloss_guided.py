#!/usr/bin/env python3
"""The DiffLoss head under its own guidance at the headline width, for rocprofv3 (eager launches: graph replay of the persistent kernels crashes the
profiler, profiles/r04_rocprof_sigsegv_analysis.md):
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dlg -- python tools/prof_diffloss_guided.py [tokens]
GPT-XL t2v, hidden head (W 1024, depth 3, 100 reverse steps), 32 videos = 64 network rows (cfg_iter 2.0): dl_persist_kernel<bf16, 2, true, 3, 8>."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd as V  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
m = V.GPT_models["GPT-XL"](block_size=32 * 32, cls_token_num=120, model_type="t2v", vae_embed_dim=8, num_frames=17, t_downsample_size=4,
                           caption_dim=2048, head="hidden")
m.to(device=dev, dtype=torch.bfloat16).eval()
m.init_random_weights(seed=1234)
m.use_graph = False
g = torch.Generator().manual_seed(0)
cond = (torch.randn(64, 120, 2048, generator=g) * 0.1).to(dev)
mask = torch.ones(64, 120, device=dev)
V.generate_t2v(m, cond, 2, mask, cfg_iter=2.0)
torch.cuda.synchronize()
out = V.generate_t2v(m, cond, n, mask, cfg_iter=2.0)
torch.cuda.synchronize()
m.status()
print("guided DiffLoss head: %d tokens x 32 videos (64 rows), finite=%s" % (n, bool(torch.isfinite(out).all())))
