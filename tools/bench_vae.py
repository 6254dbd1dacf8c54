#!/usr/bin/env python3
"""CausalVideoVAE decode timing on the benchmark shape (GPU box only): python tools/bench_vae.py [videos_per_call] [calls]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd as V  # noqa: E402

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 4
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 2
vae = V.VAE_models["VAE-16"](embed_dim=8).to("cuda", torch.bfloat16).eval()
vae.init_random_weights(seed=3)
z = torch.randn(chunk, 8, 5, 32, 32, device="cuda")
vae.decode(z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    v = vae.decode(z)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / calls
print(f"vae.decode {chunk} videos -> {tuple(v.shape)}: {dt * 1e3:.1f} ms per call, {dt / chunk * 1e3:.1f} ms per video, "
      f"{19.95 * chunk / dt:.0f} TFLOP/s (19.95 TFLOP per video)")
