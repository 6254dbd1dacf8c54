#!/usr/bin/env python3
"""sample_kernel alone (GPU box only): V = 16384 logits, 8 rows under guidance (the shape of configs 2 / 3), realistic logit spread."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd  # noqa: E402,F401
from video_llamagen_amd import _lib as L  # noqa: E402

B, V = 8, 16384
logits = (torch.randn(2 * B, V, generator=torch.Generator().manual_seed(0)) * 2.5).cuda()
idx = torch.empty((B,), dtype=torch.int32, device="cuda")
st = L.stream_ptr()
for top_k, top_p in ((0, 1.0), (1000, 1.0), (2000, 1.0), (2000, 0.9)):
    sp = L.SamplingParams(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=top_k, top_p=top_p, sample_logits=1, seed=7)

    def call():
        L.check(L.lib().vlg_sample(L.ptr(logits), B, V, 1, C.byref(sp), None, C.c_uint64(3), L.ptr(idx), None, st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(f"top_k {top_k} top_p {top_p}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call, ids {idx[:4].tolist()}")
