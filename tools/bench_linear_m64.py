#!/usr/bin/env python3
"""Skinny GEMMs of the wide models' decode (M rows x [N, K] weights, bf16): vlg_linear (slab GEMM + reduce) against torch.matmul
(hipBLASLt) on cold weights, eager launches timed with events (GPU box only).  python tools/bench_linear_m64.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_llamagen_amd import _lib as L  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda")
lib = L.lib()
for (N, K) in ((9600, 3200), (3200, 3200), (17408, 3200), (3200, 8704)):
    copies = max(2, int(600e6 // (N * K * 2)))          # > 256 MB of distinct weights between two uses of the same copy
    ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

    def run_vlg(w):
        L.check(lib.vlg_linear(L.ptr(x), L.ptr(w), L.ptr(out), M, N, K, L.VLG_BF16, L.stream_ptr(dev)))

    def run_torch(w):
        torch.matmul(x, w.t(), out=out)

    res = {}
    for name, fn in (("vlg_linear", run_vlg), ("torch.matmul", run_torch)):
        for w in ws[:2]:
            fn(w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 * copies
        e0.record()
        for i in range(reps):
            fn(ws[i % copies])
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / reps
    mb = N * K * 2 / 1e6
    print(f"M {M} N {N} K {K} ({mb:.0f} MB): " + ", ".join(f"{k} {v:.1f} us ({mb / v:.2f} TB/s)" for k, v in res.items()))
