#!/usr/bin/env python3
"""flan-t5-xl encoder (the text-conditioning step of t2i / t2v, language/t5.py:60-81) on 32 prompts x 120 tokens, random weights (GPU box only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd as V  # noqa: E402

m = V.T5EncoderModel(dict(V.t5_model.FLAN_T5_XL)).to("cuda", torch.bfloat16).eval()
m.init_random_weights(seed=1)
ids = torch.randint(1, 32000, (32, 120))
mask = torch.ones(32, 120, dtype=torch.int64)
for _ in range(2):
    m(input_ids=ids, attention_mask=mask)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    y = m(input_ids=ids, attention_mask=mask)["last_hidden_state"]
torch.cuda.synchronize()
print("flan-t5-xl encoder, 32 x 120 tokens, bf16: %.1f ms per call" % ((time.perf_counter() - t) / 5 * 1e3), tuple(y.shape))
