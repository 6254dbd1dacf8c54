#!/usr/bin/env python3
"""The reference's published serving workload (autoregressive/serve/README.md:12-16: c2i 384x384, 8 classes, cfg 4.0, top-k 2000, 576 tokens)
through the request front-end, wave engine vs iteration-level engine, plus a staggered-arrival case:  python tools/bench_serve.py [GPT-XL]"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd as V  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "GPT-XL"
args = types.SimpleNamespace(gpt_model=name, gpt_ckpt=None, gpt_type="c2i", cfg_scale=4.0, precision="bf16", image_size=384, downsample_size=16,
                             num_classes=1000, cls_token_num=1)
labels = [207, 360, 387, 974, 88, 979, 417, 279]
prompts = [[c] for c in labels] + [[1000]] * len(labels)
sp = V.SamplingParams(temperature=1.0, top_p=1.0, top_k=2000, max_tokens=576, seed=0)
for cont in (False, True):
    llm = V.LLM(args=args, model=name, seed=1, max_num_seqs=16, continuous=cont)
    llm.generate(prompt_token_ids=prompts, sampling_params=sp, use_tqdm=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    outs = llm.generate(prompt_token_ids=prompts, sampling_params=sp, use_tqdm=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("%s %s engine: 8 images (16 sequences) x 576 tokens in %.3f s" % (name, "iteration-level" if cont else "wave", dt), flush=True)
    del llm
# staggered arrival: 16 requests, 4 slots of (cond, uncond) pairs; a new pair is admitted whenever a slot frees
eng = V.ContinuousLLMEngine(V.GPT_models[name](block_size=576, cls_token_num=1, model_type="c2i").to("cuda", torch.bfloat16).init_random_weights(seed=1),
                            cfg_scale=4.0, max_num_seqs=8)
lens = [576, 144, 288, 576, 144, 288, 576, 144, 288, 576, 144, 288, 576, 144, 288, 576]
for i, n in enumerate(lens):
    eng.add_request(str(2 * i), None, V.SamplingParams(top_k=2000, max_tokens=n, seed=0), [labels[i % 8]])
    eng.add_request(str(2 * i + 1), None, V.SamplingParams(top_k=2000, max_tokens=n, seed=0), [1000])
t = time.perf_counter()
steps = done = 0
while eng.has_unfinished_requests():
    done += len(eng.step())
    steps += 1
dt = time.perf_counter() - t
print("%s iteration-level engine, 16 requests of 144/288/576 tokens in 4 slots: %d steps (waves of 4 would take %d), %.3f s, %.0f tokens/s"
      % (name, steps, sum(max(lens[i:i + 4]) for i in range(0, 16, 4)), dt, sum(lens) / dt))
