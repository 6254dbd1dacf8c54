#!/usr/bin/env python3
"""Class-conditional sampling: counterpart of the reference's (missing) autoregressive/sample/sample_c2i.py; argument
names/defaults follow autoregressive/serve/sample_c2i.py:76-95.  `torchrun --nproc-per-node N` shards the class list."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd as V  # noqa: E402
from video_llamagen_amd import dist as vd  # noqa: E402
from video_llamagen_amd.sample_common import Timer, is_rank0, load_or_init, save_images  # noqa: E402


def main(args):
    rank, world, local = vd.init_from_env()
    torch.manual_seed(args.seed)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    precision = {'none': torch.float32, 'bf16': torch.bfloat16}[args.precision]
    latent_size = args.image_size // args.downsample_size
    vq_model = V.VQ_models[args.vq_model](codebook_size=args.codebook_size, codebook_embed_dim=args.codebook_embed_dim).to(device).eval()
    print("image tokenizer:", load_or_init(vq_model, args.vq_ckpt, 1))
    gpt_model = V.GPT_models[args.gpt_model](vocab_size=args.codebook_size, block_size=latent_size ** 2, num_classes=args.num_classes,
                                             cls_token_num=args.cls_token_num, model_type=args.gpt_type).to(device=device, dtype=precision).eval()
    print("gpt model:", load_or_init(gpt_model, args.gpt_ckpt, 2))
    class_labels = [207, 360, 387, 974, 88, 979, 417, 279][: args.num_samples] if args.num_samples <= 8 else \
        torch.randint(0, args.num_classes, (args.num_samples,)).tolist()
    c_indices = torch.tensor(class_labels, device=device)
    n = len(class_labels)

    llm = None
    if args.serve:            # autoregressive/serve/sample_c2i.py:33-66: the request front-end instead of a direct generate() call
        del gpt_model
        llm = V.LLM(args=args, model=args.gpt_model, seed=2, device=device)
        print("gpt model:", llm.weights)

    def run_serve(c):
        prompt_token_ids = [[int(ci)] for ci in c.tolist()]
        if args.cfg_scale > 1.0:
            prompt_token_ids.extend([[args.num_classes] for _ in range(len(prompt_token_ids))])
        sp = V.SamplingParams(temperature=args.temperature, top_p=args.top_p, top_k=args.top_k if args.top_k > 0 else -1,
                              max_tokens=latent_size ** 2, seed=args.seed)
        outputs = llm.generate(prompt_token_ids=prompt_token_ids, sampling_params=sp, use_tqdm=False)
        return torch.tensor([o.outputs[0].token_ids for o in outputs], device=device)[:len(c)]

    def run(c):
        qzshape = [len(c), args.codebook_embed_dim, latent_size, latent_size]
        with Timer("gpt sampling"):
            index_sample = run_serve(c) if args.serve else V.generate(gpt_model, c, latent_size ** 2, cfg_scale=args.cfg_scale, cfg_interval=args.cfg_interval,
                                      temperature=args.temperature, top_k=args.top_k, top_p=args.top_p, sample_logits=True, seed=args.seed)
        with Timer("decoder"):
            return vq_model.decode_code(index_sample, qzshape)          # [-1, 1]

    samples = vd.sharded_call(run, [c_indices], n)
    if not args.serve:
        gpt_model.status()    # generate() is asynchronous: a device-side time-out of a persistent kernel surfaces here, before anything is written
    if is_rank0():
        save_images(samples, args.out)
        print("images saved to %s.npy" % args.out)


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--gpt-model", type=str, choices=list(V.GPT_models.keys()), default="GPT-B")
    p.add_argument("--gpt-ckpt", type=str, default=None)
    p.add_argument("--gpt-type", type=str, choices=['c2i', 't2i'], default="c2i")
    p.add_argument("--cls-token-num", type=int, default=1)
    p.add_argument("--precision", type=str, default='bf16', choices=["none", "bf16"])
    p.add_argument("--vq-model", type=str, choices=list(V.VQ_models.keys()), default="VQ-16")
    p.add_argument("--vq-ckpt", type=str, default=None)
    p.add_argument("--codebook-size", type=int, default=16384)
    p.add_argument("--codebook-embed-dim", type=int, default=8)
    p.add_argument("--image-size", type=int, choices=[256, 384, 512], default=384)
    p.add_argument("--downsample-size", type=int, choices=[8, 16], default=16)
    p.add_argument("--num-classes", type=int, default=1000)
    p.add_argument("--cfg-scale", type=float, default=4.0)
    p.add_argument("--cfg-interval", type=float, default=-1)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--top-k", type=int, default=2000)
    p.add_argument("--temperature", type=float, default=1.0)
    p.add_argument("--top-p", type=float, default=1.0)
    p.add_argument("--num-samples", type=int, default=8)
    p.add_argument("--out", type=str, default="sample_c2i")
    p.add_argument("--serve", action="store_true", help="go through the LLM/SamplingParams request front-end (serve/sample_c2i.py)")
    main(p.parse_args())
