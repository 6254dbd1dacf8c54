#!/usr/bin/env python3
"""Text-conditional image sampling: counterpart of autoregressive/sample/sample_t2i.py:32-169 with synthetic T5-shaped
embeddings, or (--t5-layers N) a random-init flan-t5-xl-shaped T5 encoder on synthetic token ids, in place of the pretrained T5Embedder
(language/t5.py needs the network for weights and tokenizer)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd as V  # noqa: E402
from video_llamagen_amd import dist as vd  # noqa: E402
from video_llamagen_amd.sample_common import Timer, is_rank0, load_or_init, save_images, synthetic_text  # noqa: E402


def main(args):
    rank, world, local = vd.init_from_env()
    torch.manual_seed(args.seed)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    precision = {'none': torch.float32, 'bf16': torch.bfloat16}[args.precision]
    latent_size = args.image_size // args.downsample_size
    vq_model = V.VQ_models[args.vq_model](codebook_size=args.codebook_size, codebook_embed_dim=args.codebook_embed_dim).to(device).eval()
    print("image tokenizer:", load_or_init(vq_model, args.vq_ckpt, 1))
    gpt_model = V.GPT_models[args.gpt_model](block_size=latent_size ** 2, cls_token_num=args.cls_token_num,
                                             model_type=args.gpt_type).to(device=device, dtype=precision).eval()
    print("gpt model:", load_or_init(gpt_model, args.gpt_ckpt, 2))
    if args.t5_layers > 0:
        # sample_t2i.py:88-119 with the text encoder in the loop: T5Embedder -> [B,120,2048] embeddings + mask, LEFT-padded as the
        # reference re-packs them (:105-119).  Random-init encoder and synthetic token ids (no tokenizer / weights without network).
        t5 = V.T5EncoderModel(dict(V.t5_model.FLAN_T5_XL, num_layers=args.t5_layers)).to(device, precision).init_random_weights(seed=3)
        g = torch.Generator().manual_seed(args.seed)
        lens = torch.randint(8, args.cls_token_num + 1, (args.num_samples,), generator=g)
        ids = torch.randint(1, 32128, (args.num_samples, args.cls_token_num), generator=g)
        msk = (torch.arange(args.cls_token_num)[None, :] < lens[:, None]).long()
        with Timer("text encoder"):
            embs, emb_masks = V.T5Embedder(device, t5, model_max_length=args.cls_token_num).get_text_embeddings_from_ids(ids * msk, msk)
        new_embs, new_masks = [], []
        for e, mk in zip(embs.float(), emb_masks):                  # valid tokens to the right (sample_t2i.py:105-119)
            n = int(mk.sum())
            new_embs.append(torch.cat([e[n:], e[:n]]))
            new_masks.append(torch.flip(mk, dims=[-1]))
        c_emb_masks = torch.stack(new_masks).float()
        c_indices = torch.stack(new_embs) * c_emb_masks[:, :, None]
    else:
        c_indices, c_emb_masks = synthetic_text(args.num_samples, args.cls_token_num, 2048, args.seed, device)

    def run(c, m):
        qzshape = [len(c), args.codebook_embed_dim, latent_size, latent_size]
        with Timer("Full sampling"):
            index_sample = V.generate(gpt_model, c, latent_size ** 2, m, cfg_scale=args.cfg_scale, temperature=args.temperature,
                                      top_k=args.top_k, top_p=args.top_p, sample_logits=True, seed=args.seed)
        with Timer("decoder"):
            return vq_model.decode_code(index_sample, qzshape)

    samples = vd.sharded_call(run, [c_indices, c_emb_masks], args.num_samples)
    gpt_model.status()        # generate() is asynchronous: a device-side time-out of a persistent kernel surfaces here, before anything is written
    if is_rank0():
        save_images(samples, args.out)
        print("image is saved to %s.npy" % args.out)


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--gpt-model", type=str, choices=list(V.GPT_models.keys()), default="GPT-XL")
    p.add_argument("--gpt-ckpt", type=str, default=None)
    p.add_argument("--gpt-type", type=str, choices=['c2i', 't2i'], default="t2i")
    p.add_argument("--cls-token-num", type=int, default=120)
    p.add_argument("--precision", type=str, default='bf16', choices=["none", "bf16"])
    p.add_argument("--vq-model", type=str, choices=list(V.VQ_models.keys()), default="VQ-16")
    p.add_argument("--vq-ckpt", type=str, default=None)
    p.add_argument("--codebook-size", type=int, default=16384)
    p.add_argument("--codebook-embed-dim", type=int, default=8)
    p.add_argument("--image-size", type=int, choices=[256, 384, 512], default=512)
    p.add_argument("--downsample-size", type=int, choices=[8, 16], default=16)
    p.add_argument("--cfg-scale", type=float, default=7.5)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--top-k", type=int, default=1000)
    p.add_argument("--temperature", type=float, default=1.0)
    p.add_argument("--t5-layers", type=int, default=0, help="> 0: run a (random-init) flan-t5-xl-shaped text encoder with this many layers "
                   "on synthetic token ids instead of using synthetic embeddings")
    p.add_argument("--top-p", type=float, default=1.0)
    p.add_argument("--num-samples", type=int, default=4)
    p.add_argument("--out", type=str, default="sample_t2i")
    main(p.parse_args())
