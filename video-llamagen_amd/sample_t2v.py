#!/usr/bin/env python3
"""Text-to-video sampling: counterpart of autoregressive/sample/sample_t2v_1f_diff.py:61-257 (generate -> reshape
[B,vae_t,h,w,C] -> permute -> vae.decode -> clamp -> uint8), synthetic T5-shaped embeddings.  --head hidden is that script's model
(gpt_video_diff + DiffLoss.sample, --num-sampling-steps reverse steps per token); --head adapter2 the MSE head of gpt_video.py."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_llamagen_amd as V  # noqa: E402
from video_llamagen_amd import dist as vd  # noqa: E402
from video_llamagen_amd.sample_common import Timer, is_rank0, load_or_init, synthetic_text  # noqa: E402


def main(args):
    rank, world, local = vd.init_from_env()
    torch.manual_seed(args.seed)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    precision = {'none': torch.float32, 'bf16': torch.bfloat16}[args.precision]
    latent_size = args.image_size // args.downsample_size
    vae = V.VAE_models[args.vae_model](embed_dim=args.vae_embed_dim).to(device, precision)
    print("video VAE:", load_or_init(vae, args.vae_ckpt, 3))
    vae.enable_tiling()
    vae.tile_overlap_factor = args.tile_overlap_factor
    gpt_model = V.GPT_models[args.gpt_model](block_size=latent_size ** 2, cls_token_num=args.cls_token_num, model_type=args.gpt_type,
                                             vae_embed_dim=vae.config.embed_dim, num_frames=args.num_frames,
                                             t_downsample_size=args.t_downsample_size, head=getattr(args, "head", "adapter2"),
                                             num_sampling_steps=getattr(args, "num_sampling_steps", 100)).to(device=device, dtype=precision).eval()
    print("gpt model:", load_or_init(gpt_model, args.gpt_ckpt, 2))
    cond, masks = synthetic_text(args.num_samples, args.cls_token_num, 2048, args.seed, device)
    vae_t = (args.num_frames - 1) // args.t_downsample_size + 1

    def run(c, m):
        with Timer("Full sampling"):
            lat = V.generate_t2v(gpt_model, c, vae_t * latent_size ** 2, m, cfg_scale=args.cfg_scale,
                                 temperature=getattr(args, "temperature", 1.0), seed=args.seed)
        z = lat.view(-1, vae_t, latent_size, latent_size, vae.config.embed_dim).permute(0, 4, 1, 2, 3).contiguous()
        with Timer("decoder"):
            vids = [vae.decode(z[i:i + 4]) for i in range(0, z.shape[0], 4)]
        x = torch.cat(vids, 0).clamp(-1, 1)                               # custom_to_video, sample_t2v_1f_diff.py:49-58
        return (((x + 1) / 2) * 255).to(torch.uint8).permute(0, 2, 3, 4, 1).contiguous()   # [B,T,H,W,3]

    videos = vd.sharded_call(run, [cond, masks], args.num_samples)
    gpt_model.status()        # generate() is asynchronous: a device-side time-out of a persistent kernel surfaces here, before anything is written
    if is_rank0():
        np.save(args.out + ".npy", videos.cpu().numpy())
        print("videos %s saved to %s.npy" % (tuple(videos.shape), args.out))
        try:                                                  # first clip as a playable file (mp4 with OpenCV, else gif)
            from PIL import Image
            ims = [Image.fromarray(f) for f in videos[0].cpu().numpy()]
            ims[0].save(args.out + "_0.gif", save_all=True, append_images=ims[1:], duration=500, loop=0)
        except ImportError:
            pass


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--gpt-model", type=str, choices=list(V.GPT_models.keys()), default="GPT-XL")
    p.add_argument("--gpt-ckpt", type=str, default=None)
    p.add_argument("--gpt-type", type=str, choices=['t2v'], default="t2v")
    p.add_argument("--cls-token-num", type=int, default=120)
    p.add_argument("--precision", type=str, default='bf16', choices=["none", "bf16"])
    p.add_argument("--vae-model", type=str, choices=list(V.VAE_models.keys()), default="VAE-16")
    p.add_argument("--vae-ckpt", type=str, default=None)
    p.add_argument("--vae-embed-dim", type=int, default=8)
    p.add_argument("--tile_overlap_factor", type=float, default=0.125)
    p.add_argument("--image-size", type=int, default=256)
    p.add_argument("--downsample-size", type=int, choices=[8, 16], default=8)
    p.add_argument("--num_frames", type=int, default=17)
    p.add_argument("--t-downsample-size", type=int, default=4)
    p.add_argument("--cfg-scale", type=float, default=1.0)
    p.add_argument("--head", type=str, choices=["adapter2", "hidden"], default="adapter2", help="hidden = gpt_video_diff + DiffLoss sampler")
    p.add_argument("--num-sampling-steps", type=int, default=100, help="DiffLoss reverse steps per token (gpt_video_diff.py:78)")
    p.add_argument("--temperature", type=float, default=1.0)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--num-samples", type=int, default=4)
    p.add_argument("--out", type=str, default="sample_t2v")
    main(p.parse_args())
