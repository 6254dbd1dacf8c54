"""Shared pieces of the sample entry points (counterparts of autoregressive/sample/*.py driven by synthetic conditions)."""
import os
import time

import numpy as np
import torch


def unwrap_checkpoint(ckpt):
    """sample_t2i.py:62-69: DDP "model" / DeepSpeed "module" / "state_dict" / raw FSDP dict (video_llamagen_amd.io has the strict form)."""
    for k in ("model", "module", "state_dict"):
        if isinstance(ckpt, dict) and k in ckpt:
            return ckpt[k]
    return ckpt


def load_or_init(model, path, seed):
    if path:
        sd = unwrap_checkpoint(torch.load(path, map_location="cpu"))
        sd.pop("freqs_cis", None)                        # serve/gpt_model.py:323-324
        model.load_state_dict(sd, strict=False)
        return "checkpoint " + path
    model.init_random_weights(seed=seed)
    return "random init (no checkpoint given; there is no network for the published weights)"


def synthetic_text(B, T, caption_dim, seed, device):
    """T5-shaped embeddings [B,120,2048] with random valid lengths, left-padded and masked (sample_t2i.py:105-119)."""
    g = torch.Generator().manual_seed(seed)
    emb = torch.randn(B, T, caption_dim, generator=g) * 0.1
    lens = torch.randint(8, T + 1, (B,), generator=g)
    mask = torch.zeros(B, T)
    for b in range(B):
        mask[b, T - int(lens[b]):] = 1.0
    return (emb * mask[:, :, None]).to(device), mask.to(device)


def save_images(samples, path):
    """[-1,1] float [B,3,H,W] -> uint8 .npy (and a PNG grid when PIL is importable); save_image(normalize=True, value_range=(-1,1))."""
    u8 = ((samples.clamp(-1, 1) + 1) * 127.5).round().to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
    np.save(path + ".npy", u8)
    try:
        from PIL import Image
        B, H, W, _ = u8.shape
        cols = min(4, B)
        rows = (B + cols - 1) // cols
        grid = np.zeros((rows * H, cols * W, 3), np.uint8)
        for i in range(B):
            grid[(i // cols) * H:(i // cols + 1) * H, (i % cols) * W:(i % cols + 1) * W] = u8[i]
        Image.fromarray(grid).save(path + ".png")
    except Exception:
        pass


class Timer:
    def __init__(self, label):
        self.label = label

    def __enter__(self):
        torch.cuda.synchronize()
        self.t = time.time()
        return self

    def __exit__(self, *a):
        torch.cuda.synchronize()
        print("%s takes about %.2f seconds." % (self.label, time.time() - self.t), flush=True)


def is_rank0():
    return int(os.environ.get("RANK", "0")) == 0
