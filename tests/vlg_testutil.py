"""Helpers shared by the GPU parity tests (builds product models from the deterministic test weights)."""
import numpy as np
import torch

from oracle import detweights


def product_gpt(cfg, dtype=torch.float32, sd=None):
    import video_llamagen_amd as V
    keys = ("dim", "n_layer", "n_head", "vocab_size", "block_size", "cls_token_num", "model_type", "num_classes",
            "caption_dim", "norm_eps", "rope_base", "multiple_of", "vae_embed_dim", "num_frames", "t_downsample_size")
    args = V.ModelArgs(**{k: cfg[k] for k in keys if k in cfg})
    m = V.Transformer(args).to(device="cuda", dtype=dtype).eval()
    sd = sd if sd is not None else detweights.gpt_weights(cfg)
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return m, unexpected


def to_np(t):
    return t.detach().float().cpu().numpy()


def bf16_np(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16)
