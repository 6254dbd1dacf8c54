"""Shared parity-case definitions (configs + seeded synthetic inputs) for
tests/golden/make_goldens.py, the CPU tests and the GPU parity tests.
Test infrastructure (oracle/__init__.py)."""
import numpy as np

TINY_C2I = dict(dim=128, n_layer=2, n_head=2, vocab_size=1024, block_size=16, cls_token_num=1,
                model_type="c2i", num_classes=10, caption_dim=64, norm_eps=1e-5, rope_base=10000.0,
                multiple_of=256, head="logits")
TINY_T2I = dict(TINY_C2I, model_type="t2i", cls_token_num=120)  # uncond_embedding is [120, cd] (gpt.py:96): CFG needs T=120
TINY_T2V = dict(TINY_C2I, model_type="t2v", cls_token_num=8, block_size=16, vae_embed_dim=8,
                num_frames=9, t_downsample_size=4, head="adapter2",
                adapter_in_std=0.3, adapter_out_std=0.3)
# t2v with the per-token diffusion head (gpt_video_diff.py): 10 sampling steps, width 128, depth 3
TINY_T2V_DIFF = dict(TINY_T2V, head="hidden", diffloss_w=128, diffloss_d=3, num_sampling_steps=10)
# hd = 100 (GPT-3B's head_dim, gpt.py:445) at toy width
TINY_HD100 = dict(TINY_C2I, dim=200, n_head=2, block_size=16)

# transformers.T5Config fields; flan-t5-xl (the reference's text encoder, language/t5.py:16): d_model 2048, d_kv 64, 32 heads, d_ff 5120, 24 layers
TINY_T5 = dict(d_model=64, d_kv=16, num_heads=4, d_ff=128, num_layers=2, vocab_size=100, relative_attention_num_buckets=32,
               relative_attention_max_distance=128, layer_norm_epsilon=1e-6, feed_forward_proj="gated-gelu")

GPT_B = dict(dim=768, n_layer=12, n_head=12, vocab_size=16384, block_size=256, cls_token_num=1,
             model_type="c2i", num_classes=1000, caption_dim=2048, norm_eps=1e-5, rope_base=10000.0,
             multiple_of=256, head="logits")

GPT_SIZES = {  # gpt.py:441-464
    "GPT-B": dict(n_layer=12, n_head=12, dim=768), "GPT-L": dict(n_layer=24, n_head=16, dim=1024),
    "GPT-XL": dict(n_layer=36, n_head=20, dim=1280), "GPT-XXL": dict(n_layer=48, n_head=24, dim=1536),
    "GPT-XXXL": dict(n_layer=48, n_head=40, dim=2560), "GPT-1B": dict(n_layer=22, n_head=32, dim=2048),
    "GPT-3B": dict(n_layer=24, n_head=32, dim=3200), "GPT-7B": dict(n_layer=32, n_head=32, dim=4096),
}

TINY_VIDEOVQ = dict(n_hiddens=48, embedding_dim=32, n_codes=64, n_res_layers=2, n_upsample=2)
TINY_VAE = dict(hidden_size=32, hidden_size_mult=(1, 2, 4, 4), z_channels=4, embed_dim=8, num_res_blocks=2)


def rng(seed):
    return np.random.Generator(np.random.Philox(key=seed))


def class_ids(B, num_classes, seed=0):
    return rng(seed).integers(0, num_classes, size=(B,)).astype(np.int64)


def text_cond(B, T, cd, seed=1, lens=None):
    """Left-padded caption embeddings * mask (sample_t2i.py:105-119 convention)."""
    emb = (rng(seed).standard_normal((B, T, cd), dtype=np.float32) * np.float32(0.1))
    if lens is None:
        lens = rng(seed + 1).integers(max(1, T // 4), T + 1, size=(B,))
    mask = np.zeros((B, T), np.float32)
    for b in range(B):
        mask[b, T - int(lens[b]):] = 1.0
    return (emb * mask[:, :, None]).astype(np.float32), mask


def exp_noise(shape, seed=7):
    u = rng(seed).random(shape, dtype=np.float32)
    return (-np.log1p(-u)).astype(np.float32) + np.float32(1e-20)


def sampler_logits(B=4, V=16384, seed=11):
    return (rng(seed).standard_normal((B, V), dtype=np.float32) * np.float32(3.0)).astype(np.float32)
