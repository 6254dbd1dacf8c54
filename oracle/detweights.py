"""Build-owned deterministic weight generator (numpy only).

Per-tensor seed = crc32(name) ^ base seed, Philox counter RNG, so the GPU box
regenerates bit-identical weights without the reference and without torch.
Shapes follow the reference's state-dict layout:
  GPT:   autoregressive/models/gpt.py:273-289, gpt_video.py:293-296
  VQ:    tokenizer/tokenizer_image/vq_model.py:32-39,137-167,207-212
  VAE:   CausalVideoVAE/causalvideovae/model/causal_vae/modeling_causalvae.py:151-262,369

Test infrastructure (see oracle/__init__.py).
"""
import zlib

import numpy as np


def _rng(name, seed):
    key = (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF
    return np.random.Generator(np.random.Philox(key=key))


def normal(name, shape, std, seed=1234):
    return (_rng(name, seed).standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)


def find_multiple(n, k):
    return n if n % k == 0 else n + k - (n % k)


def ffn_hidden(dim, multiple_of=256):
    # gpt.py:154-159
    return find_multiple(int(2 * (4 * dim) / 3), multiple_of)


def gpt_weights(cfg, seed=1234, std=0.02, head_std=0.02):
    """cfg: dict with dim,n_layer,n_head,vocab_size,num_classes,caption_dim,model_type,
    vae_embed_dim (t2v), head in {'logits','adapter2','diff'}. Returns name->float32 array.
    output.weight is drawn N(0, head_std) (reference zero-inits it, gpt.py:307 - SURVEY Q10)."""
    D = cfg["dim"]
    F = ffn_hidden(D, cfg.get("multiple_of", 256))
    sd = {}
    mt = cfg["model_type"]
    if mt == "c2i":
        sd["cls_embedding.embedding_table.weight"] = normal("cls_embedding.embedding_table.weight", (cfg["num_classes"] + 1, D), std, seed)
    else:
        cd = cfg["caption_dim"]
        sd["cls_embedding.cap_proj.fc1.weight"] = normal("cls_embedding.cap_proj.fc1.weight", (D, cd), std, seed)
        sd["cls_embedding.cap_proj.fc2.weight"] = normal("cls_embedding.cap_proj.fc2.weight", (D, D), std, seed)
        sd["cls_embedding.uncond_embedding"] = normal("cls_embedding.uncond_embedding", (120, cd), 1.0 / cd ** 0.5, seed)  # token_num=120 always (gpt.py:93,96)
    if mt == "t2v":
        C = cfg["vae_embed_dim"]
        # adapter std chosen so activations stay O(1) for tiny C (reference uses 0.02 for all Linear)
        sd["vae_latent_adapter.fc1.weight"] = normal("vae_latent_adapter.fc1.weight", (D, C), cfg.get("adapter_in_std", std), seed)
        sd["vae_latent_adapter.fc2.weight"] = normal("vae_latent_adapter.fc2.weight", (D, D), std, seed)
        if cfg.get("head", "adapter2") == "adapter2":
            sd["vae_latent_adapter2.fc1.weight"] = normal("vae_latent_adapter2.fc1.weight", (D, D), std, seed)
            sd["vae_latent_adapter2.fc2.weight"] = normal("vae_latent_adapter2.fc2.weight", (C, D), cfg.get("adapter_out_std", std), seed)
        if cfg.get("head") == "hidden":
            sd.update(diffloss_weights(D, C, cfg.get("diffloss_w", 1024), cfg.get("diffloss_d", 3), seed))
    sd["tok_embeddings.weight"] = normal("tok_embeddings.weight", (cfg["vocab_size"], D), std, seed)
    for i in range(cfg["n_layer"]):
        p = f"layers.{i}."
        sd[p + "attention.wqkv.weight"] = normal(p + "attention.wqkv.weight", (3 * D, D), std, seed)
        sd[p + "attention.wo.weight"] = normal(p + "attention.wo.weight", (D, D), std, seed)
        sd[p + "feed_forward.w1.weight"] = normal(p + "feed_forward.w1.weight", (F, D), std, seed)
        sd[p + "feed_forward.w3.weight"] = normal(p + "feed_forward.w3.weight", (F, D), std, seed)
        sd[p + "feed_forward.w2.weight"] = normal(p + "feed_forward.w2.weight", (D, F), std, seed)
        sd[p + "attention_norm.weight"] = (1.0 + normal(p + "attention_norm.weight", (D,), 0.1, seed)).astype(np.float32)
        sd[p + "ffn_norm.weight"] = (1.0 + normal(p + "ffn_norm.weight", (D,), 0.1, seed)).astype(np.float32)
    sd["norm.weight"] = (1.0 + normal("norm.weight", (D,), 0.1, seed)).astype(np.float32)
    sd["output.weight"] = normal("output.weight", (cfg["vocab_size"], D), head_std, seed)
    return sd


def diffloss_weights(z_channels, target_channels, width, depth, seed=1234):
    """diffloss.net.* (diffloss.py:161-190).  The reference zero-inits every adaLN / output layer (diffloss.py:206-215);
    they are drawn non-zero here so the goldens exercise the whole network."""
    sd = {}
    p = "diffloss.net."

    def lin(name, out_f, in_f, std=None):
        sd[p + name + ".weight"] = normal(p + name + ".weight", (out_f, in_f), std if std else 1.0 / np.sqrt(in_f), seed)
        sd[p + name + ".bias"] = normal(p + name + ".bias", (out_f,), 0.02, seed)

    lin("time_embed.mlp.0", width, 256)
    lin("time_embed.mlp.2", width, width)
    lin("cond_embed", width, z_channels)
    lin("input_proj", width, target_channels)
    for i in range(depth):
        b = f"res_blocks.{i}."
        sd[p + b + "in_ln.weight"] = (1.0 + normal(p + b + "in_ln.weight", (width,), 0.1, seed)).astype(np.float32)
        sd[p + b + "in_ln.bias"] = normal(p + b + "in_ln.bias", (width,), 0.05, seed)
        lin(b + "mlp.0", width, width)
        lin(b + "mlp.2", width, width, 0.5 / np.sqrt(width))
        lin(b + "adaLN_modulation.1", 3 * width, width, 0.5 / np.sqrt(width))
    lin("final_layer.adaLN_modulation.1", 2 * width, width, 0.5 / np.sqrt(width))
    lin("final_layer.linear", 2 * target_channels, width, 0.5 / np.sqrt(width))
    return sd


def _conv(sd, name, cout, cin, k, seed, nd=2, gain=1.0):
    ks = (k,) * nd if isinstance(k, int) else tuple(k)
    fan_in = cin * int(np.prod(ks))
    sd[name + ".weight"] = normal(name + ".weight", (cout, cin) + ks, gain / np.sqrt(fan_in), seed)
    sd[name + ".bias"] = normal(name + ".bias", (cout,), 0.02, seed)


def _gn(sd, name, c, seed):
    sd[name + ".weight"] = (1.0 + normal(name + ".weight", (c,), 0.1, seed)).astype(np.float32)
    sd[name + ".bias"] = normal(name + ".bias", (c,), 0.05, seed)


def vq_weights(cfg=None, seed=1234):
    """VQ-16/VQ-8 decoder side + codebook + post_quant_conv (vq_model.py:128-167,207,39)."""
    cfg = dict(cfg or {})
    ch = cfg.get("ch", 128)
    ch_mult = cfg.get("ch_mult", (1, 1, 2, 2, 4))
    zc = cfg.get("z_channels", 256)
    n_e = cfg.get("codebook_size", 16384)
    e_dim = cfg.get("codebook_embed_dim", 8)
    nrb = cfg.get("num_res_blocks", 2)
    sd = {}
    sd["quantize.embedding.weight"] = normal("quantize.embedding.weight", (n_e, e_dim), 1.0, seed)
    _conv(sd, "post_quant_conv", zc, e_dim, 1, seed)
    nres = len(ch_mult)
    block_in = ch * ch_mult[nres - 1]
    _conv(sd, "decoder.conv_in", block_in, zc, 3, seed)

    def res(prefix, cin, cout):
        _gn(sd, prefix + ".norm1", cin, seed)
        _conv(sd, prefix + ".conv1", cout, cin, 3, seed)
        _gn(sd, prefix + ".norm2", cout, seed)
        _conv(sd, prefix + ".conv2", cout, cout, 3, seed, gain=0.5)
        if cin != cout:
            _conv(sd, prefix + ".nin_shortcut", cout, cin, 1, seed)

    def attn(prefix, c):
        _gn(sd, prefix + ".norm", c, seed)
        for n in ("q", "k", "v"):
            _conv(sd, prefix + "." + n, c, c, 1, seed)
        _conv(sd, prefix + ".proj_out", c, c, 1, seed, gain=0.5)

    res("decoder.mid.0", block_in, block_in)
    attn("decoder.mid.1", block_in)
    res("decoder.mid.2", block_in, block_in)
    for li, i_level in enumerate(reversed(range(nres))):
        block_out = ch * ch_mult[i_level]
        for j in range(nrb + 1):
            res(f"decoder.conv_blocks.{li}.res.{j}", block_in, block_out)
            block_in = block_out
            if i_level == nres - 1:
                attn(f"decoder.conv_blocks.{li}.attn.{j}", block_in)
        if i_level != 0:
            _conv(sd, f"decoder.conv_blocks.{li}.upsample.conv", block_in, block_in, 3, seed)
    _gn(sd, "decoder.norm_out", block_in, seed)
    _conv(sd, "decoder.conv_out", 3, block_in, 3, seed)
    return sd


def vq_encoder_weights(cfg=None, seed=1234):
    """VQ-16 encoder + quant_conv (vq_model.py:64-103,38)."""
    cfg = dict(cfg or {})
    ch = cfg.get("ch", 128)
    ch_mult = cfg.get("ch_mult", (1, 1, 2, 2, 4))
    zc = cfg.get("z_channels", 256)
    e_dim = cfg.get("codebook_embed_dim", 8)
    sd = {}
    _conv(sd, "encoder.conv_in", ch, 3, 3, seed)

    def res(prefix, cin, cout):
        _gn(sd, prefix + ".norm1", cin, seed)
        _conv(sd, prefix + ".conv1", cout, cin, 3, seed)
        _gn(sd, prefix + ".norm2", cout, seed)
        _conv(sd, prefix + ".conv2", cout, cout, 3, seed, gain=0.5)
        if cin != cout:
            _conv(sd, prefix + ".nin_shortcut", cout, cin, 1, seed)

    def attn(prefix, c):
        _gn(sd, prefix + ".norm", c, seed)
        for n in ("q", "k", "v"):
            _conv(sd, prefix + "." + n, c, c, 1, seed)
        _conv(sd, prefix + ".proj_out", c, c, 1, seed, gain=0.5)

    in_mult = (1,) + tuple(ch_mult)
    nres = len(ch_mult)
    block_in = ch
    for li in range(nres):
        block_in = ch * in_mult[li]
        block_out = ch * ch_mult[li]
        for j in range(2):
            res(f"encoder.conv_blocks.{li}.res.{j}", block_in, block_out)
            block_in = block_out
            if li == nres - 1:
                attn(f"encoder.conv_blocks.{li}.attn.{j}", block_in)
        if li != nres - 1:
            _conv(sd, f"encoder.conv_blocks.{li}.downsample.conv", block_in, block_in, 3, seed)
    res("encoder.mid.0", block_in, block_in)
    attn("encoder.mid.1", block_in)
    res("encoder.mid.2", block_in, block_in)
    _gn(sd, "encoder.norm_out", block_in, seed)
    _conv(sd, "encoder.conv_out", zc, block_in, 3, seed)
    _conv(sd, "quant_conv", e_dim, zc, 1, seed)
    return sd


def vae_encoder_weights(cfg=None, seed=1234):
    """CausalVAEModel encoder + quant_conv (modeling_causalvae.py:26-148,368)."""
    cfg = dict(cfg or {})
    hs = cfg.get("hidden_size", 128)
    mult = cfg.get("hidden_size_mult", (1, 2, 4, 4))
    zc = cfg.get("z_channels", 4)
    ed = cfg.get("embed_dim", 4)
    nrb = cfg.get("num_res_blocks", 2)
    sdn = cfg.get("spatial_downsample", (True, True, True, False))
    sd = {}
    _conv(sd, "encoder.conv_in.conv", hs, 3, 3, seed, nd=3)

    def res(prefix, cin, cout):
        _gn(sd, prefix + ".norm1", cin, seed)
        _conv(sd, prefix + ".conv1.conv", cout, cin, 3, seed, nd=3)
        _gn(sd, prefix + ".norm2", cout, seed)
        _conv(sd, prefix + ".conv2.conv", cout, cout, 3, seed, nd=3, gain=0.5)
        if cin != cout:
            _conv(sd, prefix + ".nin_shortcut.conv", cout, cin, 1, seed, nd=3)

    in_mult = (1,) + tuple(mult)
    block_in = hs
    for lvl in range(len(mult)):
        block_in = hs * in_mult[lvl]
        block_out = hs * mult[lvl]
        for j in range(nrb):
            res(f"encoder.down.{lvl}.block.{j}", block_in, block_out)
            block_in = block_out
        if sdn[lvl]:
            _conv(sd, f"encoder.down.{lvl}.downsample.conv.conv", block_in, block_in, (1, 3, 3), seed, nd=3)
    res("encoder.mid.block_1", block_in, block_in)
    _gn(sd, "encoder.mid.attn_1.norm", block_in, seed)
    for n in ("q", "k", "v"):
        _conv(sd, f"encoder.mid.attn_1.{n}.conv", block_in, block_in, 1, seed, nd=3)
    _conv(sd, "encoder.mid.attn_1.proj_out.conv", block_in, block_in, 1, seed, nd=3, gain=0.5)
    res("encoder.mid.block_2", block_in, block_in)
    _gn(sd, "encoder.norm_out", block_in, seed)
    _conv(sd, "encoder.conv_out.conv", 2 * zc, block_in, 3, seed, nd=3)
    _conv(sd, "quant_conv.conv", 2 * ed, 2 * zc, 1, seed, nd=3)
    return sd


def vae_weights(cfg=None, seed=1234):
    """CausalVAEModel decoder side (modeling_causalvae.py:151-262,268-320,369).
    Every causal conv is <name>.conv.{weight,bias} (conv.py:90)."""
    cfg = dict(cfg or {})
    hs = cfg.get("hidden_size", 128)
    mult = cfg.get("hidden_size_mult", (1, 2, 4, 4))
    zc = cfg.get("z_channels", 4)
    ed = cfg.get("embed_dim", 4)
    nrb = cfg.get("num_res_blocks", 2)
    sup = cfg.get("spatial_upsample", (False, True, True, True))
    sd = {}
    _conv(sd, "post_quant_conv.conv", zc, ed, 1, seed, nd=3)
    nres = len(mult)
    block_in = hs * mult[nres - 1]
    _conv(sd, "decoder.conv_in.conv", block_in, zc, 3, seed, nd=3)

    def res(prefix, cin, cout):
        _gn(sd, prefix + ".norm1", cin, seed)
        _conv(sd, prefix + ".conv1.conv", cout, cin, 3, seed, nd=3)
        _gn(sd, prefix + ".norm2", cout, seed)
        _conv(sd, prefix + ".conv2.conv", cout, cout, 3, seed, nd=3, gain=0.5)
        if cin != cout:
            _conv(sd, prefix + ".nin_shortcut.conv", cout, cin, 1, seed, nd=3)

    res("decoder.mid.block_1", block_in, block_in)
    _gn(sd, "decoder.mid.attn_1.norm", block_in, seed)
    for n in ("q", "k", "v"):
        _conv(sd, f"decoder.mid.attn_1.{n}.conv", block_in, block_in, 1, seed, nd=3)
    _conv(sd, "decoder.mid.attn_1.proj_out.conv", block_in, block_in, 1, seed, nd=3, gain=0.5)
    res("decoder.mid.block_2", block_in, block_in)
    for i_level in reversed(range(nres)):
        block_out = hs * mult[i_level]
        for j in range(nrb + 1):
            res(f"decoder.up.{i_level}.block.{j}", block_in, block_out)
            block_in = block_out
        if sup[i_level]:
            _conv(sd, f"decoder.up.{i_level}.upsample.conv.conv", block_in, block_in, (1, 3, 3), seed, nd=3)
    _gn(sd, "decoder.norm_out", block_in, seed)
    _conv(sd, "decoder.conv_out.conv", 3, block_in, 3, seed, nd=3)
    return sd


def videovq_weights(cfg=None, seed=1234):
    """tokenizer_video VQVAE decode side (vqvae.py:17-31,89-125,245-272): codebook, post_vq_conv, decoder."""
    cfg = dict(cfg or {})
    nh = cfg.get("n_hiddens", 240)
    ed = cfg.get("embedding_dim", 256)
    nc = cfg.get("n_codes", 2048)
    nres = cfg.get("n_res_layers", 4)
    n_up = cfg.get("n_upsample", 2)
    sd = {}
    sd["codebook.embeddings"] = normal("codebook.embeddings", (nc, ed), 1.0, seed)
    _conv(sd, "post_vq_conv.conv", nh, ed, 1, seed, nd=3)

    def bn(name, c):
        sd[name + ".weight"] = (1.0 + normal(name + ".weight", (c,), 0.1, seed)).astype(np.float32)
        sd[name + ".bias"] = normal(name + ".bias", (c,), 0.05, seed)
        sd[name + ".running_mean"] = normal(name + ".running_mean", (c,), 0.1, seed)
        sd[name + ".running_var"] = (1.0 + np.abs(normal(name + ".running_var", (c,), 0.2, seed))).astype(np.float32)

    for i in range(nres):
        p = f"decoder.res_stack.{i}.block."
        bn(p + "0", nh)
        sd[p + "2.conv.weight"] = normal(p + "2.conv.weight", (nh // 2, nh, 3, 3, 3), 1.0 / np.sqrt(nh * 27), seed)
        bn(p + "3", nh // 2)
        sd[p + "5.conv.weight"] = normal(p + "5.conv.weight", (nh, nh // 2, 1, 1, 1), 1.0 / np.sqrt(nh // 2), seed)
        bn(p + "6", nh)
        for ax in ("attn_w", "attn_h", "attn_t"):
            q = p + "8." + ax + "."
            for n in ("w_qs", "w_ks", "w_vs"):
                sd[q + n + ".weight"] = normal(q + n + ".weight", (nh, nh), 1.0 / np.sqrt(nh), seed)
            sd[q + "fc.weight"] = normal(q + "fc.weight", (nh, nh), 0.5 / np.sqrt(nh), seed)
            sd[q + "fc.bias"] = normal(q + "fc.bias", (nh,), 0.02, seed)
    bn(f"decoder.res_stack.{nres}", nh)
    for i in range(n_up):
        co = 3 if i == n_up - 1 else nh
        sd[f"decoder.convts.{i}.convt.weight"] = normal(f"decoder.convts.{i}.convt.weight", (nh, co, 4, 4, 4), 1.0 / np.sqrt(nh * 8), seed)
        sd[f"decoder.convts.{i}.convt.bias"] = normal(f"decoder.convts.{i}.convt.bias", (co,), 0.02, seed)
    return sd


def t5_weights(cfg, seed=1234):
    """transformers.T5EncoderModel state-dict names (gated-GELU feed-forward), init roughly as T5PreTrainedModel._init_weights."""
    D, dk, H, F_, V = cfg["d_model"], cfg["d_kv"], cfg["num_heads"], cfg["d_ff"], cfg["vocab_size"]
    inner = H * dk
    sd = {"shared.weight": normal("shared.weight", (V, D), 1.0, seed)}
    sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] = normal(
        "t5.rel_bias", (cfg["relative_attention_num_buckets"], H), 0.5, seed)
    for l in range(cfg["num_layers"]):
        p = f"encoder.block.{l}.layer."
        sd[p + "0.SelfAttention.q.weight"] = normal(p + "q", (inner, D), (D * dk) ** -0.5 * 2.0, seed)
        sd[p + "0.SelfAttention.k.weight"] = normal(p + "k", (inner, D), D ** -0.5, seed)
        sd[p + "0.SelfAttention.v.weight"] = normal(p + "v", (inner, D), D ** -0.5, seed)
        sd[p + "0.SelfAttention.o.weight"] = normal(p + "o", (D, inner), inner ** -0.5, seed)
        sd[p + "0.layer_norm.weight"] = (1.0 + 0.1 * normal(p + "ln0", (D,), 1.0, seed)).astype("float32")
        sd[p + "1.DenseReluDense.wi_0.weight"] = normal(p + "wi0", (F_, D), D ** -0.5, seed)
        sd[p + "1.DenseReluDense.wi_1.weight"] = normal(p + "wi1", (F_, D), D ** -0.5, seed)
        sd[p + "1.DenseReluDense.wo.weight"] = normal(p + "wo", (D, F_), F_ ** -0.5, seed)
        sd[p + "1.layer_norm.weight"] = (1.0 + 0.1 * normal(p + "ln1", (D,), 1.0, seed)).astype("float32")
    sd["encoder.final_layer_norm.weight"] = (1.0 + 0.1 * normal("t5.final_ln", (D,), 1.0, seed)).astype("float32")
    return sd
