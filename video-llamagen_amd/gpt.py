"""Host mirror of the reference model object for the sampling path.

Reference: autoregressive/models/gpt.py:23-50 (ModelArgs), :262-332 (Transformer, setup_caches),
:441-470 (size registry); t2v fields gpt_video.py:58-61.  The object owns a vlg_gpt handle (weights +
KV cache live in HBM inside libvlg); there is no torch compute here.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L


def find_multiple(n: int, k: int):
    if n % k == 0:
        return n
    return n + k - (n % k)


@dataclass
class ModelArgs:
    dim: int = 4096
    n_layer: int = 32
    n_head: int = 32
    n_kv_head: Optional[int] = None
    multiple_of: int = 256
    ffn_dim_multiplier: Optional[float] = None
    rope_base: float = 10000
    norm_eps: float = 1e-5
    initializer_range: float = 0.02

    token_dropout_p: float = 0.1
    attn_dropout_p: float = 0.0
    resid_dropout_p: float = 0.1
    ffn_dropout_p: float = 0.1
    drop_path_rate: float = 0.0

    num_classes: int = 1000
    caption_dim: int = 2048
    class_dropout_prob: float = 0.1
    model_type: str = 'c2i'

    vocab_size: int = 16384
    cls_token_num: int = 1
    block_size: int = 256
    max_batch_size: int = 32
    max_seq_len: int = 2048

    # t2v (gpt_video.py:58-61)
    vae_embed_dim: int = 2048
    t_downsample_size: int = 4
    num_frames: int = 17
    head: str = 'auto'     # 'logits' | 'adapter2' (gpt_video.py:296,431) | 'hidden' (gpt_video_diff.py:657 + DiffLoss)
    # DiffLoss head (gpt_video_diff.py:76-78)
    diffloss_d: int = 3
    diffloss_w: int = 1024
    num_sampling_steps: int = 100


class _Embedding:
    """stand-in for `model.tok_embeddings.weight.dtype` (generate.py:154)"""

    class _W:
        def __init__(self):
            self.dtype = torch.float32

    def __init__(self):
        self.weight = self._W()


class _ClsEmbedding:
    """exposes `.uncond_embedding` (generate.py:138)"""

    def __init__(self):
        self.uncond_embedding = None


class Transformer:
    """Drop-in for the inference surface of gpt.py `Transformer`."""

    def __init__(self, config: ModelArgs):
        if config.n_kv_head is not None and config.n_kv_head != config.n_head:
            raise L.VlgError(-3, "n_kv_head != n_head is not used by any reference size (gpt.py:441-464)")
        if config.ffn_dim_multiplier is not None:
            raise L.VlgError(-3, "ffn_dim_multiplier is not used by any reference size")
        if config.model_type not in ('c2i', 't2i', 't2v'):
            raise Exception("please check model type")          # gpt.py:277
        self.config = config
        self.vocab_size = config.vocab_size
        self.n_layer = config.n_layer
        self.block_size = config.block_size
        self.num_classes = config.num_classes
        self.model_type = config.model_type
        self.cls_token_num = config.cls_token_num
        self.num_frames = config.num_frames
        self.t_downsample_size = config.t_downsample_size
        grid_size = int(self.block_size ** 0.5)
        assert grid_size * grid_size == self.block_size
        self.tok_embeddings = _Embedding()
        self.cls_embedding = _ClsEmbedding()
        self.max_batch_size = -1
        self.max_seq_length = -1
        self.causal_mask = None
        self.training = False
        self._dtype = torch.float32
        self._device = None
        self._handle = None
        self._loaded = set()
        self.use_graph = True
        self.time_attn = False
        self.fuse_gemm = True    # decode: fused skinny GEMMs (norm prologue, residual / RoPE+scatter / SwiGLU epilogues)
        self.fuse_swiglu = True  # w1/w3 GEMM with the SiLU*mul epilogue
        self.check_faults = False  # generate() returns with the work enqueued (include/vlg.h stream contract); a device-side time-out surfaces on the handle's next call or through status().  True: wait for the call and raise at once
        self.debug_spin_max = 0   # tests: spin bound of the persistent kernels' in-launch waits (0 = default)
        self.pdecode = True      # decode: all layers of a step as one persistent launch (csrc/pdecode.hip) where the shape allows (<= 16 rows)
        self.debug_pos_offset = 0  # benchmarks: decode as if this many tokens had already been generated (zeroed cache rows): late-context timing
        self.weights_fm = True   # decode GEMMs stream the fragment-major weight copies (one MFMA B fragment = 1 KB contiguous); bit-identical results
        self.act_fm = True       # the fused decode chain keeps its activations A-fragment-major; bit-identical results
        self.pd_rows = 0         # ... up to this many cache rows (0 = the library's measured default)
        self.dl_persist = True   # DiffLoss.sample as one persistent launch per token (csrc/diffloss_persist.hip); False = per-step launch chain;
                                 # 4 / 8 = persistent with that many rows per workgroup group (default: 4 up to 32 rows at W 1024, 8 up to 64)

    # ---- nn.Module-like surface ---------------------------------------------------------------------------
    def eval(self):
        self.training = False
        return self

    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        if dtype is not None and dtype != self._dtype:
            if self._handle is not None:
                raise L.VlgError(-6, "dtype must be chosen before weights are loaded")
            L.torch_dtype_code(dtype)
            self._dtype = dtype
            self.tok_embeddings.weight.dtype = dtype
        if device is not None:
            self._device = torch.device(device)
        return self

    def _head_code(self):
        h = self.config.head
        if h == 'auto':
            h = 'adapter2' if self.model_type == 't2v' else 'logits'
        return {'logits': L.VLG_HEAD_LOGITS, 'adapter2': L.VLG_HEAD_ADAPTER2, 'hidden': L.VLG_HEAD_HIDDEN}[h]

    def _ensure_handle(self):
        if self._handle is not None:
            return
        if self._device is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        if self._device.type != "cuda":
            raise L.VlgError(-3, "video_llamagen_amd runs on an MI355X only (device %s requested)" % self._device)
        c = self.config
        cfg = L.GptConfig(
            dim=c.dim, n_layer=c.n_layer, n_head=c.n_head, vocab_size=c.vocab_size, block_size=c.block_size,
            cls_token_num=c.cls_token_num, model_type={'c2i': L.VLG_C2I, 't2i': L.VLG_T2I, 't2v': L.VLG_T2V}[c.model_type],
            num_classes=c.num_classes, caption_dim=c.caption_dim if c.model_type != 'c2i' else 0,
            vae_embed_dim=c.vae_embed_dim if c.model_type == 't2v' else 0, num_frames=c.num_frames,
            t_downsample_size=c.t_downsample_size, head=self._head_code(), dtype=L.torch_dtype_code(self._dtype),
            multiple_of=c.multiple_of, norm_eps=c.norm_eps, rope_base=float(c.rope_base),
            diffloss_w=c.diffloss_w, diffloss_d=c.diffloss_d, num_sampling_steps=int(c.num_sampling_steps))
        h = C.c_void_p()
        with torch.cuda.device(self._device):
            L.check(L.lib().vlg_gpt_create(C.byref(cfg), C.byref(h)))
        self._handle = h

    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference key names.  Returns (missing, unexpected) like torch."""
        self._ensure_handle()
        unexpected = []
        with torch.cuda.device(self._device):
            for k, v in state_dict.items():
                if L.load_tensor(L.lib().vlg_gpt_load_tensor, self._handle, k, v):
                    self._loaded.add(k)
                    if k == "cls_embedding.uncond_embedding":
                        self.cls_embedding.uncond_embedding = v
                else:
                    unexpected.append(k)
        if strict and unexpected:
            raise RuntimeError("Unexpected key(s) in state_dict: %s" % ", ".join(unexpected))
        return [], unexpected

    def param_shapes(self):
        """name -> shape of every tensor the sampling path needs (reference state-dict names)."""
        c = self.config
        D = c.dim
        F = find_multiple(int(2 * (4 * D) / 3), c.multiple_of)          # gpt.py:154-159
        out = {}
        if c.model_type == 'c2i':
            out["cls_embedding.embedding_table.weight"] = (c.num_classes + 1, D)
        else:
            out["cls_embedding.cap_proj.fc1.weight"] = (D, c.caption_dim)
            out["cls_embedding.cap_proj.fc2.weight"] = (D, D)
            out["cls_embedding.uncond_embedding"] = (120, c.caption_dim)
        if c.model_type == 't2v':
            out["vae_latent_adapter.fc1.weight"] = (D, c.vae_embed_dim)
            out["vae_latent_adapter.fc2.weight"] = (D, D)
            if self._head_code() == L.VLG_HEAD_ADAPTER2:
                out["vae_latent_adapter2.fc1.weight"] = (D, D)
                out["vae_latent_adapter2.fc2.weight"] = (c.vae_embed_dim, D)
            if self._head_code() == L.VLG_HEAD_HIDDEN:                       # diffloss.py:161-190
                Wd, Cc, p = c.diffloss_w, c.vae_embed_dim, "diffloss.net."
                for n, o, i in (("time_embed.mlp.0", Wd, 256), ("time_embed.mlp.2", Wd, Wd), ("cond_embed", Wd, D), ("input_proj", Wd, Cc),
                                ("final_layer.adaLN_modulation.1", 2 * Wd, Wd), ("final_layer.linear", 2 * Cc, Wd)):
                    out[p + n + ".weight"], out[p + n + ".bias"] = (o, i), (o,)
                for b in range(c.diffloss_d):
                    q = p + "res_blocks.%d." % b
                    out[q + "in_ln.weight"], out[q + "in_ln.bias"] = (Wd,), (Wd,)
                    for n, o in (("mlp.0", Wd), ("mlp.2", Wd), ("adaLN_modulation.1", 3 * Wd)):
                        out[q + n + ".weight"], out[q + n + ".bias"] = (o, Wd), (o,)
        else:
            out["tok_embeddings.weight"] = (c.vocab_size, D)
            out["output.weight"] = (c.vocab_size, D)
        for i in range(c.n_layer):
            p = "layers.%d." % i
            out[p + "attention.wqkv.weight"] = (3 * D, D)
            out[p + "attention.wo.weight"] = (D, D)
            out[p + "feed_forward.w1.weight"] = (F, D)
            out[p + "feed_forward.w3.weight"] = (F, D)
            out[p + "feed_forward.w2.weight"] = (D, F)
            out[p + "attention_norm.weight"] = (D,)
            out[p + "ffn_norm.weight"] = (D,)
        out["norm.weight"] = (D,)
        return out

    def init_random_weights(self, seed=0, std=None):
        """Random initialisation on the device, stand-in for the reference constructor's `initialize_weights`
        (gpt.py:302-316: N(0, 0.02) Linear/Embedding; `output.weight` is drawn too instead of zeroed)."""
        self._ensure_handle()
        std = self.config.initializer_range if std is None else std
        g = torch.Generator(device=self._device).manual_seed(seed)
        for name, shape in self.param_shapes().items():
            if name.endswith("norm.weight") or name.endswith("in_ln.weight"):
                t = torch.ones(shape, device=self._device)
            elif name == "cls_embedding.uncond_embedding":
                t = torch.randn(shape, generator=g, device=self._device) / shape[1] ** 0.5
            elif name.startswith("diffloss.") and name.endswith(".bias"):
                t = torch.zeros(shape, device=self._device)
            elif name.startswith("diffloss.") and len(shape) == 2:
                t = torch.randn(shape, generator=g, device=self._device) * (0.5 / shape[1] ** 0.5)
            elif name in ("vae_latent_adapter.fc1.weight", "vae_latent_adapter2.fc2.weight"):
                t = torch.randn(shape, generator=g, device=self._device) * 0.3     # keeps O(1) latents for tiny embed dims
            else:
                t = torch.randn(shape, generator=g, device=self._device) * std
            self.load_state_dict({name: t}, strict=False)
        return self

    def setup_caches(self, max_batch_size, max_seq_length, dtype=None):
        """gpt.py:318-332.  The KV cache itself is (re)sized inside vlg_gpt_generate; this records the shape."""
        self.max_seq_length = find_multiple(max_seq_length, 8)
        self.max_batch_size = max_batch_size

    def __del__(self):
        try:
            if self._handle is not None:
                L.lib().vlg_gpt_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def attn_timing(self):
        """(total ms, total algorithmic bytes, launches) of the event-timed attention kernel (time_attn=True)."""
        ms, by, n = C.c_double(), C.c_double(), C.c_int64()
        L.check(L.lib().vlg_gpt_attn_timing(self._handle, C.byref(ms), C.byref(by), C.byref(n)))
        return ms.value, by.value, n.value

    def status(self, sync=True):
        """Raises VlgError(VLG_ERR_STATE) if a persistent kernel of this handle recorded a time-out since the last check."""
        if self._handle is not None:
            L.check(L.lib().vlg_gpt_status(self._handle, C.c_int32(1 if sync else 0)))

    def graphs_built(self):
        """decode-step graphs instantiated by this handle so far (a repeated generate() of the same shape must not add one)."""
        n = C.c_int64()
        L.check(L.lib().vlg_gpt_graphs_built(self._handle, C.byref(n)))
        return n.value

    def counter(self, key):
        """host-side counters of the handle: "pd_steps" / "chain_steps" (decode steps recorded on the persistent / per-layer path), "graphs_built"."""
        n = C.c_int64()
        L.check(L.lib().vlg_gpt_counter(self._handle, key.encode(), C.byref(n)))
        return n.value

    def attn_event_overhead_ms(self):
        """mean elapsed ms of an empty event pair on the launch stream (calibration of the timing bracket)."""
        ms = C.c_double()
        L.check(L.lib().vlg_gpt_attn_event_overhead(self._handle, C.byref(ms)))
        return ms.value

    def algorithmic_bytes(self):
        w, k, o = C.c_double(), C.c_double(), C.c_double()
        L.check(L.lib().vlg_gpt_last_algorithmic_bytes(self._handle, C.byref(w), C.byref(k), C.byref(o)))
        return w.value, k.value, o.value


#################################################################################
#                                GPT Configs (gpt.py:441-470)                   #
#################################################################################
def GPT_7B(**kwargs):
    return Transformer(ModelArgs(n_layer=32, n_head=32, dim=4096, **kwargs))


def GPT_3B(**kwargs):
    return Transformer(ModelArgs(n_layer=24, n_head=32, dim=3200, **kwargs))


def GPT_1B(**kwargs):
    return Transformer(ModelArgs(n_layer=22, n_head=32, dim=2048, **kwargs))


def GPT_XXXL(**kwargs):
    return Transformer(ModelArgs(n_layer=48, n_head=40, dim=2560, **kwargs))


def GPT_XXL(**kwargs):
    return Transformer(ModelArgs(n_layer=48, n_head=24, dim=1536, **kwargs))


def GPT_XL(**kwargs):
    return Transformer(ModelArgs(n_layer=36, n_head=20, dim=1280, **kwargs))


def GPT_L(**kwargs):
    return Transformer(ModelArgs(n_layer=24, n_head=16, dim=1024, **kwargs))


def GPT_B(**kwargs):
    return Transformer(ModelArgs(n_layer=12, n_head=12, dim=768, **kwargs))


GPT_models = {
    'GPT-B': GPT_B, 'GPT-L': GPT_L, 'GPT-XL': GPT_XL, 'GPT-XXL': GPT_XXL, 'GPT-XXXL': GPT_XXXL,
    'GPT-1B': GPT_1B, 'GPT-3B': GPT_3B, 'GPT-7B': GPT_7B,
}
