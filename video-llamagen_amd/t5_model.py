"""Text-conditioning step in front of the t2i / t2v path: mirror of what language/t5.py uses of `transformers.T5EncoderModel`
(`model(input_ids=..., attention_mask=...)['last_hidden_state']`, language/t5.py:76-80) and of `T5Embedder.get_text_embeddings`
(:62-81) over libvlg's `vlg_t5_*`.  State-dict names are transformers'; `T5EncoderModel.from_config(T5Config-like dict)`.

The tokenizer (sentencepiece `spiece.model`, downloaded from the hub by the reference) is not part of this package: `T5Embedder`
takes any callable with the Hugging Face tokenizer call convention, or ids + mask directly."""
import ctypes as C
from types import SimpleNamespace

import torch

from . import _lib as L

FLAN_T5_XL = dict(d_model=2048, d_kv=64, num_heads=32, d_ff=5120, num_layers=24, vocab_size=32128, relative_attention_num_buckets=32,
                  relative_attention_max_distance=128, layer_norm_epsilon=1e-6, feed_forward_proj="gated-gelu")


class T5EncoderModel:
    def __init__(self, config):
        cfg = dict(config) if isinstance(config, dict) else {k: getattr(config, k) for k in FLAN_T5_XL if hasattr(config, k)}
        for k, v in FLAN_T5_XL.items():
            cfg.setdefault(k, v)
        if cfg["feed_forward_proj"] != "gated-gelu":
            raise L.VlgError(-3, "only feed_forward_proj='gated-gelu' (flan-t5 / t5-v1_1, language/t5.py:16) is supported")
        self.config = SimpleNamespace(**cfg)
        self._dtype = torch.bfloat16          # language/t5.py:22: torch_dtype defaults to bfloat16
        self._device = None
        self._handle = None

    @classmethod
    def from_config(cls, config):
        return cls(config)

    def eval(self):
        return self

    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        if dtype is not None and dtype != self._dtype:
            if self._handle is not None:
                raise L.VlgError(-6, "dtype must be chosen before weights are loaded")
            L.torch_dtype_code(dtype)
            self._dtype = dtype
        if device is not None:
            self._device = torch.device(device)
        return self

    def _ensure_handle(self):
        if self._handle is not None:
            return
        if self._device is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        c = self.config
        cfg = L.T5Config(d_model=c.d_model, d_kv=c.d_kv, num_heads=c.num_heads, d_ff=c.d_ff, num_layers=c.num_layers, vocab_size=c.vocab_size,
                         relative_attention_num_buckets=c.relative_attention_num_buckets,
                         relative_attention_max_distance=c.relative_attention_max_distance, gated_gelu=1,
                         dtype=L.torch_dtype_code(self._dtype), layer_norm_epsilon=c.layer_norm_epsilon)
        h = C.c_void_p()
        with torch.cuda.device(self._device):
            L.check(L.lib().vlg_t5_create(C.byref(cfg), C.byref(h)))
        self._handle = h

    def load_state_dict(self, state_dict, strict=True):
        self._ensure_handle()
        unexpected = []
        with torch.cuda.device(self._device):
            for k, v in state_dict.items():
                if not L.load_tensor(L.lib().vlg_t5_load_tensor, self._handle, k, v) and k != "encoder.embed_tokens.weight":
                    unexpected.append(k)
        if strict and unexpected:
            raise RuntimeError("Unexpected key(s) in state_dict: " + ", ".join(unexpected))
        return [], unexpected

    def init_random_weights(self, seed=0):
        """Random initialisation on the device (stand-in for from_pretrained: the published weights need the network)."""
        self._ensure_handle()
        c = self.config
        g = torch.Generator(device=self._device).manual_seed(seed)
        D, inner, F = c.d_model, c.num_heads * c.d_kv, c.d_ff

        def put(name, shape, std):
            t = torch.ones(shape, device=self._device) if std is None else torch.randn(shape, generator=g, device=self._device) * std
            self.load_state_dict({name: t})

        put("shared.weight", (c.vocab_size, D), 1.0)
        put("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", (c.relative_attention_num_buckets, c.num_heads), 0.5)
        put("encoder.final_layer_norm.weight", (D,), None)
        for l in range(c.num_layers):
            p = "encoder.block.%d.layer." % l
            put(p + "0.SelfAttention.q.weight", (inner, D), (D * c.d_kv) ** -0.5)
            for n in ("k", "v"):
                put(p + "0.SelfAttention.%s.weight" % n, (inner, D), D ** -0.5)
            put(p + "0.SelfAttention.o.weight", (D, inner), inner ** -0.5)
            put(p + "1.DenseReluDense.wi_0.weight", (F, D), D ** -0.5)
            put(p + "1.DenseReluDense.wi_1.weight", (F, D), D ** -0.5)
            put(p + "1.DenseReluDense.wo.weight", (D, F), F ** -0.5)
            put(p + "0.layer_norm.weight", (D,), None)
            put(p + "1.layer_norm.weight", (D,), None)
        return self

    @torch.no_grad()
    def __call__(self, input_ids=None, attention_mask=None):
        self._ensure_handle()
        ids = input_ids.to(device=self._device, dtype=torch.int64).contiguous()
        if ids.dim() != 2:
            raise L.VlgError(-2, "input_ids must be [B, T]")
        B, T = ids.shape
        mask = torch.ones((B, T), device=self._device) if attention_mask is None else attention_mask
        mask = mask.to(device=self._device, dtype=torch.float32).contiguous()
        out = torch.empty((B, T, self.config.d_model), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            L.check(L.lib().vlg_t5_encode(self._handle, L.ptr(ids), L.ptr(mask), B, T, L.ptr(out), L.stream_ptr(self._device)))
        return {"last_hidden_state": out.to(self._dtype)}

    def __del__(self):
        try:
            if self._handle is not None:
                L.lib().vlg_t5_destroy(self._handle)
        except Exception:
            pass


class T5Embedder:
    """language/t5.py:14-81 without the hub download.  `model` is a loaded T5EncoderModel.  `tokenizer` is a Hugging Face style callable, or
    `tokenizer_path` a local directory holding one (spiece.model / tokenizer.json + tokenizer_config.json, as the reference's cache directory
    does): it is opened with transformers.AutoTokenizer - a host-side dependency the reference has too; nothing is fetched.  Captions go
    through the reference's cleaning first (caption.py: clean_caption twice when use_text_preprocessing, else lower().strip())."""

    def __init__(self, device, model, tokenizer=None, model_max_length=120, tokenizer_path=None, use_text_preprocessing=True):
        self.device = torch.device(device)
        self.model = model.to(self.device)
        if tokenizer is None and tokenizer_path is not None:
            import os
            if not os.path.isdir(tokenizer_path):
                raise L.VlgError(-2, "tokenizer_path %r is not a directory (no download is attempted)" % (tokenizer_path,))
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(tokenizer_path, local_files_only=True)
        self.tokenizer = tokenizer
        self.model_max_length = model_max_length
        self.use_text_preprocessing = use_text_preprocessing

    def text_preprocessing(self, text):
        from .caption import text_preprocessing
        return text_preprocessing(text, self.use_text_preprocessing)

    def get_text_embeddings(self, texts):
        if self.tokenizer is None:
            raise L.VlgError(-6, "no tokenizer given: pass tokenizer= / tokenizer_path=, or use get_text_embeddings_from_ids(input_ids, attention_mask)")
        texts = [self.text_preprocessing(t) for t in texts]
        tok = self.tokenizer(texts, max_length=self.model_max_length, padding="max_length", truncation=True,
                             return_attention_mask=True, add_special_tokens=True, return_tensors="pt")
        return self.get_text_embeddings_from_ids(tok["input_ids"], tok["attention_mask"])

    def get_text_embeddings_from_ids(self, input_ids, attention_mask):
        embs = self.model(input_ids=input_ids.to(self.device), attention_mask=attention_mask.to(self.device))["last_hidden_state"].detach()
        return embs, attention_mask.to(self.device)
