"""ctypes binding of libvlg.so (include/vlg.h).  Fails loudly when the library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VLG_LIB_PATH: load another build of the same library (tools/asan_host_check.sh points it at the host-AddressSanitizer build)
LIB_PATH = os.environ.get("VLG_LIB_PATH") or os.path.join(_HERE, "lib", "libvlg.so")

VLG_F32, VLG_BF16 = 0, 1
VLG_C2I, VLG_T2I, VLG_T2V = 0, 1, 2
VLG_HEAD_LOGITS, VLG_HEAD_ADAPTER2, VLG_HEAD_HIDDEN = 0, 1, 2

VLG_OK, VLG_ERR_BAD_ARG, VLG_ERR_BAD_SHAPE, VLG_ERR_UNSUPPORTED, VLG_ERR_OOM, VLG_ERR_HIP, VLG_ERR_STATE = 0, -1, -2, -3, -4, -5, -6
STATUS = {0: "VLG_OK", -1: "VLG_ERR_BAD_ARG", -2: "VLG_ERR_BAD_SHAPE", -3: "VLG_ERR_UNSUPPORTED",
          -4: "VLG_ERR_OOM", -5: "VLG_ERR_HIP", -6: "VLG_ERR_STATE"}


class VlgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (STATUS.get(code, code), msg))
        self.code = code


class GptConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dim", "n_layer", "n_head", "vocab_size", "block_size", "cls_token_num", "model_type", "num_classes",
        "caption_dim", "vae_embed_dim", "num_frames", "t_downsample_size", "head", "dtype", "multiple_of")] + [
        ("norm_eps", C.c_float), ("rope_base", C.c_float),
        ("diffloss_w", C.c_int32), ("diffloss_d", C.c_int32), ("num_sampling_steps", C.c_int32)]


class SamplingParams(C.Structure):
    _fields_ = [("cfg_scale", C.c_float), ("cfg_interval", C.c_int32), ("temperature", C.c_float),
                ("top_k", C.c_int32), ("top_p", C.c_float), ("sample_logits", C.c_int32), ("seed", C.c_uint64)]


class VqConfig(C.Structure):
    _fields_ = [("codebook_size", C.c_int32), ("codebook_embed_dim", C.c_int32), ("z_channels", C.c_int32),
                ("ch", C.c_int32), ("n_mult", C.c_int32), ("ch_mult", C.c_int32 * 8), ("num_res_blocks", C.c_int32),
                ("l2_norm", C.c_int32), ("dtype", C.c_int32)]


class VaeConfig(C.Structure):
    _fields_ = [("hidden_size", C.c_int32), ("z_channels", C.c_int32), ("embed_dim", C.c_int32),
                ("num_res_blocks", C.c_int32), ("n_mult", C.c_int32), ("hidden_size_mult", C.c_int32 * 8),
                ("spatial_upsample", C.c_int32 * 8), ("temporal_upsample", C.c_int32 * 8), ("dtype", C.c_int32)]


class VqvaeConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_hiddens", "embedding_dim", "n_codes", "n_res_layers", "n_head", "n_upsample", "dtype")]


class T5Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d_model", "d_kv", "num_heads", "d_ff", "num_layers", "vocab_size", "relative_attention_num_buckets",
                                         "relative_attention_max_distance", "gated_gelu", "dtype")] + [("layer_norm_epsilon", C.c_float)]


# every symbol include/vlg.h declares
SYMBOLS = [
    "vlg_last_error", "vlg_version",
    "vlg_gpt_create", "vlg_gpt_destroy", "vlg_gpt_load_tensor", "vlg_gpt_generate",
    "vlg_gpt_last_algorithmic_bytes", "vlg_gpt_graphs_built", "vlg_gpt_set_option", "vlg_gpt_attn_timing", "vlg_gpt_attn_event_overhead",
    "vlg_gpt_session_begin", "vlg_gpt_session_prefill", "vlg_gpt_session_prefill_batch", "vlg_gpt_session_step", "vlg_gpt_session_read", "vlg_gpt_session_read_latents", "vlg_gpt_session_end",
    "vlg_gpt_session_reserve", "vlg_gpt_session_release", "vlg_gpt_session_free_blocks", "vlg_gpt_set_option_f64", "vlg_gpt_set_teacher", "vlg_gpt_status", "vlg_gpt_counter",
    "vlg_rmsnorm", "vlg_linear", "vlg_rope_table", "vlg_sample", "vlg_attn_decode",
    "vlg_vq_create", "vlg_vq_destroy", "vlg_vq_load_tensor", "vlg_vq_decode_code", "vlg_vq_argmin",
    "vlg_codebook_argmin", "vlg_codebook_forward", "vlg_causal_conv3d", "vlg_group_norm", "vlg_time_upsample2x",
    "vlg_vae_create", "vlg_vae_destroy", "vlg_vae_load_tensor", "vlg_vae_decode", "vlg_vae_out_shape",
    "vlg_vq_encode", "vlg_vae_encode", "vlg_tile_blend", "vlg_conv_timing", "vlg_conv_timing_read",
    "vlg_vqvae_create", "vlg_vqvae_destroy", "vlg_vqvae_load_tensor", "vlg_vqvae_decode",
    "vlg_t5_create", "vlg_t5_destroy", "vlg_t5_load_tensor", "vlg_t5_encode",
]

_lib = None


def lib():
    """The loaded library; raises if it was never built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libvlg.so is missing at %s - build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.vlg_last_error.restype = C.c_char_p
        for s in SYMBOLS:
            getattr(_lib, s)  # AttributeError if the header and the library ever diverge
    return _lib


def check(code):
    if code != 0:
        raise VlgError(code, lib().vlg_last_error().decode(errors="replace"))


def ptr(t):
    """device/host pointer of a torch tensor (or None)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def torch_dtype_code(dt):
    import torch
    if dt == torch.float32:
        return VLG_F32
    if dt == torch.bfloat16:
        return VLG_BF16
    raise VlgError(-3, "unsupported dtype %s (float32 and bfloat16 only)" % dt)


def load_tensor(fn, handle, name, t):
    """Uploads one state-dict tensor through a *_load_tensor entry point; returns True if it was consumed."""
    import torch
    t = t.detach()
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    t = t.contiguous()
    shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
    consumed = C.c_int32(0)
    check(fn(handle, name.encode(), ptr(t), shape, C.c_int32(t.dim()), C.c_int32(torch_dtype_code(t.dtype)),
             C.c_int32(1 if t.is_cuda else 0), C.byref(consumed)))
    return bool(consumed.value)
