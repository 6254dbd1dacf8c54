"""video_llamagen_amd - MI355X-native KV-cached visual-token sampling + VQ / CausalVideoVAE decode.

Host-side mirror of the reference's Python call surface for ONE hot path, over the
C-ABI library libvlg.so (hand-written HIP for gfx950).  There is no CPU fallback: importing the compute
entry points without the built library raises.
"""
from . import _lib  # noqa: F401
from .gpt import GPT_models, ModelArgs, Transformer  # noqa: F401
from .generate import generate, generate_t2v  # noqa: F401
from .vq_model import VQ_models, VQModel, codebook_argmin  # noqa: F401
from .vae_model import VAE_models, CausalVAEModel  # noqa: F401
from .vqvae_video import VQVAE, Codebook  # noqa: F401
from .t5_model import T5EncoderModel, T5Embedder  # noqa: F401
from .serve import LLM, LLMEngine, ContinuousLLMEngine, SamplingParams, RequestOutput  # noqa: F401

__all__ = ["GPT_models", "ModelArgs", "Transformer", "generate", "generate_t2v", "VQ_models", "VQModel", "codebook_argmin",
           "VAE_models", "CausalVAEModel", "VQVAE", "Codebook", "T5EncoderModel", "T5Embedder", "LLM", "LLMEngine", "ContinuousLLMEngine", "SamplingParams", "RequestOutput"]
