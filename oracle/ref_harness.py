"""Loads the REAL reference (/root/reference) on CPU to generate golden vectors.

Runs only in the build container (the reference does not travel to the GPU box);
used by tests/golden/make_goldens.py.  Test infrastructure (oracle/__init__.py).

Shims: a no-op `ipdb` (breakpoints at gpt.py:350, generate.py:121,172);
gpt_video.py is exec'd up to its training-script imports (gpt_video.py:613); the
CausalVAE `Decoder` class is AST-extracted from modeling_causalvae.py:151-262 because the
file's module-level imports need diffusers/decord; `causalvideovae.model` is registered as
a bare package so `modules/*` import without model/__init__.py.
"""
import ast
import importlib
import os
import sys
import types

REF = os.environ.get("VLG_REFERENCE", "/root/reference")


def _install():
    if "ipdb" not in sys.modules:
        sys.modules["ipdb"] = types.SimpleNamespace(set_trace=lambda *a, **k: None)
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load_gpt():
    _install()
    from autoregressive.models import gpt, generate
    return gpt, generate


def load_vq():
    _install()
    from tokenizer.tokenizer_image import vq_model
    return vq_model


def load_gpt_video():
    """exec gpt_video.py:1-612 (class part) into a fresh module."""
    _install()
    path = os.path.join(REF, "autoregressive/models/gpt_video.py")
    src = open(path).read()
    cut = src.index("from einops import rearrange, repeat")
    mod = types.ModuleType("ref_gpt_video")
    mod.__file__ = path
    exec(compile(src[:cut], path, "exec"), mod.__dict__)
    return mod


def load_gpt_video_diff():
    """exec gpt_video_diff.py:1-1091 (class part; needs autoregressive/models on sys.path for `from diffloss import`)."""
    _install()
    mdir = os.path.join(REF, "autoregressive/models")
    if mdir not in sys.path:
        sys.path.insert(0, mdir)
    path = os.path.join(mdir, "gpt_video_diff.py")
    src = open(path).read()
    cut = src.index("from einops import rearrange, repeat")
    mod = types.ModuleType("ref_gpt_video_diff")
    mod.__file__ = path
    exec(compile(src[:cut], path, "exec"), mod.__dict__)
    gen = importlib.import_module("autoregressive.models.generate_video_diff")
    return mod, gen


def load_vae_modules():
    _install()
    root = os.path.join(REF, "CausalVideoVAE")
    if root not in sys.path:
        sys.path.insert(0, root)
    for name, sub in (("causalvideovae", "causalvideovae"), ("causalvideovae.model", "causalvideovae/model")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(root, sub)]
            sys.modules[name] = m
    return importlib.import_module("causalvideovae.model.modules")


def load_vae_decoder_cls():
    mods = load_vae_modules()
    import torch
    import torch.nn as nn
    from typing import Tuple
    path = os.path.join(REF, "CausalVideoVAE/causalvideovae/model/causal_vae/modeling_causalvae.py")
    tree = ast.parse(open(path).read())
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("Decoder", "Encoder")]
    code = compile(ast.Module(body=node, type_ignores=[]), path, "exec")
    mu = importlib.import_module("causalvideovae.model.utils.module_utils")
    ns = dict(nn=nn, torch=torch, Tuple=Tuple, Module=str, Normalize=mods.Normalize,
              nonlinearity=importlib.import_module("causalvideovae.model.modules.ops").nonlinearity,
              resolve_str_to_obj=mu.resolve_str_to_obj)
    exec(code, ns)
    mods.RefEncoder = ns["Encoder"]
    return ns["Decoder"], mods


def load_videovq_classes():
    """tokenizer/tokenizer_video/vqvae.py imports pytorch_lightning (absent): AST-extract the decode-side classes and exec
    them with the (importable) attention/utils helpers in scope."""
    _install()
    import importlib.util
    import math
    import numpy as np
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    base = os.path.join(REF, "tokenizer/tokenizer_video")

    def load_file(name):
        spec = importlib.util.spec_from_file_location("ref_tv_" + name, os.path.join(base, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        return spec, mod

    ut_src = open(os.path.join(base, "utils.py")).read()
    ns_ut = {}
    tree = ast.parse(ut_src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("shift_dim", "view_range", "tensor_slice")]
    exec(compile(ast.Module(body=keep, type_ignores=[]), "utils.py", "exec"), ns_ut)
    at_src = open(os.path.join(base, "attention.py")).read()
    tree = ast.parse(at_src)
    keep = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name in ("MultiHeadAttention", "AxialAttention", "FullAttention"))
            or (isinstance(n, ast.FunctionDef) and n.name == "scaled_dot_product_attention")]
    ns = dict(nn=nn, torch=torch, F=F, np=np, math=math, **ns_ut)
    ns["SparseAttention"] = None
    exec(compile(ast.Module(body=keep, type_ignores=[]), "attention.py", "exec"), ns)
    vq_src = open(os.path.join(base, "vqvae.py")).read()
    tree = ast.parse(vq_src)
    keep = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in
            ("AxialBlock", "AttentionResidualBlock", "Decoder", "SamePadConv3d", "SamePadConvTranspose3d")]
    exec(compile(ast.Module(body=keep, type_ignores=[]), "vqvae.py", "exec"), ns)
    return ns


def load_vae_tiling_methods():
    """CausalVAEModel's tiled_decode / tiled_decode2d / tiled_encode / tiled_encode2d / blend_v / blend_h as plain functions (the class
    itself needs diffusers).  tiled_encode returns the moments tensor (DiagonalGaussianDistribution is replaced by the identity)."""
    import torch
    path = os.path.join(REF, "CausalVideoVAE/causalvideovae/model/causal_vae/modeling_causalvae.py")
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CausalVAEModel"][0]
    keep = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("tiled_decode", "tiled_decode2d", "tiled_encode", "tiled_encode2d",
                                                                                 "blend_v", "blend_h")]
    stub = ast.ClassDef(name="TilingStub", bases=[], keywords=[], body=keep, decorator_list=[])
    mod = ast.Module(body=[stub], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(torch=torch, DiagonalGaussianDistribution=lambda moments: moments)
    exec(compile(mod, path, "exec"), ns)
    return ns["TilingStub"]
