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
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Decoder"][0]
    code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
    mu = importlib.import_module("causalvideovae.model.utils.module_utils")
    ns = dict(nn=nn, torch=torch, Tuple=Tuple, Module=str, Normalize=mods.Normalize,
              nonlinearity=importlib.import_module("causalvideovae.model.modules.ops").nonlinearity,
              resolve_str_to_obj=mu.resolve_str_to_obj)
    exec(code, ns)
    return ns["Decoder"], mods
