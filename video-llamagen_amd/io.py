"""Checkpoint containers and sample writers around the hot path.  Pure host code: nothing here touches the GPU.

Reference behaviour restated:
* GPT checkpoints (sample_t2i.py:62-69, serve/model_runner.py:184-191): `torch.save` dicts whose weights sit under "model" (DDP
  training), "module" (DeepSpeed) or "state_dict"; `--from-fsdp` files are the raw state dict.
* CausalVAE directories (modeling_videobase.py:42-53): `config.json` + `*.ckpt`; the LAST ckpt in glob order is used; a directory without
  any *.ckpt goes to `super().from_pretrained` = diffusers `ModelMixin.from_pretrained` (third-party, not vendored: restated from its
  published behaviour): `config.json` + `diffusion_pytorch_model[.variant].safetensors`, else `diffusion_pytorch_model[.variant].bin`,
  optionally under `subfolder`; the file holds the plain state dict (no "state_dict" / EMA wrapper);
  `init_from_ckpt` (modeling_causalvae.py:578-601) prefers a non-empty "ema_state_dict" (unless NOT_USE_EMA_MODEL is set), strips
  "module." prefixes, else takes "state_dict" (its "gen_model" entry when present), drops `ignore_keys` prefixes, loads strictly.
* custom_to_video (sample_t2v_1f_diff.py:49-58): clamp to [-1, 1], (x + 1) / 2, [C,T,H,W] -> [T,H,W,C], (255 * x) truncated to uint8.
* ADM evaluator input (GETTING_STARTED.md:60): an .npz whose "arr_0" is uint8 [N, H, W, 3].
"""
import glob
import json
import os

import numpy as np
import torch


def select_gpt_state_dict(checkpoint, from_fsdp=False):
    """sample_t2i.py:60-69."""
    if from_fsdp:
        return checkpoint
    for key in ("model", "module", "state_dict"):
        if key in checkpoint:
            return checkpoint[key]
    raise Exception("please check model weight, maybe add --from-fsdp to run command")


def load_gpt_checkpoint(path, from_fsdp=False):
    sd = dict(select_gpt_state_dict(torch.load(path, map_location="cpu"), from_fsdp))
    sd.pop("freqs_cis", None)           # serve/gpt_model.py:323-324; the RoPE table is rebuilt by the engine
    return sd


def select_vae_state_dict(sd, ignore_keys=(), use_ema=None):
    """modeling_causalvae.py:578-599 without the final load."""
    if use_ema is None:
        use_ema = os.environ.get("NOT_USE_EMA_MODEL", 0) == 0
    if "ema_state_dict" in sd and len(sd["ema_state_dict"]) > 0 and use_ema:
        sd = {key.replace("module.", ""): value for key, value in sd["ema_state_dict"].items()}
    elif "state_dict" in sd:
        sd = sd["state_dict"]["gen_model"] if "gen_model" in sd["state_dict"] else sd["state_dict"]
    sd = dict(sd)
    for k in list(sd.keys()):
        if any(k.startswith(ik) for ik in ignore_keys):
            del sd[k]
    return sd


DIFFUSERS_WEIGHTS = "diffusion_pytorch_model"     # diffusers.utils.{SAFETENSORS_,}WEIGHTS_NAME stem


def find_vae_checkpoint(directory, config_name="config.json", subfolder=None, variant=None):
    """modeling_videobase.py:44-53 -> (config dict, path of the weight file to load).  `*.ckpt` directories first (the last file in glob
    order, as the reference); otherwise the diffusers layout its `super().from_pretrained` reads: safetensors preferred, then .bin."""
    ckpt_files = glob.glob(os.path.join(directory, "*.ckpt"))
    if ckpt_files:
        path, cdir = ckpt_files[-1], directory
    else:
        cdir = os.path.join(directory, subfolder) if subfolder else directory
        stem = DIFFUSERS_WEIGHTS + ("." + variant if variant else "")
        cands = [os.path.join(cdir, stem + ext) for ext in (".safetensors", ".bin")]
        path = next((c for c in cands if os.path.isfile(c)), None)
        if path is None:
            raise FileNotFoundError("no *.ckpt under %s and none of %s" % (directory, ", ".join(os.path.basename(c) for c in cands)))
    with open(os.path.join(cdir, config_name)) as f:
        cfg = json.load(f)
    cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}       # diffusers bookkeeping keys (_class_name, _diffusers_version)
    return cfg, path


def load_vae_weight_file(path):
    """-> the object `select_vae_state_dict` takes: a torch.load'ed checkpoint dict, or the plain state dict of a diffusers file."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    return torch.load(path, map_location="cpu")


def video_to_uint8(x):
    """custom_to_video's array step: float [C,T,H,W] in [-1,1] -> uint8 [T,H,W,C] (truncating, as the reference does)."""
    x = torch.clamp(x.detach().float().cpu(), -1, 1)
    x = (x + 1) / 2
    return (255 * x.permute(1, 2, 3, 0).numpy()).astype(np.uint8)


def custom_to_video(x, fps=2.0, output_file="output_video.mp4"):
    """Writes the clip.  mp4 needs OpenCV (as in the reference); without it the frames go to `<stem>.npy` and, when Pillow is
    importable, an animated `<stem>.gif`.  Returns the list of files written."""
    frames = video_to_uint8(x)
    stem = os.path.splitext(output_file)[0]
    try:
        import cv2                                         # sample_t2v_1f_diff.py:37-47
        h, w = frames[0].shape[:2]
        vw = cv2.VideoWriter(output_file, cv2.VideoWriter_fourcc(*"mp4v"), float(fps), (w, h))
        for im in frames:
            vw.write(cv2.cvtColor(im, cv2.COLOR_RGB2BGR))
        vw.release()
        return [output_file]
    except ImportError:
        pass
    written = [stem + ".npy"]
    np.save(written[0], frames)
    try:
        from PIL import Image
        ims = [Image.fromarray(f) for f in frames]
        ims[0].save(stem + ".gif", save_all=True, append_images=ims[1:], duration=int(1000 / max(fps, 1e-3)), loop=0)
        written.append(stem + ".gif")
    except ImportError:
        pass
    return written


def save_samples_npz(samples_uint8, path):
    """uint8 [N,H,W,3] -> `path` (.npz with arr_0), the file evaluations/c2i/evaluator.py reads."""
    arr = np.asarray(samples_uint8)
    if arr.dtype != np.uint8 or arr.ndim != 4 or arr.shape[-1] != 3:
        raise ValueError("expected uint8 [N,H,W,3], got %s %s" % (arr.dtype, arr.shape))
    np.savez(path, arr_0=arr)
    return path
