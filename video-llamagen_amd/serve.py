"""Request-level serving front-end over the same decode kernels.

Counterpart of the reference's `autoregressive/serve/` (a patched vLLM 0.4.1: llm.py:20-267 `LLM`, llm_engine.py `LLMEngine.add_request/step`,
sampler.py:46-125 classifier-free guidance inside the sampler, sample_c2i.py:33-66 the calling convention).  The call surface is kept:

    llm = LLM(args=args, model="GPT-XL")                       # args: gpt_model / gpt_ckpt / cfg_scale / precision / ... (sample_c2i.py:72-95)
    sp = SamplingParams(temperature=1.0, top_p=1.0, top_k=2000, max_tokens=latent_size ** 2)
    outs = llm.generate(prompt_token_ids=[[c] for c in labels] + [[1000]] * len(labels), sampling_params=sp, use_tqdm=False)
    ids = [o.outputs[0].token_ids for o in outs]

Scheduling.  Two engines share the call surface:
* `LLMEngine` (default of `LLM`): WAVES.  Every request of the reference's workload produces exactly `max_tokens` tokens, so sequences
  that start together end together; `step()` takes up to `max_num_seqs` waiting requests with identical sampling parameters, runs
  them through `vlg_gpt_generate` as one batch and finishes them all.
* `ContinuousLLMEngine` (`LLM(..., continuous=True)`): ITERATION-LEVEL batching over `vlg_gpt_session_*` - a fixed set of KV-cache slots,
  every `step()` advances each occupied slot by one token at its own position; a finished request frees its slot and a waiting one
  (any `max_tokens` up to the session's) starts in it on the next step, mid-flight of the others.  One hipGraph per session.

Classifier-free guidance follows sampler.py:54-58,106-108: with `args.cfg_scale > 1` the second half of the prompts are the null-class
rows; logits are combined `u + (c - u) * s`, one token is drawn per pair and written to both members.  Here the pair shares one row of
the user batch (the engine doubles it internally), and both requests of the pair report the same token ids.

The sampling filter is the eager path's (generate.py:16-54); it coincides with vLLM's `_apply_top_k_top_p` for `top_p = 1` (the
setting of serve/sample_c2i.py) — top-k keeps ties in both.
"""
import collections
import dataclasses
import time
from typing import Dict, List, Optional, Union

import torch

from .generate import generate as _generate
from .gpt import GPT_models
from .sample_common import load_or_init


@dataclasses.dataclass(frozen=True)
class SamplingParams:
    """The subset of vllm.SamplingParams the reference's script sets (sample_c2i.py:47-49) plus a seed."""
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = -1            # -1 = all tokens (vLLM convention)
    max_tokens: int = 16
    seed: Optional[int] = None

    def __post_init__(self):
        if self.temperature < 0:
            raise ValueError(f"temperature must be non-negative, got {self.temperature}.")
        if not 0.0 < self.top_p <= 1.0:
            raise ValueError(f"top_p must be in (0, 1], got {self.top_p}.")
        if self.top_k < -1 or self.top_k == 0:
            raise ValueError(f"top_k must be -1 (disable), or at least 1, got {self.top_k}.")
        if self.max_tokens < 1:
            raise ValueError(f"max_tokens must be at least 1, got {self.max_tokens}.")


@dataclasses.dataclass
class CompletionOutput:
    index: int
    token_ids: List[int]
    text: str = ""
    finish_reason: Optional[str] = "length"
    latents: Optional[torch.Tensor] = None           # continuous-latent video models: float [max_tokens, vae_embed_dim] (token_ids stays empty)


@dataclasses.dataclass
class RequestOutput:
    request_id: str
    prompt: Optional[str]
    prompt_token_ids: List[int]
    outputs: List[CompletionOutput]
    finished: bool = True


@dataclasses.dataclass
class _Request:
    request_id: str
    prompt_token_ids: List[int]
    params: SamplingParams
    arrival: float
    prompt_embeds: Optional[torch.Tensor] = None     # text-conditioned models: [cls_token_num, caption_dim] features (already * mask)
    emb_mask: Optional[torch.Tensor] = None          # [cls_token_num], 1 = valid, left-padded
    order: int = 0                                   # arrival order inside an engine (preemption picks the youngest running request)


class Scheduler:
    """FIFO wave scheduler (host logic only; unit-tested on CPU).  A wave = requests with equal SamplingParams, at most
    `max_num_seqs` of them; with guidance a wave holds whole (cond, uncond) pairs, `max_num_seqs` counting both members
    (the reference feeds 2B prompts for B images, sample_c2i.py:35-37)."""

    def __init__(self, max_num_seqs=256, cfg=False, null_token=None):
        if max_num_seqs < (2 if cfg else 1):
            raise ValueError("max_num_seqs too small")
        self.max_num_seqs = max_num_seqs
        self.cfg = cfg
        self.null_token = null_token
        self.waiting = collections.deque()

    def add(self, req):
        self.waiting.append(req)

    def __len__(self):
        return len(self.waiting)

    def _is_null(self, req):
        return self.cfg and len(req.prompt_token_ids) == 1 and req.prompt_token_ids[0] == self.null_token

    def next_wave(self):
        """-> (cond requests, partner uncond requests or None per cond).  Requests stay queued when their partner has not arrived."""
        if not self.waiting:
            return [], []
        head = self.waiting[0].params
        same = [r for r in self.waiting if r.params == head]
        if not self.cfg:
            wave = same[:self.max_num_seqs]
            for r in wave:
                self.waiting.remove(r)
            return wave, [None] * len(wave)
        conds = [r for r in same if not self._is_null(r)]
        nulls = [r for r in same if self._is_null(r)]
        n = min(len(conds), len(nulls), self.max_num_seqs // 2)
        if n == 0:
            raise ValueError("classifier-free guidance needs one null-class prompt [[%s]] per conditional prompt (sample_c2i.py:36-37)"
                             % self.null_token)
        conds, nulls = conds[:n], nulls[:n]          # i-th conditional pairs with the i-th null prompt, in arrival order
        for r in conds + nulls:
            self.waiting.remove(r)
        return conds, nulls


class LLMEngine:
    """add_request / step / has_unfinished_requests, as llm_engine.py exposes them to llm.py:222-260."""

    def __init__(self, model, cfg_scale=1.0, cfg_interval=-1, max_num_seqs=256, seed=0):
        self.model = model
        self.cfg_scale = float(cfg_scale)
        self.cfg_interval = cfg_interval
        self.seed = seed
        self.scheduler = Scheduler(max_num_seqs, cfg=self.cfg_scale > 1.0, null_token=getattr(model, "num_classes", None))
        self.waves_run = 0

    def add_request(self, request_id, prompt, sampling_params, prompt_token_ids=None, **_):
        if prompt_token_ids is None:
            raise ValueError("prompt_token_ids is required (skip_tokenizer_init=True, sample_c2i.py:43)")
        if self.model.model_type == "c2i" and len(prompt_token_ids) != 1:
            raise ValueError("class-conditional prompts hold exactly one class id")
        self.scheduler.add(_Request(str(request_id), list(prompt_token_ids), sampling_params or SamplingParams(), time.time()))

    def get_num_unfinished_requests(self):
        return len(self.scheduler)

    def has_unfinished_requests(self):
        return len(self.scheduler) > 0

    def step(self) -> List[RequestOutput]:
        conds, nulls = self.scheduler.next_wave()
        if not conds:
            return []
        sp = conds[0].params
        dev = self.model._device
        if self.model.model_type != "c2i":
            raise NotImplementedError("the serving front-end covers class-conditional prompts (serve/sample_c2i.py is the only caller)")
        c_indices = torch.tensor([r.prompt_token_ids[0] for r in conds], device=dev, dtype=torch.long)
        seed = sp.seed if sp.seed is not None else self.seed + self.waves_run
        ids = _generate(self.model, c_indices, sp.max_tokens, cfg_scale=self.cfg_scale, cfg_interval=self.cfg_interval,
                        temperature=sp.temperature, top_k=0 if sp.top_k == -1 else sp.top_k, top_p=sp.top_p,
                        sample_logits=sp.temperature > 0, seed=seed)
        self.waves_run += 1
        rows = ids.cpu().tolist()
        self.model.status(sync=False)      # generate() is asynchronous; the copy above waited for it: a device-side time-out surfaces here
        outs = []
        for r, row in zip(conds, rows):
            outs.append(RequestOutput(r.request_id, None, r.prompt_token_ids, [CompletionOutput(0, row)]))
        for r, row in zip(nulls, rows):
            if r is not None:                          # sampler.py:106-108: the unconditional member repeats its partner's tokens
                outs.append(RequestOutput(r.request_id, None, r.prompt_token_ids, [CompletionOutput(0, list(row))]))
        return outs


class ContinuousLLMEngine:
    """Iteration-level scheduler (vLLM's scheduler + model runner loop, llm_engine.py `step`): slots instead of waves.  All requests of
    a session share the sampling parameters of the first one except `max_tokens` (<= the session's)."""

    def __init__(self, model, cfg_scale=1.0, cfg_interval=-1, max_num_seqs=256, seed=0, max_tokens=None, kv_block_size=0,
                 num_kv_blocks=0, kv_policy="reserve"):
        """kv_block_size > 0: block-granular KV cache (vLLM's `block_size` / `num_gpu_blocks`); num_kv_blocks = 0 sizes the pool for
        every slot at full length.  kv_policy:
          "reserve"  every request reserves blocks for ITS whole length at admission and returns them when it finishes; a request the
                     pool cannot cover yet stays queued (no running request is ever disturbed);
          "grow"     vLLM's scheduler policy (the block manager behind autoregressive/serve/: blocks are appended as a sequence grows,
                     and when the pool runs dry the most recently arrived running sequence is PREEMPTED by recomputation): a request is
                     admitted with the blocks of its condition + first token, grows block by block, and on exhaustion the youngest
                     running request gives its blocks back and returns to the head of the queue to start over.  More requests in flight
                     for the same pool; greedy results are unchanged, sampled ones depend on the slot a request ends up in."""
        import ctypes as C
        from . import _lib as L
        self._C, self._L = C, L
        self.model = model
        self.cfg_scale, self.cfg_interval, self.seed = float(cfg_scale), cfg_interval, seed
        self.cfg = self.cfg_scale > 1.0
        self.null_token = model.num_classes
        self.text = model.model_type in ("t2i", "t2v")   # conditions are caption features: prefilled per slot, guidance partner internal
        self.latent = model.model_type == "t2v"       # continuous-latent video models: a request's output is latents [max_tokens, vae_embed_dim]
        if self.latent and self.cfg:
            raise ValueError("sessions of the continuous-latent models run without transformer guidance (cfg_scale 1), as generate_t2v's shipped mode")
        self.slots_n = max(1, max_num_seqs // 2 if (self.cfg and not self.text) else max_num_seqs)
        self.max_tokens = max_tokens
        self.kv_block_size, self.num_kv_blocks = int(kv_block_size), int(num_kv_blocks)
        if kv_policy not in ("reserve", "grow"):
            raise ValueError("kv_policy must be 'reserve' or 'grow'")
        self.kv_policy = kv_policy if self.kv_block_size > 0 else "reserve"
        self._arrivals = 0
        self.preempted = 0                            # running requests sent back to the queue because the pool ran dry (kv_policy "grow")
        self.deferred = 0                             # admissions postponed because the KV pool was full
        self.waiting = collections.deque()
        self.pending_null = collections.deque()      # null-class partner requests (reported with their partner's tokens)
        self.slots = [None] * self.slots_n           # (request, tokens done, partner request or None)
        self._open = False
        self._params = None
        self.steps_run = 0

    def add_request(self, request_id, prompt, sampling_params, prompt_token_ids=None, prompt_embeds=None, emb_mask=None, **_):
        if self.text:
            # text-conditioned request: T5 features [cls_token_num, caption_dim] (* mask) + left-padded mask, as sample_t2i.py:105-119
            # builds them; under guidance the unconditional partner (uncond_embedding, generate.py:138) lives inside the slot
            if prompt_embeds is None:
                raise ValueError("text-conditioned requests carry prompt_embeds [cls_token_num, caption_dim]")
            want = (self.model.cls_token_num, self.model.config.caption_dim)
            if tuple(prompt_embeds.shape) != want:
                raise ValueError("prompt_embeds must be %s, got %s" % (want, tuple(prompt_embeds.shape)))
            self._arrivals += 1
            self.waiting.append(_Request(str(request_id), [], sampling_params or SamplingParams(), time.time(), prompt_embeds, emb_mask, self._arrivals))
            return
        if prompt_token_ids is None or len(prompt_token_ids) != 1:
            raise ValueError("class-conditional prompts hold exactly one class id")
        self._arrivals += 1
        r = _Request(str(request_id), list(prompt_token_ids), sampling_params or SamplingParams(), time.time(), order=self._arrivals)
        if self.cfg and r.prompt_token_ids[0] == self.null_token:
            self.pending_null.append(r)
        else:
            self.waiting.append(r)

    def get_num_unfinished_requests(self):
        return len(self.waiting) + len(self.pending_null) + sum(1 + (s[2] is not None) for s in self.slots if s)

    def has_unfinished_requests(self):
        return bool(self.waiting) or any(self.slots)

    def _begin(self, sp):
        L, C = self._L, self._C
        self.model._ensure_handle()
        n = self.max_tokens or max([sp.max_tokens] + [r.params.max_tokens for r in self.waiting])
        c = L.SamplingParams(cfg_scale=self.cfg_scale, cfg_interval=int(self.cfg_interval), temperature=float(sp.temperature),
                             top_k=0 if sp.top_k == -1 else int(sp.top_k), top_p=float(sp.top_p), sample_logits=1 if sp.temperature > 0 else 0,
                             seed=int(sp.seed if sp.seed is not None else self.seed))
        with torch.cuda.device(self.model._device):
            L.check(L.lib().vlg_gpt_set_option(self.model._handle, b"kv_block", C.c_int64(self.kv_block_size)))
            L.check(L.lib().vlg_gpt_set_option(self.model._handle, b"kv_pool_blocks", C.c_int64(self.num_kv_blocks)))
            if self.latent:   # the handle keeps options across calls: the session runs the mirror's current kernel switches, and no DiffLoss guidance
                m = self.model
                L.check(L.lib().vlg_gpt_set_option(m._handle, b"graph", C.c_int64(1 if m.use_graph else 0)))
                L.check(L.lib().vlg_gpt_set_option(m._handle, b"fuse_gemm", C.c_int64(1 if m.fuse_gemm else 0)))
                L.check(L.lib().vlg_gpt_set_option(m._handle, b"dl_persist", C.c_int64(int(getattr(m, "dl_persist", True)))))
                if m._head_code() == L.VLG_HEAD_HIDDEN:
                    L.check(L.lib().vlg_gpt_set_option_f64(m._handle, b"cfg_iter", C.c_double(1.0)))
            L.check(L.lib().vlg_gpt_session_begin(self.model._handle, self.slots_n, n, C.byref(c)))
        self._open, self._params, self.session_tokens = True, sp, n

    def free_kv_blocks(self):
        """Blocks free in the pool right now (-1: the session has contiguous slots)."""
        n, bs = self._C.c_int32(0), self._C.c_int32(0)
        self._L.check(self._L.lib().vlg_gpt_session_free_blocks(self.model._handle, self._C.byref(n), self._C.byref(bs)))
        return n.value

    def _reserve(self, slot, r, tokens=None):
        """True when the slot now owns KV blocks for `tokens` tokens of the request (default: all of them); False = pool full."""
        n = int(r.params.max_tokens) if tokens is None else max(1, min(int(tokens), int(r.params.max_tokens)))
        rc = self._L.lib().vlg_gpt_session_reserve(self.model._handle, slot, n)
        if rc == self._L.VLG_ERR_OOM:
            return False
        self._L.check(rc)
        return True

    def _preempt(self, slot):
        """vLLM's preemption by recomputation: the request leaves its slot, its blocks return to the pool, and it goes back to the HEAD of
        the queue (with its null-class partner) to start over when blocks are free again."""
        r, _, partner = self.slots[slot]
        self._L.check(self._L.lib().vlg_gpt_session_release(self.model._handle, slot))
        self.slots[slot] = None
        self.waiting.appendleft(r)
        if partner is not None:
            self.pending_null.appendleft(partner)
        self.preempted += 1

    def _grow(self):
        """kv_policy "grow": before an iteration every running request must own the block its next position falls into.  Oldest requests
        first; when the pool cannot serve one, the YOUNGEST running request is preempted (possibly the asker itself).  Returns True when
        something was preempted (no admission in this iteration then: the freed blocks belong to the survivors)."""
        order = sorted((i for i, s in enumerate(self.slots) if s is not None), key=lambda i: self.slots[i][0].order)
        hit = False
        for i in order:
            while self.slots[i] is not None and not self._reserve(i, self.slots[i][0], self.slots[i][1] + 1):
                running = [j for j, s in enumerate(self.slots) if s is not None]
                victim = max(running, key=lambda j: self.slots[j][0].order)
                if len(running) == 1:
                    raise ValueError("request %s does not fit the KV pool even when it runs alone" % self.slots[i][0].request_id)
                self._preempt(victim)
                hit = True
        return hit

    def _prefill(self, starts):
        """Conditions of the requests admitted in this iteration into their slots' KV rows: runs of CONSECUTIVE slots go through one
        vlg_gpt_session_prefill_batch call (one prefill of n x 119 rows instead of n of 119)."""
        L, dev = self._L, self.model._device
        runs, cur = [], [starts[0]]
        for item in starts[1:]:
            if item[0] == cur[-1][0] + 1 and (item[1].emb_mask is None) == (cur[0][1].emb_mask is None):
                cur.append(item)
            else:
                runs.append(cur)
                cur = [item]
        runs.append(cur)
        with torch.cuda.device(dev):
            torch.cuda.current_stream(dev).synchronize()            # the features may come from another stream's work
            for run in runs:
                emb = torch.stack([r.prompt_embeds.to(device=dev, dtype=torch.float32) for _, r in run]).contiguous()
                msk = None if run[0][1].emb_mask is None else torch.stack([r.emb_mask.to(device=dev, dtype=torch.float32) for _, r in run]).contiguous()
                L.check(L.lib().vlg_gpt_session_prefill_batch(self.model._handle, run[0][0], len(run), L.ptr(emb), L.ptr(msk)))

    def close(self):
        if self._open:
            self._L.check(self._L.lib().vlg_gpt_session_end(self.model._handle))
            self._open = False

    def step(self) -> List[RequestOutput]:
        L, C = self._L, self._C
        if not self._open:
            if not self.waiting:
                return []
            self._begin(self.waiting[0].params)
        row_class = (C.c_int32 * self.slots_n)()
        grow = self.kv_policy == "grow"
        blocked = self._grow() if grow else False                   # growth first; after a preemption nobody is admitted in this iteration
        starts = []                                                 # text-conditioned requests admitted in this iteration: (slot, request)
        # the head of the queue waits for KV blocks: nobody overtakes it (FIFO),
        for i, s in enumerate(self.slots):                          # but every later slot still gets its own code (-1 running, -2 idle)
            if s is not None:
                row_class[i] = -1                                   # continue
                continue
            row_class[i] = -2                                       # idle unless a waiting request fits
            if self.waiting and not blocked:
                r = self.waiting[0]
                if r.params.max_tokens > self.session_tokens:
                    raise ValueError("request %s asks for %d tokens, the session holds %d" % (r.request_id, r.params.max_tokens, self.session_tokens))
                if self.cfg and not self.text and not self.pending_null:
                    continue                                        # its null-class partner has not arrived yet
                if not self._reserve(i, r, 1 if grow else None):
                    self.deferred += 1
                    if not any(self.slots):
                        raise ValueError("request %s does not fit the KV pool even when it is empty" % r.request_id)
                    blocked = True                                  # FIFO: wait for blocks instead of overtaking
                    continue
                if self.text:
                    self.waiting.popleft()
                    starts.append((i, r))                           # prefilled below, consecutive slots in one call
                    self.slots[i] = [r, 0, None]
                    row_class[i] = -3                               # first iteration: the last condition token, samples token 0
                    continue
                self.waiting.popleft()
                partner = self.pending_null.popleft() if self.cfg else None
                self.slots[i] = [r, 0, partner]
                row_class[i] = r.prompt_token_ids[0]
        if not any(self.slots):
            if self.waiting:
                raise ValueError("classifier-free guidance needs one null-class prompt [[%s]] per conditional prompt" % self.null_token)
            return []
        if starts:
            self._prefill(starts)
        with torch.cuda.device(self.model._device):
            L.check(L.lib().vlg_gpt_session_step(self.model._handle, row_class))
        self.steps_run += 1
        outs = []
        for i, s in enumerate(self.slots):
            if s is None:
                continue
            s[1] += 1
            r, done, partner = s
            if done == r.params.max_tokens:
                if self.latent:
                    cdim = self.model.config.vae_embed_dim
                    fbuf = (C.c_float * (done * cdim))()
                    L.check(L.lib().vlg_gpt_session_read_latents(self.model._handle, i, done, fbuf))
                    lat = torch.frombuffer(fbuf, dtype=torch.float32).clone().view(done, cdim)
                    self.model.status(sync=False)      # a time-out inside the persistent DiffLoss sampler surfaces here, not as NaN latents
                    outs.append(RequestOutput(r.request_id, None, r.prompt_token_ids, [CompletionOutput(0, [], latents=lat)]))
                    L.check(L.lib().vlg_gpt_session_release(self.model._handle, i))
                    self.slots[i] = None
                    continue
                buf = (C.c_int32 * done)()
                L.check(L.lib().vlg_gpt_session_read(self.model._handle, i, done, buf))
                toks = list(buf)
                outs.append(RequestOutput(r.request_id, None, r.prompt_token_ids, [CompletionOutput(0, toks)]))
                if partner is not None:
                    outs.append(RequestOutput(partner.request_id, None, partner.prompt_token_ids, [CompletionOutput(0, list(toks))]))
                L.check(L.lib().vlg_gpt_session_release(self.model._handle, i))
                self.slots[i] = None
        if not self.has_unfinished_requests():
            self.close()
        return outs


class LLM:
    """llm.py:20-267.  `model` names the GPT size ("GPT-XL" or a path ending in "GPT-XL.json", the reference's fake_json convention);
    `args` carries gpt_ckpt / gpt_type / cfg_scale / precision / image_size / downsample_size / num_classes / cls_token_num / from_fsdp."""

    def __init__(self, args, model, skip_tokenizer_init=True, seed=0, gpu_memory_utilization=0.9, max_num_seqs=256, dtype="auto",
                 device="cuda", continuous=False, **kwargs):
        name = model.split("/")[-1]
        name = name[:-5] if name.endswith(".json") else name
        if name not in GPT_models:
            raise ValueError("unknown GPT model %r" % model)
        precision = {"none": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[getattr(args, "precision", "bf16")]
        if precision is torch.float16:
            raise ValueError("fp16 is not supported: the engine computes in bf16 or fp32")
        latent = getattr(args, "image_size", 384) // getattr(args, "downsample_size", 16)
        gpt = GPT_models[name](block_size=latent ** 2, num_classes=getattr(args, "num_classes", 1000),
                               cls_token_num=getattr(args, "cls_token_num", 1), model_type=getattr(args, "gpt_type", "c2i"))
        gpt = gpt.to(device=device, dtype=precision)
        self.weights = load_or_init(gpt, getattr(args, "gpt_ckpt", None), seed)
        gpt.eval()
        engine = ContinuousLLMEngine if continuous else LLMEngine
        self.llm_engine = engine(gpt, cfg_scale=getattr(args, "cfg_scale", 1.0), cfg_interval=getattr(args, "cfg_interval", -1),
                                 max_num_seqs=max_num_seqs, seed=seed)
        self._counter = 0

    def generate(self, prompts=None, sampling_params: Optional[Union[SamplingParams, List[SamplingParams]]] = None,
                 prompt_token_ids: Optional[List[List[int]]] = None, use_tqdm=True, **_) -> List[RequestOutput]:
        if prompts is not None:
            raise ValueError("prompts must be None if skip_tokenizer_init is True")            # llm.py:176-179
        if prompt_token_ids is None:
            raise ValueError("Either prompts or prompt_token_ids must be provided.")             # llm.py:173-175
        n = len(prompt_token_ids)
        if sampling_params is None:
            sampling_params = SamplingParams()
        elif isinstance(sampling_params, list) and len(sampling_params) != n:
            raise ValueError("The lengths of prompts and sampling_params must be the same.")     # llm.py:199-202
        for i in range(n):
            sp = sampling_params[i] if isinstance(sampling_params, list) else sampling_params
            self.llm_engine.add_request(str(self._counter), None, sp, prompt_token_ids[i])
            self._counter += 1
        return self._run_engine(use_tqdm)

    def _run_engine(self, use_tqdm) -> List[RequestOutput]:
        outputs: Dict[str, RequestOutput] = {}
        total = self.llm_engine.get_num_unfinished_requests()
        t0, toks = time.time(), 0
        while self.llm_engine.has_unfinished_requests():
            for out in self.llm_engine.step():
                outputs[out.request_id] = out
                toks += len(out.outputs[0].token_ids)
            if use_tqdm:
                print("Processed prompts: %d/%d, Generation Speed: %.2f toks/s" % (len(outputs), total, toks / max(time.time() - t0, 1e-9)),
                      flush=True)
        return sorted(outputs.values(), key=lambda x: int(x.request_id))                          # llm.py:263-266
