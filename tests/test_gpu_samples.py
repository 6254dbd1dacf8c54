"""End-to-end sample entry points (counterparts of the reference's sample_*.py) on the GPU, random-init weights."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ns(**kw):
    return argparse.Namespace(**kw)


def test_sample_c2i(tmp_path):
    import video_llamagen_amd  # noqa: F401
    from video_llamagen_amd import sample_c2i
    out = str(tmp_path / "c2i")
    sample_c2i.main(_ns(gpt_model="GPT-B", gpt_ckpt=None, gpt_type="c2i", cls_token_num=1, precision="bf16", vq_model="VQ-16", vq_ckpt=None,
                        codebook_size=16384, codebook_embed_dim=8, image_size=256, downsample_size=16, num_classes=1000, cfg_scale=4.0,
                        cfg_interval=-1, seed=0, top_k=2000, temperature=1.0, top_p=1.0, num_samples=2, out=out, serve=False))
    img = np.load(out + ".npy")
    assert img.shape == (2, 256, 256, 3) and img.dtype == np.uint8 and img.std() > 0
    # the same run through the request front-end (serve/sample_c2i.py) gives the same images
    sample_c2i.main(_ns(gpt_model="GPT-B", gpt_ckpt=None, gpt_type="c2i", cls_token_num=1, precision="bf16", vq_model="VQ-16", vq_ckpt=None,
                        codebook_size=16384, codebook_embed_dim=8, image_size=256, downsample_size=16, num_classes=1000, cfg_scale=4.0,
                        cfg_interval=-1, seed=0, top_k=2000, temperature=1.0, top_p=1.0, num_samples=2, out=out + "_serve", serve=True))
    assert np.array_equal(np.load(out + "_serve.npy"), img)


def test_sample_t2i_and_t2v(tmp_path):
    import video_llamagen_amd  # noqa: F401
    from video_llamagen_amd import sample_t2i, sample_t2v
    out = str(tmp_path / "t2i")
    sample_t2i.main(_ns(gpt_model="GPT-B", gpt_ckpt=None, gpt_type="t2i", cls_token_num=120, precision="bf16", vq_model="VQ-16", vq_ckpt=None,
                        codebook_size=16384, codebook_embed_dim=8, image_size=256, downsample_size=16, cfg_scale=7.5, seed=0, top_k=1000,
                        temperature=1.0, top_p=1.0, num_samples=2, out=out, t5_layers=0))
    assert np.load(out + ".npy").shape == (2, 256, 256, 3)
    # with the text encoder in the loop (one flan-t5-xl-shaped layer, random init): T5 -> left padding -> generate -> VQ decode
    sample_t2i.main(_ns(gpt_model="GPT-B", gpt_ckpt=None, gpt_type="t2i", cls_token_num=120, precision="bf16", vq_model="VQ-16", vq_ckpt=None,
                        codebook_size=16384, codebook_embed_dim=8, image_size=256, downsample_size=16, cfg_scale=7.5, seed=0, top_k=1000,
                        temperature=1.0, top_p=1.0, num_samples=2, out=out + "_t5", t5_layers=1))
    img = np.load(out + "_t5.npy")
    assert img.shape == (2, 256, 256, 3) and img.std() > 0
    out = str(tmp_path / "t2v")
    sample_t2v.main(_ns(gpt_model="GPT-B", gpt_ckpt=None, gpt_type="t2v", cls_token_num=120, precision="bf16", vae_model="VAE-16", vae_ckpt=None,
                        vae_embed_dim=8, tile_overlap_factor=0.125, image_size=64, downsample_size=8, num_frames=5, t_downsample_size=4,
                        cfg_scale=1.0, seed=0, num_samples=2, out=out))
    vid = np.load(out + ".npy")
    # the reference script's own model: hidden head + DiffLoss sampler (10 reverse steps here)
    sample_t2v.main(_ns(gpt_model="GPT-B", gpt_ckpt=None, gpt_type="t2v", cls_token_num=120, precision="bf16", vae_model="VAE-16", vae_ckpt=None,
                        vae_embed_dim=8, tile_overlap_factor=0.125, image_size=64, downsample_size=8, num_frames=5, t_downsample_size=4,
                        cfg_scale=1.0, seed=0, num_samples=2, out=out + "_diff", head="hidden", num_sampling_steps=10, temperature=0.9))
    assert np.load(out + "_diff.npy").shape == vid.shape
    vid = np.load(out + ".npy")
    assert vid.shape == (2, 5, 64, 64, 3) and vid.dtype == np.uint8      # 2 latent frames -> 2T-1 = 3 -> 5 frames


def test_serve_llm_matches_generate():
    """serve/sample_c2i.py calling convention: 2B prompts under guidance, outputs in request order, same ids as generate()."""
    import types
    import video_llamagen_amd as V
    args = types.SimpleNamespace(gpt_model="GPT-B", gpt_ckpt=None, gpt_type="c2i", cfg_scale=4.0, precision="bf16", image_size=64,
                                 downsample_size=16, num_classes=1000, cls_token_num=1)
    llm = V.LLM(args=args, model="autoregressive/serve/fake_json/GPT-B.json", seed=5, max_num_seqs=6)
    labels = [207, 360, 387, 974, 88]
    sp = V.SamplingParams(temperature=1.0, top_p=1.0, top_k=2000, max_tokens=16, seed=11)
    outs = llm.generate(prompt_token_ids=[[c] for c in labels] + [[1000]] * len(labels), sampling_params=sp, use_tqdm=False)
    assert [int(o.request_id) for o in outs] == list(range(10)) and all(o.finished for o in outs)
    ids = torch.tensor([o.outputs[0].token_ids for o in outs])
    assert ids.shape == (10, 16) and torch.equal(ids[:5], ids[5:])                     # sampler.py:106-108
    m = llm.llm_engine.model
    c = torch.tensor(labels, device="cuda")
    # max_num_seqs 6 -> waves of 3 pairs then 2 pairs; an explicit seed is used by every wave
    ref = torch.cat([V.generate(m, c[:3], 16, cfg_scale=4.0, temperature=1.0, top_k=2000, top_p=1.0, seed=11),
                     V.generate(m, c[3:], 16, cfg_scale=4.0, temperature=1.0, top_k=2000, top_p=1.0, seed=11)]).cpu()
    assert torch.equal(ids[:5], ref.long())
    # no guidance, greedy (temperature 0), top_k -1
    args.cfg_scale = 1.0
    llm2 = V.LLM(args=args, model="GPT-B", seed=5)
    outs = llm2.generate(prompt_token_ids=[[c] for c in labels], sampling_params=V.SamplingParams(temperature=0.0, max_tokens=8), use_tqdm=False)
    ref = V.generate(llm2.llm_engine.model, c, 8, sample_logits=False).cpu()
    assert torch.equal(torch.tensor([o.outputs[0].token_ids for o in outs]), ref.long())
    with pytest.raises(ValueError):
        llm2.generate(prompts=["a cat"], sampling_params=sp)


def test_continuous_engine_matches_generate():
    """Iteration-level batching (vlg_gpt_session_*): requests of different lengths share KV slots, a waiting request starts in the
    slot of a finished one while the others are mid-flight, every row at its own position.  Greedy fp32 ids of each request equal a
    stand-alone generate() of that request - with and without classifier-free guidance."""
    import video_llamagen_amd as V
    from oracle import cases
    from vlg_testutil import product_gpt
    m, _ = product_gpt(cases.TINY_C2I, torch.float32)
    greedy = lambda n: V.SamplingParams(temperature=0.0, max_tokens=n)
    reqs = [(3, 6), (9, 3), (0, 5), (7, 4), (5, 6)]                     # (class id, max_tokens); 3 slots -> 2 requests join later
    eng = V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=3)
    for i, (c, n) in enumerate(reqs):
        eng.add_request(str(i), None, greedy(n), [c])
    outs, steps = {}, 0
    while eng.has_unfinished_requests():
        for o in eng.step():
            outs[int(o.request_id)] = o.outputs[0].token_ids
        steps += 1
    # slots: [6 tokens] | [3, then 4 (starts at step 4)] | [5, then 6 (starts at step 6, ends at step 11)]; waves of 3 would need 6 + 6 = 12
    assert steps == 11 and sorted(outs) == list(range(5))
    for i, (c, n) in enumerate(reqs):
        ref = V.generate(m, torch.tensor([c]), n, sample_logits=False).cpu().tolist()[0]
        assert outs[i] == ref, (i, outs[i], ref)
    # guidance: 2 slots of (cond, uncond) pairs; the null-class requests report their partner's tokens
    eng = V.ContinuousLLMEngine(m, cfg_scale=2.0, max_num_seqs=4)
    labels = [4, 9, 2]
    null = cases.TINY_C2I["num_classes"]
    for i, c in enumerate(labels + [null] * 3):
        eng.add_request(str(i), None, greedy(5), [c])
    outs = {}
    while eng.has_unfinished_requests():
        for o in eng.step():
            outs[int(o.request_id)] = o.outputs[0].token_ids
    for i, c in enumerate(labels):
        ref = V.generate(m, torch.tensor([c]), 5, cfg_scale=2.0, sample_logits=False).cpu().tolist()[0]
        assert outs[i] == ref and outs[i + 3] == ref
    # sampling path through LLM(continuous=True): in range, reproducible under a fixed seed
    import types
    args = types.SimpleNamespace(gpt_model="GPT-B", gpt_ckpt=None, gpt_type="c2i", cfg_scale=1.5, precision="bf16", image_size=64,
                                 downsample_size=16, num_classes=1000, cls_token_num=1)
    runs = []
    for _ in range(2):
        llm = V.LLM(args=args, model="GPT-B", seed=3, max_num_seqs=4, continuous=True)
        sp = V.SamplingParams(temperature=1.0, top_k=100, max_tokens=16, seed=9)
        o = llm.generate(prompt_token_ids=[[207], [360], [387], [1000], [1000], [1000]], sampling_params=sp, use_tqdm=False)
        runs.append([x.outputs[0].token_ids for x in o])
    assert runs[0] == runs[1] and all(len(t) == 16 and min(t) >= 0 and max(t) < 16384 for t in runs[0]) and runs[0][0] == runs[0][3]


def test_session_engines_vs_reference_golden(golden):
    """The request front-ends against ids produced by the REFERENCE's generate() (tests/golden/gpt.npz, TINY_C2I, fp32 greedy): the
    wave engine, the iteration-level engine with all requests started together, and the iteration-level engine with fewer slots
    than requests (the third request starts in a freed slot, mid-flight of nothing else - its KV slot is reused).  With and without
    classifier-free guidance (scale 2.5, cfg_interval 6 as in the golden run)."""
    import video_llamagen_amd as V
    from oracle import cases
    from vlg_testutil import product_gpt
    g = golden("gpt")
    cfg = cases.TINY_C2I
    m, _ = product_gpt(cfg, torch.float32)
    labels = [int(c) for c in cases.class_ids(3, cfg["num_classes"])]
    N = cfg["block_size"]
    sp = V.SamplingParams(temperature=0.0, max_tokens=N)
    null = cfg["num_classes"]

    def run(engine, with_null):
        for i, c in enumerate(labels + ([null] * 3 if with_null else [])):
            engine.add_request(str(i), None, sp, [c])
        outs = {}
        while engine.has_unfinished_requests():
            for o in engine.step():
                outs[int(o.request_id)] = o.outputs[0].token_ids
        return np.array([outs[i] for i in range(3)])

    for slots in (3, 2):
        assert (run(V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=slots), False) == g["c2i_fp32_greedy_ids"]).all(), slots
        assert (run(V.ContinuousLLMEngine(m, cfg_scale=2.5, cfg_interval=6, max_num_seqs=2 * slots), True) == g["c2i_fp32_cfg_ids"]).all(), slots
    assert (run(V.LLMEngine(m, cfg_scale=1.0, max_num_seqs=3), False) == g["c2i_fp32_greedy_ids"]).all()
    assert (run(V.LLMEngine(m, cfg_scale=2.5, cfg_interval=6, max_num_seqs=6), True) == g["c2i_fp32_cfg_ids"]).all()


def test_t2i_sessions_vs_reference_golden(golden):
    """Text-conditioned requests through the iteration-level engine (vlg_gpt_session_prefill + session_step): each request's 120-token
    condition is prefilled into its slot, then it joins the running batch.  Ids equal the REFERENCE's generate() on the same captions
    (tests/golden/gpt.npz, TINY_T2I fp32 greedy; ragged left-padded masks 120 / 3 / 57 valid tokens), without guidance and with
    guidance 2.5 / cfg_interval 6, with as many slots as requests and with fewer (the third request starts in a reused slot while
    the first two are mid-flight)."""
    import video_llamagen_amd as V
    from oracle import cases
    from vlg_testutil import product_gpt
    g = golden("gpt")
    cfg = cases.TINY_T2I
    m, _ = product_gpt(cfg, torch.float32)
    c, mk = cases.text_cond(3, cfg["cls_token_num"], cfg["caption_dim"], lens=[120, 3, 57])
    N = cfg["block_size"]
    sp = V.SamplingParams(temperature=0.0, max_tokens=N)

    def run(engine):
        for i in range(3):
            engine.add_request(str(i), None, sp, prompt_embeds=torch.from_numpy(c[i]), emb_mask=torch.from_numpy(mk[i]))
        outs = {}
        while engine.has_unfinished_requests():
            for o in engine.step():
                outs[int(o.request_id)] = o.outputs[0].token_ids
        return np.array([outs[i] for i in range(3)])

    for slots in (3, 2):
        assert (run(V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=slots)) == g["t2i_fp32_greedy_ids"]).all(), slots
        assert (run(V.ContinuousLLMEngine(m, cfg_scale=2.5, cfg_interval=6, max_num_seqs=slots)) == g["t2i_fp32_cfg_ids"]).all(), slots
    eng = V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=2)
    with pytest.raises(ValueError):
        eng.add_request("x", None, sp, prompt_token_ids=[3])           # a text-conditioned engine takes features, not class ids


def test_kv_block_growth_and_preemption_vs_reference_golden(golden):
    """kv_policy "grow" (vLLM's scheduler policy behind autoregressive/serve/: blocks appended as a sequence grows, the youngest running
    sequence preempted by recomputation when the pool runs dry): requests are admitted with one block, grow, get preempted and start
    over - and every request's greedy ids still equal the REFERENCE's generate().  17 positions = 3 blocks of 8 per row."""
    import video_llamagen_amd as V
    from oracle import cases
    from vlg_testutil import product_gpt
    g = golden("gpt")
    cfg = cases.TINY_C2I
    m, _ = product_gpt(cfg, torch.float32)
    labels = [int(c) for c in cases.class_ids(3, cfg["num_classes"])]
    N, null = cfg["block_size"], cfg["num_classes"]

    def run(engine, with_null):
        for i, c in enumerate(labels):
            engine.add_request(str(i), None, V.SamplingParams(temperature=0.0, max_tokens=N), [c])
        for i in range(3 if with_null else 0):
            engine.add_request(str(3 + i), None, V.SamplingParams(temperature=0.0, max_tokens=N), [null])
        outs, steps = {}, 0
        while engine.has_unfinished_requests():
            for o in engine.step():
                outs[int(o.request_id)] = o.outputs[0].token_ids
            steps += 1
            assert steps < 400
        return [np.array(outs[i]) for i in range(3 + (3 if with_null else 0))]

    # scratch + 5 blocks, three slots: all three start on one block each (the "reserve" policy would admit ONE: 3 blocks each), two more
    # blocks are handed out as they grow, then the pool is dry: the youngest goes back to the queue, later the next one
    e = V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=3, max_tokens=N, kv_block_size=8, num_kv_blocks=6, kv_policy="grow")
    out = run(e, False)
    assert e.preempted >= 1 and all((out[i] == g["c2i_fp32_greedy_ids"][i]).all() for i in range(3))
    # under guidance every slot owns two rows (conditional + null-class partner): 6 blocks per request, pool of 10 + scratch
    e = V.ContinuousLLMEngine(m, cfg_scale=2.5, cfg_interval=6, max_num_seqs=6, max_tokens=N, kv_block_size=8, num_kv_blocks=11, kv_policy="grow")
    out = run(e, True)
    assert e.preempted >= 1
    for i in range(3):
        assert (out[i] == g["c2i_fp32_cfg_ids"][i]).all() and (out[3 + i] == out[i]).all()      # null-class members repeat their partner's tokens
    # a pool that holds everything: the policy never preempts and needs no more iterations than "reserve"
    e = V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=3, max_tokens=N, kv_block_size=8, num_kv_blocks=10, kv_policy="grow")
    out = run(e, False)
    assert e.preempted == 0 and e.steps_run == N and all((out[i] == g["c2i_fp32_greedy_ids"][i]).all() for i in range(3))
    # one request alone must fit: 2 blocks + scratch cannot hold 3 blocks
    e = V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=2, max_tokens=N, kv_block_size=8, num_kv_blocks=3, kv_policy="grow")
    e.add_request("0", None, V.SamplingParams(temperature=0.0, max_tokens=N), [labels[0]])
    with pytest.raises(ValueError):
        while e.has_unfinished_requests():
            e.step()
    e.close()
    with pytest.raises(ValueError):
        V.ContinuousLLMEngine(m, kv_block_size=8, kv_policy="swap")

    # text-conditioned requests (condition prefilled per slot): 120 condition positions + 16 tokens = 9 blocks of 16 per row.  Two slots,
    # 8 + 8 blocks at admission (120 + 1 positions), the 9th block of the older request is the last free one: the younger is preempted
    # when it needs its own, re-prefilled and run again later
    cfg = cases.TINY_T2I
    m, _ = product_gpt(cfg, torch.float32)
    c, mk = cases.text_cond(3, cfg["cls_token_num"], cfg["caption_dim"], lens=[120, 3, 57])
    N = cfg["block_size"]
    e = V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=2, kv_block_size=16, num_kv_blocks=1 + 17, kv_policy="grow")
    for i in range(3):
        e.add_request(str(i), None, V.SamplingParams(temperature=0.0, max_tokens=N), prompt_embeds=torch.from_numpy(c[i]), emb_mask=torch.from_numpy(mk[i]))
    outs, steps = {}, 0
    while e.has_unfinished_requests():
        for o in e.step():
            outs[int(o.request_id)] = o.outputs[0].token_ids
        steps += 1
        assert steps < 400
    assert e.preempted >= 1 and (np.array([outs[i] for i in range(3)]) == g["t2i_fp32_greedy_ids"]).all()


def test_block_granular_kv_sessions_vs_reference_golden(golden):
    """The iteration-level engine on a block-granular KV cache (vlg_gpt_session_reserve / release, paged attention + paged KV append):
    ids equal the REFERENCE's generate().  Pools are sized so that (a) blocks are handed out in scrambled order and reused by later
    requests, (b) an admission has to wait for a release (deferred > 0), (c) three requests of different lengths run side by side in
    a pool that could not hold three full-length slots - shorter greedy requests are prefixes of the golden ids."""
    import video_llamagen_amd as V
    from oracle import cases
    from vlg_testutil import product_gpt
    g = golden("gpt")
    cfg = cases.TINY_C2I
    m, _ = product_gpt(cfg, torch.float32)
    labels = [int(c) for c in cases.class_ids(3, cfg["num_classes"])]
    N = cfg["block_size"]
    null = cfg["num_classes"]

    def run(engine, with_null, lens=(N, N, N)):
        for i, c in enumerate(labels):
            engine.add_request(str(i), None, V.SamplingParams(temperature=0.0, max_tokens=lens[i]), [c])
        for i in range(3 if with_null else 0):
            engine.add_request(str(3 + i), None, V.SamplingParams(temperature=0.0, max_tokens=lens[i]), [null])
        outs = {}
        while engine.has_unfinished_requests():
            for o in engine.step():
                outs[int(o.request_id)] = o.outputs[0].token_ids
        return [np.array(outs[i]) for i in range(3)]

    # 17 positions per request = 3 blocks of 8 per row.  7 blocks = scratch + two requests: the third waits for a release and
    # then runs on blocks the first two returned
    e = V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=3, max_tokens=N, kv_block_size=8, num_kv_blocks=7)
    out = run(e, False)
    assert all((out[i] == g["c2i_fp32_greedy_ids"][i]).all() for i in range(3)) and e.deferred > 0
    e = V.ContinuousLLMEngine(m, cfg_scale=2.5, cfg_interval=6, max_num_seqs=6, max_tokens=N, kv_block_size=8, num_kv_blocks=13)
    out = run(e, True)
    assert all((out[i] == g["c2i_fp32_cfg_ids"][i]).all() for i in range(3)) and e.deferred > 0
    # default pool (every slot at full length), block 16
    e = V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=2, max_tokens=N, kv_block_size=16)
    out = run(e, False)
    assert all((out[i] == g["c2i_fp32_greedy_ids"][i]).all() for i in range(3))
    # lengths 16 / 7 / 3 tokens need 3 + 1 + 1 blocks: they run together in 6 blocks, where three full-length slots would need 10
    e = V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=3, max_tokens=N, kv_block_size=8, num_kv_blocks=6)
    lens = (N, 7, 3)
    out = run(e, False, lens)
    assert e.deferred == 0
    for i in range(3):
        assert len(out[i]) == lens[i] and (out[i] == g["c2i_fp32_greedy_ids"][i][:lens[i]]).all(), i
    # a deferral at slot 0 while slot 1 is mid-request: two slots, 1 + 4 blocks; the 3-token request (1 block) finishes first, the
    # queued full-length one (3 blocks) then has to wait for slot 1's request - which must keep running undisturbed meanwhile
    e = V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=2, max_tokens=N, kv_block_size=8, num_kv_blocks=5)
    lens = (3, N, N)
    out = run(e, False, lens)
    assert e.deferred > 0
    for i in range(3):
        assert len(out[i]) == lens[i] and (out[i] == g["c2i_fp32_greedy_ids"][i][:lens[i]]).all(), i

    # text-conditioned: 120 condition positions + 16 tokens = 9 blocks of 16 per row (18 per request under guidance)
    cfg = cases.TINY_T2I
    m, _ = product_gpt(cfg, torch.float32)
    c, mk = cases.text_cond(3, cfg["cls_token_num"], cfg["caption_dim"], lens=[120, 3, 57])
    N = cfg["block_size"]
    sp = V.SamplingParams(temperature=0.0, max_tokens=N)

    def run_t(engine):
        for i in range(3):
            engine.add_request(str(i), None, sp, prompt_embeds=torch.from_numpy(c[i]), emb_mask=torch.from_numpy(mk[i]))
        outs = {}
        while engine.has_unfinished_requests():
            for o in engine.step():
                outs[int(o.request_id)] = o.outputs[0].token_ids
        return np.array([outs[i] for i in range(3)])

    e = V.ContinuousLLMEngine(m, cfg_scale=1.0, max_num_seqs=3, kv_block_size=16, num_kv_blocks=1 + 2 * 9)
    assert (run_t(e) == g["t2i_fp32_greedy_ids"]).all() and e.deferred > 0
    e = V.ContinuousLLMEngine(m, cfg_scale=2.5, cfg_interval=6, max_num_seqs=3, kv_block_size=16, num_kv_blocks=1 + 2 * 18)
    assert (run_t(e) == g["t2i_fp32_cfg_ids"]).all() and e.deferred > 0
    e = V.ContinuousLLMEngine(m, cfg_scale=2.5, cfg_interval=6, max_num_seqs=2, kv_block_size=32)
    assert (run_t(e) == g["t2i_fp32_cfg_ids"]).all()


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_t2v_sessions_vs_reference_golden(golden, dt):
    """Round 4: the iteration-level engine for the continuous-latent VIDEO models (vlg_gpt_session_* with model_type t2v, adapter2 head;
    the reference's serving path stops at class-conditional images).  Every request's latents equal the REFERENCE's generate_t2v golden
    (t2v.npz), whatever slot and iteration it starts in: three requests on 3 / 2 / 1 slots, a shorter request (a prefix of the golden), and
    the block-granular KV cache with a pool that makes an admission wait."""
    import video_llamagen_amd as V
    from oracle import cases
    from vlg_testutil import product_gpt, to_np
    g = golden("t2v")
    cfg = cases.TINY_T2V
    m, _ = product_gpt(cfg, torch.float32 if dt == "fp32" else torch.bfloat16)
    c, mk = cases.text_cond(2, cfg["cls_token_num"], cfg["caption_dim"], lens=[8, 4])
    N = 3 * cfg["block_size"]
    ref = g[f"t2v_{dt}_latents"]
    tol = (3e-4 if dt == "fp32" else 8e-2) * max(1.0, np.abs(ref).max())
    which = [0, 1, 0]

    def run(engine, lens=(N, N, N)):
        for i, w in enumerate(which):
            engine.add_request(str(i), None, V.SamplingParams(temperature=1.0, max_tokens=lens[i]), prompt_embeds=torch.from_numpy(c[w]),
                               emb_mask=torch.from_numpy(mk[w]))
        outs, steps = {}, 0
        while engine.has_unfinished_requests():
            for o in engine.step():
                assert o.outputs[0].token_ids == [] and o.outputs[0].latents is not None
                outs[int(o.request_id)] = to_np(o.outputs[0].latents)
            steps += 1
            assert steps < 1000
        return outs

    for slots in (3, 2, 1):
        outs = run(V.ContinuousLLMEngine(m, max_num_seqs=slots))
        for i, w in enumerate(which):
            assert outs[i].shape == (N, cfg["vae_embed_dim"])
            assert np.abs(outs[i] - ref[w]).max() < tol, (slots, i, np.abs(outs[i] - ref[w]).max())
    lens = (N, 20, 33)                                      # requests of different lengths side by side: prefixes of the golden
    outs = run(V.ContinuousLLMEngine(m, max_num_seqs=2), lens)
    for i, w in enumerate(which):
        assert outs[i].shape[0] == lens[i] and np.abs(outs[i] - ref[w][:lens[i]]).max() < tol, i
    # block-granular KV: 8 + 48 = 56 positions = 4 blocks of 16 per request; a pool of 1 + 8 blocks holds two requests, the third waits
    e = V.ContinuousLLMEngine(m, max_num_seqs=3, kv_block_size=16, num_kv_blocks=1 + 8)
    outs = run(e)
    assert e.deferred > 0
    for i, w in enumerate(which):
        assert np.abs(outs[i] - ref[w]).max() < tol, i
    with pytest.raises(ValueError):
        V.ContinuousLLMEngine(m, cfg_scale=2.0, max_num_seqs=2)   # no transformer guidance for the continuous-latent sessions


def test_t2v_diffloss_sessions_match_generate():
    """Sessions of the DiffLoss (hidden) head: the persistent sampler with every slot at ITS token index (DlPersist::row_step).  The noise
    stream is keyed by (seed, slot, token index), the key of generate_t2v's (seed, batch row, token index): B requests started together on B
    slots reproduce generate_t2v(seed) row for row; with fewer slots than requests, the late request equals the row of the generate() call
    that puts it at its slot's row.  fp32: both sides run the same kernels, tolerance for the fused-chain instance chosen by the row count."""
    import video_llamagen_amd as V
    from oracle import cases, detweights
    from vlg_testutil import to_np
    cfg = dict(cases.TINY_T2V_DIFF, num_sampling_steps=10, diffloss_w=256)
    keys = ("dim", "n_layer", "n_head", "vocab_size", "block_size", "cls_token_num", "model_type", "caption_dim", "vae_embed_dim",
            "num_frames", "t_downsample_size", "head", "diffloss_w", "diffloss_d", "num_sampling_steps")
    m = V.Transformer(V.ModelArgs(**{k: cfg[k] for k in keys})).to("cuda", torch.float32).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in detweights.gpt_weights(cfg).items()}, strict=False)
    c, mk = cases.text_cond(3, cfg["cls_token_num"], cfg["caption_dim"], lens=[8, 2, 5])
    N = 12
    sp = V.SamplingParams(temperature=0.9, max_tokens=N, seed=5)

    def run(engine, order):
        for i in order:
            engine.add_request(str(i), None, sp, prompt_embeds=torch.from_numpy(c[i]), emb_mask=torch.from_numpy(mk[i]))
        outs = {}
        while engine.has_unfinished_requests():
            for o in engine.step():
                outs[int(o.request_id)] = to_np(o.outputs[0].latents)
        return outs

    ref = to_np(V.generate_t2v(m, torch.from_numpy(c), N, torch.from_numpy(mk), temperature=0.9, seed=5))
    scale = max(1.0, np.abs(ref).max())
    outs = run(V.ContinuousLLMEngine(m, max_num_seqs=3), [0, 1, 2])
    for i in range(3):
        assert np.isfinite(outs[i]).all() and np.abs(outs[i] - ref[i]).max() < 2e-4 * scale, (i, np.abs(outs[i] - ref[i]).max())
    # two slots: requests 0 and 1 run first, request 2 then starts in slot 0 = row 0 of a generate() call over [2, 1]
    outs = run(V.ContinuousLLMEngine(m, max_num_seqs=2), [0, 1, 2])
    ref21 = to_np(V.generate_t2v(m, torch.from_numpy(c[[2, 1]]), N, torch.from_numpy(mk[[2, 1]]), temperature=0.9, seed=5))
    assert np.abs(outs[0] - ref[0]).max() < 2e-4 * scale and np.abs(outs[1] - ref[1]).max() < 2e-4 * scale
    assert np.abs(outs[2] - ref21[0]).max() < 2e-4 * scale
    m.dl_persist = False                                     # without the persistent sampler the session is refused, loudly
    e = V.ContinuousLLMEngine(m, max_num_seqs=2)
    e.add_request("0", None, sp, prompt_embeds=torch.from_numpy(c[0]), emb_mask=torch.from_numpy(mk[0]))
    with pytest.raises(RuntimeError):
        e.step()
    m.dl_persist = True


@pytest.mark.parametrize("slots", [8, 34])
def test_t2v_diffloss_sessions_full_width_bf16(slots):
    """The benchmark's DiffLoss head (W 1024, depth 3, 100 reverse steps, bf16) inside a session: 8 slots = groups of four rows, 34 slots = the
    eight-row groups (dl_persist_kernel<bf16, 2, true, 3, 8> with per-row token indices).  All slots start together, so every request must
    reproduce its row of generate_t2v(seed) - same kernels, same noise keys; bf16 tolerance on the first token, finite throughout."""
    import video_llamagen_amd as V
    from vlg_testutil import to_np
    m = V.Transformer(V.ModelArgs(dim=256, n_layer=2, n_head=4, block_size=64, cls_token_num=8, model_type="t2v", vae_embed_dim=8,
                                  num_frames=17, t_downsample_size=4, caption_dim=64, head="hidden", diffloss_w=1024, diffloss_d=3,
                                  num_sampling_steps=100)).to("cuda", torch.bfloat16)
    m.init_random_weights(seed=4)
    g = torch.Generator().manual_seed(2)
    cond = torch.randn(slots, 8, 64, generator=g) * 0.1
    mask = torch.ones(slots, 8)
    N = 3
    sp = V.SamplingParams(temperature=1.0, max_tokens=N, seed=11)
    eng = V.ContinuousLLMEngine(m, max_num_seqs=slots)
    for i in range(slots):
        eng.add_request(str(i), None, sp, prompt_embeds=cond[i], emb_mask=mask[i])
    outs = {}
    while eng.has_unfinished_requests():
        for o in eng.step():
            outs[int(o.request_id)] = to_np(o.outputs[0].latents)
    got = np.stack([outs[i] for i in range(slots)])
    ref = to_np(V.generate_t2v(m, cond, N, mask, temperature=1.0, seed=11))
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert np.abs(got[:, 0] - ref[:, 0]).max() < 8e-2 * max(1.0, np.abs(ref[:, 0]).max())


def test_block_granular_kv_is_the_same_arithmetic():
    """Paging only changes where a cache row lives: sampled (top-k, temperature 1, guidance) bf16 sessions produce the same ids on 16-position
    blocks handed out of a tight pool as on contiguous slots."""
    import video_llamagen_amd as V
    from oracle import cases
    from vlg_testutil import product_gpt
    cfg = cases.TINY_C2I
    m, _ = product_gpt(cfg, torch.bfloat16)
    N = cfg["block_size"]
    labels = [int(c) for c in cases.class_ids(5, cfg["num_classes"])]
    sp = V.SamplingParams(temperature=1.0, top_k=50, max_tokens=N, seed=11)

    def run(**kw):
        e = V.ContinuousLLMEngine(m, cfg_scale=2.0, max_num_seqs=4, max_tokens=N, **kw)
        for i, c in enumerate(labels + [cfg["num_classes"]] * 5):
            e.add_request(str(i), None, sp, [c])
        outs = {}
        while e.has_unfinished_requests():
            for o in e.step():
                outs[int(o.request_id)] = o.outputs[0].token_ids
        return [outs[i] for i in range(5)], e

    a, _ = run()
    b, eng = run(kv_block_size=16, num_kv_blocks=1 + 2 * 2 * 2)      # 2 slots' worth of blocks (2 rows x 2 blocks each) for 2 running pairs
    assert a == b


def test_block_granular_kv_error_paths():
    import ctypes as C
    import video_llamagen_amd as V
    from video_llamagen_amd import _lib as L
    from oracle import cases
    from vlg_testutil import product_gpt
    m, _ = product_gpt(cases.TINY_C2I, torch.float32)
    m._ensure_handle()
    lib, h = L.lib(), m._handle
    sp = L.SamplingParams(cfg_scale=1.0, cfg_interval=-1, temperature=1.0, top_k=0, top_p=1.0, sample_logits=0, seed=0)
    with pytest.raises(L.VlgError):
        L.check(lib.vlg_gpt_set_option(h, b"kv_block", C.c_int64(24)))          # not a power of two
    L.check(lib.vlg_gpt_set_option(h, b"kv_block", C.c_int64(8)))
    L.check(lib.vlg_gpt_set_option(h, b"kv_pool_blocks", C.c_int64(4)))         # scratch + 3
    L.check(lib.vlg_gpt_session_begin(h, 2, 16, C.byref(sp)))
    nfree, bs = C.c_int32(0), C.c_int32(0)
    L.check(lib.vlg_gpt_session_free_blocks(h, C.byref(nfree), C.byref(bs)))
    assert (nfree.value, bs.value) == (3, 8)
    rc = (C.c_int32 * 2)(5, -2)
    assert lib.vlg_gpt_session_step(h, rc) == L.VLG_ERR_STATE                   # start without a reservation
    L.check(lib.vlg_gpt_session_reserve(h, 0, 7))                               # 8 positions: one block
    assert lib.vlg_gpt_session_reserve(h, 1, 16) == L.VLG_ERR_OOM               # 3 blocks wanted, 2 free: nothing changes
    L.check(lib.vlg_gpt_session_free_blocks(h, C.byref(nfree), None))
    assert nfree.value == 2
    L.check(lib.vlg_gpt_session_step(h, rc))
    rc[0] = -1
    for _ in range(6):
        L.check(lib.vlg_gpt_session_step(h, rc))                                # positions 1..6 (tokens 1..6 sampled, 7 in all)
    L.check(lib.vlg_gpt_session_step(h, rc))                                    # position 7: the last one of the block
    assert lib.vlg_gpt_session_step(h, rc) == L.VLG_ERR_STATE                   # position 8 is outside the reservation
    L.check(lib.vlg_gpt_session_reserve(h, 0, 16))                              # grow: 2 more blocks
    L.check(lib.vlg_gpt_session_free_blocks(h, C.byref(nfree), None))
    assert nfree.value == 0
    L.check(lib.vlg_gpt_session_release(h, 0))
    L.check(lib.vlg_gpt_session_free_blocks(h, C.byref(nfree), None))
    assert nfree.value == 3
    L.check(lib.vlg_gpt_session_end(h))
    L.check(lib.vlg_gpt_set_option(h, b"kv_block", C.c_int64(0)))
    L.check(lib.vlg_gpt_set_option(h, b"kv_pool_blocks", C.c_int64(0)))
    L.check(lib.vlg_gpt_session_begin(h, 2, 16, C.byref(sp)))                   # contiguous slots: reserve / release are no-ops
    L.check(lib.vlg_gpt_session_reserve(h, 0, 16))
    L.check(lib.vlg_gpt_session_free_blocks(h, C.byref(nfree), C.byref(bs)))
    assert (nfree.value, bs.value) == (-1, 0)
    L.check(lib.vlg_gpt_session_end(h))


def test_session_and_t5_error_paths():
    """Loud failures instead of silent fallbacks: sessions on a text-conditioned model, stepping without a session, slot overrun,
    T5 configurations / sequence lengths that are not built."""
    import ctypes as C
    import video_llamagen_amd as V
    from video_llamagen_amd import _lib as L
    from oracle import cases
    from vlg_testutil import product_gpt
    sp = L.SamplingParams(cfg_scale=1.0, cfg_interval=-1, temperature=1.0, top_k=0, top_p=1.0, sample_logits=0, seed=0)
    t2v, _ = product_gpt(cases.TINY_T2V, torch.float32)
    spg = L.SamplingParams(cfg_scale=2.0, cfg_interval=-1, temperature=1.0, top_k=0, top_p=1.0, sample_logits=0, seed=0)
    with pytest.raises(L.VlgError, match="without transformer guidance"):      # continuous-latent sessions: cfg_scale 1 only
        L.check(L.lib().vlg_gpt_session_begin(t2v._handle, 2, 4, C.byref(spg)))
    L.check(L.lib().vlg_gpt_session_begin(t2v._handle, 2, 4, C.byref(sp)))
    with pytest.raises(L.VlgError, match="read_latents"):                      # ... and their results are latents, not ids
        L.check(L.lib().vlg_gpt_session_read(t2v._handle, 0, 1, (C.c_int32 * 1)()))
    L.check(L.lib().vlg_gpt_session_end(t2v._handle))
    t2i, _ = product_gpt(cases.TINY_T2I, torch.float32)
    L.check(L.lib().vlg_gpt_session_begin(t2i._handle, 2, 4, C.byref(sp)))
    with pytest.raises(L.VlgError, match="samples token ids"):
        L.check(L.lib().vlg_gpt_session_read_latents(t2i._handle, 0, 1, (C.c_float * 8)()))
    with pytest.raises(L.VlgError, match="no prefilled condition"):
        L.check(L.lib().vlg_gpt_session_step(t2i._handle, (C.c_int32 * 2)(-3, -2)))
    with pytest.raises(L.VlgError, match="does not fit"):
        L.check(L.lib().vlg_gpt_session_step(t2i._handle, (C.c_int32 * 2)(5, -2)))
    L.check(L.lib().vlg_gpt_session_end(t2i._handle))
    c2i, _ = product_gpt(cases.TINY_C2I, torch.float32)
    rows = (C.c_int32 * 2)(1, -2)
    with pytest.raises(L.VlgError, match="no open session"):
        L.check(L.lib().vlg_gpt_session_step(c2i._handle, rows))
    with pytest.raises(L.VlgError, match="RoPE table"):
        L.check(L.lib().vlg_gpt_session_begin(c2i._handle, 2, 17, C.byref(sp)))
    L.check(L.lib().vlg_gpt_session_begin(c2i._handle, 2, 3, C.byref(sp)))
    with pytest.raises(L.VlgError, match="text-conditioned"):
        L.check(L.lib().vlg_gpt_session_prefill(c2i._handle, 0, L.ptr(torch.zeros(1, device="cuda")), None))
    L.check(L.lib().vlg_gpt_session_step(c2i._handle, rows))
    cont = (C.c_int32 * 2)(-1, -2)
    L.check(L.lib().vlg_gpt_session_step(c2i._handle, cont))
    L.check(L.lib().vlg_gpt_session_step(c2i._handle, cont))
    with pytest.raises(L.VlgError, match="max_new_tokens"):
        L.check(L.lib().vlg_gpt_session_step(c2i._handle, cont))           # a 4th token in a 3-token slot
    buf = (C.c_int32 * 3)()
    L.check(L.lib().vlg_gpt_session_read(c2i._handle, 0, 3, buf))
    assert list(buf) == V.generate(c2i, torch.tensor([1]), 3, sample_logits=False).cpu().tolist()[0]
    with pytest.raises(L.VlgError):
        L.check(L.lib().vlg_gpt_session_read(c2i._handle, 5, 1, buf))
    L.check(L.lib().vlg_gpt_session_end(c2i._handle))
    with pytest.raises(L.VlgError, match="gated-gelu"):
        V.T5EncoderModel(dict(cases.TINY_T5, feed_forward_proj="relu"))
    t5 = V.T5EncoderModel(cases.TINY_T5).to("cuda", torch.float32).init_random_weights(seed=1)
    with pytest.raises(L.VlgError):
        t5(input_ids=torch.zeros(1, 300, dtype=torch.long))
    with pytest.raises(RuntimeError, match="Unexpected key"):
        t5.load_state_dict({"decoder.block.0.layer.0.layer_norm.weight": torch.ones(64)})
