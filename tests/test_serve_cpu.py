"""Host logic of the serving front-end (video_llamagen_amd.serve; reference: autoregressive/serve/llm.py, sampler.py:46-125)."""
import pytest

import video_llamagen_amd as V
from video_llamagen_amd.serve import Scheduler, _Request, SamplingParams


def req(i, tok, sp):
    return _Request(str(i), [tok], sp, 0.0)


def test_sampling_params_validation():
    SamplingParams(temperature=1.0, top_p=1.0, top_k=2000, max_tokens=576)
    for kw in ({"temperature": -1}, {"top_p": 0.0}, {"top_p": 1.5}, {"top_k": 0}, {"top_k": -2}, {"max_tokens": 0}):
        with pytest.raises(ValueError):
            SamplingParams(**kw)


def test_waves_without_guidance():
    s = Scheduler(max_num_seqs=3)
    a, b = SamplingParams(max_tokens=4), SamplingParams(max_tokens=8)
    for i, sp in enumerate([a, a, b, a, a]):
        s.add(req(i, 10 + i, sp))
    w, partners = s.next_wave()
    assert [r.request_id for r in w] == ["0", "1", "3"] and partners == [None] * 3     # same params as the head, FIFO, capped
    w, _ = s.next_wave()
    assert [r.request_id for r in w] == ["2"]                                           # now the head is the other parameter set
    w, _ = s.next_wave()
    assert [r.request_id for r in w] == ["4"]
    assert s.next_wave() == ([], []) and len(s) == 0


def test_guidance_pairs_in_arrival_order():
    """sample_c2i.py:35-37: B conditional prompts followed by B null-class prompts; the i-th of each form a pair."""
    s = Scheduler(max_num_seqs=4, cfg=True, null_token=1000)
    sp = SamplingParams(max_tokens=4)
    labels = [207, 360, 387]
    for i, c in enumerate(labels + [1000] * 3):
        s.add(req(i, c, sp))
    conds, nulls = s.next_wave()
    assert [r.request_id for r in conds] == ["0", "1"] and [r.request_id for r in nulls] == ["3", "4"]   # 4 sequences = 2 pairs
    conds, nulls = s.next_wave()
    assert [r.request_id for r in conds] == ["2"] and [r.request_id for r in nulls] == ["5"]
    assert len(s) == 0


def test_guidance_needs_partners():
    s = Scheduler(max_num_seqs=8, cfg=True, null_token=1000)
    s.add(req(0, 5, SamplingParams()))
    with pytest.raises(ValueError):
        s.next_wave()
    with pytest.raises(ValueError):
        Scheduler(max_num_seqs=1, cfg=True, null_token=1000)


def test_exports():
    assert V.LLM is V.serve.LLM and V.SamplingParams is SamplingParams
