#!/usr/bin/env python3
"""One-GPU timings of the per-GPU shards of the 32-video job (strong scaling, SURVEY.md 8d: 32 videos over 1 / 2 / 4 / 8 GPUs = 32 / 16 / 8 / 4
per GPU), and the speed-ups they project.  Each shard = `bench.py --batch b --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-roofline`
(sampling + the shard's own VAE decode) in its own process.  Writes gpurun_out/r04_shard_timings.json (copied to profiles/ by hand:
bench.py's N > 1 line quotes it).    python tools/bench_shards.py [batches, default 32,16,8,4]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32,16,8,4").split(",")]
out = {"source": "tools/bench_shards.py on one MI355X: bench.py --batch b --steps 1 --warmup 1 (sampling of b videos x 5120 tokens + their VAE decode), one process per shard size",
       "s_per_step": {}, "tokens_per_s": {}}
for b in batches:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(b), "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline",
                        "--no-roofline"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    out["s_per_step"][str(b)] = j["ms_per_step"] / 1e3
    out["tokens_per_s"][str(b)] = j["value"]
    print("batch", b, "->", round(j["ms_per_step"] / 1e3, 3), "s per step", flush=True)
full = max(batches)
out["speedup_vs_one_gpu"] = {str(full // b): out["s_per_step"][str(full)] / out["s_per_step"][str(b)] for b in batches if b != full}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_shard_timings.json"), "w"), indent=1)
print(json.dumps(out["speedup_vs_one_gpu"]))
