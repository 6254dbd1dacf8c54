#!/usr/bin/env python3
"""Runs bench.py in-process while a helper thread rewrites a copy of /proc/self/maps every second (argument 1 = the file), so that a
crash inside a profiler or runtime thread can be mapped to libraries and buffers afterwards:
    rocprofv3 --kernel-trace --stats -- python3 tools/run_with_maps.py gpurun_out/maps.txt --steps 1 --warmup 0 --batch 4 --no-vae ..."""
import os
import runpy
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]


def dump():
    while True:
        try:
            data = open("/proc/self/maps").read()
            tmp = out + ".tmp"
            with open(tmp, "w") as f:
                f.write(data)
            os.replace(tmp, out)
        except Exception:
            pass
        time.sleep(1.0)


threading.Thread(target=dump, daemon=True).start()
runpy.run_path(sys.argv[0], run_name="__main__")
