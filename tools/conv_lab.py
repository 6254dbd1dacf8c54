#!/usr/bin/env python3
"""Where does a conv_halo_kernel workgroup spend its time?  In-kernel phase stamps (lab build of libvlg: -DVLG_CONV_LAB).

  python tools/conv_lab.py build [-D...]   # here (hipcc cross-compiles): tools/microbench/bin/libvlg_convlab.so
                                          # timing ablations: CONV_LAB_SO=nomfma.so python tools/conv_lab.py build -DVLG_CONV_LAB_NOMFMA
  python tools/conv_lab.py            # GPU box: one 4-video CausalVideoVAE decode, stamps of every 61st workgroup, by layer shape
"""
import ctypes
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "tools", "microbench", "bin", os.environ.get("CONV_LAB_SO", "libvlg_convlab.so"))   # CONV_LAB_SO: an ablation build

if len(sys.argv) > 1 and sys.argv[1] == "build":
    pkg = os.path.join(ROOT, "video-llamagen_amd")
    obj = LAB[:-3] + ".o"
    extra = sys.argv[2:]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DVLG_CONV_LAB", *extra, "-c",
                           os.path.join(pkg, "csrc", "conv_kernels.hip"), "-o", obj])
    others = [os.path.join(pkg, "lib", "obj", f) for f in sorted(os.listdir(os.path.join(pkg, "lib", "obj"))) if f.endswith(".o") and f != "conv_kernels.o"]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LAB, obj, *others])
    print("built", LAB)
    sys.exit(0)

os.environ["VLG_LIB_PATH"] = LAB
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import video_llamagen_amd as V  # noqa: E402

lib = ctypes.CDLL(LAB)
vae = V.VAE_models["VAE-16"](embed_dim=8).to("cuda", torch.bfloat16).eval()
vae.init_random_weights(seed=3)
z = torch.randn(4, 8, 5, 32, 32, device="cuda")
vae.decode(z)
torch.cuda.synchronize()
buf = torch.zeros(32768 * 24, dtype=torch.int64, device="cuda")
GAPS = len(sys.argv) > 1 and sys.argv[1] == "gaps"   # every workgroup of the Cin 128 / 256 x 256 / kt 3 layers: idle time of a CU between workgroups
assert lib.vlg_conv_lab_mode(1 if GAPS else 0) == 0
assert lib.vlg_conv_lab_set(ctypes.c_void_p(buf.data_ptr())) == 0
vae.decode(z)
torch.cuda.synchronize()
n = ctypes.c_uint(0)
lib.vlg_conv_lab_count(ctypes.byref(n))
assert lib.vlg_conv_lab_set(ctypes.c_void_p(0)) == 0
rec = buf.cpu().numpy().reshape(-1, 24)[: min(n.value, 32768)]
print(f"{n.value} records (10 ns ticks -> us)")
if GAPS:
    cus = defaultdict(list)
    for r in rec:
        hw, xcc = int(r[11]) & 0xFFFFFFFF, int(r[11]) >> 32
        cus[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)].append((int(r[4]), int(r[8])))
    gaps, durs = [], []
    for v in cus.values():
        v.sort()
        for (s0, e0), (s1, e1) in zip(v, v[1:]):
            if s1 - e0 < 5000:   # the same launch
                gaps.append((s1 - e0) / 100.0)
            durs.append((e0 - s0) / 100.0)
    print(f"{len(cus)} CUs seen; workgroup (wave 0 start -> its last stamp): median {np.median(durs):.1f} us; "
          f"gap to the next workgroup's wave 0 on the same CU: median {np.median(gaps):.2f} us, p10 {np.percentile(gaps, 10):.2f}, p90 {np.percentile(gaps, 90):.2f}")
    sys.exit(0)
groups = defaultdict(list)
for r in rec:
    groups[(int(r[0]), int(r[1]), int(r[2]), int(r[3]))].append(r)
print("Cin   Wo kt steps |  WGs | roles  prologue  loop (per step; MFMA pipe busy)  epilogue | total | clock GHz | first 12 steps (us)")
for key in sorted(groups):
    a = np.array(groups[key], dtype=np.float64)
    us = lambda i, j: np.median(a[:, j] - a[:, i]) / 100.0
    steps = key[3]
    st = a[:, 12:24]
    prev = np.concatenate([a[:, 6:7], st[:, :-1]], axis=1)
    nst = min(12, steps)
    per = np.median(st[:, :nst] - prev[:, :nst], axis=0) / 100.0
    cyc = np.median(a[:, 10] - a[:, 9])                      # shader cycles of the loop
    ghz = cyc / (np.median(a[:, 7] - a[:, 6]) * 10.0)        # 10 ns ticks
    busy = steps * 2 * 24 * 32 / cyc                         # two waves per SIMD x 24 MFMAs x 32 cycles per step
    print(f"{key[0]:4d} {key[1]:4d} {key[2]:2d} {steps:5d} | {len(a):4d} | {us(4, 5):5.2f} {us(5, 6):8.2f} {us(6, 7):7.1f} ({us(6, 7) / steps:5.2f}; {busy:4.2f}) {us(7, 8):8.2f} | "
          f"{us(4, 8):6.1f} | {ghz:5.2f} | " + " ".join(f"{x:.2f}" for x in per))
