#!/usr/bin/env python3
"""Where and when the workgroups of stats_mfma_kernel ran (measurement tool, MI355X only).

Builds svt-av1-psy_amd/csrc/lr_stats.hip with -DSVT_HIP_STATS_CENSUS into gpurun_out/census/libsvtav1_hip_census.so (every workgroup leaves its start / end on the
100 MHz clock, HW_ID and XCC_ID in a lower-triangle row of its unit's H; the finalize launch is skipped), runs the bench leg's 3840x2160 10-bit plane once and prints how many
workgroups were resident per CU over time.  Answers: does the launch really hold two workgroups per CU?"""
import collections
import ctypes as C
import importlib.util
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "svt-av1-psy_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "census")
os.makedirs(OUT, exist_ok=True)
lib_path = os.path.join(OUT, "libsvtav1_hip_census.so")
objs = [os.path.join(CS, "_build", f) for f in sorted(os.listdir(os.path.join(CS, "_build"))) if f.endswith(".o") and f != "lr_stats.o"]
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off".split()
subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-DSVT_HIP_STATS_CENSUS", "-c", os.path.join(CS, "lr_stats.hip"), "-o", os.path.join(OUT, "lr_stats_census.o")], check=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, os.path.join(OUT, "lr_stats_census.o"), *objs], check=True)
spec = importlib.util.spec_from_file_location("svt_av1_psy_amd", os.path.join(ROOT, "svt-av1-psy_amd", "__init__.py"))
pkg = importlib.util.module_from_spec(spec)
sys.modules["svt_av1_psy_amd"] = pkg
spec.loader.exec_module(pkg)
lib = pkg.bind(C.CDLL(lib_path))

w, h, bd, us, pad = 3840, 2160, 10, 256, 4
g = np.random.default_rng(8)
dgd = g.integers(0, 1 << bd, (h + 2 * pad, w + 2 * pad), dtype=np.uint16)
src = np.clip(dgd.astype(np.int32) + g.integers(-9, 10, dgd.shape), 0, (1 << bd) - 1).astype(np.uint16)
rects = []
nvu, nhu = max((h + us // 2) // us, 1), max((w + us // 2) // us, 1)
for r in range(nvu):
    for c in range(nhu):
        rects.append((pad + c * us, pad + (w if c == nhu - 1 else (c + 1) * us), pad + r * us, pad + (h if r == nvu - 1 else (r + 1) * us)))
rr = np.array(rects, np.int32)
d_dgd, d_src, d_r = torch.from_numpy(dgd).cuda(), torch.from_numpy(src).cuda(), torch.from_numpy(rr).cuda()
n = len(rects)
d_M = torch.zeros(n * 49, dtype=torch.int64, device="cuda")
d_H = torch.zeros(n * 49 * 49, dtype=torch.int64, device="cuda")
mw, mh = int((rr[:, 1] - rr[:, 0]).max()), int((rr[:, 3] - rr[:, 2]).max())
stride = w + 2 * pad
for _ in range(3):
    d_H.zero_()
    lib.svt_hip_lr_compute_stats_batch(d_dgd.data_ptr(), d_src.data_ptr(), d_r.data_ptr(), n, mw, mh, stride, stride, 7, bd, d_M.data_ptr(), d_H.data_ptr(), None)
    torch.cuda.synchronize()
H = d_H.cpu().numpy().reshape(n, 49, 49)
wgs, phases = [], []
for u in range(n):
    for y in range(30):
        t0, t1, hw, xcc = (int(v) for v in H[u, 10 + y, :4])
        if t1 > t0 > 0:
            wgs.append((t0, t1, (xcc & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf), u, y))
            phases.append([int(v) for v in H[u, 10 + y, 4:10]] + [t1 - t0])
ph = np.array(phases, dtype=np.float64)
tot = ph[:, :6].sum(axis=1)
print("thread 0's clock64() ticks per workgroup, mean (share): " + ", ".join("%s %.0f (%.0f %%)" % (nm, ph[:, k].mean(), 100 * ph[:, k].mean() / tot.mean())
      for k, nm in enumerate(("prologue", "barrier+digits", "copies", "prefetch issue", "K loop", "merge"))))
print("ticks per workgroup %.0f over %.2f us of the 100 MHz clock -> %.0f MHz" % (tot.mean(), ph[:, 6].mean() / 100.0, tot.mean() / (ph[:, 6].mean() / 100.0)))
base = min(x[0] for x in wgs)
print("%d workgroups; first start -> last end %.1f us; mean life %.1f us (min %.1f, max %.1f)" % (
    len(wgs), (max(x[1] for x in wgs) - base) / 100.0, np.mean([x[1] - x[0] for x in wgs]) / 100.0, min(x[1] - x[0] for x in wgs) / 100.0, max(x[1] - x[0] for x in wgs) / 100.0))
per_cu = collections.defaultdict(list)
for t0, t1, cu, u, y in wgs:
    per_cu[cu].append((t0 - base, t1 - base))
print("%d distinct (xcc, se, sh, cu); workgroups per CU: %s" % (len(per_cu), dict(collections.Counter(len(v) for v in per_cu.values()))))
peak = collections.Counter()
for cu, v in per_cu.items():
    ev = sorted([(a, 1) for a, _ in v] + [(b, -1) for _, b in v])
    cur = mx = 0
    for _, d in ev:
        cur += d
        mx = max(mx, cur)
    peak[mx] += 1
print("peak resident workgroups per CU -> number of CUs:", dict(peak))
starts = sorted(x[0] - base for x in wgs)
print("start times (us), deciles:", [round(starts[int(k * (len(starts) - 1) / 10)] / 100.0, 1) for k in range(11)])
