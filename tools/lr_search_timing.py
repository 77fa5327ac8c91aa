#!/usr/bin/env python3
"""Wall time of svt_hip_lr_search_plane on bench_legs.lr_search's 4K 10-bit plane, per half of the stage (self-guided only / Wiener only / both) and per form of the
self-guided projection walk (SVT_HIP_LR_SG_WALK = default | table | line).  Small enough (a few dozen launches) to run under `rocprofv3 --kernel-trace --stats`.
    python tools/lr_search_timing.py [reps]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry._pkg()
lib = pkg.load(init_device=0)
stream = torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = np.random.default_rng(19)
W, H, PAD, bd = 3840, 2160, 8, 10
amp = (1 << bd) - 1
yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
tex = 0.5 + 0.2 * np.sin(xx / 2.3) * np.cos(yy / 3.1) + 0.15 * np.sin((xx + 2 * yy) / 6.7) + 0.1 * np.sign(np.sin(xx / 9.0) * np.sin(yy / 7.0))
src = np.clip(tex * amp + g.normal(0, amp / 120, tex.shape), 0, amp).astype(np.uint16)
blur = (src.astype(np.float32) * 4 + np.roll(src, 1, 0) + np.roll(src, -1, 0) + np.roll(src, 1, 1) + np.roll(src, -1, 1)) / 8
dgd = np.pad(np.clip(np.round(blur / 6) * 6 + g.normal(0, 4, blur.shape), 0, amp).astype(np.uint16), PAD, mode="edge")
d_src = torch.from_numpy(src.view(np.uint8).reshape(-1)).cuda()
d_dgd = torch.from_numpy(dgd.view(np.uint8).reshape(-1)).cuda()
res = {}
for walk in (os.environ.get("WALKS", "default,table,line").split(",")):
    os.environ["SVT_HIP_LR_SG_WALK"] = walk
    for name, wn, sg in (("sg16", (0, 7, 1, 0), (1, 0, 16, 1, 1)), ("sg16_norefine", (0, 7, 1, 0), (1, 0, 16, 1, 0)), ("sg2", (0, 7, 1, 0), (1, 0, 16, 8, 1)), ("wn7", (1, 7, 1, 0), (0, 0, 16, 1, 1)),
                         ("full", (1, 7, 1, 0), (1, 0, 16, 1, 1)), ("fast", (1, 5, 1, 1), (1, 0, 16, 8, 1))):
        if (walk != "default" and name.startswith("wn")) or (os.environ.get("CONFIGS") and name not in os.environ["CONFIGS"].split(",")):
            continue
        P = pkg.LrSearchParams()
        P.src, P.dgd = d_src.data_ptr(), d_dgd.data_ptr() + (PAD * dgd.shape[1] + PAD) * 2
        P.dgd_stride, P.src_stride, P.width, P.height, P.unit_size, P.ss_y, P.highbd, P.bit_depth = dgd.shape[1], W, W, H, 256, 0, 1, bd
        P.wn_enabled, P.wiener_win, P.wn_use_refinement, P.wn_max_one_refinement_step = wn
        P.sg_enabled, P.sg_start_ep, P.sg_end_ep, P.sg_ep_inc, P.sg_refine = sg
        n = ((H + 128) // 256) * ((W + 128) // 256)
        ws = torch.zeros(lib.svt_hip_lr_search_workspace(C.addressof(P)), dtype=torch.uint8, device="cuda")
        d_out = torch.zeros(n * 72, dtype=torch.uint8, device="cuda")
        ts = []
        for it in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            assert lib.svt_hip_lr_search_plane(C.addressof(P), None, d_out.data_ptr(), ws.data_ptr(), stream) == 0
            torch.cuda.synchronize()
            if it:
                ts.append(time.perf_counter() - t0)
        out = d_out.cpu().numpy().tobytes()
        if name in res:
            assert res[name] == out, "the walk forms disagree on " + name
        res[name] = out
        print("%-5s %-14s %8.3f ms" % (walk, name, float(np.median(ts)) * 1e3), flush=True)
