#!/usr/bin/env python3
"""Run selected bench legs only (for rocprofv3 passes that should see just a few kernels).
   python tools/microbench.py cdef|lr|txfm|hme|sad|fwd32 [--steps N]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bench_legs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("legs", nargs="+")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-parity-check", action="store_true")
    a = ap.parse_args()
    bench.NO_CHECK = a.no_parity_check
    import torch
    torch.cuda.set_device(0)
    pkg = bench.entry._pkg()
    lib = pkg.load(init_device=0)
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    for leg in a.legs:
        if leg == "cdef":
            out.update(bench.bench_cdef(torch, lib, pkg, stream, a, cpu=False))
        elif leg == "lr":
            out.update(bench_legs.lr_frames(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "txfm":
            out["txfm_quant_roundtrip"] = bench_legs.txfm_roundtrip(torch, lib, pkg, stream, a.steps, a.warmup)
        elif leg == "hme":
            out.update(bench_legs.hme_sad_loop(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "picprep":
            out.update(bench_legs.picprep(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "deblock":
            out.update(bench_legs.deblock(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "stats":
            out.update(bench_legs.lr_stats(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "cdefchain":
            out.update(bench_legs.cdef_chain(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "mesession":
            out.update(bench_legs.me_session(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "hmechain":
            out.update(bench_legs.hme_chain(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "mestage":
            out.update(bench_legs.me_stage(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "mesessionstage":
            out.update(bench_legs.me_session_stage(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "tf":
            out.update(bench_legs.tf_frames(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "tfsubpel":
            out.update(bench_legs.tf_subpel(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "tfmc":
            out.update(bench_legs.tf_inter_pred(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "lrsearch":
            out.update(bench_legs.lr_search(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "meresults":
            out.update(bench_legs.me_results(torch, lib, pkg, stream, a.steps, a.warmup))
        elif leg == "sad":
            out["sad64x64_pairs"] = bench.bench_sad_pairs(torch, lib, pkg, stream, a, False)
        elif leg == "fwd32":
            out.update(bench.bench_fwd_txfm(torch, lib, pkg, stream, a, cpu=False))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
