#!/usr/bin/env python3
"""Is a bitstream difference of the 10-bit preset-8 encode caused by a device stage's RESULTS or by its TIMING?  (round 5; DESIGN.md 0, "10-bit preset 8")

The reference encoder does not reproduce its own 10-bit preset-8 bitstream when it runs multi-threaded (six md5s in six `--lp 4` runs of the reference
built with its own CMake).  This probe shows the same latent race at `--lp 1` -- one thread per pipeline stage is still several threads -- as soon as one stage's
duration changes:

  c-only            the reference alone, N runs                                               -> one md5
  c-only + delay    the reference alone, its CDEF / LR stage entries sleeping D microseconds   (integration/seam_cpu.h: SVT_HIP_SEAM_DELAY_US; no HIP library loaded)
  cdef              the CDEF stage on the device (SVT_HIP_CDEF_SEAM=1)
  cdef verify       the same with SVT_HIP_CDEF_SEAM_VERIFY=1: the reference's own search and filter run as well, every device result is compared with them
                    (differences would be printed) and the picture CONTINUES WITH THE REFERENCE'S results -- a flip here cannot come from device data

    python tools/race_probe_10bit.py [--lib svt-av1-psy_amd/libsvtav1_hip.so] [--runs 10]
"""
import argparse, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import enc_identity as e

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=os.path.join(ROOT, "svt-av1-psy_amd", "libsvtav1_hip.so"))
ap.add_argument("--runs", type=int, default=10)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "race_probe"))
a = ap.parse_args()
os.makedirs(a.out, exist_ok=True)
w, h, n = 448, 264, 8
extra = ["--preset", "8", "--lp", "1"]
for bd in (10, 8):
    clip = os.path.join(a.out, "clip%d.yuv" % bd)
    e.make_clip(clip, w, h, n, bd)

    def run(tag, env):
        r, _ = e.encode(clip, w, h, n, bd, extra, os.path.join(a.out, tag), env, timeout=600)
        bad = [ln for ln in r.stderr.splitlines() if "VERIFY" in ln and "differ" in ln]
        if bad:
            print("\n".join(bad[:6]), flush=True)
        return hashlib.md5(open(os.path.join(a.out, tag + ".ivf"), "rb").read()).hexdigest()[:10] if r.returncode == 0 else "rc%d" % r.returncode

    def row(name, env, runs=a.runs):
        got = [run("%s%d" % (name.replace(" ", "_"), i), env) for i in range(runs)]
        print("%2d-bit %-22s %s" % (bd, name, " ".join(got)), flush=True)
    row("c-only", None, 3)
    for us in (2000, 20000, 60000):
        row("c-only delay %d us" % us, {"SVT_HIP_SEAM_DELAY_US": str(us)})
    if os.path.exists(a.lib):
        base = {"SVT_HIP": "0", "SVT_HIP_LIB": a.lib, "SVT_HIP_ONLY": "-", "SVT_HIP_CDEF_SEAM": "1"}
        row("cdef", base)
        row("cdef verify", dict(base, SVT_HIP_CDEF_SEAM_VERIFY="1"))
