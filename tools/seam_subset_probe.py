#!/usr/bin/env python3
"""Which stage seams pay at 1080p preset 8 (60 frames)?  fps of the AVX2 / AVX-512 host alone and with subsets of the stage seams on the MI355X, medians of `reps` runs, every
bitstream compared with the C-only encoder's.    python tools/seam_subset_probe.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import enc_identity as e
lib = os.path.join(ROOT, "svt-av1-psy_amd", "libsvtav1_hip.so")
out = os.path.join(ROOT, "gpurun_out", "seam_subsets")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ME, TF, TPL, DLF, CDEF, LR = ["+seam"], ["+tfseam", "+tfsubpel", "+tfdriver"], ["+tplseam", "+tplrecon"], ["+dlfseam"], ["+cdefseam"], ["+lrseam"]
SUBSETS = {"all": ME + TF + TPL + DLF + CDEF + LR, "no_lr": ME + TF + TPL + DLF + CDEF, "no_lr_dlf": ME + TF + TPL + CDEF, "me_tf_tpl": ME + TF + TPL, "me_tpl": ME + TPL, "me_tf": ME + TF,
           "me": ME, "tpl_needs_me": ME + TPL + CDEF, "lr_only": LR, "dlf_only": DLF, "paying": ME + TF + TPL + CDEF}
if os.environ.get("SUBSETS"):
    SUBSETS = {k: SUBSETS[k] for k in os.environ["SUBSETS"].split(",")}
base = (1920, 1080, 60, 8, ["--preset", "8"])
for name, seams in SUBSETS.items():
    e.CASES["fps_sub_" + name] = base[:4] + (base[4] + seams,)
med = lambda v: sorted(v)[len(v) // 2]
k0 = next(iter(SUBSETS))
first = e.run_case("fps_sub_" + k0, lib, out, timeout=600, host="c")
want = open(os.path.join(out, "fps_sub_" + k0 + "_c.ivf"), "rb").read()
print("c-only fps", first.get("fps_c"), "identical", first.get("identical"), flush=True)
for host in ("avx2", "avx512"):
    if not os.path.exists(e.HOST_ENC[host]):
        continue
    for name in SUBSETS:
        r = e.repeat_pairs("fps_sub_" + name, lib, out, want, pairs=reps, host=host)
        print("%-7s %-14s alone %6.1f (%s)   with %6.1f (%s)   identical %s" % (host, name, med(r["fps_alone"]), " ".join("%.0f" % v for v in r["fps_alone"]), med(r["fps_with_stages"]),
                                                                             " ".join("%.0f" % v for v in r["fps_with_stages"]), r["identical"]), flush=True)
