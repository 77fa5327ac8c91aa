#!/usr/bin/env python3
"""Per-leg comparison of two bench_detail.json files (markdown): python tools/round_delta.py profiles/r05_final_bench_detail.json profiles/r06_final_bench_detail.json"""
import json
import sys


def legs(path):
    d = json.load(open(path))
    out = {}
    rf = d.get("roofline") or {}
    out["me_fullpel_search_16x9 (headline)"] = (rf.get("kernel_us"), rf.get("moved_over_algorithmic"), rf.get("binds"), rf.get("valu_frac"), rf.get("frac"))
    for n, k in (d.get("kernels") or {}).items():
        if not isinstance(k, dict) or n.startswith("_"):
            continue
        r = k.get("roofline") or {}
        us = r.get("kernel_us") or (k.get("ms") * 1e3 if k.get("ms") else None)
        out[n] = (us, r.get("moved_over_algorithmic"), r.get("binds"), r.get("valu_frac"), r.get("frac"))
    return out


def f(v, sig=3):
    return "-" if v is None else (("%.*g" % (sig, v)) if isinstance(v, (int, float)) else str(v))


a, b = legs(sys.argv[1]), legs(sys.argv[2])
print("| leg | µs, round 5 | µs, round 6 | moved ÷ algorithmic, r5 → r6 | VALU frac r6 | roof frac r6 | binds |")
print("|---|---|---|---|---|---|---|")
for n in b:
    ua = a.get(n, (None,) * 5)
    ub = b[n]
    if not ub[0]:
        continue
    mark = ""
    if ua[0] and ub[0] and abs(ub[0] - ua[0]) / ua[0] > 0.08:
        mark = " **" + ("%+.0f %%" % (100.0 * (ub[0] - ua[0]) / ua[0])) + "**"
    print("| %s | %s | %s%s | %s → %s | %s | %s | %s |" % (n, f(ua[0], 4) if n in a else "(new)", f(ub[0], 4), mark, f(ua[1]), f(ub[1]), f(ub[3]), f(ub[4]), ub[2] or "-"))
