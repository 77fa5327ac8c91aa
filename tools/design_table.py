#!/usr/bin/env python3
"""DESIGN.md section 6's per-leg table, generated from a bench_detail.json ("DESIGN §6 table regenerated from it").
    python tools/design_table.py profiles/r05_final_bench_detail.json > /tmp/table.md"""
import json
import sys


def g(d, *ks, default=None):
    for k in ks:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def num(v, sig=4):
    if v is None:
        return "-"
    if isinstance(v, (int, float)):
        return ("%.*g" % (sig, v))
    return str(v)


def cpu(k, key):
    c = k.get(key)
    if not isinstance(c, dict) or "value" not in c:
        return "-"
    return "%s (%s core%s%s)" % (num(c["value"]), c.get("cores", "?"), "" if c.get("cores") == 1 else "s", "" if c.get("kind") == "reference" else ", " + str(c.get("kind")))


def main(path):
    d = json.load(open(path))
    rf = d.get("roofline") or {}
    rows = [("me_fullpel_search_16x9 (headline)", {"value": d.get("value"), "unit": "Mblocks/s", "roofline": rf, "cpu_baseline": d.get("cpu_baseline")})]
    rows += [(n, k) for n, k in (d.get("kernels") or {}).items() if isinstance(k, dict) and not n.startswith("_")]
    print("| leg | GPU figure | µs per launch / call | HBM frac (algorithmic) | moved ÷ algorithmic | VALU frac | binds | reference CPU, AVX2 | reference CPU, AVX-512 | other CPU figures |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for n, k in rows:
        r = k.get("roofline") or {}
        val = k.get("value")
        fig = "%s %s" % (num(val), k.get("unit", "")) if val is not None else (("%s ms" % num(k.get("ms"))) if k.get("ms") is not None else "-")
        moved = r.get("moved_over_algorithmic")
        if moved is None and r.get("traffic") and r.get("algorithmic_bytes_per_launch"):
            moved = r["traffic"] / r["algorithmic_bytes_per_launch"]
        others = ", ".join("%s: %s" % (x.replace("cpu_baseline_", ""), cpu(k, x)) for x in k if x.startswith("cpu_baseline_") and x not in ("cpu_baseline_avx512",) and isinstance(k[x], dict))
        print("| %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (n, fig, num(r.get("kernel_us")), num(r.get("frac"), 3), num(moved, 3), num(r.get("valu_frac"), 3), r.get("binds") or "-",
                                                                 cpu(k, "cpu_baseline"), cpu(k, "cpu_baseline_avx512"), others or "-"))


if __name__ == "__main__":
    main(sys.argv[1])
