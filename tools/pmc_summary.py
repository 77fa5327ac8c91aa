#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/, scratch) into the small tracked files under profiles/.

  python tools/pmc_summary.py <round-tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> "<command that was profiled>"
  python tools/pmc_summary.py <round-tag> <stats_dir> - - "<command>"        (kernel statistics only)

Writes profiles/<tag>_kernel_stats.txt (the --kernel-trace --stats table, our kernels first) and profiles/<tag>_pmc_traffic.json
(HBM bytes per launch and kernel).  Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are
collected in separate --pmc passes, are reported in KiB, and on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide
coalesced reads, so it is doubled; WRITE_SIZE is used as reported (it matched the known output byte counts of every kernel here)."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    tag, stats_dir, fdir, wdir, cmd = sys.argv[1:6]
    os.makedirs("profiles", exist_ok=True)
    rows = list(csv.DictReader(open(glob.glob(os.path.join(stats_dir, "*kernel_stats.csv"))[0])))
    with open("profiles/%s_kernel_stats.txt" % tag, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- %s   (MI355X)\n" % cmd)
        f.write("# %-58s %6s %12s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for r in rows:
            f.write("%-60s %6s %12.1f %12.2f %7s\n" % (short(r["Name"])[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                    float(r["AverageNs"]) / 1e3, r["Percentage"]))
    if fdir == "-":  # kernel statistics only
        return
    traffic = collections.defaultdict(dict)
    for d, ctr in ((fdir, "FETCH_SIZE"), (wdir, "WRITE_SIZE")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(glob.glob(os.path.join(d, "*counter_collection.csv"))[0])):
            if r["Counter_Name"] == ctr:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if k.startswith("at::") or "rocclr" in k:
                continue
            kib = sum(v) / len(v)
            traffic[k][ctr + "_KiB_raw"] = kib
            traffic[k]["launches_" + ctr] = len(v)
    for k, t in traffic.items():
        rd = t.get("FETCH_SIZE_KiB_raw", 0.0) * 1024 * 2  # gfx950 correction, see module docstring
        wr = t.get("WRITE_SIZE_KiB_raw", 0.0) * 1024
        t["hbm_read_bytes_per_launch"], t["hbm_write_bytes_per_launch"], t["hbm_bytes_per_launch"] = rd, wr, rd + wr
    json.dump({"command": cmd, "kernels": traffic}, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1, sort_keys=True)
    print(open("profiles/%s_kernel_stats.txt" % tag).read())
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
