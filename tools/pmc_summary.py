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
    # One kernel may serve several legs of the command with different launch sizes (the headline ME batch, the one-picture frame-partition step, the
    # session legs): launches are grouped by (kernel, grid size); the group with the largest total time keeps the plain kernel name -- that is the
    # workload bench.py's `roofline` of this kernel refers to -- the others are listed as "kernel @grid=N".
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(stats_dir, "*kernel_trace.csv"))[0])):
        groups[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    dominant = {}
    for (k, gsz), v in groups.items():
        if k not in dominant or sum(v) > sum(groups[(k, dominant[k])]):
            dominant[k] = gsz
    label = lambda k, gsz: k if dominant.get(k) == gsz else "%s @grid=%d" % (k, gsz)  # noqa: E731
    total = sum(sum(v) for v in groups.values())
    with open("profiles/%s_kernel_stats.txt" % tag, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- %s   (MI355X)\n" % cmd)
        f.write("# launches grouped by (kernel, grid size); the plain name = the group with the largest total time\n")
        f.write("# %-72s %6s %12s %12s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "median_us", "pct"))
        for (k, gsz), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            if k.startswith("at::") or "rocclr" in k:
                continue
            v = sorted(v)
            f.write("%-74s %6d %12.1f %12.2f %12.2f %7.2f\n" % (label(k, gsz)[:74], len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, 100.0 * sum(v) / total))
    if fdir == "-":  # kernel statistics only
        return
    traffic = collections.defaultdict(dict)
    for d, ctr in ((fdir, "FETCH_SIZE"), (wdir, "WRITE_SIZE")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(glob.glob(os.path.join(d, "*counter_collection.csv"))[0])):
            if r["Counter_Name"] == ctr:
                agg[label(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
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
