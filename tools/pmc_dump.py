#!/usr/bin/env python3
"""Print per-kernel averages of every counter found under the given rocprofv3 output directories (glob patterns allowed)."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for pat in sys.argv[1:]:
    for d in glob.glob(pat):
        for f in glob.glob(d + "/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                if k.startswith("at::") or "rocclr" in k:
                    continue
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-24s %14.0f" % (c, sum(vals) / len(vals)))
    if "SQ_LDS_IDX_ACTIVE" in v and "SQ_INSTS_LDS" in v:
        print("   -> LDS cycles per LDS instruction: %.1f" % (sum(v["SQ_LDS_IDX_ACTIVE"]) / sum(v["SQ_INSTS_LDS"])))
    if "SQ_INSTS_VALU" in v and "SQ_WAVES" in v:
        print("   -> VALU instructions per wave: %.0f" % (sum(v["SQ_INSTS_VALU"]) / sum(v["SQ_WAVES"])))
