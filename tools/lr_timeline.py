#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV directory of `bench.py --legs lrsearch` into the launch timeline of one svt_hip_lr_search_plane call per setting
(which kernel ran on which queue, when, for how long).  usage: tools/lr_timeline.py <trace_dir> <out.txt>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if any(k in r['Kernel_Name'] for k in ('lr_', 'stats_', 'rocclr'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
calls, cur, last = [], [], None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'lr_rects_kernel' in r['Kernel_Name'] and cur:
        calls.append(cur)
        cur = []
    cur.append(r)
calls.append(cur)
calls = [c for c in calls if any('lr_rects' in r['Kernel_Name'] for r in c)]
out = open(sys.argv[2], 'w')
seen = {}
for c in calls:  # the LAST call of each distinct launch count (= setting) is printed
    seen[len(c) // 8] = c
for key, c in sorted(seen.items()):
    t0 = int(c[0]['Start_Timestamp'])
    t1 = max(int(r['End_Timestamp']) for r in c)
    out.write('call with %d launches, first start -> last end %.1f us\n' % (len(c), (t1 - t0) / 1e3))
    for r in c:
        out.write('  %9.1f %9.1f q%-3s %s grid=%s\n' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?'),
                                                       r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:56], r['Grid_Size_X']))
out.close()
