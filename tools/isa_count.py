#!/usr/bin/env python3
"""Static instruction mix of one gfx950 kernel, per basic block (a desk check before spending GPU minutes on SQ_INSTS_* counters).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S x.hip -o x.s
  python tools/isa_count.py x.s lr_frame_kernel [--dump]

Prints, for every basic block of the first kernel whose mangled name contains the substring, the number of VALU / SALU / LDS / VMEM
instructions and the branch targets, so loop bodies can be weighted by their trip counts by hand.
"""
import re
import sys


def main():
    path, name = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(name), l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = [], ["entry", {"v": 0, "s": 0, "ds": 0, "mem": 0}, []]
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\w+):", t)
            if m:
                blocks.append(cur)
                cur = [m.group(1), {"v": 0, "s": 0, "ds": 0, "mem": 0}, []]
            continue
        op = t.split()[0]
        if dump:
            print("   ", cur[0], t.split(";")[0].strip())
        if op.startswith("v_"): cur[1]["v"] += 1
        elif op.startswith("ds_"): cur[1]["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cur[1]["mem"] += 1
        elif op.startswith("s_"):
            cur[1]["s"] += 1
            if "branch" in op:  # a branch ends the basic block even when the fall-through block carries no label
                cur[2].append(op.replace("s_", "") + "->" + t.split()[1])
                blocks.append(cur)
                cur = [cur[0].rstrip("+") + "+", {"v": 0, "s": 0, "ds": 0, "mem": 0}, []]
    blocks.append(cur)
    tot = {"v": 0, "s": 0, "ds": 0, "mem": 0}
    for b, c, br in blocks:
        for k in tot: tot[k] += c[k]
        print(f"{b:12s} valu {c['v']:4d} salu {c['s']:4d} lds {c['ds']:3d} vmem {c['mem']:3d}  {' '.join(br)}")
    print("static total", tot)


if __name__ == "__main__":
    main()
