#!/usr/bin/env python3
"""Per-kernel resource table of the product library, from the compiler's own metadata (no GPU needed).

  python tools/kernel_resources.py [out.txt]

Compiles every svt-av1-psy_amd/csrc/*.hip to gfx950 assembly (device side only) and lists, per kernel: VGPRs, SGPRs, static LDS bytes, scratch bytes
(anything but 0 is a spill or a runtime-indexed private array: a bug to fix), the waves per SIMD the register count allows (512 / allocated VGPRs, at most 8),
and the static instruction mix (VALU / SALU / LDS / VMEM: code size by class, not an execution count -- loops and branches are not weighted).
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "") for o in out]


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(ROOT, "svt-av1-psy_amd", "csrc", "*.hip"))):
            asm = os.path.join(tmp, os.path.basename(src) + ".s")
            subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", src, "-o", asm],
                           check=True, stderr=subprocess.DEVNULL)
            text = open(asm).read()
            # static instruction mix per kernel body
            mix = {}
            for m in re.finditer(r"^([A-Za-z_]\w*):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S | re.M):  # (a kernel may hold several s_endpgm: early returns)
                c = {"v": 0, "s": 0, "ds": 0, "mem": 0}
                for line in m.group(2).split("\n"):
                    op = line.strip().split(" ")[0].split("\t")[0]
                    if op.startswith("v_"): c["v"] += 1
                    elif op.startswith("ds_"): c["ds"] += 1
                    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["mem"] += 1
                    elif op.startswith("s_"): c["s"] += 1
                mix[m.group(1)] = c
            for m in re.finditer(r"- \.agpr_count:.*?\.name:\s+(\S+).*?\.wavefront_size", text, re.S):
                blk = m.group(0)
                get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))  # noqa: E731
                name = m.group(1)
                vg = get("vgpr_count")
                alloc = (vg + 7) // 8 * 8
                rows.append((os.path.basename(src), name, vg, get("sgpr_count"), get("group_segment_fixed_size"), get("private_segment_fixed_size"),
                             min(8, 512 // max(alloc, 8)), mix.get(name, {})))
    names = demangle([r[1] for r in rows])
    lines = ["# gfx950 kernel resources of libsvtav1_hip.so (hipcc -O3, compiler metadata; tools/kernel_resources.py)",
             "# %-16s %-62s %5s %5s %8s %8s %6s   %s" % ("file", "kernel", "VGPR", "SGPR", "LDS(B)", "scratch", "waves", "static VALU/SALU/LDS/VMEM")]
    for (f, _, vg, sg, lds, scr, occ, mx), n in zip(rows, names):
        lines.append("%-18s %-62s %5d %5d %8d %8d %6d   %s" % (f, n[:62], vg, sg, lds, scr, occ, "%d/%d/%d/%d" % (mx.get("v", 0), mx.get("s", 0), mx.get("ds", 0), mx.get("mem", 0))))
    spills = [n for (f, _, vg, sg, lds, scr, occ, mx), n in zip(rows, names) if scr]
    lines.append("# %d kernels, %d with scratch%s" % (len(rows), len(spills), (": " + ", ".join(spills)) if spills else ""))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
