"""Worker of tests/test_partition.py: runs in a process of its own because the emulator fixes its device count (SVT_HIPEMU_DEVICES) at first use.
argv: repo root, number of devices.  ONE picture over N emulated devices through the C ABI of csrc/partition.hip -- ME descriptor strips, CDEF search / apply / apply
with the search's directions on luma and a chroma plane, loop-restoration stripes -- each compared with the single-device call on the same inputs and, for the filters,
with the CPU checker (oracle/)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT, N = sys.argv[1], int(sys.argv[2])
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import EmuBackend, p  # noqa: E402
from test_cdef import synth_plane  # noqa: E402
from test_oracle_pin_restoration import make_units, unit_grid  # noqa: E402

be = EmuBackend()
lib, pkg = be.lib, be.pkg
assert lib.svt_hip_device_count() == N, lib.svt_hip_device_count()
oracle = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
devs = (C.c_int * N)(*range(N))
assert lib.svt_hip_frame_partition_create(devs, 0) is None and lib.svt_hip_frame_partition_create((C.c_int * 2)(0, 0), 2) is None  # empty / repeated lists are refused
assert lib.svt_hip_frame_partition_create((C.c_int * 1)(N), 1) is None                                                            # a device that does not exist
part = lib.svt_hip_frame_partition_create(devs, N)
assert part and lib.svt_hip_frame_partition_size(part) == N

# ---- ME: 7 (SB, reference) items over N strips ----
g = np.random.default_rng(7)
n, stride, rows = 7, 64 * 7 + 120, 64 + 24
planes = g.integers(0, 256, (2, rows, stride), dtype=np.uint8)
descs = np.zeros(n, dtype=pkg.MeSearchDesc)
for i in range(n):
    descs[i] = (i * 64, rows * stride + i * 64 + 2, stride, stride, -4, -1, 8, 3)
one = [np.zeros(n * 85, np.uint32) for _ in range(2)]
lib.svt_hip_me_fullpel_search_batch(p(planes), p(planes), p(descs), n, 8, 3, 0, p(one[0]), p(one[1]), None, None)
got = [np.zeros(n * 85, np.uint32) for _ in range(2)]
assert lib.svt_hip_frame_partition_me(part, p(planes), planes.nbytes, p(planes), planes.nbytes, p(descs), n, 8, 3, 0, p(got[0]), p(got[1]), None, None) == 0
assert np.array_equal(got[0], one[0]) and np.array_equal(got[1], one[1]) and one[0].any()

# ---- CDEF + LR on one 10-bit plane: 3 x 4 filter blocks, 4 stripes ----
bd, W, H, us = 10, 136, 200, 64
rec = synth_plane(np.random.default_rng(5), W, H, bd).astype(np.uint16)
src = np.clip(rec.astype(np.int32) + np.random.default_rng(8).integers(-5, 6, rec.shape), 0, 1023).astype(np.uint16)
nhfb, nvfb = (W + 63) // 64, (H + 63) // 64
nfb = nhfb * nvfb
g2 = np.random.default_rng(6)
skip = (g2.random((nvfb * 8, nhfb * 8)) < 0.2).astype(np.uint8)
apri, asec = g2.choice(np.array([0, 4, 9], np.int32), nfb).astype(np.int32), g2.choice(np.array([0, 1, 2, 4], np.int32), nfb).astype(np.int32)
cands = [(pr, sc) for pr in (0, 1, 3, 7, 12) for sc in (0, 1, 2, 4)]
cpri, csec = np.array([c[0] for c in cands], np.int32), np.array([c[1] for c in cands], np.int32)


def cdef(mode, use_part, pli=0, dirs=None, var=None):
    out = rec.copy()
    d = np.zeros(nfb * 64, np.uint8) if dirs is None else dirs.copy()
    v = np.zeros(nfb * 64, np.int32) if var is None else var.copy()
    mse = np.zeros(nfb * len(cands), np.uint64)
    pr, sc = (cpri, csec) if mode == 1 else (apri, asec)
    P = pkg.CdefParams(rec.ctypes.data, src.ctypes.data, out.ctypes.data, W, W, W, W, H, 0, 0, pli, 1, bd - 8, 5, 5, 1, len(cands) if mode == 1 else 0, skip.ctypes.data,
                       pr.ctypes.data, sc.ctypes.data, d.ctypes.data, v.ctypes.data, mse.ctypes.data)
    if use_part:
        assert lib.svt_hip_frame_partition_cdef(part, mode, C.byref(P), None) == 0
    else:
        lib.svt_hip_cdef_frame(mode, C.byref(P), None)
    return out, d, v, mse


for mode in (1, 0):
    a, b = cdef(mode, False), cdef(mode, True)
    for x, y, what in zip(a, b, ("out", "dir", "var", "mse")):
        assert np.array_equal(x, y), ("cdef mode %d" % mode, what, np.argwhere(x != y)[:4])
s_out, s_dir, s_var, s_mse = cdef(1, True)
assert s_mse.any() and s_dir.any()
# mode 2 (apply with the search's directions) and a "chroma" plane (pli = 1: directions are an input, mirrored to the peers)
for mode, pli in ((2, 0), (0, 1), (1, 1)):
    a, b = cdef(mode, False, pli, s_dir, s_var), cdef(mode, True, pli, s_dir, s_var)
    for x, y, what in zip(a, b, ("out", "dir", "var", "mse")):
        assert np.array_equal(x, y), ("cdef mode %d pli %d" % (mode, pli), what)
want_c, o_dir, o_var, o_mse = rec.copy(), np.zeros(nfb * 64, np.uint8), np.zeros(nfb * 64, np.int32), np.zeros(1, np.uint64)
oracle.oracle_cdef_frame(0, p(rec), W, p(rec), W, p(want_c), W, W, H, 0, 0, 0, 1, bd - 8, 5, 5, 1, p(skip), p(apri), p(asec), 0, p(o_dir), p(o_var), p(o_mse))
applied = cdef(0, True)[0]
assert np.array_equal(applied, want_c)

nstripes = (H + 8 + 63) // 64
above, below = g2.integers(0, 1 << bd, (2 * nstripes, W)).astype(np.uint16), g2.integers(0, 1 << bd, (2 * nstripes, W)).astype(np.uint16)
nvu, nhu = unit_grid(W, H, us)
units = make_units(g2, nvu, nhu, pkg.LrUnit)
out = np.zeros((H, W), np.uint16)
L = pkg.LrParams(applied.ctypes.data, above.ctypes.data, below.ctypes.data, out.ctypes.data, W, W, W, W, H, us, 0, 0, 1, bd, units.ctypes.data)
assert lib.svt_hip_frame_partition_lr(part, C.byref(L), None) == 0
want_l = np.zeros((H, W), np.uint16)
oracle.oracle_lr_filter_frame(p(want_c), W, p(above), p(below), W, p(want_l), W, W, H, 0, us, p(units), bd, 1)
assert np.array_equal(out, want_l), np.argwhere(out != want_l)[:5]

calls, bin_, bout = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
lib.svt_hip_frame_partition_stats(part, C.byref(calls), C.byref(bin_), C.byref(bout))
assert calls.value >= 9 and (N == 1 or (bin_.value > 0 and bout.value > 0)), (calls.value, bin_.value, bout.value)
lib.svt_hip_frame_partition_destroy(part)
print("PARTITION_OK", N, calls.value, bin_.value, bout.value)
