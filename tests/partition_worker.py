"""Worker of tests/test_partition.py: runs in a process of its own because the device count (emulated devices: SVT_HIPEMU_DEVICES; logical devices of one GPU:
SVT_HIP_VIRTUAL_DEVICES) is fixed at first use.
argv: repo root, number of devices, backend ("emu" | "gpu"), repetitions, jitter in microseconds.
ONE picture over N devices through the C ABI of csrc/partition.hip -- ME descriptor strips, CDEF search / apply / apply with the search's directions on luma and a
chroma plane, loop-restoration stripes -- each compared with the single-device call on the same inputs and, for the filters, with the CPU checker (oracle/).

Ordering (backend gpu, SVT_HIP_VIRTUAL_DEVICES=N: N logical devices with their own streams, events and arenas on the one MI355X -- a really asynchronous device):
every repetition
  * CLEARS the inputs on the home stream, delays the home stream, and only then restores them (device-to-device, home stream) right before the partition call:
    a peer that did not wait for `ready` mirrors zeros;
  * snapshots the outputs on the home stream right after the call returns (everything is only enqueued then): a home stream that did not wait for `done k`
    snapshots strip k before it arrived;
  * runs with svt_hip_frame_partition_set_jitter: delays of random length between the protocol's steps on every stream, a new interleaving per call."""
import ctypes as C
import os
import sys

import numpy as np

ROOT, N = sys.argv[1], int(sys.argv[2])
BACKEND = sys.argv[3] if len(sys.argv) > 3 else "emu"
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
JITTER_US = int(sys.argv[5]) if len(sys.argv) > 5 else 0
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import EmuBackend, GpuBackend, p  # noqa: E402
from test_cdef import synth_plane  # noqa: E402
from test_oracle_pin_restoration import make_units, unit_grid  # noqa: E402

be = GpuBackend() if BACKEND == "gpu" else EmuBackend()
lib, pkg = be.lib, be.pkg
assert lib.svt_hip_device_count() == N, (lib.svt_hip_device_count(), N)
if BACKEND == "gpu":
    assert lib.svt_hip_physical_device_count() == 1 or N <= lib.svt_hip_physical_device_count() * int(os.environ.get("SVT_HIP_VIRTUAL_DEVICES", "1"))
oracle = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
devs = (C.c_int * N)(*range(N))
assert lib.svt_hip_frame_partition_create(devs, 0) is None and lib.svt_hip_frame_partition_create((C.c_int * 2)(0, 0), 2) is None  # empty / repeated lists are refused
assert lib.svt_hip_frame_partition_create((C.c_int * 1)(N), 1) is None                                                            # a device that does not exist
part = lib.svt_hip_frame_partition_create(devs, N)
assert part and lib.svt_hip_frame_partition_size(part) == N


# ---- backend-neutral device buffers ----
def dev(a):
    return be.dev(a)


def ptr(h):
    return be.ptr(h)


def host(h):
    return np.asarray(be.host(h))


def clear(h):  # on the home stream
    if be.is_gpu:
        h.zero_()
    else:
        h[...] = 0


def restore(h, backup):  # device-to-device on the home stream
    if be.is_gpu:
        h.copy_(backup)
    else:
        h[...] = backup


def snapshot(h):  # a copy taken on the home stream, in stream order
    if be.is_gpu:
        t = h.clone()
        t._np_dtype, t._np_shape = h._np_dtype, h._np_shape
        return t
    return h.copy()


def shake_inputs(pairs):
    """inputs become final LATE on the home stream: cleared, a delay, restored from their backups"""
    if REPS == 1 and not JITTER_US:
        return
    for h, _ in pairs:
        clear(h)
    lib.svt_hip_debug_spin(be.stream, 150)
    for h, b in pairs:
        restore(h, b)


# ---- ME: 7 (SB, reference) items over N strips ----
g = np.random.default_rng(7)
n, stride, rows = 7, 64 * 7 + 120, 64 + 24
planes = g.integers(0, 256, (2, rows, stride), dtype=np.uint8)
descs = np.zeros(n, dtype=pkg.MeSearchDesc)
for i in range(n):
    descs[i] = (i * 64, rows * stride + i * 64 + 2, stride, stride, -4, -1, 8, 3)
d_planes, d_planes_b, d_descs, d_descs_b = dev(planes), dev(planes), dev(descs.view(np.uint8)), dev(descs.view(np.uint8))
want_me = [np.zeros(n * 85, np.uint32) for _ in range(2)]
for i in range(n):
    oracle.oracle_me_fullpel_search(C.c_void_p(planes.ctypes.data + int(descs[i]["src_off"])), stride, C.c_void_p(planes.ctypes.data + int(descs[i]["ref_off"])), stride,
                                    -4, -1, 8, 3, 0, C.c_void_p(want_me[0].ctypes.data + i * 85 * 4), C.c_void_p(want_me[1].ctypes.data + i * 85 * 4))
d_s, d_m = dev(np.zeros(n * 85, np.uint32)), dev(np.zeros(n * 85, np.uint32))
lib.svt_hip_me_fullpel_search_batch(ptr(d_planes), ptr(d_planes), ptr(d_descs), n, 8, 3, 0, ptr(d_s), ptr(d_m), None, be.stream)
be.sync()
one = [host(d_s).copy(), host(d_m).copy()]
assert np.array_equal(one[0], want_me[0]) and np.array_equal(one[1], want_me[1]) and one[0].any()


def me_partition():
    clear(d_s), clear(d_m)
    shake_inputs([(d_planes, d_planes_b), (d_descs, d_descs_b)])
    assert lib.svt_hip_frame_partition_me(part, ptr(d_planes), planes.nbytes, ptr(d_planes), planes.nbytes, ptr(d_descs), n, 8, 3, 0, ptr(d_s), ptr(d_m), None, be.stream) == 0
    s_s, s_m = snapshot(d_s), snapshot(d_m)
    be.sync()
    return host(s_s), host(s_m)


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
d_rec, d_rec_b, d_src, d_src_b, d_skip, d_skip_b = dev(rec), dev(rec), dev(src), dev(src), dev(skip), dev(skip)
d_apri, d_asec, d_cpri, d_csec = dev(apri), dev(asec), dev(cpri), dev(csec)


def cdef(mode, use_part, pli=0, dirs=None, var=None):
    d_out = dev(rec)
    d_d = dev(np.zeros(nfb * 64, np.uint8) if dirs is None else dirs)
    d_v = dev(np.zeros(nfb * 64, np.int32) if var is None else var)
    d_mse = dev(np.zeros(nfb * len(cands), np.uint64))
    pr, sc = (d_cpri, d_csec) if mode == 1 else (d_apri, d_asec)
    P = pkg.CdefParams(ptr(d_rec), ptr(d_src), ptr(d_out), W, W, W, W, H, 0, 0, pli, 1, bd - 8, 5, 5, 1, len(cands) if mode == 1 else 0, ptr(d_skip),
                       ptr(pr), ptr(sc), ptr(d_d), ptr(d_v), ptr(d_mse))
    if use_part:
        shake_inputs([(d_rec, d_rec_b), (d_src, d_src_b), (d_skip, d_skip_b)])
        assert lib.svt_hip_frame_partition_cdef(part, mode, C.byref(P), be.stream) == 0
    else:
        lib.svt_hip_cdef_frame(mode, C.byref(P), be.stream)
    snaps = [snapshot(x) for x in (d_out, d_d, d_v, d_mse)]
    be.sync()
    return tuple(host(x).copy() for x in snaps)


want_c, o_dir, o_var, o_mse = rec.copy(), np.zeros(nfb * 64, np.uint8), np.zeros(nfb * 64, np.int32), np.zeros(1, np.uint64)
oracle.oracle_cdef_frame(0, p(rec), W, p(rec), W, p(want_c), W, W, H, 0, 0, 0, 1, bd - 8, 5, 5, 1, p(skip), p(apri), p(asec), 0, p(o_dir), p(o_var), p(o_mse))
nstripes = (H + 8 + 63) // 64
above, below = g2.integers(0, 1 << bd, (2 * nstripes, W)).astype(np.uint16), g2.integers(0, 1 << bd, (2 * nstripes, W)).astype(np.uint16)
nvu, nhu = unit_grid(W, H, us)
units = make_units(g2, nvu, nhu, pkg.LrUnit)
want_l = np.zeros((H, W), np.uint16)
oracle.oracle_lr_filter_frame(p(want_c), W, p(above), p(below), W, p(want_l), W, W, H, 0, us, p(units), bd, 1)
d_applied, d_applied_b = dev(want_c), dev(want_c)
d_above, d_below, d_units = dev(above), dev(below), dev(units.view(np.uint8))

single = {}
for rep in range(REPS):
    if JITTER_US:
        lib.svt_hip_frame_partition_set_jitter(part, 0x1234567 + 977 * rep, JITTER_US)
    got = me_partition()
    assert np.array_equal(got[0], one[0]) and np.array_equal(got[1], one[1]), ("ME", rep, np.argwhere(got[0] != one[0])[:4])
    for mode in (1, 0):
        if (mode, 0) not in single:
            single[(mode, 0)] = cdef(mode, False)
        a, b = single[(mode, 0)], cdef(mode, True)
        for x, y, what in zip(a, b, ("out", "dir", "var", "mse")):
            assert np.array_equal(x, y), ("cdef mode %d rep %d" % (mode, rep), what, np.argwhere(x != y)[:4])
    s_out, s_dir, s_var, s_mse = single[(1, 0)]
    assert s_mse.any() and s_dir.any()
    # mode 2 (apply with the search's directions) and a "chroma" plane (pli = 1: directions are an input, mirrored to the peers)
    for mode, pli in ((2, 0), (0, 1), (1, 1)):
        if (mode, pli) not in single:
            single[(mode, pli)] = cdef(mode, False, pli, s_dir, s_var)
        a, b = single[(mode, pli)], cdef(mode, True, pli, s_dir, s_var)
        for x, y, what in zip(a, b, ("out", "dir", "var", "mse")):
            assert np.array_equal(x, y), ("cdef mode %d pli %d rep %d" % (mode, pli, rep), what)
    applied = cdef(0, True)[0]
    assert np.array_equal(applied, want_c), ("cdef apply vs the checker", rep)

    d_out = dev(np.zeros((H, W), np.uint16))
    L = pkg.LrParams(ptr(d_applied), ptr(d_above), ptr(d_below), ptr(d_out), W, W, W, W, H, us, 0, 0, 1, bd, ptr(d_units))
    shake_inputs([(d_applied, d_applied_b)])
    assert lib.svt_hip_frame_partition_lr(part, C.byref(L), be.stream) == 0
    s_l = snapshot(d_out)
    be.sync()
    out = host(s_l)
    assert np.array_equal(out, want_l), ("LR", rep, np.argwhere(out != want_l)[:5])

calls, bin_, bout = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
lib.svt_hip_frame_partition_stats(part, C.byref(calls), C.byref(bin_), C.byref(bout))
assert calls.value >= 8 * REPS and (N == 1 or (bin_.value > 0 and bout.value > 0)), (calls.value, bin_.value, bout.value)
lib.svt_hip_frame_partition_destroy(part)
print("PARTITION_OK", N, calls.value, bin_.value, bout.value)
