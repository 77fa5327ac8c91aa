"""SURVEY 8e, the frame-partition case from a C host (csrc/partition.hip): one picture over 1, 2 and 3 emulated devices -- ME strips, CDEF (search, apply, apply with
the search's directions, a chroma plane) and loop-restoration stripes -- bit-identical to the single-device calls and to the CPU checker; peer copies go through the
emulator's hipMemcpyPeerAsync, which aborts when a pointer does not live on the device it is said to live on.  (GPU: the driver's boxes have one device; the 1-device
partition runs the same entry points there.)"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.parametrize("n", [1, 2, 3])
def test_frame_partition_emulated_devices(n):
    from conftest import EmuBackend  # builds the emulator library if needed
    EmuBackend()
    env = dict(os.environ, SVT_HIPEMU_DEVICES=str(n))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "partition_worker.py"), ROOT, str(n)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PARTITION_OK %d" % n in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


@pytest.mark.parametrize("n", [2, 4])
def test_frame_partition_virtual_devices_emulator(n):
    """the LOGICAL-device layer (SVT_HIP_VIRTUAL_DEVICES: n logical devices on the one emulated device) through the same worker"""
    from conftest import EmuBackend
    EmuBackend()
    env = dict(os.environ, SVT_HIPEMU_DEVICES="1", SVT_HIP_VIRTUAL_DEVICES=str(n))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "partition_worker.py"), ROOT, str(n), "emu", "2", "50"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PARTITION_OK %d" % n in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3, 4])
def test_frame_partition_virtual_peers_gpu(n):
    """The multi-device code on a really asynchronous device: n logical devices on the MI355X (SVT_HIP_VIRTUAL_DEVICES), each peer with its own stream, `done` event
    and arena, hipMemcpyPeerAsync between them; 50 repetitions, inputs that become final late on the home stream, outputs snapshotted on the home stream right after
    the call, random delays between the protocol's steps (tests/partition_worker.py).  Compared with the single-device calls and the CPU checker."""
    lib = os.path.join(ROOT, "svt-av1-psy_amd", "libsvtav1_hip.so")
    assert os.path.exists(lib), "libsvtav1_hip.so missing (no CPU fallback)"
    env = dict(os.environ, SVT_HIP_VIRTUAL_DEVICES=str(n))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "partition_worker.py"), ROOT, str(n), "gpu", "50", "300"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PARTITION_OK %d" % n in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_frame_partition_home_device_only(be):
    """a partition of ONE device (what the driver's single-GPU boxes can run): the same entry points, every strip on the home device -- ME and CDEF search results equal
    the plain calls'"""
    import ctypes as C

    import numpy as np

    from test_cdef import synth_plane
    lib, pkg = be.lib, be.pkg
    part = lib.svt_hip_frame_partition_create((C.c_int * 1)(0), 1)
    assert part and lib.svt_hip_frame_partition_size(part) == 1
    g = np.random.default_rng(11)
    n, stride, rows = 5, 64 * 5 + 120, 64 + 24
    planes = g.integers(0, 256, (2, rows, stride), dtype=np.uint8)
    descs = np.zeros(n, dtype=pkg.MeSearchDesc)
    for i in range(n):
        descs[i] = (i * 64, rows * stride + i * 64 + 2, stride, stride, -4, -1, 8, 3)
    d_pl, d_d = be.dev(planes), be.dev(descs.view(np.uint8))
    res = []
    for use_part in (False, True):
        d_s, d_m = be.dev(np.zeros(n * 85, np.uint32)), be.dev(np.zeros(n * 85, np.uint32))
        if use_part:
            assert lib.svt_hip_frame_partition_me(part, be.ptr(d_pl), planes.nbytes, be.ptr(d_pl), planes.nbytes, be.ptr(d_d), n, 8, 3, 0, be.ptr(d_s), be.ptr(d_m), None, be.stream) == 0
        else:
            lib.svt_hip_me_fullpel_search_batch(be.ptr(d_pl), be.ptr(d_pl), be.ptr(d_d), n, 8, 3, 0, be.ptr(d_s), be.ptr(d_m), None, be.stream)
        be.sync()
        res.append((be.host(d_s), be.host(d_m)))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.asarray(res[0][0]).any()
    bd, W, H = 10, 136, 200
    rec = synth_plane(np.random.default_rng(5), W, H, bd).astype(np.uint16)
    nhfb, nvfb = (W + 63) // 64, (H + 63) // 64
    nfb = nhfb * nvfb
    skip = np.zeros((nvfb * 8, nhfb * 8), np.uint8)
    cands = [(pr, sc) for pr in (0, 2, 5, 9) for sc in (0, 1, 2, 4)]
    pri, sec = np.array([c[0] for c in cands], np.int32), np.array([c[1] for c in cands], np.int32)
    d_rec, d_skip, d_pri, d_sec = be.dev(rec), be.dev(skip), be.dev(pri), be.dev(sec)
    mses = []
    for use_part in (False, True):
        d_dir, d_var, d_mse = be.dev(np.zeros(nfb * 64, np.uint8)), be.dev(np.zeros(nfb * 64, np.int32)), be.dev(np.zeros(nfb * len(cands), np.uint64))
        P = pkg.CdefParams(be.ptr(d_rec), be.ptr(d_rec), None, W, W, W, W, H, 0, 0, 0, 1, bd - 8, 5, 5, 1, len(cands), be.ptr(d_skip), be.ptr(d_pri), be.ptr(d_sec), be.ptr(d_dir),
                           be.ptr(d_var), be.ptr(d_mse))
        if use_part:
            assert lib.svt_hip_frame_partition_cdef(part, 1, C.byref(P), be.stream) == 0
        else:
            lib.svt_hip_cdef_frame(1, C.byref(P), be.stream)
        be.sync()
        mses.append((np.asarray(be.host(d_mse)).copy(), np.asarray(be.host(d_dir)).copy()))
    assert np.array_equal(mses[0][0], mses[1][0]) and np.array_equal(mses[0][1], mses[1][1])
    lib.svt_hip_frame_partition_destroy(part)
