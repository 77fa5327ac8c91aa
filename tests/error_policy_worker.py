"""Worker of tests/test_error_policy.py (a process of its own: the switch-off is sticky).  argv: repo root, backend ("emu" | "gpu").
The library's error policy (include/svtav1_hip.h; SURVEY 8b "errors"): after the first HIP error -- here svt_hip_debug_inject_failure(), which takes the same path as a
failed hipMalloc inside an entry point -- svt_hip_last_error() names it, every host-form / stage entry point returns SVT_HIP_E_DEVICE (or NULL) without touching the
device or the caller's buffers, nothing aborts; svt_hip_shutdown() + svt_hip_init() start clean."""
import ctypes as C
import os
import sys

import numpy as np

ROOT, BACKEND = sys.argv[1], sys.argv[2]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import EmuBackend, GpuBackend  # noqa: E402

be = GpuBackend() if BACKEND == "gpu" else EmuBackend()
lib, pkg = be.lib, be.pkg
E_DEVICE = -100
assert lib.svt_hip_last_error() is None and lib.svt_hip_failed() == 0

# a host form that works ...
W, H = 64, 64
plane = np.random.default_rng(3).integers(0, 1024, (H, W)).astype(np.uint16)
units = np.zeros(1, dtype=pkg.LrUnit)
out = np.zeros((H, W), np.uint16)
above, below = np.zeros((4, W), np.uint16), np.zeros((4, W), np.uint16)
L = pkg.LrParams(plane.ctypes.data, above.ctypes.data, below.ctypes.data, out.ctypes.data, W, W, W, W, H, 64, 0, 0, 1, 10, units.ctypes.data)
assert lib.svt_hip_lr_filter_frame_host(C.byref(L)) == 0
assert np.array_equal(out, plane)  # (one unit of type RESTORE_NONE: a copy)

# ... the failure ...
lib.svt_hip_debug_inject_failure()
msg = lib.svt_hip_last_error()
assert lib.svt_hip_failed() == 1 and msg and b"injected failure" in msg, msg

# ... and every entry point declines, leaving the caller's memory alone
out[:] = 7
assert lib.svt_hip_lr_filter_frame_host(C.byref(L)) == E_DEVICE and (out == 7).all()
A = pkg.CdefApplyHost() if hasattr(pkg, "CdefApplyHost") else None
if A is not None:
    assert lib.svt_hip_cdef_apply_host(C.byref(A)) == E_DEVICE
assert lib.svt_hip_me_session_create(64, 64, 64, 0, 0, 64, 4, 1, 64, 64, 2) is None
assert lib.svt_hip_frame_partition_create((C.c_int * 1)(0), 1) is None
assert lib.svt_hip_host_register(plane.ctypes.data, plane.nbytes) != 0
assert lib.svt_hip_set_thread_device(0) == E_DEVICE
assert lib.svt_hip_setup_rtcd(0) == 0  # nothing is installed any more
lib.svt_hip_warmup()  # (void: returns)

# a fresh start
lib.svt_hip_shutdown()
assert lib.svt_hip_init(0) == 0 and lib.svt_hip_last_error() is None
out[:] = 0
assert lib.svt_hip_lr_filter_frame_host(C.byref(L)) == 0 and np.array_equal(out, plane)
print("ERROR_POLICY_OK")
