"""Drop-in boundary: svt_hip_setup_rtcd() must overwrite the REFERENCE's own dispatch pointers (weak symbols resolved against
libsvtref.so = the reference's Codec objects) and calls made THROUGH those pointers must land in our variants."""
import ctypes as C
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import EMU_LIB, REF_LIB, ROOT

CHILD = textwrap.dedent('''
    import ctypes as C, os, sys
    import numpy as np
    ref = C.CDLL(sys.argv[1], mode=os.RTLD_GLOBAL | os.RTLD_NOW)          # defines the RTCD pointer globals
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0)); ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    before = C.c_void_p.in_dll(ref, "svt_nxm_sad_kernel").value
    hip = C.CDLL(sys.argv[2], mode=os.RTLD_GLOBAL | os.RTLD_NOW)          # weak references bind to the globals above
    hip.svt_hip_init(0)
    n = hip.svt_hip_setup_rtcd(C.c_uint64(0))
    after = C.c_void_p.in_dll(ref, "svt_nxm_sad_kernel").value
    ours = C.cast(hip.svt_nxm_sad_kernel_hip, C.c_void_p).value
    assert n >= 170, n
    assert before != after and after == ours, (before, after, ours)
    for name in ("svt_av1_fwd_txfm2d_32x32", "svt_av1_inv_txfm2d_add_16x64", "svt_aom_quantize_b", "svt_cdef_filter_block", "svt_av1_wiener_convolve_add_src",
                 "svt_aom_cdef_find_dir", "svt_av1_compute_stats", "svt_handle_transform64x64", "svt_ext_all_sad_calculation_8x8_16x16", "hadamard_path"):
        assert C.c_void_p.in_dll(ref, name).value not in (None, 0)
    # call through the reference's pointer: 64x64 SAD and a 16x16 forward transform
    g = np.random.default_rng(0)
    a, b = g.integers(0, 256, 64 * 64, dtype=np.uint8), g.integers(0, 256, 64 * 64, dtype=np.uint8)
    fp = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32)(after)
    assert fp(a.ctypes.data, 64, b.ctypes.data, 64, 64, 64) == int(np.abs(a.astype(np.int32) - b).sum())
    res = g.integers(-255, 256, 256).astype(np.int16)
    o1, o2 = np.zeros(256, np.int32), np.zeros(256, np.int32)
    ft = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint8, C.c_uint8)(C.c_void_p.in_dll(ref, "svt_av1_fwd_txfm2d_16x16").value)
    ft(res.ctypes.data, o1.ctypes.data, 16, 3, 8)
    ref.svt_av1_transform_two_d_16x16_c(C.c_void_p(res.ctypes.data), C.c_void_p(o2.ctypes.data), 16, 3, 8)
    assert np.array_equal(o1, o2)
    print("HOOK_OK", n)
''')


def test_rtcd_hook_overwrites_reference_pointers(be):
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libsvtref.so not available")
    lib = EMU_LIB if not be.is_gpu else os.path.join(ROOT, "svt-av1-psy_amd", "libsvtav1_hip.so")
    r = subprocess.run([sys.executable, "-c", CHILD, REF_LIB, lib], capture_output=True, text=True, timeout=300)
    assert "HOOK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
