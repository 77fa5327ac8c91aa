"""Pins oracle/oracle_sad.c against (a) the real reference built from /root/reference (oracle/_ref, when present) and
(b) golden vectors under tests/golden/ produced from (a) by tools/gen_golden.py.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, p, rng
from test_sad import BLOCK_SIZES, make_planes


def fptr(lib, name):
    return C.cast(getattr(lib, name), C.c_void_p)


def ref_me(oracle, ref, src, ss, refp, rs, xo, yo, w, h, sub):
    bs, bm = np.zeros(85, np.uint32), np.zeros(85, np.uint32)
    oracle.oracle_drive_ref_me_search.argtypes = [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p, C.c_uint32] + [C.c_int] * 5 + [C.c_void_p] * 2
    oracle.oracle_drive_ref_me_search(fptr(ref, "svt_ext_all_sad_calculation_8x8_16x16_c"),
                                      fptr(ref, "svt_ext_eight_sad_calculation_32x32_64x64_c"),
                                      fptr(ref, "svt_ext_sad_calculation_8x8_16x16_c"),
                                      fptr(ref, "svt_ext_sad_calculation_32x32_64x64_c"), src, ss, refp, rs, xo, yo, w, h, sub, p(bs), p(bm))
    return bs, bm


def test_sad_nxm_and_loop_vs_reference(oracle, ref):
    g = rng()
    oracle.oracle_sad_nxm.restype = C.c_uint32
    ref.svt_nxm_sad_kernel_helper_c.restype = C.c_uint32
    for pattern in ["REF_MAX", "SRC_MAX", "RANDOM", "UNALIGN"]:
        for (w, h) in BLOCK_SIZES:
            src, ss, r, rs = make_planes(pattern, g, 64, 128, ref_extra=48)
            assert oracle.oracle_sad_nxm(p(src), ss, p(r), rs, h, w) == ref.svt_nxm_sad_kernel_helper_c(p(src), ss, p(r), rs, h, w)
            for skip in (0, 1):
                for (aw, ah) in [(8, 15), (15, 6), (48, 24)]:
                    a = [C.c_uint64(0), C.c_int16(0), C.c_int16(0)]
                    b = [C.c_uint64(0), C.c_int16(0), C.c_int16(0)]
                    oracle.oracle_sad_loop(p(src), ss, p(r), rs, h, w, C.byref(a[0]), C.byref(a[1]), C.byref(a[2]), rs, skip, aw, ah)
                    ref.svt_sad_loop_kernel_c(p(src), ss, p(r), rs, h, w, C.byref(b[0]), C.byref(b[1]), C.byref(b[2]), rs, skip, aw, ah)
                    assert [v.value for v in a] == [v.value for v in b], (pattern, w, h, skip, aw, ah)


def test_me_search_vs_reference_driver(oracle, ref):
    """oracle_me_fullpel_search == the reference's ext_* `_c` kernels driven as open_loop_me_fullpel_search_sblock does."""
    g = rng(2)
    for pattern in ["RANDOM", "SRC_MAX", "REF_MAX"]:
        for (aw, ah) in [(16, 9), (15, 6), (8, 3), (21, 5), (3, 2)]:
            for sub in (0, 1):
                src, ss, r, rs = make_planes(pattern, g, 64, 96, ref_extra=80)
                bs, bm = np.zeros(85, np.uint32), np.zeros(85, np.uint32)
                oracle.oracle_me_fullpel_search(p(src), ss, p(r), rs, -7, -3, aw, ah, sub, p(bs), p(bm))
                ws, wm = ref_me(oracle, ref, p(src), ss, p(r), rs, -7, -3, aw, ah, sub)
                assert np.array_equal(bs, ws) and np.array_equal(bm, wm), (pattern, aw, ah, sub)


def test_sad_oracle_vs_golden(oracle):
    """Golden vectors (inputs + the reference's outputs) committed under tests/golden/sad.npz."""
    path = os.path.join(GOLDEN, "sad.npz")
    assert os.path.exists(path), "run tools/gen_golden.py in the build container"
    z = np.load(path)
    src, r = z["src"], z["ref"]
    ss, rs = int(z["src_stride"]), int(z["ref_stride"])
    oracle.oracle_sad_nxm.restype = C.c_uint32
    for i, (w, h) in enumerate(z["nxm_sizes"]):
        assert oracle.oracle_sad_nxm(p(src), ss, p(r), rs, int(h), int(w)) == z["nxm_out"][i]
    for i, (w, h, aw, ah, skip) in enumerate(z["loop_cfg"]):
        a = [C.c_uint64(0), C.c_int16(0), C.c_int16(0)]
        oracle.oracle_sad_loop(p(src), ss, p(r), rs, int(h), int(w), C.byref(a[0]), C.byref(a[1]), C.byref(a[2]), rs, int(skip), int(aw), int(ah))
        assert [v.value for v in a] == z["loop_out"][i].tolist()
    for i, (aw, ah, sub) in enumerate(z["me_cfg"]):
        bs, bm = np.zeros(85, np.uint32), np.zeros(85, np.uint32)
        oracle.oracle_me_fullpel_search(p(src), ss, p(r), rs, -5, -2, int(aw), int(ah), int(sub), p(bs), p(bm))
        assert np.array_equal(bs, z["me_sad"][i]) and np.array_equal(bm, z["me_mv"][i])
