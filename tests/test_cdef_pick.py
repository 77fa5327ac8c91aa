"""CDEF strength selection: svt_search_one_dual.  Oracle pinned against the reference's svt_search_one_dual_c (pointer-array
tables as pcs->mse_seg); HIP device form and RTCD single-call form compared with the oracle, including the greedy use of
joint_strength_search_dual (enc_cdef.c:697-726: add strengths one at a time)."""
import ctypes as C

import numpy as np
import pytest

from conftest import p, rng


def tables(g, n, flat=False):
    m0 = g.integers(0, 1 << 34, (n, 64), dtype=np.uint64)
    m1 = g.integers(0, 1 << 33, (n, 64), dtype=np.uint64)
    if flat:  # many ties: the first (j, k) in raster order must win
        m0[:], m1[:] = 1000, 500
        m0[:, 7] = 999
    return m0, m1


def ptr_tables(m0, m1):
    rows = [(C.POINTER(C.c_uint64) * len(m))(*[C.cast(m[i].ctypes.data, C.POINTER(C.c_uint64)) for i in range(len(m))]) for m in (m0, m1)]
    arr = (C.POINTER(C.POINTER(C.c_uint64)) * 2)(C.cast(rows[0], C.POINTER(C.POINTER(C.c_uint64))), C.cast(rows[1], C.POINTER(C.POINTER(C.c_uint64))))
    return arr, rows


CASES = [(510, 0, 64, False), (37, 0, 64, False), (100, 0, 16, False), (64, 8, 40, False), (12, 0, 64, True), (0, 0, 64, False)]


def test_search_one_dual_oracle_vs_reference(oracle, ref):
    g = rng(31)
    oracle.oracle_search_one_dual.restype = C.c_uint64
    ref.svt_search_one_dual_c.restype = C.c_uint64
    for (n, s, e, flat) in CASES:
        m0, m1 = tables(g, max(n, 1), flat)
        arr, keep = ptr_tables(m0, m1)
        la, lb = np.zeros(9, np.int32), np.zeros(9, np.int32)
        ra, rb = np.zeros(9, np.int32), np.zeros(9, np.int32)
        for nb in range(4):
            a = oracle.oracle_search_one_dual(p(la), p(lb), nb, p(m0), p(m1), n, s, e)
            b = ref.svt_search_one_dual_c(p(ra), p(rb), nb, arr, n, s, e)
            assert a == b and np.array_equal(la, ra) and np.array_equal(lb, rb), (n, s, e, nb)


def test_search_one_dual_hip(be, oracle):
    g = rng(32)
    oracle.oracle_search_one_dual.restype = C.c_uint64
    for (n, s, e, flat) in (CASES if be.is_gpu else CASES[1:]):
        m0, m1 = tables(g, max(n, 1), flat)
        arr, keep = ptr_tables(m0, m1)
        la, lb = np.zeros(9, np.int32), np.zeros(9, np.int32)
        ha, hb = np.zeros(9, np.int32), np.zeros(9, np.int32)
        d0, d1 = be.dev(m0), be.dev(m1)
        dla, dlb = be.dev(np.zeros(65, np.int32)), be.dev(np.zeros(65, np.int32))
        dbest, ws = be.empty(1, np.uint64), be.empty(4096 + max(n, 1), np.uint64)
        for nb in range(3):
            want = oracle.oracle_search_one_dual(p(la), p(lb), nb, p(m0), p(m1), n, s, e)
            got = be.lib.svt_search_one_dual_hip(p(ha), p(hb), nb, arr, n, s, e)  # RTCD form, host pointer arrays
            assert got == want and np.array_equal(la, ha) and np.array_equal(lb, hb), (n, s, e, nb)
            be.lib.svt_hip_cdef_search_one_dual(be.ptr(d0), be.ptr(d1), be.ptr(dla), be.ptr(dlb), nb, n, s, e, be.ptr(dbest), be.ptr(ws), be.stream)  # device-resident form
            assert int(be.host(dbest)[0]) == want
            assert np.array_equal(be.host(dla)[:nb + 1], la[:nb + 1]) and np.array_equal(be.host(dlb)[:nb + 1], lb[:nb + 1])


def test_joint_strength_search_and_assignment(be, oracle):
    """joint_strength_search_dual + the per-filter-block assignment of finish_cdef_search, device-resident on the search tables."""
    g = rng(33)
    oracle.oracle_joint_strength_search.restype = C.c_uint64
    for (n, s, e, nb) in ([(510, 0, 64, 8), (77, 0, 16, 4), (40, 4, 20, 2), (9, 0, 64, 1)] if be.is_gpu else [(40, 0, 16, 2), (9, 0, 8, 1)]):
        m0, m1 = tables(g, n)
        la, lb = np.zeros(9, np.int32), np.zeros(9, np.int32)
        want = oracle.oracle_joint_strength_search(p(la), p(lb), nb, p(m0), p(m1), n, s, e)
        wgi = np.zeros(n, np.int8)
        oracle.oracle_assign_fb_strengths(p(m0), p(m1), p(la), p(lb), nb, n, p(wgi))
        d0, d1 = be.dev(m0), be.dev(m1)
        dla, dlb = be.dev(np.zeros(65, np.int32)), be.dev(np.zeros(65, np.int32))
        dbest, ws, dgi = be.empty(1, np.uint64), be.empty(4096 + max(n, 1), np.uint64), be.empty(n, np.int8)
        be.lib.svt_hip_cdef_joint_strength_search(be.ptr(d0), be.ptr(d1), be.ptr(dla), be.ptr(dlb), nb, n, s, e, be.ptr(dbest), be.ptr(ws), be.stream)
        be.lib.svt_hip_cdef_assign_fb_strengths(be.ptr(d0), be.ptr(d1), be.ptr(dla), be.ptr(dlb), nb, n, be.ptr(dgi), be.stream)
        assert int(be.host(dbest)[0]) == want, (n, s, e, nb)
        assert np.array_equal(be.host(dla)[:nb], la[:nb]) and np.array_equal(be.host(dlb)[:nb], lb[:nb]), (n, s, e, nb)
        assert np.array_equal(be.host(dgi), wgi), (n, s, e, nb)
