"""svt_pme_sad_loop_kernel: the oracle restatement is pinned against the reference's svt_pme_sad_loop_kernel_c (compiled in
place, oracle/_ref), and the HIP entry point is compared with the oracle.  Cases follow test/SadTest.cc PmeSadLoopTest (:1595): random /
extreme planes, every MV_COST_TYPE, sparse search steps, partial 8-position groups, improving and non-improving incumbents."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_pkg, p, rng

CASES = [(16, 16, 24, 9, 1), (8, 8, 8, 8, 1), (32, 16, 21, 6, 2), (64, 64, 17, 5, 4), (4, 8, 40, 3, 8), (12, 10, 9, 7, 1), (16, 16, 7, 4, 1), (128, 64, 8, 2, 1)]


def make_case(g, bw, bh, aw, ah, kind):
    rs = bw + aw + 11
    src = g.integers(0, 256, (bh, bw + 3), dtype=np.uint8)
    ref = g.integers(0, 256, (bh + ah + 2, rs), dtype=np.uint8)
    if kind == 1:
        src[:], ref[:] = 255, 0
    elif kind == 2:  # a planted exact match away from the origin
        y0, x0 = min(ah - 1, 2), min(max(aw - 8, 0), 5)
        ref[y0:y0 + bh, x0:x0 + bw] = src[:, :bw]
    return src, ref, rs


def tables(g):
    t = [g.integers(0, 4000, 2 * 16384 + 1).astype(np.int32) for _ in range(2)]
    j = g.integers(0, 900, 4).astype(np.int32)
    return t, j


def run_oracle(oracle, src, ref, rs, bw, bh, aw, ah, step, best0, start, mv, refmv, ctype, t, j, epb):
    bc, bx, by = C.c_uint32(best0), C.c_int16(-7), C.c_int16(-9)
    oracle.oracle_pme_sad_loop(p(src), src.shape[1], p(ref), rs, bh, bw, C.byref(bc), C.byref(bx), C.byref(by), start[0], start[1], aw, ah, step, mv[0], mv[1],
                               refmv[0], refmv[1], ctype, p(j), C.c_void_p(t[0].ctypes.data + 16384 * 4), C.c_void_p(t[1].ctypes.data + 16384 * 4), epb)
    return bc.value, bx.value, by.value


def cost_params(pkg, refmv, ctype, t, j, epb):
    mvs = pkg.Mv(refmv[0], refmv[1])
    M = pkg.MvCostParams()
    M.ref_mv = C.pointer(mvs)
    M.mv_cost_type = ctype
    M.mvjcost = C.cast(j.ctypes.data, C.POINTER(C.c_int))
    M.mvcost[0] = C.cast(t[0].ctypes.data + 16384 * 4, C.POINTER(C.c_int))
    M.mvcost[1] = C.cast(t[1].ctypes.data + 16384 * 4, C.POINTER(C.c_int))
    M.error_per_bit = epb
    return M, mvs


def test_pme_oracle_vs_reference(oracle, ref):
    pkg = load_pkg()
    g = rng(77)
    f = ref.svt_pme_sad_loop_kernel_c
    f.restype = None
    for ci, (bw, bh, aw, ah, step) in enumerate(CASES):
        for ctype in range(6):
            for kind in range(3):
                src, rf, rs = make_case(g, bw, bh, aw, ah, kind)
                t, j = tables(g)
                start, mv, refmv = (int(g.integers(-20, 20)), int(g.integers(-20, 20))), (int(g.integers(-300, 300)), int(g.integers(-300, 300))), (int(g.integers(-200, 200)), int(g.integers(-200, 200)))
                epb = int(g.integers(1, 300))
                best0 = [0xffffffff, 40, 5000][kind]
                want = run_oracle(oracle, src, rf, rs, bw, bh, aw, ah, step, best0, start, mv, refmv, ctype, t, j, epb)
                M, keep = cost_params(pkg, refmv, ctype, t, j, epb)
                bc, bx, by = C.c_uint32(best0), C.c_int16(-7), C.c_int16(-9)
                f(C.byref(M), p(src), C.c_uint32(src.shape[1]), p(rf), C.c_uint32(rs), C.c_uint32(bh), C.c_uint32(bw), C.byref(bc), C.byref(bx), C.byref(by),
                  C.c_int16(start[0]), C.c_int16(start[1]), C.c_int16(aw), C.c_int16(ah), C.c_int16(step), C.c_int16(mv[0]), C.c_int16(mv[1]))
                assert (bc.value, bx.value, by.value) == want, (ci, ctype, kind)


def test_pme_sad_loop_single_call(be, oracle):
    g = rng(78)
    cases = CASES if be.is_gpu else CASES[:5]
    for ci, (bw, bh, aw, ah, step) in enumerate(cases):
        for ctype in range(6):
            for kind in range(3):
                if not be.is_gpu and (ci + ctype + kind) % 3:
                    continue
                src, rf, rs = make_case(g, bw, bh, aw, ah, kind)
                t, j = tables(g)
                start, mv, refmv = (int(g.integers(-20, 20)), int(g.integers(-20, 20))), (int(g.integers(-300, 300)), int(g.integers(-300, 300))), (int(g.integers(-200, 200)), int(g.integers(-200, 200)))
                epb = int(g.integers(1, 300))
                best0 = [0xffffffff, 40, 5000][kind]
                want = run_oracle(oracle, src, rf, rs, bw, bh, aw, ah, step, best0, start, mv, refmv, ctype, t, j, epb)
                M, keep = cost_params(be.pkg, refmv, ctype, t, j, epb)
                bc, bx, by = C.c_uint32(best0), C.c_int16(-7), C.c_int16(-9)
                be.lib.svt_pme_sad_loop_kernel_hip(C.byref(M), p(src), src.shape[1], p(rf), rs, bh, bw, C.byref(bc), C.byref(bx), C.byref(by), start[0], start[1], aw, ah,
                                                   step, mv[0], mv[1])
                assert (bc.value, bx.value, by.value) == want, (ci, ctype, kind)
