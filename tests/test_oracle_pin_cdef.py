"""Pins oracle/oracle_cdef.c against the real reference: svt_aom_cdef_find_dir_c, svt_cdef_filter_block_c,
svt_cdef_filter_fb (the per-64x64 driver, RTCD forced to the C variants), svt_aom_compute_cdef_dist_c / _8bit_c."""
import ctypes as C

import numpy as np
import pytest

from conftest import p, rng

BSTRIDE, VERY_LARGE = 144, 0x7f7f
BLOCK = {(4, 4): 0, (4, 8): 1, (8, 4): 2, (8, 8): 3}  # BlockSize enum: BLOCK_4X4, 4X8, 8X4, 8X8


def make_tile(g, bd, edges):
    """(64+6) x 144 u16 tile with `in` at (3, 8); optional VERY_LARGE borders like a frame edge."""
    t = g.integers(0, 1 << bd, (70, BSTRIDE)).astype(np.uint16)
    # smooth-ish content so directions are meaningful
    yy, xx = np.mgrid[0:70, 0:BSTRIDE]
    t = ((t >> 2) + (((xx * 3 + yy * 5) & 63) << (bd - 6))).clip(0, (1 << bd) - 1).astype(np.uint16)
    if edges & 1: t[:3, :] = VERY_LARGE
    if edges & 2: t[:, :8] = VERY_LARGE
    if edges & 4: t[67:, :] = VERY_LARGE
    if edges & 8: t[:, 72:] = VERY_LARGE
    return t


def in_ptr(t, off=0):
    return C.c_void_p(t.ctypes.data + 2 * (3 * BSTRIDE + 8 + off))


def test_find_dir_and_filter_block_vs_reference(oracle, ref):
    g = rng(1)
    for bd in (8, 10, 12):
        cs = bd - 8
        for edges in (0, 1, 2, 5, 15):
            t = make_tile(g, bd, edges)
            for (by, bx) in ((0, 0), (3, 4), (7, 7), (0, 7)):
                off = by * 8 * BSTRIDE + bx * 8
                v0, v1 = C.c_int32(0), C.c_int32(0)
                d0 = oracle.oracle_cdef_find_dir(in_ptr(t, off), BSTRIDE, C.byref(v0), cs)
                d1 = ref.svt_aom_cdef_find_dir_c(in_ptr(t, off), BSTRIDE, C.byref(v1), cs)
                assert (d0 & 255, v0.value) == (d1 & 255, v1.value)
                for (bw, bh) in BLOCK:
                    for pri in (0, 1, 4, 15):
                        for sec in (0, 1, 2, 4):
                            for damp in (3, 4, 6):
                                for sub in (1, 2):
                                    for dirn in (0, 3, 7):
                                        o0 = np.zeros(8 * 8, np.uint16); o1 = o0.copy()
                                        oracle.oracle_cdef_filter_block(None, p(o0), 8, in_ptr(t, off), pri << cs, sec << cs, dirn, damp + cs, damp + cs - 1, bw, bh, cs, sub)
                                        ref.svt_cdef_filter_block_c(None, p(o1), 8, in_ptr(t, off), pri << cs, sec << cs, dirn, damp + cs, damp + cs - 1, BLOCK[(bw, bh)], cs, sub)
                                        assert np.array_equal(o0, o1), (bd, edges, by, bx, bw, bh, pri, sec, damp, sub, dirn)
                                        if bd == 8:
                                            b0 = np.zeros(64, np.uint8); b1 = b0.copy()
                                            oracle.oracle_cdef_filter_block(p(b0), None, 8, in_ptr(t, off), pri, sec, dirn, damp, damp, bw, bh, 0, sub)
                                            ref.svt_cdef_filter_block_c(p(b1), None, 8, in_ptr(t, off), pri, sec, dirn, damp, damp, BLOCK[(bw, bh)], 0, sub)
                                            assert np.array_equal(b0, b1)


def test_filter_fb_and_dist_vs_reference(oracle, ref):
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))  # every pointer -> *_c
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    ref.svt_aom_compute_cdef_dist_c.restype = C.c_uint64
    ref.svt_aom_compute_cdef_dist_8bit_c.restype = C.c_uint64
    oracle.oracle_cdef_dist.restype = C.c_uint64
    g = rng(2)
    for bd in (8, 10):
        cs = bd - 8
        for (xdec, ydec, pli) in ((0, 0, 0), (1, 1, 1), (1, 0, 1), (0, 1, 2)):
            for edges in (0, 3, 12):
                t = make_tile(g, bd, edges)
                keep = g.random(64) < 0.7
                units = [(by, bx) for by in range(8) for bx in range(8) if keep[by * 8 + bx]]
                dl = np.array(units, np.uint8).reshape(-1)
                dl_ref = np.array(units, np.uint8).reshape(-1)  # CdefList {uint8 by, bx}
                for (lvl, sec) in ((0, 0), (3, 0), (0, 2), (7, 4), (15, 1)):
                    for dstride in (0, 80):
                        for sub in ((1, 2) if dstride == 0 else (1,)):
                            outs = []
                            dirs = []
                            for which in (0, 1):
                                dir_ = np.zeros((16, 16), np.uint8); var = np.zeros((16, 16), np.int32)
                                if pli:
                                    dir_[:8, :8] = np.arange(64).reshape(8, 8) % 8
                                    var[:8, :8] = 100 * np.arange(64).reshape(8, 8)
                                di = C.c_int32(0)
                                o = np.full(80 * 80, 7, np.uint16) if bd > 8 else np.full(80 * 80, 7, np.uint8)
                                a8, a16 = (None, p(o)) if bd > 8 else (p(o), None)
                                f = oracle.oracle_cdef_filter_fb if which == 0 else ref.svt_cdef_filter_fb
                                f(a8, a16, dstride, in_ptr(t), xdec, ydec, p(dir_), C.byref(di), p(var), pli, p(dl if which == 0 else dl_ref), len(units), lvl, sec, 3, 3, cs, sub)
                                outs.append(o); dirs.append((dir_.copy(), var.copy(), di.value))
                            assert np.array_equal(outs[0], outs[1]), (bd, xdec, ydec, pli, edges, lvl, sec, dstride, sub)
                            assert np.array_equal(dirs[0][0], dirs[1][0]) and np.array_equal(dirs[0][1], dirs[1][1]) and dirs[0][2] == dirs[1][2]
                            if dstride == 0:
                                bw, bh = 8 >> xdec, 8 >> ydec
                                plane = g.integers(0, 1 << bd, 64 * 96).astype(np.uint16 if bd > 8 else np.uint8)
                                d0 = oracle.oracle_cdef_dist(p(plane), 96, p(outs[0]), p(dl), len(units), bw, bh, cs, pli, sub, 1 if bd > 8 else 0)
                                fr = ref.svt_aom_compute_cdef_dist_c if bd > 8 else ref.svt_aom_compute_cdef_dist_8bit_c
                                d1 = fr(p(plane), 96, p(outs[1]), p(dl_ref), len(units), BLOCK[(bw, bh)], cs, pli, sub)
                                assert d0 == d1, (bd, xdec, ydec, pli, lvl, sec, sub)
