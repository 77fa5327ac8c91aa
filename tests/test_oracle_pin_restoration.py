"""Pins oracle/oracle_restoration.c against the real reference: svt_av1_wiener_convolve_add_src_c (+highbd),
svt_av1_selfguided_restoration_c, svt_apply_selfguided_restoration_c, the x_by_xplus1 / one_by_x / sgr_params tables,
and the stripe / unit driver svt_av1_loop_restoration_filter_unit (RTCD forced to C)."""
import ctypes as C

import numpy as np
import pytest

from conftest import p, rng


class ConvolveParams(C.Structure):  # definitions.h:572-585
    _fields_ = [("ref", C.c_int32), ("do_average", C.c_int32), ("dst", C.c_void_p), ("dst_stride", C.c_int32), ("round_0", C.c_int32),
                ("round_1", C.c_int32), ("plane", C.c_int32), ("is_compound", C.c_int32), ("use_jnt_comp_avg", C.c_int32), ("fwd_offset", C.c_int32),
                ("bck_offset", C.c_int32), ("use_dist_wtd_comp_avg", C.c_int32)]


def conv_params(bd):
    r0, r1 = 3, 11
    rng_ = bd + 7 - r0 + 2
    if rng_ > 16:
        r0 += rng_ - 16
        r1 -= rng_ - 16
    cp = ConvolveParams()
    cp.round_0, cp.round_1 = r0, r1
    return cp


def wiener_taps(g, extreme=0):
    """Legal AV1 Wiener taps (wiener_convolve_test.cc:143-172 ranges): 3 free taps, symmetric, centre = -2 * sum."""
    lim = [(-5, 10), (-23, 8), (-17, 46)]
    f = [lim[i][extreme - 1] if extreme else int(g.integers(lim[i][0], lim[i][1] + 1)) for i in range(3)]
    k = np.zeros(64, np.int16)  # generously aligned so that the reference's 256-byte filter-base trick sees offset 0
    k[:8] = [f[0], f[1], f[2], -2 * sum(f), f[2], f[1], f[0], 0]
    return k


def aligned_i16(vals):
    raw = np.zeros(256, np.int16)
    off = (-raw.ctypes.data % 256) // 2
    a = raw[off:off + 8]
    a[:] = vals
    return a


def byteptr(a, off_elems=0):  # CONVERT_TO_BYTEPTR
    return C.c_void_p((a.ctypes.data + 2 * off_elems) >> 1)


def test_sgr_tables(oracle, ref):
    xb = (C.c_int32 * 256).in_dll(ref, "svt_aom_eb_x_by_xplus1")
    ob = (C.c_int32 * 25).in_dll(ref, "svt_aom_eb_one_by_x")
    assert [oracle.oracle_x_by_xplus1(z) for z in range(256)] == list(xb)
    assert [oracle.oracle_one_by_x(n) for n in range(1, 26)] == list(ob)

    class Sgr(C.Structure):
        _fields_ = [("r", C.c_int32 * 2), ("s", C.c_int32 * 2)]
    tab = (Sgr * 16).in_dll(ref, "svt_aom_eb_sgr_params")
    for i in range(16):
        assert [oracle.oracle_sgr_r(i, 0), oracle.oracle_sgr_r(i, 1), oracle.oracle_sgr_s(i, 0), oracle.oracle_sgr_s(i, 1)] == list(tab[i].r) + list(tab[i].s)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_wiener_and_sgr_kernels_vs_reference(oracle, ref, bd):
    g = rng(bd)
    hb = bd > 8
    dt = np.uint16 if hb else np.uint8
    S = 96
    for (w, h) in ((64, 64), (16, 8), (32, 56), (48, 20), (64, 17)):
        src = g.integers(0, 1 << bd, (h + 12, S)).astype(dt)
        if w == 16:
            src[:] = (1 << bd) - 1
        org = 6 * S + 8
        for ext in (0, 1, 2):
            fx, fy = aligned_i16(wiener_taps(g, ext)[:8]), aligned_i16(wiener_taps(g, ext)[:8])
            d0, d1 = np.zeros((h, S), dt), np.zeros((h, S), dt)
            oracle.oracle_wiener_convolve_add_src(C.c_void_p(src.ctypes.data + org * src.itemsize), S, p(d0), S, p(fx), p(fy), w, h, bd, int(hb))
            cp = conv_params(bd)
            if hb:
                ref.svt_av1_highbd_wiener_convolve_add_src_c(byteptr(src, org), C.c_ssize_t(S), byteptr(d1), C.c_ssize_t(S), p(fx), p(fy), w, h, C.byref(cp), bd)
            else:
                ref.svt_av1_wiener_convolve_add_src_c(C.c_void_p(src.ctypes.data + org), C.c_ssize_t(S), p(d1), C.c_ssize_t(S), p(fx), p(fy), w, h, C.byref(cp))
            assert np.array_equal(d0[:, :w], d1[:, :w]), ("wiener", bd, w, h, ext)
        tmp = np.zeros(2 * 161 * 161 * 4 + 1024, np.int32)  # SGRPROJ_TMPBUF_SIZE worth of int32
        for idx in range(16):
            a0, a1 = np.full((h, w), -7, np.int32), np.full((h, w), -7, np.int32)
            b0, b1 = a0.copy(), a1.copy()
            sp = C.c_void_p(src.ctypes.data + org * src.itemsize)
            oracle.oracle_selfguided_restoration(sp, w, h, S, p(a0), p(a1), w, idx, bd, int(hb))
            ref.svt_av1_selfguided_restoration_c(byteptr(src, org) if hb else sp, w, h, S, p(b0), p(b1), w, idx, bd, int(hb))
            assert np.array_equal(a0, b0) and np.array_equal(a1, b1), ("sgr", bd, w, h, idx)
            xqd = np.array([int(g.integers(-96, 32)), int(g.integers(-32, 96))], np.int32)
            d0, d1 = np.zeros((h, S), dt), np.zeros((h, S), dt)
            oracle.oracle_apply_selfguided_restoration(sp, w, h, S, idx, p(xqd), p(d0), S, bd, int(hb))
            ref.svt_apply_selfguided_restoration_c(byteptr(src, org) if hb else sp, w, h, S, idx, p(xqd), byteptr(d1) if hb else p(d1), S, p(tmp), bd, int(hb))
            assert np.array_equal(d0[:, :w], d1[:, :w]), ("sgr apply", bd, w, h, idx)


class Limits(C.Structure):
    _fields_ = [("h_start", C.c_int32), ("h_end", C.c_int32), ("v_start", C.c_int32), ("v_end", C.c_int32)]


class Rect(C.Structure):
    _fields_ = [("left", C.c_int32), ("top", C.c_int32), ("right", C.c_int32), ("bottom", C.c_int32)]


class Rsb(C.Structure):
    _fields_ = [("above", C.c_void_p), ("below", C.c_void_p), ("stride", C.c_int32), ("size", C.c_int32)]


class Rui(C.Structure):  # RestorationUnitInfo: the 16-byte aligned WienerInfo pushes the kernels to offset 16
    _fields_ = [("rtype", C.c_int32), ("pad", C.c_int32 * 3), ("vfilter", C.c_int16 * 8), ("hfilter", C.c_int16 * 8), ("ep", C.c_int32), ("xqd", C.c_int32 * 2),
                ("tail", C.c_int32)]


def unit_grid(w, h, us):
    return max((h + us // 2) // us, 1), max((w + us // 2) // us, 1)


def make_units(g, nvu, nhu, pkg_dtype):
    u = np.zeros(nvu * nhu, dtype=pkg_dtype)
    for i in range(len(u)):
        t = [1, 2, 0, 2, 1][i % 5]
        fy, fx = wiener_taps(g)[:8], wiener_taps(g)[:8]
        u[i] = (t, fy, fx, int(g.integers(0, 16)), (int(g.integers(-96, 32)), int(g.integers(-32, 96))))
    return u


@pytest.mark.parametrize("cfg", [(8, 0, 200, 150, 64), (10, 0, 136, 200, 128), (10, 1, 100, 68, 32), (8, 1, 70, 90, 64)])
def test_lr_frame_driver_vs_reference_filter_unit(oracle, ref, cfg):
    """oracle_lr_filter_frame == svt_av1_loop_restoration_filter_unit applied to every restoration unit of a plane."""
    from conftest import load_pkg
    pkg = load_pkg()
    bd, ss, w, h, us = cfg
    g = rng(sum(cfg))
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    hb = bd > 8
    dt = np.uint16 if hb else np.uint8
    off, sh = 8 >> ss, 64 >> ss
    nstripes = (h + off + sh - 1) // sh
    plane = g.integers(0, 1 << bd, (h, w)).astype(dt)
    above = g.integers(0, 1 << bd, (2 * nstripes, w)).astype(dt)
    below = g.integers(0, 1 << bd, (2 * nstripes, w)).astype(dt)
    nvu, nhu = unit_grid(w, h, us)
    units = make_units(g, nvu, nhu, pkg.LrUnit)
    out = np.zeros((h, w), dt)
    oracle.oracle_lr_filter_frame(p(plane), w, p(above), p(below), w, p(out), w, w, h, ss, us, p(units), bd, int(hb))
    # reference: frame with a replicated 8-pixel border (svt_extend_frame), boundary buffers with the 4-pixel horizontal extension
    B = 8
    ext = np.pad(plane, B, mode="edge")
    es = ext.shape[1]
    ab = np.ascontiguousarray(np.pad(above, ((0, 0), (4, 4)), mode="edge"))
    bl = np.ascontiguousarray(np.pad(below, ((0, 0), (4, 4)), mode="edge"))
    bs = ab.shape[1]
    # buffer column c holds frame column c - RESTORATION_EXTRA_HORZ (restoration.c:296-300: buf_x0_off = h_start, data_x0 = h_start - 4)
    rsb = Rsb(ab.ctypes.data, bl.ctypes.data, bs, 0)
    rlbs = np.zeros(1 << 16, np.uint16)
    tmp = np.zeros(2 * 161 * 161 * 4 + 1024, np.int32)
    dst = np.zeros_like(ext)
    rect = Rect(0, 0, w, h)
    for ur in range(nvu):
        for uc in range(nhu):
            v0 = max(0, ur * us - off)
            v1 = h if ur == nvu - 1 else (ur + 1) * us - off
            h0, h1 = uc * us, (w if uc == nhu - 1 else (uc + 1) * us)
            u = units[ur * nhu + uc]
            rui = Rui()
            rui.rtype = int(u["rtype"])
            rui.vfilter[:] = [int(x) for x in u["vfilter"]]
            rui.hfilter[:] = [int(x) for x in u["hfilter"]]
            rui.ep = int(u["ep"])
            rui.xqd[:] = [int(x) for x in u["xqd"]]
            lim = Limits(h0, h1, v0, v1)
            d_org = ext.ctypes.data + (B * es + B) * ext.itemsize
            o_org = dst.ctypes.data + (B * es + B) * ext.itemsize
            ref.svt_av1_loop_restoration_filter_unit(C.c_uint8(1), C.byref(lim), C.byref(rui), C.byref(rsb), p(rlbs), C.byref(rect), 0, ss, ss, int(hb), bd,
                                                     C.c_void_p(d_org >> 1) if hb else C.c_void_p(d_org), es, C.c_void_p(o_org >> 1) if hb else C.c_void_p(o_org), es,
                                                     p(tmp), 0)
    got = dst[B:B + h, B:B + w]
    assert np.array_equal(out, got), np.argwhere(out != got)[:8]
