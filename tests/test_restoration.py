"""Loop restoration: HIP path vs the oracle (pinned against the reference's `_c` filters and
svt_av1_loop_restoration_filter_unit).  Frame level: Wiener / self-guided / none units mixed over a plane, stripe
boundaries from saved deblocked lines; 4K 10-bit on the GPU, small planes on the CPU interpreter."""
import ctypes as C

import numpy as np
import pytest

from conftest import p, rng
from test_oracle_pin_restoration import aligned_i16, byteptr, make_units, unit_grid, wiener_taps


@pytest.mark.parametrize("cfg", [(8, 0, 200, 150, 64), (10, 0, 136, 200, 128), (10, 1, 100, 68, 32), (8, 1, 70, 90, 64), (10, 0, 3840, 2160, 256), (10, 1, 1920, 1080, 128)])
def test_lr_filter_frame(be, oracle, cfg):
    bd, ss, w, h, us = cfg
    if not be.is_gpu and w * h > 100000:
        pytest.skip("full-size frames run on the GPU only")
    g = rng(sum(cfg))
    hb = bd > 8
    dt = np.uint16 if hb else np.uint8
    off, sh = 8 >> ss, 64 >> ss
    nstripes = (h + off + sh - 1) // sh
    yy, xx = np.mgrid[0:h, 0:w]
    plane = np.clip(((xx * 3 + yy * 2) % (1 << bd)) // 2 + g.integers(0, 1 << (bd - 2), (h, w)), 0, (1 << bd) - 1).astype(dt)
    above = g.integers(0, 1 << bd, (2 * nstripes, w)).astype(dt)
    below = g.integers(0, 1 << bd, (2 * nstripes, w)).astype(dt)
    nvu, nhu = unit_grid(w, h, us)
    units = make_units(g, nvu, nhu, be.pkg.LrUnit)
    want = np.zeros((h, w), dt)
    oracle.oracle_lr_filter_frame(p(plane), w, p(above), p(below), w, p(want), w, w, h, ss, us, p(units), bd, int(hb))
    d_pl, d_ab, d_bl, d_un = be.dev(plane), be.dev(above), be.dev(below), be.dev(units)
    d_out = be.empty((h, w), dt)
    P = be.pkg.LrParams(be.ptr(d_pl), be.ptr(d_ab), be.ptr(d_bl), be.ptr(d_out), w, w, w, w, h, us, ss, ss, int(hb), bd, be.ptr(d_un))
    be.lib.svt_hip_lr_filter_frame(C.byref(P), be.stream)
    got = be.host(d_out)
    assert np.array_equal(got, want), np.argwhere(got != want)[:8]


def test_lr_filter_frame_in_stripe_ranges(be, oracle):
    """A picture split over several GPUs: svt_hip_lr_filter_frame_stripes over ranges of 64-row stripes, every range reading its halos and the saved
    boundary lines from the full inputs; together == the whole-frame result."""
    bd, ss, us = 10, 0, 64
    w, h = (1920, 1080) if be.is_gpu else (136, 328)
    g = rng(808)
    dt = np.uint16
    nstripes = (h + 8 + 63) // 64
    yy, xx = np.mgrid[0:h, 0:w]
    plane = np.clip(((xx * 3 + yy * 2) % (1 << bd)) // 2 + g.integers(0, 1 << (bd - 2), (h, w)), 0, (1 << bd) - 1).astype(dt)
    above, below = g.integers(0, 1 << bd, (2 * nstripes, w)).astype(dt), g.integers(0, 1 << bd, (2 * nstripes, w)).astype(dt)
    nvu, nhu = unit_grid(w, h, us)
    units = make_units(g, nvu, nhu, be.pkg.LrUnit)
    want = np.zeros((h, w), dt)
    oracle.oracle_lr_filter_frame(p(plane), w, p(above), p(below), w, p(want), w, w, h, ss, us, p(units), bd, 1)
    d_pl, d_ab, d_bl, d_un = be.dev(plane), be.dev(above), be.dev(below), be.dev(units)
    d_out = be.empty((h, w), dt)
    P = be.pkg.LrParams(be.ptr(d_pl), be.ptr(d_ab), be.ptr(d_bl), be.ptr(d_out), w, w, w, w, h, us, ss, ss, 1, bd, be.ptr(d_un))
    cuts = [0, nstripes // 3, nstripes // 3 + 1, nstripes]
    for a, b in reversed(list(zip(cuts[:-1], cuts[1:]))):
        be.lib.svt_hip_lr_filter_frame_stripes(C.byref(P), a, b, be.stream)
    got = be.host(d_out)
    assert np.array_equal(got, want), np.argwhere(got != want)[:8]


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_restoration_single_call_symbols(be, oracle, bd):
    """svt_av1_[highbd_]wiener_convolve_add_src, svt_av1_selfguided_restoration, svt_apply_selfguided_restoration
    (wiener_convolve_test.cc, selfguided_filter_test.cc shapes)."""
    g = rng(30 + bd)
    hb = bd > 8
    dt = np.uint16 if hb else np.uint8
    S = 160
    sizes = [(64, 64), (16, 8), (32, 56), (96, 96), (128, 24)] if be.is_gpu else [(16, 8), (32, 56), (80, 20)]
    for (w, h) in sizes:
        src = g.integers(0, 1 << bd, (h + 12, S)).astype(dt)
        org = 6 * S + 8
        sp = C.c_void_p(src.ctypes.data + org * src.itemsize)
        fx, fy = aligned_i16(wiener_taps(g)[:8]), aligned_i16(wiener_taps(g)[:8])
        d0, d1 = np.zeros((h, S), dt), np.zeros((h, S), dt)
        oracle.oracle_wiener_convolve_add_src(sp, S, p(d0), S, p(fx), p(fy), w, h, bd, int(hb))
        if hb:
            be.lib.svt_av1_highbd_wiener_convolve_add_src_hip(byteptr(src, org), S, byteptr(d1), S, p(fx), p(fy), w, h, None, bd)
        else:
            be.lib.svt_av1_wiener_convolve_add_src_hip(sp, S, p(d1), S, p(fx), p(fy), w, h, None)
        assert np.array_equal(d0[:, :w], d1[:, :w]), ("wiener", bd, w, h)
        for idx in ((0, 7, 10, 13, 14, 15) if not be.is_gpu else range(16)):
            a0, a1 = np.full((h, w), -7, np.int32), np.full((h, w), -7, np.int32)
            b0, b1 = a0.copy(), a1.copy()
            oracle.oracle_selfguided_restoration(sp, w, h, S, p(a0), p(a1), w, idx, bd, int(hb))
            be.lib.svt_av1_selfguided_restoration_hip(byteptr(src, org) if hb else sp, w, h, S, p(b0), p(b1), w, idx, bd, int(hb))
            assert np.array_equal(a0, b0) and np.array_equal(a1, b1), ("sgr", bd, w, h, idx)
            xqd = np.array([int(g.integers(-96, 32)), int(g.integers(-32, 96))], np.int32)
            d0, d1 = np.zeros((h, S), dt), np.zeros((h, S), dt)
            oracle.oracle_apply_selfguided_restoration(sp, w, h, S, idx, p(xqd), p(d0), S, bd, int(hb))
            be.lib.svt_apply_selfguided_restoration_hip(byteptr(src, org) if hb else sp, w, h, S, idx, p(xqd), byteptr(d1) if hb else p(d1), S, None, bd, int(hb))
            assert np.array_equal(d0[:, :w], d1[:, :w]), ("sgr apply", bd, w, h, idx)
