"""CDEF: HIP path vs the oracle (which is pinned against the reference's svt_cdef_filter_fb + `_c`
kernels).  Frame-level apply and strength search over synthetic frames with blocking steps + noise (SURVEY 8d config 4:
4K 10-bit on the GPU; a small frame on the CPU interpreter), partial bottom/right filter blocks, 25 % skipped units."""
import ctypes as C

import numpy as np
import pytest

from conftest import p, rng
from test_oracle_pin_cdef import BLOCK, BSTRIDE, in_ptr, make_tile


def synth_plane(g, w, h, bd):
    yy, xx = np.mgrid[0:h, 0:w]
    base = ((xx * 2 + yy * 3) % (1 << bd)).astype(np.int32) // 2 + (((xx // 8 + yy // 8) % 5) << (bd - 5))  # gradient + 8x8 blocking steps
    return np.clip(base + g.integers(-(1 << (bd - 6)), (1 << (bd - 6)) + 1, (h, w)), 0, (1 << bd) - 1)


def run_frame(be, oracle, mode, recon, source, xdec, ydec, pli, bd, skip, pri, sec, dir_in, var_in, sub=1, damping=5, dev_mode=None):
    is16 = bd > 8
    dt = np.uint16 if is16 else np.uint8
    h, w = recon.shape
    rec, src = recon.astype(dt), source.astype(dt)
    bw, bh = 64 >> xdec, 64 >> ydec
    nhfb, nvfb = (w + bw - 1) // bw, (h + bh - 1) // bh
    nfb = nhfb * nvfb
    ncand = len(pri) if mode == 1 else 0
    # oracle
    o_out = rec.copy()
    o_dir, o_var = dir_in.copy(), var_in.copy()
    o_mse = np.zeros(max(nfb * max(ncand, 1), 1), np.uint64)
    oracle.oracle_cdef_frame(mode, p(rec), w, p(src), w, p(o_out), w, w, h, xdec, ydec, pli, int(is16), bd - 8, damping, damping, sub, p(skip),
                             p(pri), p(sec), ncand, p(o_dir), p(o_var), p(o_mse))
    # device
    d_rec, d_src, d_out = be.dev(rec), be.dev(src), be.dev(rec)
    d_skip, d_pri, d_sec = be.dev(skip), be.dev(pri), be.dev(sec)
    d_dir, d_var = be.dev(dir_in), be.dev(var_in)
    d_mse = be.empty(max(nfb * max(ncand, 1), 1), np.uint64)
    P = be.pkg.CdefParams(be.ptr(d_rec), be.ptr(d_src), be.ptr(d_out), w, w, w, w, h, xdec, ydec, pli, int(is16), bd - 8, damping, damping, sub, ncand,
                          be.ptr(d_skip), be.ptr(d_pri), be.ptr(d_sec), be.ptr(d_dir), be.ptr(d_var), be.ptr(d_mse))
    be.lib.svt_hip_cdef_frame(mode if dev_mode is None else dev_mode, C.byref(P), be.stream)
    g_out, g_dir, g_var, g_mse = be.host(d_out), be.host(d_dir), be.host(d_var), be.host(d_mse)
    if pli == 0:
        assert np.array_equal(g_dir, o_dir), np.nonzero(g_dir != o_dir)[0][:10]
        assert np.array_equal(g_var, o_var)
    if mode == 0:
        assert np.array_equal(g_out.reshape(h, w), o_out), np.argwhere(g_out.reshape(h, w) != o_out)[:6]
    else:
        assert np.array_equal(g_mse[:nfb * ncand], o_mse[:nfb * ncand]), np.nonzero(g_mse[:nfb * ncand] != o_mse[:nfb * ncand])[0][:10]
    return o_dir, o_var


@pytest.mark.parametrize("bd,damping", [(8, 3), (10, 6), (12, 4)])
def test_cdef_apply_single_strength_blocks_extreme_content(be, oracle, bd, damping):
    """Filter blocks with ONE non-zero strength take the 4- / 8-tap forms that leave out the reference's clamp to the taps' [min, max] (it cannot bind
    there).  Adversarial content for that claim: every sample 0, the maximum or uniform noise, so the constrained differences reach their limits in both
    directions; every primary level 1..15 alone and every secondary strength alone, a quarter of the blocks skipped, luma (variance-adjusted primary)
    and 4:2:0 chroma."""
    g = rng(900 + bd + damping)
    W, H = (704, 392) if be.is_gpu else (200, 136)
    pm = (1 << bd) - 1
    kind = g.integers(0, 3, (H, W))
    luma = np.where(kind == 0, 0, np.where(kind == 1, pm, g.integers(0, pm + 1, (H, W)))).astype(np.int64)
    luma[H // 3:H // 3 + 24, :] = np.where(g.random((24, W)) < 0.5, 0, pm)  # a band of pure extremes
    nhfb, nvfb = (W + 63) // 64, (H + 63) // 64
    nfb = nhfb * nvfb
    skip = (g.random((nvfb * 8, nhfb * 8)) < 0.25).astype(np.uint8)
    one = g.random(nfb) < 0.5
    apri = np.where(one, g.integers(1, 16, nfb), 0).astype(np.int32)
    asec = np.where(one, 0, g.choice(np.array([1, 2, 4], np.int32), nfb)).astype(np.int32)
    dir0, var0 = np.zeros(nfb * 64, np.uint8), np.zeros(nfb * 64, np.int32)
    d, v = run_frame(be, oracle, 0, luma, luma, 0, 0, 0, bd, skip, apri, asec, dir0, var0, damping=damping)
    chroma = luma[::2, ::2].copy()
    run_frame(be, oracle, 0, chroma, chroma, 1, 1, 1, bd, skip, apri, asec, d, v, damping=damping)


@pytest.mark.parametrize("bd,damping", [(8, 5), (10, 5), (12, 6), (10, 3)])
def test_cdef_frame_apply_and_search(be, oracle, bd, damping):
    g = rng(50 + bd + damping)
    W, H = ((3840, 2160) if (bd, damping) == (10, 5) else (1920, 1080)) if be.is_gpu else (200, 136)
    luma = synth_plane(g, W, H, bd)
    src_l = np.clip(luma + g.integers(-6, 7, luma.shape), 0, (1 << bd) - 1)
    nhfb, nvfb = (W + 63) // 64, (H + 63) // 64
    nfb = nhfb * nvfb
    for skip_frac in (0.0, 0.25):
        skip = (g.random((nvfb * 8, nhfb * 8)) < skip_frac).astype(np.uint8)
        dir0, var0 = np.zeros(nfb * 64, np.uint8), np.zeros(nfb * 64, np.int32)
        # search: all 64 luma strengths on the GPU (pri 0..15 x sec {0,1,2,4}), a subset on the interpreter
        cands = [(pr, sc) for pr in range(16) for sc in (0, 1, 2, 4)] if be.is_gpu else [(0, 0), (4, 2), (15, 4), (1, 0), (0, 1), (7, 1), (9, 0), (2, 4), (5, 2)]
        pri, sec = np.array([c[0] for c in cands], np.int32), np.array([c[1] for c in cands], np.int32)
        d, v = run_frame(be, oracle, 1, luma, src_l, 0, 0, 0, bd, skip, pri, sec, dir0, var0, sub=2 if skip_frac else 1, damping=damping)
        # apply: per-block strengths, (4, 2) with some zero-strength and some secondary-only blocks
        apri = g.choice(np.array([0, 4, 9, 15], np.int32), nfb, p=[0.2, 0.4, 0.2, 0.2]).astype(np.int32)
        # level 0 with a secondary strength filters along dir 0; a primary level alone / a secondary strength alone take the 4- / 8-tap forms without the clamp
        asec = np.where(apri == 0, (g.random(nfb) < 0.5) * 1, np.where(g.random(nfb) < 0.4, 0, g.choice(np.array([1, 2, 4], np.int32), nfb))).astype(np.int32)
        run_frame(be, oracle, 0, luma, src_l, 0, 0, 0, bd, skip, apri, asec, dir0, var0, damping=damping)
        run_frame(be, oracle, 0, luma, src_l, 0, 0, 0, bd, skip, apri, asec, d, v, damping=damping, dev_mode=2)  # apply with the search pass's directions
        # chroma 4:2:0 uses the luma directions
        cw, ch = W // 2, H // 2
        chroma = synth_plane(g, cw, ch, bd)
        src_c = np.clip(chroma + g.integers(-6, 7, chroma.shape), 0, (1 << bd) - 1)
        run_frame(be, oracle, 1, chroma, src_c, 1, 1, 1, bd, skip, pri[:9], sec[:9], d, v, damping=damping)
        run_frame(be, oracle, 0, chroma, src_c, 1, 1, 2, bd, skip, apri, asec, d, v, damping=damping)
        if not be.is_gpu or (bd, damping) == (8, 5):  # 4:2:2 and 4:4:0 chroma: 4x8 / 8x4 units and the direction remap of cdef.c:388-395
            for (xd, yd) in ((1, 0), (0, 1)):
                c2 = synth_plane(g, W >> xd, H >> yd, bd)
                s2 = np.clip(c2 + g.integers(-6, 7, c2.shape), 0, (1 << bd) - 1)
                run_frame(be, oracle, 1, c2, s2, xd, yd, 1, bd, skip, pri[:9], sec[:9], d, v, damping=damping)
                run_frame(be, oracle, 0, c2, s2, xd, yd, 2, bd, skip, apri, asec, d, v, damping=damping)


@pytest.mark.parametrize("gpw", ["2", "4"])
def test_cdef_search_groups_per_workgroup(be, oracle, gpw):
    """Search with several primary-level groups per workgroup (what large frames run: SVT_HIP_CDEF_GPW forces it on a small one): the later groups take the
    secondary sums from the LDS cache the first group's pass filled.  Luma (weighted distortion), 4:2:0 chroma (4x4 units), skipped units, row sub-sampling."""
    import os
    g = rng(77 + int(gpw))
    bd, (W, H) = 10, ((704, 392) if be.is_gpu else (200, 136))
    luma = synth_plane(g, W, H, bd)
    src_l = np.clip(luma + g.integers(-6, 7, luma.shape), 0, (1 << bd) - 1)
    nhfb, nvfb = (W + 63) // 64, (H + 63) // 64
    nfb = nhfb * nvfb
    cands = [(pr, sc) for pr in range(16) for sc in (0, 1, 2, 4)] if be.is_gpu else [(pr, sc) for pr in (0, 1, 2, 3, 5, 7, 8, 11, 12, 15) for sc in (0, 2, 4)] + [(4, 1)]
    pri, sec = np.array([c[0] for c in cands], np.int32), np.array([c[1] for c in cands], np.int32)
    os.environ["SVT_HIP_CDEF_GPW"] = gpw
    be.lib.svt_hip_tuning_reload()
    try:
        for skip_frac, sub in ((0.0, 1), (0.25, 2)):
            skip = (g.random((nvfb * 8, nhfb * 8)) < skip_frac).astype(np.uint8)
            dir0, var0 = np.zeros(nfb * 64, np.uint8), np.zeros(nfb * 64, np.int32)
            d, v = run_frame(be, oracle, 1, luma, src_l, 0, 0, 0, bd, skip, pri, sec, dir0, var0, sub=sub)
            chroma = synth_plane(g, W // 2, H // 2, bd)
            src_c = np.clip(chroma + g.integers(-6, 7, chroma.shape), 0, (1 << bd) - 1)
            run_frame(be, oracle, 1, chroma, src_c, 1, 1, 1, bd, skip, pri, sec, d, v)
    finally:
        del os.environ["SVT_HIP_CDEF_GPW"]
        be.lib.svt_hip_tuning_reload()


def test_cdef_frame_in_row_strips(be, oracle):
    """A picture split over several GPUs: svt_hip_cdef_frame_rows over the strips 3,2,... of filter-block rows, each strip reading its 3-row halos from the
    full input plane and writing only its own rows; the strips together == the whole-frame result, for the search tables and for the applied plane."""
    g = rng(4242)
    bd, (W, H) = 10, ((1920, 1080) if be.is_gpu else (200, 328))  # (sizes are multiples of 8: the reference aligns pictures to 8x8 units)
    luma = synth_plane(g, W, H, bd)
    src = np.clip(luma + g.integers(-6, 7, luma.shape), 0, (1 << bd) - 1)
    nhfb, nvfb = (W + 63) // 64, (H + 63) // 64
    nfb = nhfb * nvfb
    skip = (g.random((nvfb * 8, nhfb * 8)) < 0.2).astype(np.uint8)
    cands = [(0, 0), (4, 2), (15, 4), (1, 0), (0, 1), (7, 1), (9, 0)]
    pri, sec = np.array([c[0] for c in cands], np.int32), np.array([c[1] for c in cands], np.int32)
    apri, asec = g.choice(np.array([0, 4, 9], np.int32), nfb).astype(np.int32), g.choice(np.array([0, 1, 2, 4], np.int32), nfb).astype(np.int32)
    rec, so = luma.astype(np.uint16), src.astype(np.uint16)
    world = 3
    base, rem = divmod(nvfb, world)
    strips = [(k * base + min(k, rem), k * base + min(k, rem) + base + (1 if k < rem else 0)) for k in range(world)]
    for mode, P_, S_, ncand in ((1, pri, sec, len(cands)), (0, apri, asec, 0)):
        o_out, o_dir, o_var, o_mse = rec.copy(), np.zeros(nfb * 64, np.uint8), np.zeros(nfb * 64, np.int32), np.zeros(max(nfb * max(ncand, 1), 1), np.uint64)
        oracle.oracle_cdef_frame(mode, p(rec), W, p(so), W, p(o_out), W, W, H, 0, 0, 0, 1, bd - 8, 5, 5, 1, p(skip), p(P_), p(S_), ncand, p(o_dir), p(o_var), p(o_mse))
        d_rec, d_src, d_out = be.dev(rec), be.dev(so), be.dev(rec)
        d_skip, d_pri, d_sec = be.dev(skip), be.dev(P_), be.dev(S_)
        d_dir, d_var, d_mse = be.dev(np.zeros(nfb * 64, np.uint8)), be.dev(np.zeros(nfb * 64, np.int32)), be.empty(max(nfb * max(ncand, 1), 1), np.uint64)
        P = be.pkg.CdefParams(be.ptr(d_rec), be.ptr(d_src), be.ptr(d_out), W, W, W, W, H, 0, 0, 0, 1, bd - 8, 5, 5, 1, ncand, be.ptr(d_skip), be.ptr(d_pri), be.ptr(d_sec),
                              be.ptr(d_dir), be.ptr(d_var), be.ptr(d_mse))
        for (r0, r1) in reversed(strips):  # any order
            be.lib.svt_hip_cdef_frame_rows(mode, C.byref(P), r0, r1, be.stream)
        assert np.array_equal(be.host(d_dir), o_dir) and np.array_equal(be.host(d_var), o_var)
        if mode:
            assert np.array_equal(be.host(d_mse)[:nfb * ncand], o_mse[:nfb * ncand])
        else:
            assert np.array_equal(be.host(d_out).reshape(H, W), o_out)


def test_cdef_single_call_symbols(be, oracle):
    """svt_aom_cdef_find_dir(_dual), svt_cdef_filter_block, svt_compute_cdef_dist_*, copy_rect8_8bit_to_16bit (CdefTest.cc)."""
    g = rng(3)
    oracle.oracle_cdef_dist.restype = C.c_uint64
    for bd in (8, 10, 12):
        cs = bd - 8
        t = make_tile(g, bd, 5)
        for (by, bx) in ((0, 0), (2, 5), (7, 7)):
            off = by * 8 * BSTRIDE + bx * 8
            v0, v1, v2 = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            d0 = oracle.oracle_cdef_find_dir(in_ptr(t, off), BSTRIDE, C.byref(v0), cs)
            d1 = be.lib.svt_aom_cdef_find_dir_hip(in_ptr(t, off), BSTRIDE, C.cast(C.byref(v1), C.c_void_p), cs)
            assert (d0 & 255, v0.value) == (d1, v1.value)
            o1, o2 = C.c_uint8(0), C.c_uint8(0)
            be.lib.svt_aom_cdef_find_dir_dual_hip(in_ptr(t, off), in_ptr(t, 8), BSTRIDE, C.cast(C.byref(v1), C.c_void_p), C.cast(C.byref(v2), C.c_void_p), cs,
                                                  C.cast(C.byref(o1), C.c_void_p), C.cast(C.byref(o2), C.c_void_p))
            d2 = oracle.oracle_cdef_find_dir(in_ptr(t, 8), BSTRIDE, C.byref(v0), cs)
            assert (o1.value, o2.value, v2.value) == (d0 & 255, d2 & 255, v0.value)
            for (bw, bh) in BLOCK:
                for (pri, sec, dirn, damp, sub) in ((0, 0, 0, 3, 1), (4, 2, 3, 5, 1), (15, 4, 7, 6, 2), (1, 1, 5, 3, 1)):
                    o0 = np.full(64, 9, np.uint16); o1_ = o0.copy()
                    oracle.oracle_cdef_filter_block(None, p(o0), 8, in_ptr(t, off), pri << cs, sec << cs, dirn, damp + cs, damp + cs, bw, bh, cs, sub)
                    be.lib.svt_cdef_filter_block_hip(None, p(o1_), 8, in_ptr(t, off), pri << cs, sec << cs, dirn, damp + cs, damp + cs, BLOCK[(bw, bh)], cs, sub)
                    assert np.array_equal(o0, o1_), (bd, bw, bh, pri, sec, dirn)
    units = np.array([(0, 0), (1, 3), (7, 7), (4, 2)], np.uint8).reshape(-1)
    for (is16, bw, bh, pli, sub) in ((1, 8, 8, 0, 1), (0, 8, 8, 0, 2), (1, 4, 4, 1, 1), (0, 8, 4, 2, 1), (1, 4, 8, 1, 2)):
        dt = np.uint16 if is16 else np.uint8
        plane = g.integers(0, 1024 if is16 else 256, 64 * 80).astype(dt)
        packed = g.integers(0, 1024 if is16 else 256, 4 * bw * bh).astype(dt)
        want = oracle.oracle_cdef_dist(p(plane), 80, p(packed), p(units), 4, bw, bh, 2 if is16 else 0, pli, sub, is16)
        f = be.lib.svt_compute_cdef_dist_16bit_hip if is16 else be.lib.svt_compute_cdef_dist_8bit_hip
        assert f(p(plane), 80, p(packed), p(units), 4, BLOCK[(bw, bh)], 2 if is16 else 0, pli, sub) == want
    src = g.integers(0, 256, 20 * 40).astype(np.uint8)
    dst = np.zeros(20 * 48, np.uint16)
    be.lib.svt_aom_copy_rect8_8bit_to_16bit_hip(p(dst), 48, p(src), 40, 20, 33)
    assert np.array_equal(dst.reshape(20, 48)[:, :33], src.reshape(20, 40)[:, :33].astype(np.uint16))
