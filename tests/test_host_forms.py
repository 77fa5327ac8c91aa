"""The host-pointer forms of the stage entry points -- what the seams inside the reference encoder call (oracle/ref_wrap/*_seam.c): each must give exactly
what the device-pointer form / the oracle gives for the same input.  (Emulator here, MI355X with -m gpu.)"""
import ctypes as C

import numpy as np
import pytest

from conftest import p, rng
from test_cdef import synth_plane
from test_lr_search import PrevUnit, SearchUnit, make_planes, random_prev, search_params


@pytest.mark.parametrize("bd", [8, 10])
def test_cdef_host_forms(be, oracle, bd):
    """svt_hip_cdef_search_host / svt_hip_cdef_apply_host (4:2:0, three planes) == oracle_cdef_frame per plane"""
    pkg = be.pkg
    g = rng(800 + bd)
    W, H = (448, 264) if be.is_gpu else (136, 72)
    is16 = bd > 8
    dt = np.uint16 if is16 else np.uint8
    planes = [synth_plane(g, W, H, bd).astype(dt), synth_plane(g, W // 2, H // 2, bd).astype(dt), synth_plane(g, W // 2, H // 2, bd).astype(dt)]
    srcs = [np.clip(pl.astype(np.int32) + g.integers(-6, 7, pl.shape), 0, (1 << bd) - 1).astype(dt) for pl in planes]
    nhfb, nvfb = (W + 63) // 64, (H + 63) // 64
    nfb = nhfb * nvfb
    skip = (g.random((nvfb * 8, nhfb * 8)) < 0.2).astype(np.uint8)
    damping, sub = 5, (2, 1)
    cand_y = [(0, 0), (4, 2), (15, 4), (1, 0), (7, 1)] if not be.is_gpu else [(pr, sc) for pr in range(0, 16, 3) for sc in (0, 1, 2, 4)]
    cand_uv = cand_y[:3]
    py, sy = np.array([c[0] for c in cand_y], np.int32), np.array([c[1] for c in cand_y], np.int32)
    pu, su = np.array([c[0] for c in cand_uv], np.int32), np.array([c[1] for c in cand_uv], np.int32)
    # oracle: search per plane (luma first: directions)
    o_dir, o_var = np.zeros(nfb * 64, np.uint8), np.zeros(nfb * 64, np.int32)
    o_mse = []
    for pli in range(3):
        pr, sc = (py, sy) if pli == 0 else (pu, su)
        m = np.zeros(nfb * len(pr), np.uint64)
        h, w = planes[pli].shape
        oracle.oracle_cdef_frame(1, p(planes[pli]), w, p(srcs[pli]), w, p(planes[pli].copy()), w, w, h, int(pli > 0), int(pli > 0), pli, int(is16), bd - 8, damping, damping,
                                 sub[min(pli, 1)], p(skip), p(pr), p(sc), len(pr), p(o_dir), p(o_var), p(m))
        o_mse.append(m)
    A = pkg.CdefSearchHost()
    mse = [np.zeros(nfb * len(py), np.uint64), np.zeros(nfb * len(pu), np.uint64), np.zeros(nfb * len(pu), np.uint64)]
    g_dir, g_var = np.zeros(nfb * 64, np.uint8), np.zeros(nfb * 64, np.int32)
    for pli in range(3):
        A.recon[pli], A.source[pli] = planes[pli].ctypes.data, srcs[pli].ctypes.data
        A.recon_stride[pli] = A.source_stride[pli] = planes[pli].shape[1]
    A.width, A.height, A.is_16bit, A.coeff_shift, A.damping = W, H, int(is16), bd - 8, damping
    A.subsampling[0], A.subsampling[1] = sub
    A.skip, A.ncand_y, A.ncand_uv = skip.ctypes.data, len(py), len(pu)
    A.pri_y, A.sec_y, A.pri_uv, A.sec_uv = py.ctypes.data, sy.ctypes.data, pu.ctypes.data, su.ctypes.data
    A.mse_y, A.mse_u, A.mse_v, A.dir, A.var = mse[0].ctypes.data, mse[1].ctypes.data, mse[2].ctypes.data, g_dir.ctypes.data, g_var.ctypes.data
    be.lib.svt_hip_cdef_search_host(C.byref(A))
    for pli in range(3):
        assert np.array_equal(mse[pli], o_mse[pli]), pli
    keep = np.repeat(np.repeat(skip == 0, 1, 0), 1, 1)  # directions are defined for the non-skipped units
    unit_ok = np.zeros(nfb * 64, bool)
    for fb in range(nfb):
        blk = keep[(fb // nhfb) * 8:(fb // nhfb) * 8 + 8, (fb % nhfb) * 8:(fb % nhfb) * 8 + 8]
        unit_ok[fb * 64:(fb + 1) * 64] = blk.reshape(-1)
    assert np.array_equal(g_dir[unit_ok], o_dir[unit_ok]) and np.array_equal(g_var[unit_ok], o_var[unit_ok])
    # apply: per-filter-block strengths, in place
    apri_y = np.where(g.random(nfb) < 0.2, 0, 4).astype(np.int32)
    asec_y = np.where(apri_y == 0, 1, 2).astype(np.int32)
    apri_uv, asec_uv = (apri_y // 2).astype(np.int32), np.where(g.random(nfb) < 0.5, 0, 4).astype(np.int32)
    want = []
    o_dir2, o_var2 = np.zeros(nfb * 64, np.uint8), np.zeros(nfb * 64, np.int32)
    for pli in range(3):
        pr, sc = (apri_y, asec_y) if pli == 0 else (apri_uv, asec_uv)
        h, w = planes[pli].shape
        out = planes[pli].copy()
        oracle.oracle_cdef_frame(0, p(planes[pli]), w, p(srcs[pli]), w, p(out), w, w, h, int(pli > 0), int(pli > 0), pli, int(is16), bd - 8, damping, damping, 1, p(skip),
                                 p(pr), p(sc), 0, p(o_dir2), p(o_var2), p(np.zeros(1, np.uint64)))
        want.append(out)
    B = pkg.CdefApplyHost()
    work = [pl.copy() for pl in planes]
    for pli in range(3):
        B.plane[pli], B.stride[pli] = work[pli].ctypes.data, work[pli].shape[1]
    B.width, B.height, B.num_planes, B.is_16bit, B.coeff_shift, B.damping = W, H, 3, int(is16), bd - 8, damping
    B.skip, B.pri_y, B.sec_y, B.pri_uv, B.sec_uv = skip.ctypes.data, apri_y.ctypes.data, asec_y.ctypes.data, apri_uv.ctypes.data, asec_uv.ctypes.data
    be.lib.svt_hip_cdef_apply_host(C.byref(B))
    for pli in range(3):
        assert np.array_equal(work[pli], want[pli]), (pli, np.argwhere(work[pli] != want[pli])[:5])


@pytest.mark.parametrize("bd", [8, 10])
def test_lr_host_forms(be, oracle, bd):
    """svt_hip_lr_search_plane_host == oracle_lr_search_plane; svt_hip_lr_filter_frame_host == oracle_lr_filter_frame (in place, with the saved boundary lines)"""
    if bd == 10 and not be.is_gpu:
        pytest.skip("the 10-bit case runs on the GPU only (the CPU suite's time budget; tests/test_lr_search.py covers 10 bit on the emulator)")
    pkg = be.pkg
    g = rng(820 + bd)
    w, h, unit = (300, 200, 64) if be.is_gpu else (96, 40, 64)  # (a 64-column processing unit never spans two restoration units: unit >= 64 >> ss_x)
    src, dgd, pad = make_planes(g, w, h, bd)
    P = search_params(src, dgd, pad, w, h, bd, unit, 0, (1, 7, 1, 0), (1, 2, 12, 3, 1) if be.is_gpu else (1, 2, 6, 4, 1))
    n = oracle.oracle_lr_unit_rect(C.byref(P), -1, None)
    prev = random_prev(g, n, 7)
    want = np.zeros(n, SearchUnit)
    oracle.oracle_lr_search_plane(C.byref(P), p(prev), p(want), None)
    got = np.zeros(n, pkg.LrSearchUnit)
    PH = pkg.LrSearchParams.from_buffer_copy(bytes(P))
    assert be.lib.svt_hip_lr_search_plane_host(C.byref(PH), p(prev), p(got)) == 0
    for k in ("sse", "vfilter", "hfilter", "ep", "xqd"):
        assert np.array_equal(got[k], want[k]), (k, got[k], want[k])
    # frame filter with the units the search found (Wiener where accepted, self-guided elsewhere), boundary lines = random saved rows
    dt = src.dtype
    plane = np.ascontiguousarray(dgd[pad:pad + h, pad:pad + w])
    nstripes = (h + 8 + 63) // 64
    above, below = g.integers(0, 1 << bd, (2 * nstripes, w)).astype(dt), g.integers(0, 1 << bd, (2 * nstripes, w)).astype(dt)
    units = np.zeros(n, pkg.LrUnit)
    wn_ok = want["sse"][:, 1] != np.iinfo(np.int64).max
    units["rtype"] = np.where(wn_ok & (np.arange(n) % 3 != 2), 1, 2)
    units["vfilter"], units["hfilter"], units["ep"], units["xqd"] = want["vfilter"], want["hfilter"], want["ep"], want["xqd"]
    out = np.zeros((h, w), dt)
    oracle.oracle_lr_filter_frame(p(plane), w, p(above), p(below), w, p(out), w, w, h, 0, unit, p(units), bd, int(bd > 8))
    import os
    for ur in (("32", "16", "64") if be.is_gpu or bd == 8 else ("32",)):  # rows of a stripe per workgroup (SVT_HIP_LR_UR: 32 is the default; the other two instantiations stay covered)
        os.environ["SVT_HIP_LR_UR"] = ur
        be.lib.svt_hip_tuning_reload()  # the knob is read once, not per launch
        try:
            work = plane.copy()
            L = pkg.LrParams(work.ctypes.data, above.ctypes.data, below.ctypes.data, work.ctypes.data, w, w, w, w, h, unit, 0, 0, int(bd > 8), bd, units.ctypes.data)
            be.lib.svt_hip_lr_filter_frame_host(C.byref(L))
        finally:
            del os.environ["SVT_HIP_LR_UR"]
            be.lib.svt_hip_tuning_reload()
        assert np.array_equal(work, out), (ur, np.argwhere(work != out)[:5])
