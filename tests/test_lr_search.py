"""Loop-restoration SEARCH, per-unit half (restoration_pick.c:1205-1420 = restoration_seg_search): SSE of the unrestored unit, the Wiener
coefficient solve + refinement search, the self-guided parameter search, for every restoration unit of a plane.
  * oracle (oracle/oracle_lr_search.c) pinned against the reference's own static functions (search_norestore_seg / search_wiener_seg / search_sgrproj_seg and
    everything below), compiled where they lie through oracle/ref_wrap/ref_lr_search.c;
  * device: svt_hip_lr_search_plane vs the oracle (emulator here, MI355X with -m gpu)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import REF_LIB, ROOT, p, rng

REF_ME_LIB = os.path.join(ROOT, "oracle", "_ref", "libsvtref_me.so")
INT64_MAX = (1 << 63) - 1


class SearchParams(C.Structure):
    _fields_ = [("dgd", C.c_void_p), ("src", C.c_void_p), ("dgd_stride", C.c_uint32), ("src_stride", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("unit_size", C.c_uint32), ("ss_y", C.c_uint8), ("highbd", C.c_uint8), ("bit_depth", C.c_uint8), ("wn_enabled", C.c_uint8), ("wiener_win", C.c_uint8),
                ("wn_use_refinement", C.c_uint8), ("wn_max_one_refinement_step", C.c_uint8), ("sg_enabled", C.c_uint8), ("sg_start_ep", C.c_uint8),
                ("sg_end_ep", C.c_uint8), ("sg_ep_inc", C.c_uint8), ("sg_refine", C.c_uint8), ("pad", C.c_uint8 * 3)]


SearchUnit = np.dtype([("sse", "<i8", (3,)), ("vfilter", "<i2", (8,)), ("hfilter", "<i2", (8,)), ("ep", "<i4"), ("xqd", "<i4", (2,)), ("pad", "<i4")])
PrevUnit = np.dtype([("use", "<i4"), ("vfilter", "<i2", (8,)), ("hfilter", "<i2", (8,))])
assert SearchUnit.itemsize == 72 and PrevUnit.itemsize == 36


def make_planes(g, w, h, bd, pad=8, blur=2.0, noise=6.0):
    """source = textured picture; dgd = the source degraded by a separable blur + quantisation-like noise (so that restoration has something to find),
    edge-extended by `pad` samples like svt_extend_frame leaves it"""
    amp = (1 << bd) - 1
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    tex = 0.5 + 0.2 * np.sin(xx / 2.3) * np.cos(yy / 3.1) + 0.15 * np.sin((xx + 2 * yy) / 6.7) + 0.1 * np.sign(np.sin(xx / 9.0) * np.sin(yy / 7.0))
    src = np.clip(tex * amp + g.normal(0, amp / 120, tex.shape), 0, amp)
    k = np.array([1, blur, 4, blur, 1], np.float64)
    k /= k.sum()
    d = np.apply_along_axis(lambda r: np.convolve(np.pad(r, 2, mode="edge"), k, "valid"), 1, src)
    d = np.apply_along_axis(lambda c: np.convolve(np.pad(c, 2, mode="edge"), k, "valid"), 0, d)
    d = np.clip(np.round(d / (noise * amp / 255)) * (noise * amp / 255) + g.normal(0, amp / 200, d.shape), 0, amp)
    dt = np.uint16 if bd > 8 else np.uint8
    dgd = np.pad(d.astype(dt), pad, mode="edge")
    return np.ascontiguousarray(src.astype(dt)), np.ascontiguousarray(dgd), pad


def search_params(src, dgd, pad, w, h, bd, unit, ss_y=0, wn=(1, 7, 1, 0), sg=(1, 0, 16, 1, 1)):
    P = SearchParams()
    P.dgd = dgd.ctypes.data + (pad * dgd.shape[1] + pad) * dgd.itemsize
    P.src = src.ctypes.data
    P.dgd_stride, P.src_stride, P.width, P.height, P.unit_size = dgd.shape[1], src.shape[1], w, h, unit
    P.ss_y, P.highbd, P.bit_depth = ss_y, int(bd > 8), bd
    P.wn_enabled, P.wiener_win, P.wn_use_refinement, P.wn_max_one_refinement_step = wn
    P.sg_enabled, P.sg_start_ep, P.sg_end_ep, P.sg_ep_inc, P.sg_refine = sg
    return P


def unit_rects(oracle, P):
    n = oracle.oracle_lr_unit_rect(C.byref(P), -1, None)
    rects = np.zeros((n, 4), np.int32)
    for u in range(n):
        oracle.oracle_lr_unit_rect(C.byref(P), u, p(rects[u]))
    return rects


# (width, height, bit depth, unit size, ss_y, wiener (enabled, win, refinement, one step), self-guided (enabled, start, end, inc, refine), previous-frame taps)
CASES = [(150, 100, 8, 64, 0, (1, 7, 1, 0), (1, 0, 16, 1, 1), False),
         (150, 100, 10, 64, 0, (1, 7, 1, 0), (1, 0, 16, 1, 1), False),
         (96, 70, 8, 32, 1, (1, 5, 1, 1), (1, 2, 14, 4, 0), False),
         (130, 64, 10, 64, 0, (1, 3, 0, 0), (1, 10, 16, 1, 1), True),
         (100, 90, 8, 64, 0, (1, 5, 1, 0), (0, 0, 0, 1, 0), True),
         (96, 70, 12, 64, 0, (1, 7, 1, 0), (1, 4, 12, 4, 1), False)]  # 12 bit (the device keeps int32 flt planes there)


@pytest.mark.parametrize("case", range(len(CASES)))
def test_lr_search_oracle_vs_reference(oracle, ref, case):
    if not os.path.exists(REF_ME_LIB):
        pytest.skip("oracle/_ref/libsvtref_me.so not available")
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    C.CDLL(REF_LIB, mode=C.RTLD_GLOBAL)
    refme = C.CDLL(REF_ME_LIB)
    w, h, bd, unit, ss_y, wn, sg, use_prev = CASES[case]
    g = rng(700 + case)
    src, dgd, pad = make_planes(g, w, h, bd)
    P = search_params(src, dgd, pad, w, h, bd, unit, ss_y, wn, sg)
    rects = unit_rects(oracle, P)
    n = len(rects)
    assert n >= 2 and rects[:, 1].max() == w and rects[:, 3].max() == h
    prev = None
    if use_prev:
        prev = np.zeros(n, PrevUnit)
        prev["use"] = np.arange(n) % 2
        off = (7 - wn[1]) // 2
        for u in range(n):
            for f in ("vfilter", "hfilter"):
                t = [int(g.integers(lo, hi + 1)) for lo, hi in ((-5, 10), (-23, 8), (-17, 46))]
                for i in range(off):
                    t[i] = 0
                prev[f][u][:7] = t + [-2 * sum(t)] + t[::-1]
    a, b = np.zeros(n, SearchUnit), np.zeros(n, SearchUnit)
    trials = np.zeros(n, np.int32)
    oracle.oracle_lr_search_plane(C.byref(P), p(prev) if use_prev else None, p(a), p(trials))
    refme.ref_lr_search_plane(C.byref(P), p(prev) if use_prev else None, p(b), p(rects), n)
    cols = [0] + ([1] if wn[0] else []) + ([2] if sg[0] else [])  # (the reference leaves the sse of a disabled tool untouched; oracle and device report INT64_MAX there)
    assert np.array_equal(a["sse"][:, cols], b["sse"][:, cols]), (a["sse"], b["sse"])
    for c in (1, 2):
        if c not in cols:
            assert (a["sse"][:, c] == np.iinfo(np.int64).max).all()
    for k in ("vfilter", "hfilter", "ep", "xqd"):
        assert np.array_equal(a[k], b[k]), (k, a[k], b[k])
    if wn[0] and wn[2]:
        assert trials.max() > (4 if wn[3] else 6)  # the refinement really runs
    if sg[0]:
        assert (a["sse"][:, 2] <= a["sse"][:, 0]).mean() > 0.4  # and restoration finds something on this content


def test_wiener_solve_oracle_vs_reference(oracle, ref):
    """wiener_decompose_sep_sym / finalize_sym_filter / compute_score on random statistics of realistic magnitude, all three window sizes"""
    if not os.path.exists(REF_ME_LIB):
        pytest.skip("oracle/_ref/libsvtref_me.so not available")
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))  # (svt_memcpy is a dispatch pointer)
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    C.CDLL(REF_LIB, mode=C.RTLD_GLOBAL)
    refme = C.CDLL(REF_ME_LIB)
    refme.ref_wiener_solve.restype = C.c_int64
    oracle.oracle_wiener_compute_score.restype = C.c_int64
    g = rng(720)
    for it in range(60):
        win = (7, 5, 3)[it % 3]
        w2 = win * win
        # statistics of a smooth signal: y = shifted copies of x + noise
        npx = 4096
        x = np.cumsum(g.normal(0, 6, npx + 64)).astype(np.int64) + g.integers(-40, 41, npx + 64)
        Y = np.stack([np.roll(x, k - w2 // 2)[32:32 + npx] + g.integers(-3, 4, npx) for k in range(w2)])
        M = (Y @ (x[32:32 + npx] + g.integers(-2, 3, npx))).astype(np.int64)
        H = (Y @ Y.T).astype(np.int64)
        if it % 7 == 6:
            H[:] = 0  # singular: linsolve fails, the initial filter stays
        M1, H1, M2, H2 = M.copy(), H.copy().reshape(-1), M.copy(), H.copy().reshape(-1)
        vd1, hd1, vd2, hd2 = (np.zeros(7, np.int32) for _ in range(4))
        vf1, hf1, vf2, hf2 = (np.zeros(8, np.int16) for _ in range(4))
        oracle.oracle_wiener_decompose_sep_sym(win, p(M1), p(H1), p(vd1), p(hd1))
        oracle.oracle_wiener_finalize_sym_filter(win, p(vd1), p(vf1))
        oracle.oracle_wiener_finalize_sym_filter(win, p(hd1), p(hf1))
        s1 = oracle.oracle_wiener_compute_score(win, p(M1), p(H1), p(vf1), p(hf1))
        s2 = refme.ref_wiener_solve(win, p(M2), p(H2), p(vf2), p(hf2), p(vd2), p(hd2))
        assert np.array_equal(vd1, vd2) and np.array_equal(hd1, hd2), (it, win, vd1, vd2)
        assert np.array_equal(vf1, vf2) and np.array_equal(hf1, hf2) and s1 == s2, (it, win, vf1, vf2, s1, s2)


def run_device(be, P, n, prev):
    pkg = be.pkg
    ws = be.empty((be.lib.svt_hip_lr_search_workspace(C.byref(P)),), np.uint8)
    d_out = be.empty((n,), pkg.LrSearchUnit)
    d_prev = be.dev(prev) if prev is not None else None
    rc = be.lib.svt_hip_lr_search_plane(C.byref(P), be.ptr(d_prev) if d_prev is not None else None, be.ptr(d_out), be.ptr(ws), be.stream)
    assert rc == 0
    return be.host(d_out)


def random_prev(g, n, win):
    prev = np.zeros(n, PrevUnit)
    prev["use"] = np.arange(n) % 2
    off = (7 - win) // 2
    for u in range(n):
        for f in ("vfilter", "hfilter"):
            t = [int(g.integers(lo, hi + 1)) for lo, hi in ((-5, 10), (-23, 8), (-17, 46))]
            for i in range(off):
                t[i] = 0
            prev[f][u][:7] = t + [-2 * sum(t)] + t[::-1]
    return prev


# small planes (the lock-step emulator runs every trial of every unit): (width, height, bit depth, unit, ss_y, wiener, self-guided, previous-frame taps)
DEV_CASES = [(48, 40, 8, 32, 0, (1, 7, 1, 0), (1, 0, 16, 6, 1), False),
             (72, 40, 10, 64, 0, (1, 5, 1, 0), (1, 12, 16, 1, 1), True),
             (60, 50, 8, 32, 1, (1, 3, 1, 1), (1, 3, 4, 1, 0), False),
             (44, 36, 10, 64, 0, (1, 7, 1, 0), (0, 0, 0, 1, 0), False),   # one unit (plane smaller than 3/2 unit), Wiener only
             (50, 40, 8, 64, 0, (0, 7, 0, 0), (1, 9, 10, 1, 1), False),   # one unit, self-guided only, a single parameter set
             (48, 40, 12, 32, 0, (0, 7, 0, 0), (1, 6, 15, 4, 1), False)]  # 12-bit: the int32 flt planes (<= 10 bit packs int16 differences)
GPU_CASES = [(500, 300, 8, 64, 0, (1, 7, 1, 0), (1, 0, 16, 1, 1), False),
             (420, 260, 10, 128, 0, (1, 7, 1, 0), (1, 0, 16, 1, 1), True),
             (300, 200, 10, 64, 1, (1, 5, 1, 1), (1, 2, 14, 4, 0), False),
             (330, 170, 8, 64, 0, (1, 3, 0, 0), (1, 10, 16, 1, 1), True),
             (70, 50, 10, 64, 0, (1, 7, 1, 0), (0, 0, 0, 1, 0), False),
             (1000, 90, 8, 256, 0, (0, 7, 0, 0), (1, 9, 10, 1, 1), False),  # one row of wide units (the last one 1.4 units wide)
             (300, 200, 12, 64, 0, (1, 7, 1, 0), (1, 0, 16, 3, 1), False),   # 12-bit: the int32 flt planes
             (300, 380, 10, 256, 0, (0, 7, 0, 0), (1, 0, 16, 5, 1), False)]  # one unit of 114 000 samples (111 per thread of the projection kernel)


@pytest.mark.parametrize("walk", ["line", "fallback", "table"])
@pytest.mark.parametrize("case", [1, 3])
def test_lr_search_plane_hip_walk_forms(be, oracle, case, walk, monkeypatch):
    """the self-guided projection refinement's other forms give the same units: SVT_HIP_LR_SG_WALK=line (every step size line by line, the round-4 form), =table (the
    largest step from a one-pass error table too) and =fallback (as table, then the line-by-line redo a walk that leaves the table takes -- forced for every set)"""
    monkeypatch.setenv("SVT_HIP_LR_SG_WALK", walk)
    test_lr_search_plane_hip(be, oracle, case, 0, monkeypatch)


@pytest.mark.parametrize("group", [0, 1, 3])
@pytest.mark.parametrize("case", range(max(len(GPU_CASES), len(DEV_CASES))))
def test_lr_search_plane_hip(be, oracle, case, group, monkeypatch):
    """svt_hip_lr_search_plane == oracle_lr_search_plane: SSEs, Wiener taps, self-guided parameter set and projection for every unit.  group: the self-guided parameter
    sets are filtered and projected in groups that share one set of plane buffers (0 = as many as fit 240 MB -- all of them on these planes --, 1 and 3 forced)"""
    cases = GPU_CASES if be.is_gpu else DEV_CASES
    if group and (case >= len(cases) or not cases[case][6][0] or case in (0, 2, 4)):  # (case 0 has three sets: covered by cases 1 and 5)
        pytest.skip("grouping only matters with several self-guided parameter sets")
    if group:
        monkeypatch.setenv("SVT_HIP_LR_SG_GROUP", str(group))
    else:
        monkeypatch.delenv("SVT_HIP_LR_SG_GROUP", raising=False)
    if case >= len(cases):
        pytest.skip("GPU-sized case")
    w, h, bd, unit, ss_y, wn, sg, use_prev = cases[case]
    g = rng(740 + case)
    src, dgd, pad = make_planes(g, w, h, bd)
    P = search_params(src, dgd, pad, w, h, bd, unit, ss_y, wn, sg)
    n = oracle.oracle_lr_unit_rect(C.byref(P), -1, None)
    prev = random_prev(g, n, wn[1]) if use_prev else None
    want = np.zeros(n, SearchUnit)
    oracle.oracle_lr_search_plane(C.byref(P), p(prev) if use_prev else None, p(want), None)
    d_src, d_dgd = be.dev(src), be.dev(dgd)
    PD = be.pkg.LrSearchParams.from_buffer_copy(bytes(P))
    PD.src = be.ptr(d_src)
    PD.dgd = be.ptr(d_dgd) + (pad * dgd.shape[1] + pad) * dgd.itemsize
    got = run_device(be, PD, n, prev)
    for k in ("sse", "vfilter", "hfilter", "ep", "xqd"):
        assert np.array_equal(got[k], want[k]), (k, got[k], want[k])
