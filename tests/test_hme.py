"""One hierarchical-ME level for a whole picture (the callers of svt_sad_loop_kernel): hme_level_0 / hme_level_1 / hme_level_2 geometry + search +
rescaling.  The oracle is pinned against the reference's own leaf drivers (static; reached through oracle/_ref/libsvtref_me.so), the HIP stage
against the oracle through the C-ABI."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, load_pkg, p, rng

REF_ME_LIB = os.path.join(ROOT, "oracle", "_ref", "libsvtref_me.so")
ORG = {0: 16, 1: 32, 2: 68}  # plane padding per level (enc_handle.c:1260-1278, :4084)


def make_planes(g, W, H, level, n_refs):
    """source + n_refs reference planes at the level's resolution, padded; references = shifted source + noise so that the search has structure"""
    sh = {0: 2, 1: 1, 2: 0}[level]
    w, h, org = W >> sh, H >> sh, ORG[level]
    stride, rows = w + 2 * org, h + 2 * org
    base = g.integers(0, 256, (rows + 16, stride + 16), dtype=np.uint8)
    base = (base // 4 + (np.add.outer(np.arange(rows + 16), np.arange(stride + 16)) * 3 % 160)).astype(np.uint8)
    src = np.ascontiguousarray(base[8:8 + rows, 8:8 + stride])
    refs = []
    for r in range(n_refs):
        dx, dy = int(g.integers(-5, 6)), int(g.integers(-4, 5))
        a = base[8 + dy:8 + dy + rows, 8 + dx:8 + dx + stride].astype(np.int32) + g.integers(-3, 4, (rows, stride))
        refs.append(np.clip(a, 0, 255).astype(np.uint8))
    return src, refs, w, h, org, stride


def vp(a, off=0):
    return C.c_void_p(a.ctypes.data + off)


def cpu_level(fn, level, sub, nw, nh, src, refs, w, h, org, stride, W, H, sa_w, sa_h, prev):
    """every (ref, SB, region) item with a per-item C function (oracle or reference); returns (sad [items], sc [items][2])"""
    sh = {0: 2, 1: 1, 2: 0}[level]
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    sbs_x, sbs_y = (aw + 63) // 64, (ah + 63) // 64
    n = len(refs) * sbs_x * sbs_y * nw * nh
    sad, sc = np.zeros(n, np.uint64), np.zeros((n, 2), np.int16)
    i = 0
    for r in range(len(refs)):
        for sb in range(sbs_x * sbs_y):
            fx, fy = (sb % sbs_x) * 64, (sb // sbs_x) * 64
            bw, bh = min(64, aw - fx) >> sh, min(64, ah - fy) >> sh
            ox, oy = fx >> sh, fy >> sh
            for sr_h in range(nh):
                for sr_w in range(nw):
                    bs, x, y = C.c_uint64(0), C.c_int16(0), C.c_int16(0)
                    fn(level, sub, nw, nh, sr_w, sr_h, vp(src, (org + oy) * stride + org + ox), stride, vp(refs[r]), stride, org, org, w, h, ox, oy, bw, bh, sa_w,
                       sa_h, int(prev[i, 0]), int(prev[i, 1]), C.byref(bs), C.byref(x), C.byref(y))
                    sad[i], sc[i] = bs.value, (x.value, y.value)
                    i += 1
    return sad, sc


CASES = [(0, 0, 2, 2, 16, 8), (0, 1, 2, 2, 24, 12), (0, 0, 1, 1, 40, 20), (1, 0, 2, 2, 8, 5), (1, 1, 2, 2, 16, 8), (2, 0, 2, 2, 8, 3), (2, 1, 1, 2, 13, 7)]


@pytest.mark.parametrize("case", CASES)
def test_hme_level_oracle_vs_reference(oracle, ref, case):
    if not os.path.exists(REF_ME_LIB):
        pytest.skip("oracle/_ref/libsvtref_me.so not available")
    refme = C.CDLL(REF_ME_LIB)
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))  # svt_sad_loop_kernel
    level, sub, nw, nh, sa_w, sa_h = case
    g = rng(1600 + sum(case))
    W, H = 200, 136  # 4 x 3 SBs, partial last column / row
    src, refs, w, h, org, stride = make_planes(g, W, H, level, 2)
    n = 2 * 4 * 3 * nw * nh
    prev = np.stack([g.integers(-3 * w // 4, 3 * w // 4, n), g.integers(-3 * h // 4, 3 * h // 4, n)], 1).astype(np.int16)  # also far outside: clipping
    prev[::3] //= 8
    a = cpu_level(oracle.oracle_hme_level, level, sub, nw, nh, src, refs, w, h, org, stride, W, H, sa_w, sa_h, prev)
    b = cpu_level(refme.ref_hme_level, level, sub, nw, nh, src, refs, w, h, org, stride, W, H, sa_w, sa_h, prev)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), case
    assert len(np.unique(a[1], axis=0)) > 4


@pytest.mark.parametrize("case", CASES)
def test_hme_level_hip(be, oracle, case):
    pkg = load_pkg()
    level, sub, nw, nh, sa_w, sa_h = case
    g = rng(1700 + sum(case))
    W, H = (200, 136) if not be.is_gpu else (712, 400)
    n_refs = 2
    src, refs, w, h, org, stride = make_planes(g, W, H, level, n_refs)
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    sbs_x, sbs_y = (aw + 63) // 64, (ah + 63) // 64
    n = n_refs * sbs_x * sbs_y * nw * nh
    prev = np.stack([g.integers(-3 * w // 4, 3 * w // 4, n), g.integers(-3 * h // 4, 3 * h // 4, n)], 1).astype(np.int16)
    prev[::3] //= 8
    want = cpu_level(oracle.oracle_hme_level, level, sub, nw, nh, src, refs, w, h, org, stride, W, H, sa_w, sa_h, prev)
    planes = np.stack([src] + refs)
    P = pkg.HmeLevelParams()
    P.level, P.sub_sampled, P.num_hme_sa_w, P.num_hme_sa_h, P.sa_width, P.sa_height = level, sub, nw, nh, sa_w, sa_h
    P.sbs_x, P.sbs_y, P.n_refs, P.aligned_width, P.aligned_height = sbs_x, sbs_y, n_refs, aw, ah
    P.src_off, P.src_stride = org * stride + org, stride
    P.ref_stride, P.ref_org_x, P.ref_org_y, P.ref_width, P.ref_height = stride, org, org, w, h
    for r in range(n_refs):
        P.ref_off[r] = (1 + r) * src.size
    d_pl, d_prev = be.dev(planes), be.dev(prev)
    d_sad, d_sc = be.empty(n, np.uint64), be.dev(np.zeros((n, 2), np.int16))
    d_ws = be.empty(be.lib.svt_hip_hme_level_workspace(C.addressof(P)), np.uint8)
    be.lib.svt_hip_hme_level_batch(C.addressof(P), be.ptr(d_pl), be.ptr(d_pl), be.ptr(d_prev), None, be.ptr(d_sad), be.ptr(d_sc), be.ptr(d_ws), be.stream)
    assert np.array_equal(be.host(d_sad), want[0]), case
    assert np.array_equal(be.host(d_sc), want[1]), case


def test_hme_three_level_chain_hip(be, oracle):
    """Level 0 -> 1 -> 2 entirely on the device (level-0 centres >> 1 feed level 1, hme_level1_b64 :2105-2110) vs the same chain through the oracle."""
    pkg, g = load_pkg(), rng(1800)
    W, H = (200, 136) if not be.is_gpu else (712, 400)
    n_refs, nw, nh = 2, 2, 2
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    sbs_x, sbs_y = (aw + 63) // 64, (ah + 63) // 64
    n = n_refs * sbs_x * sbs_y * nw * nh
    full = make_planes(g, W, H, 2, n_refs)
    # decimated planes of the same content (plain 2x2 / 4x4 subsampling is enough here: the test is about the chaining)
    levels = {2: full}
    for lv, sh in ((1, 1), (0, 2)):
        org = ORG[lv]
        def dec(a, o2=ORG[2]):
            core = a[o2:o2 + H, o2:o2 + W][::1 << sh, ::1 << sh]
            return np.ascontiguousarray(np.pad(core, org, mode="edge"))
        s = dec(full[0])
        levels[lv] = (s, [dec(r) for r in full[1]], W >> sh, H >> sh, org, s.shape[1])
    sa = {0: (16, 8), 1: (8, 3), 2: (8, 3)}
    prev_o = np.zeros((n, 2), np.int16)
    d_prev = be.dev(prev_o)
    fused = []
    for lv in (0, 1, 2):
        src, refs, w, h, org, stride = levels[lv]
        pin = prev_o >> 1 if lv == 1 else prev_o
        sad_o, prev_o = cpu_level(oracle.oracle_hme_level, lv, 0, nw, nh, src, refs, w, h, org, stride, W, H, sa[lv][0], sa[lv][1], pin)
        planes = np.stack([src] + refs)
        P = pkg.HmeLevelParams()
        P.level, P.sub_sampled, P.num_hme_sa_w, P.num_hme_sa_h, P.sa_width, P.sa_height = lv, 0, nw, nh, sa[lv][0], sa[lv][1]
        P.sbs_x, P.sbs_y, P.n_refs, P.prev_shift, P.aligned_width, P.aligned_height = sbs_x, sbs_y, n_refs, int(lv == 1), aw, ah
        P.src_off, P.src_stride = org * stride + org, stride
        P.ref_stride, P.ref_org_x, P.ref_org_y, P.ref_width, P.ref_height = stride, org, org, w, h
        for r in range(n_refs):
            P.ref_off[r] = (1 + r) * src.size
        d_pl = be.dev(planes)
        d_sad, d_sc = be.empty(n, np.uint64), be.dev(np.zeros((n, 2), np.int16))
        d_ws = be.empty(be.lib.svt_hip_hme_level_workspace(C.addressof(P)), np.uint8)
        be.lib.svt_hip_hme_level_batch(C.addressof(P), be.ptr(d_pl), be.ptr(d_pl), be.ptr(d_prev), None, be.ptr(d_sad), be.ptr(d_sc), be.ptr(d_ws), be.stream)
        assert np.array_equal(be.host(d_sad), sad_o) and np.array_equal(be.host(d_sc), prev_o), lv
        d_prev = d_sc
        fused.append((P, d_pl, sad_o, prev_o))
    assert np.abs(prev_o).max() > 0
    # the same chain in one launch
    PA = (pkg.HmeLevelParams * 3)(*[f[0] for f in fused])
    planes_p = (C.c_void_p * 3)(*[be.ptr(f[1]) for f in fused])
    d_sads, d_scs = [be.empty(n, np.uint64) for _ in range(3)], [be.dev(np.zeros((n, 2), np.int16)) for _ in range(3)]
    sad_p, sc_p = (C.c_void_p * 3)(*[be.ptr(x) for x in d_sads]), (C.c_void_p * 3)(*[be.ptr(x) for x in d_scs])
    be.lib.svt_hip_hme_chain_batch(C.addressof(PA), C.addressof(planes_p), C.addressof(planes_p), None, C.addressof(sad_p), C.addressof(sc_p), be.stream)
    for lv in range(3):
        assert np.array_equal(be.host(d_sads[lv]), fused[lv][2]) and np.array_equal(be.host(d_scs[lv]), fused[lv][3]), ("fused", lv)
    # SvtHipHmeChainInputs by name: n_levels = 2 (enable_hme_level2_flag = 0: level 2 buffers untouched) and list1_no_hme = 1 (temporal layer 0: list 1's
    # items -- the second reference here -- take no part in HME and their entries are left as the caller filled them)
    for p_ in PA:
        p_.n_refs_list0 = 1
    X = pkg.HmeChainInputs()
    X.n_levels, X.list1_no_hme = 2, 1
    mark_sad, mark_sc = np.full(n, 0x1234567, np.uint64), np.full((n, 2), -77, np.int16)
    d_sads, d_scs = [be.dev(mark_sad) for _ in range(3)], [be.dev(mark_sc) for _ in range(3)]
    sad_p, sc_p = (C.c_void_p * 3)(*[be.ptr(x) for x in d_sads]), (C.c_void_p * 3)(*[be.ptr(x) for x in d_scs])
    be.lib.svt_hip_hme_chain_batch(C.addressof(PA), C.addressof(planes_p), C.addressof(planes_p), C.addressof(X), C.addressof(sad_p), C.addressof(sc_p), be.stream)
    l0 = n // n_refs  # items are reference-major: the first n / n_refs belong to list 0
    for lv in range(2):
        gs, gc = be.host(d_sads[lv]), be.host(d_scs[lv])
        assert np.array_equal(gs[:l0], fused[lv][2][:l0]) and np.array_equal(gc[:l0], fused[lv][3][:l0]), ("two-level chain, list 0", lv)
        assert np.array_equal(gs[l0:], mark_sad[l0:]) and np.array_equal(gc[l0:], mark_sc[l0:]), ("two-level chain, list 1 untouched", lv)
    assert np.array_equal(be.host(d_sads[2]), mark_sad) and np.array_equal(be.host(d_scs[2]), mark_sc), "level 2 untouched"


# ------------------------------------------------------------------ integer ME from HME results (set_final_seach_centre_sb + integer_search_b64)
class IntSearch(C.Structure):
    _fields_ = [("sa_min_w", C.c_int16), ("sa_min_h", C.c_int16), ("sa_max_w", C.c_int16), ("sa_max_h", C.c_int16), ("dist", C.c_uint16),
                ("mv_adj_enabled", C.c_uint8), ("mv_adj_nearest_ref_only", C.c_uint8), ("ref_pic_index", C.c_uint8), ("sub_sad", C.c_uint8),
                ("mv_size_th", C.c_uint16), ("sa_multiplier", C.c_uint16), ("divisor", C.c_uint32)]


INT_CASES = [dict(sa=(8, 3, 16, 9), dist=1, adj=(0, 0, 0, 1), r=0, sub=0, div=1), dict(sa=(8, 5, 64, 32), dist=3, adj=(1, 0, 6, 2), r=1, sub=0, div=1),
             dict(sa=(16, 9, 48, 24), dist=2, adj=(1, 1, 4, 2), r=0, sub=1, div=2), dict(sa=(8, 3, 24, 12), dist=5, adj=(1, 1, 4, 3), r=2, sub=0, div=4),
             dict(sa=(12, 7, 30, 20), dist=1, adj=(0, 0, 0, 1), r=0, sub=1, div=3)]


def int_params(c):
    P = IntSearch()
    P.sa_min_w, P.sa_min_h, P.sa_max_w, P.sa_max_h = c["sa"]
    P.dist, P.ref_pic_index, P.sub_sad, P.divisor = c["dist"], c["r"], c["sub"], c["div"]
    P.mv_adj_enabled, P.mv_adj_nearest_ref_only, P.mv_size_th, P.sa_multiplier = c["adj"]
    return P


def make_hme_results(g, n_items, regions, W, H):
    sad = g.integers(100, 100000, (n_items, regions)).astype(np.uint64)
    sad[::4, 1:] = sad[::4, :1]  # ties: the first region must win
    sc = np.stack([g.integers(-W // 3, W // 3, (n_items, regions)), g.integers(-H // 3, H // 3, (n_items, regions))], 2).astype(np.int16)
    sc[1::3] //= 16
    return sad, sc


@pytest.mark.parametrize("ci", range(len(INT_CASES)))
def test_me_integer_search_oracle_vs_reference(oracle, ref, ci):
    if not os.path.exists(REF_ME_LIB):
        pytest.skip("oracle/_ref/libsvtref_me.so not available")
    refme = C.CDLL(REF_ME_LIB)
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    c, g = INT_CASES[ci], rng(1900 + ci)
    P = int_params(c)
    W, H = 200, 136
    src, refs, w, h, org, stride = make_planes(g, W, H, 2, 1)
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    sbs_x, sbs_y, nw, nh = (aw + 63) // 64, (ah + 63) // 64, 2, 2
    sad, sc = make_hme_results(g, sbs_x * sbs_y, nw * nh, W, H)
    for sb in range(sbs_x * sbs_y):
        fx, fy = (sb % sbs_x) * 64, (sb // sbs_x) * 64
        o_sc, r_sc, o_sad, r_sad = np.zeros(2, np.int16), np.zeros(2, np.int16), C.c_uint64(0), C.c_uint64(0)
        area = np.zeros(4, np.int16)
        bs0, bm0, bs1, bm1 = (np.zeros(85, np.uint32) for _ in range(4))
        oracle.oracle_me_integer_search(C.byref(P), nw * nh, p(sad[sb]), p(sc[sb]), p(src), stride, org, org, p(refs[0]), stride, org, org, fx, fy, aw, ah, p(o_sc),
                                        C.byref(o_sad), p(area), p(bs0), p(bm0))
        refme.ref_me_integer_search(C.byref(P), nw, nh, p(sad[sb]), p(sc[sb]), p(src), stride, org, org, p(refs[0]), stride, org, org, W, H, fx, fy, aw, ah,
                                    p(r_sc), C.byref(r_sad), p(bs1), p(bm1))
        assert np.array_equal(o_sc, r_sc) and o_sad.value == r_sad.value, (ci, sb)
        assert np.array_equal(bs0, bs1) and np.array_equal(bm0, bm1), (ci, sb, area)


@pytest.mark.parametrize("ci", range(len(INT_CASES)))
def test_me_integer_search_hip(be, oracle, ci):
    pkg, c, g = load_pkg(), INT_CASES[ci], rng(2000 + ci)
    W, H = (200, 136) if not be.is_gpu else (712, 400)
    n_refs, nw, nh = 3, 2, 2
    src, refs, w, h, org, stride = make_planes(g, W, H, 2, n_refs)
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    sbs_x, sbs_y = (aw + 63) // 64, (ah + 63) // 64
    n_sb = sbs_x * sbs_y
    sad, sc = make_hme_results(g, n_refs * n_sb, nw * nh, W, H)
    dist, rpi = [c["dist"], max(1, c["dist"] - 1), c["dist"] + 1], [c["r"], 0, 1]
    do_ref = (g.random((n_sb, n_refs)) < 0.85).astype(np.uint8)
    do_ref8 = np.ones((n_sb, 2, 4), np.uint8)  # search_results[list][ref] layout: slot 0 = list 0, slots 1, 2 = list 1
    for r in range(n_refs):
        do_ref8[:, int(r >= 1), rpi[r]] = do_ref[:, r]
    div = np.where(g.random((n_sb, n_refs)) < 0.5, c["div"], 1).astype(np.uint32)
    P = pkg.MeIntegerSearchParams()
    P.sbs_x, P.sbs_y, P.n_refs, P.regions, P.aligned_width, P.aligned_height = sbs_x, sbs_y, n_refs, nw * nh, aw, ah
    P.sa_min_width, P.sa_min_height, P.sa_max_width, P.sa_max_height = c["sa"]
    P.sub_sad, P.mv_adj_enabled, P.mv_adj_nearest_ref_only, P.mv_adj_mv_size_th, P.mv_adj_sa_multiplier = c["sub"], *c["adj"]
    for r in range(n_refs):
        P.dist[r], P.ref_pic_index[r], P.ref_off[r] = dist[r], rpi[r], (1 + r) * src.size
    P.src_off, P.src_stride, P.ref_stride, P.ref_org_x, P.ref_org_y = org * stride + org, stride, stride, org, org
    P.n_refs_list0 = 1
    planes = np.stack([src] + refs)
    d_pl, d_sad, d_sc, d_do, d_div = be.dev(planes), be.dev(sad), be.dev(sc), be.dev(do_ref8), be.dev(div)
    d_bs, d_bm = be.empty((n_refs, n_sb, 85), np.uint32), be.empty((n_refs, n_sb, 85), np.uint32)
    d_sco, d_sado = be.empty((n_refs, n_sb, 2), np.int16), be.empty((n_refs, n_sb), np.uint64)
    d_ws = be.empty(be.lib.svt_hip_me_integer_search_workspace(C.addressof(P)), np.uint8)
    be.lib.svt_hip_me_integer_search_batch(C.addressof(P), be.ptr(d_pl), be.ptr(d_pl), be.ptr(d_sad), be.ptr(d_sc), be.ptr(d_do), be.ptr(d_div), None, be.ptr(d_bs),
                                           be.ptr(d_bm), be.ptr(d_sco), be.ptr(d_sado), be.ptr(d_ws), be.stream)
    bs, bm, sco, sado = be.host(d_bs), be.host(d_bm), be.host(d_sco), be.host(d_sado)
    checked = 0
    for r in range(n_refs):
        Q = int_params(dict(c, dist=dist[r], r=rpi[r]))
        for sb in (range(n_sb) if not be.is_gpu else g.choice(n_sb, 12, replace=False)):
            sb = int(sb)
            Q.divisor = int(div[sb, r])
            i = r * n_sb + sb
            o_sc, o_sad, area = np.zeros(2, np.int16), C.c_uint64(0), np.zeros(4, np.int16)
            ws_, wm_ = np.zeros(85, np.uint32), np.zeros(85, np.uint32)
            oracle.oracle_me_integer_search(C.byref(Q), nw * nh, p(sad[i]), p(sc[i]), p(src), stride, org, org, p(refs[r]), stride, org, org, (sb % sbs_x) * 64,
                                            (sb // sbs_x) * 64, aw, ah, p(o_sc), C.byref(o_sad), p(area), p(ws_), p(wm_))
            assert np.array_equal(sco[r, sb], o_sc) and int(sado[r, sb]) == o_sad.value, (ci, r, sb)
            if do_ref[sb, r]:
                assert np.array_equal(bs[r, sb], ws_) and np.array_equal(bm[r, sb], wm_), (ci, r, sb, area)
                checked += 1
    assert checked > 10


def test_me_session_stage_from_host_pictures(be, oracle):
    """The whole open-loop ME stage from host pictures (upload -> decimation -> HME 0-2 -> integer search -> MeSbResults) vs the same chain composed
    from the oracle's pieces (each pinned against the reference separately)."""
    import test_me_results as M
    pkg, g, lib = load_pkg(), rng(2100), be.lib
    W, H, PAD = (192, 136, 68) if not be.is_gpu else (448, 264, 68)
    stride, rows = W + 2 * PAD, H + 2 * PAD + 64
    n_pics, nw, nh = 4, 2, 2
    base = g.integers(0, 256, (rows + 16, stride + 16), dtype=np.uint8) // 3 + (np.add.outer(np.arange(rows + 16), np.arange(stride + 16)) * 5 % 170).astype(np.uint8)
    pics = []
    for k in range(n_pics):
        a = np.zeros((rows, stride), np.uint8)
        a[PAD:PAD + H, PAD:PAD + W] = base[8 + k:8 + k + H, 8 + 2 * k:8 + 2 * k + W] + g.integers(0, 4, (H, W), dtype=np.uint8)
        oracle.oracle_generate_padding(p(a), stride, W, H, PAD, PAD)  # what the encoder hands over: the padded input picture
        pics.append(a)
    sess = lib.svt_hip_me_session_create(W, H, stride, PAD, PAD, rows, 4, 3, 32, 16, 2)
    assert lib.svt_hip_me_session_enable_stage(sess, 32, 16, nw * nh, 32, 16) == 0
    S = pkg.MeStageParams()
    S.num_hme_sa_w, S.num_hme_sa_h, S.hme_sub_sampled, S.me_sub_sad = nw, nh, 0, 0
    for lv, (a, b) in enumerate(((16, 8), (8, 3), (8, 3))):
        S.hme_sa_width[lv], S.hme_sa_height[lv] = a, b
    S.me_sa_min_width, S.me_sa_min_height, S.me_sa_max_width, S.me_sa_max_height = 8, 3, 24, 12
    S.mv_adj_enabled, S.mv_adj_nearest_ref_only, S.mv_adj_mv_size_th, S.mv_adj_sa_multiplier = 1, 1, 4, 2
    dist, rpi = [1, 2, 3], [0, 1, 0]
    S.temporal_layer_gt0 = 1  # both lists go through HME (0 = base layer: list 1 starts the integer search at (0, 0))
    for r in range(3):
        S.dist[r], S.ref_pic_index[r] = dist[r], rpi[r]
    cfg = (2, 2, 1, 1, 1, 0, 0, 1, 30, 40, 0, 1, 1)
    R = M.make_params(pkg, cfg, 0, g)
    C.memmove(C.addressof(S.results), C.addressof(R), C.sizeof(R))
    for k in range(3):
        assert lib.svt_hip_me_session_submit_stage(sess, k, p(pics[k]), None, 0, C.addressof(S), None) >= 0
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    sbs_x, sbs_y = (aw + 63) // 64, (ah + 63) // 64
    n_sb = sbs_x * sbs_y
    out = dict(total=np.full((n_sb, 85), 7, np.uint8), mv=np.full((n_sb, 85 * R.max_refs), 7, np.uint32), cand=np.full((n_sb, 85 * R.max_cand), 7, np.uint8),
               stats=np.zeros(n_sb, pkg.MeSbStats), bs=np.zeros((3, n_sb, 85), np.uint32), bm=np.zeros((3, n_sb, 85), np.uint32))
    Hst = pkg.MeResultsHost(None, p(out["total"]), p(out["mv"]), p(out["cand"]), p(out["stats"]), p(out["bs"]), p(out["bm"]))
    refs = np.array([2, 1, 0], np.int64)  # list 0: pictures 2, 1; list 1: picture 0
    slot = lib.svt_hip_me_session_submit_stage(sess, 3, p(pics[3]), p(refs), 3, C.addressof(S), C.addressof(Hst))
    assert slot >= 0
    lib.svt_hip_me_session_wait(sess, slot)
    lib.svt_hip_me_session_destroy(sess)

    # ---- the same chain through the oracle
    def decimate(full):
        qw, qh, sw, sh = W // 2, H // 2, W // 4, H // 4
        q = np.zeros((qh + 64 + 32, qw + 64), np.uint8)
        oracle.oracle_downsample_2d(vp(full, PAD * stride + PAD), stride, W, H, vp(q, 32 * (qw + 64) + 32), qw + 64, 2)
        oracle.oracle_generate_padding(p(q), qw + 64, qw, qh, 32, 32)
        x = np.zeros((sh + 32 + 16, sw + 32), np.uint8)
        oracle.oracle_downsample_2d(vp(q, 32 * (qw + 64) + 32), qw + 64, qw, qh, vp(x, 16 * (sw + 32) + 16), sw + 32, 2)
        oracle.oracle_generate_padding(p(x), sw + 32, sw, sh, 16, 16)
        return {2: full, 1: q, 0: x}
    lv_planes = [decimate(pc) for pc in pics]
    order = [2, 1, 0]
    n_items = 3 * n_sb * nw * nh
    prev = np.zeros((n_items, 2), np.int16)
    for lv in (0, 1, 2):
        sh_ = 2 - lv
        org = {0: 16, 1: 32, 2: PAD}[lv]
        src = lv_planes[3][lv]
        rf = [lv_planes[i][lv] for i in order]
        pin = prev >> 1 if lv == 1 else prev
        sad, prev = cpu_level(oracle.oracle_hme_level, lv, 0, nw, nh, src, rf, W >> sh_, H >> sh_, org, src.shape[1], W, H, S.hme_sa_width[lv], S.hme_sa_height[lv], pin)
    sad = sad.reshape(3 * n_sb, nw * nh)
    sc = prev.reshape(3 * n_sb, nw * nh, 2)
    want_bs, want_bm = np.zeros((3, n_sb, 85), np.uint32), np.zeros((3, n_sb, 85), np.uint32)
    for r in range(3):
        Q = int_params(dict(sa=(8, 3, 24, 12), dist=dist[r], adj=(1, 1, 4, 2), r=rpi[r], sub=0, div=1))
        for sb in range(n_sb):
            i = r * n_sb + sb
            o_sc, o_sad, area = np.zeros(2, np.int16), C.c_uint64(0), np.zeros(4, np.int16)
            oracle.oracle_me_integer_search(C.byref(Q), nw * nh, p(sad[i]), p(sc[i]), p(pics[3]), stride, PAD, PAD, p(pics[order[r]]), stride, PAD, PAD,
                                            (sb % sbs_x) * 64, (sb // sbs_x) * 64, aw, ah, p(o_sc), C.byref(o_sad), p(area), vp(want_bs[r, sb]), vp(want_bm[r, sb]))
    assert np.array_equal(out["bs"], want_bs) and np.array_equal(out["bm"], want_bm)
    sb_size = np.array([[min(64, aw - 64 * (i % sbs_x)), min(64, ah - 64 * (i // sbs_x))] for i in range(n_sb)], np.uint8)
    R.n_sb = n_sb
    want = M.run_cpu(oracle.oracle_me_results_sb, pkg, R, cfg, want_bs, want_bm, np.ones((n_sb, 2, 4), np.uint8), sb_size, 0)
    M.same(want, (out["total"], out["mv"], out["cand"], out["stats"], None), "stage session")


def test_me_session_stage_pictures_in_flight(be, oracle):
    """Two pictures in flight (each on its own stream; the next upload only waits for a picture that still reads the ring entry it replaces) give the
    same MeSbResults as one picture at a time, for a ring with and without such conflicts."""
    import test_me_results as M
    pkg, g, lib = load_pkg(), rng(2101), be.lib
    W, H, PAD = (192, 136, 68) if not be.is_gpu else (640, 360, 68)
    stride, rows = W + 2 * PAD, H + 2 * PAD + 64
    n_pics = 6 if not be.is_gpu else 12
    pics = []
    for k in range(n_pics):
        a = np.zeros((rows, stride), np.uint8)
        a[PAD:PAD + H, PAD:PAD + W] = g.integers(0, 256, (H, W), dtype=np.uint8) // 2 + ((np.add.outer(np.arange(H), np.arange(W)) + 3 * k) % 120).astype(np.uint8)
        oracle.oracle_generate_padding(p(a), stride, W, H, PAD, PAD)
        pics.append(a)
    S = pkg.MeStageParams()
    S.num_hme_sa_w, S.num_hme_sa_h, S.hme_sub_sampled, S.me_sub_sad = 2, 2, 1, 1
    for lv, (a, b) in enumerate(((16, 8), (8, 3), (8, 3))):
        S.hme_sa_width[lv], S.hme_sa_height[lv] = a, b
    S.me_sa_min_width, S.me_sa_min_height, S.me_sa_max_width, S.me_sa_max_height = 8, 3, 24, 12
    S.me_early_exit_th, S.temporal_layer_gt0, S.is_ref = 64 * 64 * 8, 1, 1
    S.prehme_enabled, S.prehme_skip_search_line, S.prehme_l1_early_exit = 1, 1, 1
    for k, (mw, mh, xw, xh) in enumerate(((8, 20, 8, 60), (16, 5, 48, 5))):
        S.prehme_sa_min_width[k], S.prehme_sa_min_height[k], S.prehme_sa_max_width[k], S.prehme_sa_max_height[k] = mw, mh, xw, xh
    for r, (d, i) in enumerate(((1, 0), (2, 1), (3, 0))):
        S.dist[r], S.ref_pic_index[r] = d, i
    cfg = (2, 2, 1, 1, 1, 0, 0, 1, 30, 40, 0, 1, 1)
    R = M.make_params(pkg, cfg, 0, g)
    C.memmove(C.addressof(S.results), C.addressof(R), C.sizeof(R))
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    n_sb = ((aw + 63) // 64) * ((ah + 63) // 64)

    def run(ring, in_flight):
        sess = lib.svt_hip_me_session_create(W, H, stride, PAD, PAD, rows, ring, 3, 32, 16, 2)
        assert lib.svt_hip_me_session_enable_stage(sess, 32, 16, 4, 48, 16) == 0
        res, pending = [], []
        for k in range(n_pics):
            out = dict(total=np.full((n_sb, 85), 7, np.uint8), mv=np.full((n_sb, 85 * R.max_refs), 7, np.uint32), cand=np.full((n_sb, 85 * R.max_cand), 7, np.uint8),
                       stats=np.zeros(n_sb, pkg.MeSbStats), bs=np.zeros((3, n_sb, 85), np.uint32), bm=np.zeros((3, n_sb, 85), np.uint32))
            Hst = pkg.MeResultsHost(None, p(out["total"]), p(out["mv"]), p(out["cand"]), p(out["stats"]), p(out["bs"]), p(out["bm"]))
            refs = np.array([k - 1, k - 2, k - 3], np.int64)
            slot = lib.svt_hip_me_session_submit_stage(sess, k, p(pics[k]), p(refs) if k >= 3 else None, 3 if k >= 3 else 0, C.addressof(S),
                                                       C.addressof(Hst) if k >= 3 else None)
            assert slot >= 0, (ring, k, slot)
            res.append((out, Hst, refs))
            pending.append(slot)
            while len(pending) >= in_flight:
                lib.svt_hip_me_session_wait(sess, pending.pop(0))
        for slot in pending:
            lib.svt_hip_me_session_wait(sess, slot)
        lib.svt_hip_me_session_destroy(sess)
        return [r[0] for r in res]
    serial = run(6, 1)
    for ring in (4, 6):  # 4: every upload replaces an entry the picture before it still reads; 6: never
        got = run(ring, 2)
        for k in range(3, n_pics):
            for key in ("total", "mv", "cand", "stats", "bs", "bm"):
                assert np.array_equal(serial[k][key], got[k][key]), (ring, k, key)
    assert any(serial[k]["total"].min() < 7 for k in range(3, n_pics))


# ------------------------------------------------------ the device stage vs the reference's TOP-LEVEL svt_aom_motion_estimation_b64, SB by SB
class RefPlane(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("stride", C.c_uint32), ("org_x", C.c_uint32), ("org_y", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]


class RefPicture(C.Structure):
    _fields_ = [("lvl", RefPlane * 3), ("picture_number", C.c_uint64)]


class RefMeStageOptions(C.Structure):
    _fields_ = [("num_hme_sa_w", C.c_uint8), ("num_hme_sa_h", C.c_uint8), ("hme_sub_sampled", C.c_uint8), ("me_sub_sad", C.c_uint8),
                ("hme_l0_min_w", C.c_uint16), ("hme_l0_min_h", C.c_uint16), ("hme_l0_max_w", C.c_uint16), ("hme_l0_max_h", C.c_uint16),
                ("hme_l1_w", C.c_uint16), ("hme_l1_h", C.c_uint16), ("hme_l2_w", C.c_uint16), ("hme_l2_h", C.c_uint16),
                ("me_min_w", C.c_uint16), ("me_min_h", C.c_uint16), ("me_max_w", C.c_uint16), ("me_max_h", C.c_uint16),
                ("mv_adj_enabled", C.c_uint8), ("mv_adj_nearest_ref_only", C.c_uint8), ("mv_adj_mv_size_th", C.c_uint16), ("mv_adj_sa_multiplier", C.c_uint16),
                ("temporal_layer_index", C.c_uint8), ("is_ref", C.c_uint8), ("me_early_exit_th", C.c_uint32),
                ("sr_adjustment", C.c_uint8), ("me_8x8_var_enabled", C.c_uint8), ("me_sr_div4_th", C.c_uint32), ("me_sr_div2_th", C.c_uint32),
                ("me_sr_mult2_th", C.c_uint32), ("hme_prune_enabled", C.c_uint8), ("prune_ref_if_hme_sad_dev_bigger_than_th", C.c_uint16),
                ("reduce_me_sr_based_on_mv_length_th", C.c_uint16), ("stationary_hme_sad_abs_th", C.c_uint16), ("stationary_me_sr_divisor", C.c_uint16),
                ("reduce_me_sr_based_on_hme_sad_abs_th", C.c_uint16), ("me_sr_divisor_for_low_hme_sad", C.c_uint16), ("distance_based_hme_resizing", C.c_uint8),
                ("prehme_enabled", C.c_uint8), ("prehme_skip_search_line", C.c_uint8), ("prehme_l1_early_exit", C.c_uint8),
                ("prehme_sa_min_width", C.c_uint16 * 2), ("prehme_sa_min_height", C.c_uint16 * 2), ("prehme_sa_max_width", C.c_uint16 * 2),
                ("prehme_sa_max_height", C.c_uint16 * 2), ("zz_sad_th", C.c_uint32), ("phme_sad_th", C.c_uint32), ("zz_sad_pct", C.c_uint16),
                ("phme_sad_pct", C.c_uint16), ("prev_me_stage_based_exit_th", C.c_uint32), ("me_safe_limit_zz_th", C.c_uint32), ("me_type_mctf", C.c_uint32), ("tf_me_exit_th", C.c_uint32),
                ("hme_level2_off", C.c_uint8), ("pad", C.c_uint8), ("reduce_hme_l0_sr_th_min", C.c_uint16), ("reduce_hme_l0_sr_th_max", C.c_uint16)]


def scaled_distance(d):  # svt_aom_get_scaled_picture_distance (motion_estimation.c:1239-1243)
    return d * 5 // 8 + (1 if d % 8 else 0)


STAGE_OPTS = [dict(name="baseline"),
              dict(name="early_exit", me_early_exit_th=64 * 64 * 8),
              dict(name="early_exit_low", me_early_exit_th=64 * 64 * 3, l0=(32, 16, 96, 48)),
              dict(name="hme_prune_sr_adjust", hme_prune=25, sr=(1, 6, 9000, 4, 20000, 2), l0=(32, 16, 64, 32)),
              dict(name="all", me_early_exit_th=64 * 64 * 6, hme_prune=40, sr=(1, 4, 12000, 3, 30000, 2), l0=(48, 24, 96, 48)),
              dict(name="is_ref", is_ref=1),
              dict(name="var_probe_lvl2", var=(80000, 150000, 0xffffffff), me=(16, 9, 48, 24)),
              dict(name="var_probe_lvl1", var=(0, 0, 900000), me=(16, 9, 32, 16)),
              dict(name="var_probe_mid", var=(2000, 20000, 200000), me=(16, 9, 32, 16), is_ref=1, hme_prune=40, sr=(1, 4, 12000, 3, 30000, 2)),
              dict(name="preset8_like", me_early_exit_th=64 * 64 * 8, var=(80000, 150000, 0xffffffff), me=(16, 9, 32, 16), is_ref=1, hme_prune=30,
                   sr=(1, 4, 12000, 3, 30000, 2), l0=(32, 16, 96, 48)),
              dict(name="zz_gate_inactive", zz=(20 * 64 * 64, 5)),  # init_zz_sad only runs with me_early_exit_th
              dict(name="zz_gate", zz=(20 * 64 * 64, 30), me_early_exit_th=64 * 64 * 3),
              dict(name="prehme", prehme=dict(skip=0, l1=0, sa=((8, 20, 8, 40), (24, 3, 48, 3)))),
              dict(name="prehme_lvl4", prehme=dict(skip=1, l1=1, sa=((8, 24, 8, 48), (16, 7, 32, 7)), phme=(10 * 64 * 64, 5)), zz=(20 * 64 * 64, 5)),
              dict(name="prev_stage_exit", prev_stage=64 * 64 * 4),
              dict(name="prev_stage_exit_high", prev_stage=64 * 64 * 40, me_early_exit_th=64 * 64 * 3, sub=1),
              dict(name="prev_stage_exit_prehme", prev_stage=64 * 64 * 24, prehme=dict(skip=1, l1=1, sa=((8, 24, 8, 48), (16, 7, 32, 7)), phme=(10 * 64 * 64, 5)),
                   hme_prune=30, sr=(1, 4, 12000, 3, 30000, 2)),
              dict(name="tf_me_like", prev_stage=64 * 64 * 4, me_early_exit_th=0, sub=1, me=(8, 5, 16, 9), l0=(16, 16, 32, 32)),
              dict(name="safe_limit_zz", safe_zz=64 * 64 * 6),
              dict(name="safe_limit_zz_with_gate", safe_zz=64 * 64 * 5, zz=(20 * 64 * 64, 5), me_early_exit_th=64 * 64 * 3),
              # the temporal filter's form of the call (ME_MCTF): unscaled distances, no pruning, raw tables only, early exit on the first reference's HME SAD
              dict(name="mctf", mctf=0, prev_stage=64 * 64 * 4, sub=1, me=(8, 5, 16, 9), l0=(16, 16, 32, 32)),
              dict(name="mctf_exit", mctf=12000, prev_stage=64 * 64 * 4, me=(8, 5, 16, 9)),
              dict(name="mctf_exit_prehme", mctf=20000, sub=1, prehme=dict(skip=1, l1=1, sa=((8, 24, 8, 48), (16, 7, 32, 7))), me_early_exit_th=64 * 64 * 3),
              dict(name="preset8", me_early_exit_th=64 * 64 * 8, var=(80000, 150000, 0xffffffff), me=(16, 9, 32, 16), is_ref=1, hme_prune=5,
                   sr=(1, 4, 12000, 8, 12000, 8), l0=(32, 32, 96, 96), zz=(20 * 64 * 64, 5), sub=1,
                   prehme=dict(skip=1, l1=1, sa=((8, 24, 8, 48), (16, 7, 32, 7)), phme=(10 * 64 * 64, 5))),
              # what the reference's own derivation gives at preset 8 (svt_aom_sig_deriv_me, enc_mode_config.c:681-815; HME flags :1630-1640): HME level 2 is OFF,
              # the level-0 area shrinks with the reference index (distance_based_hme_resizing), base-layer pictures search two lists without HME for list 1
              dict(name="two_levels", levels=2),
              dict(name="level0_only", levels=1),
              dict(name="mctf_level0_only", levels=1, mctf=2000, me=(8, 5, 16, 9)),  # the temporal filter's ME at tf_ctrls.hme_me_level 3 / 4 (enc_mode_config.c:1655-1661)
              dict(name="two_levels_exits", levels=2, prev_stage=64 * 64 * 24, me_early_exit_th=64 * 64 * 3, sub=1, hme_prune=30, sr=(1, 4, 3000, 8, 3000, 8)),
              dict(name="l0_resize_by_ref_index", dbr=1, sr=(1, 4, 12000, 8, 12000, 8), l0=(32, 32, 96, 96)),
              # the low-delay settings: level-0 areas of every slot but (list 0, reference 0) resized from that slot's level-0 motion of the same SB (enc_mode_config.c:702-714)
              dict(name="l0_resize_from_list0_motion", dbr=1, sr=(1, 4, 12000, 8, 12000, 8), l0=(64, 48, 128, 96), l0_th=(8, 12)),
              dict(name="l0_resize_from_list0_motion_mixed", dbr=1, sr=(1, 4, 12000, 8, 12000, 8), l0=(16, 8, 32, 16), l0_th=(30, 2), levels=2, motion=5),  # thresholds under which SBs take all four width / height combinations
              # enable_me_sr_adjustment == 2 (the screen-content levels 4 / 5, enc_mode_config.c:485-505): the height halves on a good HME result, else both sides halve for the
              # slots after the first when the first slot's final 64x64 SAD is small (motion_estimation.c:1349-1364) -- the first slot is searched before the others' geometry
              dict(name="sr2", sr=(2, 16, 20000, 8, 20000, 8), me=(16, 9, 48, 24)),
              dict(name="sr2_is_ref_clean", sr=(2, 16, 20000, 8, 20000, 8), me=(16, 10, 48, 24), is_ref=1, noise=(1, 1)),  # noise-free pictures: HME SADs below 24 * 24, check_00_center's too
              dict(name="sr2_is_ref_var_probe", sr=(2, 4, 12000, 2, 6000, 2), me=(24, 12, 48, 24), is_ref=1, noise=(2, 1), var=(2000, 20000, 200000), hme_prune=40, levels=2),
              dict(name="sr2_early_exit", sr=(2, 16, 20000, 8, 20000, 8), me=(16, 9, 48, 24), me_early_exit_th=64 * 64 * 8, is_ref=1),  # with me_early_exit_th the two rules are off
              dict(name="sr2_l0_still", dbr=1, sr=(2, 16, 20000, 8, 20000, 8), l0=(64, 48, 128, 96), l0_th=(3, 12), me=(16, 9, 48, 24)),  # + level-0 areas / (4 + index) where list 0's motion is small
              dict(name="sr2_l0_still_small_areas", dbr=1, sr=(2, 16, 20000, 8, 20000, 8), l0=(16, 8, 32, 16), l0_th=(6, 30), motion=4, levels=1, me=(8, 3, 24, 12)),  # ... areas small enough for the (4 + index) divisor to lose the motion (the mutation without the rule fails here)
              dict(name="base_layer", tl=0),
              dict(name="base_layer_prune", tl=0, hme_prune=25, sr=(1, 4, 12000, 8, 12000, 8), me_early_exit_th=64 * 64 * 8, zz=(20 * 64 * 64, 5), is_ref=1,
                   prehme=dict(skip=1, l1=1, sa=((8, 24, 8, 48), (16, 7, 32, 7)), phme=(10 * 64 * 64, 5))),
              dict(name="base_layer_is_ref", tl=0, is_ref=1, var=(80000, 150000, 0xffffffff), me=(16, 9, 32, 16), hme_prune=5, levels=2),
              dict(name="preset8_derived", levels=2, dbr=1, me_early_exit_th=64 * 64 * 8, var=(80000, 150000, 0xffffffff), me=(8, 3, 8, 4), is_ref=1, hme_prune=5,
                   sr=(1, 4, 3000, 8, 3000, 8), l0=(16, 16, 192, 192), zz=(20 * 64 * 64, 5), sub=1,
                   prehme=dict(skip=1, l1=1, sa=((8, 100, 8, 350), (32, 7, 128, 7)), phme=(10 * 64 * 64, 5))),
              dict(name="preset8_derived_base_layer", tl=0, levels=2, dbr=1, me_early_exit_th=64 * 64 * 8, var=(80000, 150000, 0xffffffff), me=(8, 3, 8, 4), is_ref=1,
                   hme_prune=80, sr=(1, 4, 3000, 8, 3000, 8), l0=(16, 16, 192, 192), sub=1,
                   prehme=dict(skip=1, l1=1, sa=((8, 100, 8, 350), (32, 7, 128, 7))))]


@pytest.mark.parametrize("oi", range(len(STAGE_OPTS)))
def test_me_stage_vs_reference_motion_estimation_b64(be, oracle, ref, oi):
    """End to end: host pictures -> svt_hip_me_session_submit_stage vs the reference's own svt_aom_motion_estimation_b64 run on every SB (zero-motion
    SAD gating, HME 0-2 with per-reference level-0 areas, final centre, HME-based reference pruning and search-range divisors, integer search,
    reference pruning on ME SADs, MeSbResults, distortion statistics, GM flags)."""
    opt = STAGE_OPTS[oi]
    if not os.path.exists(REF_ME_LIB):
        pytest.skip("oracle/_ref/libsvtref_me.so not available")
    import test_me_results as M
    refme = C.CDLL(REF_ME_LIB)
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    pkg, g, lib = load_pkg(), rng(2200), be.lib
    W, H, PAD = (192, 136, 68) if not be.is_gpu else (448, 264, 68)
    stride, rows = W + 2 * PAD, H + 2 * PAD + 64
    nw, nh = 2, 2
    base = g.integers(0, 256, (rows + 80, stride + 80), dtype=np.uint8) // 3 + (np.add.outer(np.arange(rows + 80), np.arange(stride + 80)) * 5 % 170).astype(np.uint8)
    pics = []
    mo = opt.get("motion", 1)  # displacement between consecutive pictures: (mo, 2 mo) samples (large values: the level-0 centres leave zero)
    for k in range(4):
        a = np.zeros((rows, stride), np.uint8)
        nz = opt.get("noise", (4, 2))  # noise amplitudes of the moving / the static region (1 = none)
        a[PAD:PAD + H, PAD:PAD + W] = base[8 + k * mo:8 + k * mo + H, 8 + 2 * k * mo:8 + 2 * k * mo + W] + g.integers(0, nz[0], (H, W), dtype=np.uint8)
        # a static region (zero-motion SAD small enough for the early exits) next to the moving one
        a[PAD:PAD + H, PAD:PAD + 100] = base[8:8 + H, 8:8 + 100] + g.integers(0, nz[1], (H, 100), dtype=np.uint8)
        oracle.oracle_generate_padding(p(a), stride, W, H, PAD, PAD)
        pics.append(a)
    numbers = {0: 41, 1: 38, 2: 39, 3: 40}  # picture 3 is the source; list 0 = pictures 2, 1 (past), list 1 = picture 0 (future)
    order = [2, 1, 0]
    dist = [scaled_distance(abs(numbers[3] - numbers[i])) for i in order]
    sess = lib.svt_hip_me_session_create(W, H, stride, PAD, PAD, rows, 4, 3, 48, 24, 2)
    assert lib.svt_hip_me_session_enable_stage(sess, 32, 16, nw * nh, 96, 48) == 0
    S = pkg.MeStageParams()
    S.num_hme_sa_w, S.num_hme_sa_h, S.hme_sub_sampled, S.me_sub_sad = nw, nh, 0, 0
    for lv, (a, b) in enumerate(((16, 8), (8, 3), (8, 3))):
        S.hme_sa_width[lv], S.hme_sa_height[lv] = a, b
    me_sa = opt.get("me", (8, 3, 24, 12))
    S.me_sa_min_width, S.me_sa_min_height, S.me_sa_max_width, S.me_sa_max_height = me_sa
    S.mv_adj_enabled, S.mv_adj_nearest_ref_only, S.mv_adj_mv_size_th, S.mv_adj_sa_multiplier = 1, 1, 4, 2
    S.is_ref = opt.get("is_ref", 0)
    S.temporal_layer_gt0 = 1 if opt.get("tl", 1) > 0 else 0
    S.hme_levels = opt.get("levels", 0)
    S.hme_sub_sampled = S.me_sub_sad = opt.get("sub", 0)
    if "zz" in opt:
        S.zz_sad_th, S.zz_sad_pct = opt["zz"]
    if "prehme" in opt:
        ph = opt["prehme"]
        S.prehme_enabled, S.prehme_skip_search_line, S.prehme_l1_early_exit = 1, ph["skip"], ph["l1"]
        for k in range(2):
            S.prehme_sa_min_width[k], S.prehme_sa_min_height[k], S.prehme_sa_max_width[k], S.prehme_sa_max_height[k] = ph["sa"][k]
        if "phme" in ph:
            S.phme_sad_th, S.phme_sad_pct = ph["phme"]
    S.prev_me_stage_based_exit_th = opt.get("prev_stage", 0)
    if "mctf" in opt:
        S.me_type_mctf, S.tf_me_exit_th = 1, opt["mctf"]
    S.me_safe_limit_zz_th = opt.get("safe_zz", 0)
    if "var" in opt:
        S.me_8x8_var_enabled, (S.me_sr_div4_th, S.me_sr_div2_th, S.me_sr_mult2_th) = 1, opt["var"]
    rpi = [0, 1, 0]
    for r in range(3):
        S.dist[r], S.ref_pic_index[r] = (abs(numbers[3] - numbers[order[r]]) if "mctf" in opt else dist[r]), rpi[r]
    l0 = opt.get("l0", (32, 16, 32, 16))  # total level-0 area: min w, min h, max w, max h (hme_l0_sa)
    S.hme_l0_per_ref = 1
    for r in range(3):  # get_hme_l0_search_area (:1806-1866); distance-based resizing (non-RTC form, :1847-1854) divides the base area by 1 + the reference index
        b = [v // (1 + rpi[r]) for v in l0] if opt.get("dbr") else l0
        S.hme_l0_sa_width_ref[r] = min((((b[0] // nw) * dist[r]) + 15) & ~15, ((b[2] // nw) + 15) & ~15)
        S.hme_l0_sa_height_ref[r] = min((b[1] // nh) * dist[r], b[3] // nh)
        b2 = [v // (2 + rpi[r]) for v in l0]  # the (2 + index) divisors of the low-delay resizing (:1829-1850)
        S.hme_l0_sa_width_ref2[r] = min((((b2[0] // nw) * dist[r]) + 15) & ~15, ((b2[2] // nw) + 15) & ~15)
        S.hme_l0_sa_height_ref2[r] = min((b2[1] // nh) * dist[r], b2[3] // nh)
        b4 = [v // (4 + rpi[r]) for v in l0]  # enable_me_sr_adjustment == 2: small motion on both axes (:1836-1841)
        S.hme_l0_sa_width_ref4[r] = min((((b4[0] // nw) * dist[r]) + 15) & ~15, ((b4[2] // nw) + 15) & ~15)
        S.hme_l0_sa_height_ref4[r] = min((b4[1] // nh) * dist[r], b4[3] // nh)
    if "l0_th" in opt:
        S.reduce_hme_l0_sr_th_min, S.reduce_hme_l0_sr_th_max = opt["l0_th"]
    S.me_early_exit_th = opt.get("me_early_exit_th", 0)
    if "hme_prune" in opt:
        S.hme_prune_enabled, S.prune_ref_if_hme_sad_dev_bigger_than_th = 1, opt["hme_prune"]
    if "sr" in opt:
        (S.sr_adjustment, S.reduce_me_sr_based_on_mv_length_th, S.stationary_hme_sad_abs_th, S.stationary_me_sr_divisor, S.reduce_me_sr_based_on_hme_sad_abs_th,
         S.me_sr_divisor_for_low_hme_sad) = opt["sr"]
    cfg = (2, 2, 1, 1, 1, 0, 0, 1, 30, 40, 0, 1, 1)
    R = M.make_params(pkg, cfg, 0, g)
    R.picture_number = numbers[3]
    for r in range(2):
        R.ref_picture_number[0][r] = numbers[order[r]]
    R.ref_picture_number[1][0] = numbers[order[2]]
    C.memmove(C.addressof(S.results), C.addressof(R), C.sizeof(R))
    for k in range(3):
        assert lib.svt_hip_me_session_submit_stage(sess, k, p(pics[k]), None, 0, C.addressof(S), None) >= 0
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    sbs_x, sbs_y = (aw + 63) // 64, (ah + 63) // 64
    n_sb = sbs_x * sbs_y
    out = dict(total=np.zeros((n_sb, 85), np.uint8), mv=np.zeros((n_sb, 85 * R.max_refs), np.uint32), cand=np.zeros((n_sb, 85 * R.max_cand), np.uint8),
               stats=np.zeros(n_sb, pkg.MeSbStats), bs=np.zeros((3, n_sb, 85), np.uint32), bm=np.zeros((3, n_sb, 85), np.uint32),
               do_ref=np.ones((n_sb, 2, 4), np.uint8))
    out["hme_sc"], out["hme_sad"] = np.full((3, n_sb, 2), 77, np.int16), np.zeros((3, n_sb), np.uint64)
    Hst = pkg.MeResultsHost(p(out["do_ref"]), p(out["total"]), p(out["mv"]), p(out["cand"]), p(out["stats"]), p(out["bs"]), p(out["bm"]), p(out["hme_sc"]), p(out["hme_sad"]))
    if "mctf" in opt:  # raw tables only
        Hst = pkg.MeResultsHost(None, None, None, None, None, p(out["bs"]), p(out["bm"]), p(out["hme_sc"]), p(out["hme_sad"]))
    slot = lib.svt_hip_me_session_submit_stage(sess, 3, p(pics[3]), p(np.array(order, np.int64)), 3, C.addressof(S), C.addressof(Hst))
    assert slot >= 0
    lib.svt_hip_me_session_wait(sess, slot)
    lib.svt_hip_me_session_destroy(sess)

    # ---- the reference, SB by SB, on planes decimated by the (pinned) oracle
    def decimate(full):
        qw, qh, sw, sh = W // 2, H // 2, W // 4, H // 4
        q = np.zeros((qh + 64 + 32, qw + 64), np.uint8)
        oracle.oracle_downsample_2d(vp(full, PAD * stride + PAD), stride, W, H, vp(q, 32 * (qw + 64) + 32), qw + 64, 2)
        oracle.oracle_generate_padding(p(q), qw + 64, qw, qh, 32, 32)
        x = np.zeros((sh + 32 + 16, sw + 32), np.uint8)
        oracle.oracle_downsample_2d(vp(q, 32 * (qw + 64) + 32), qw + 64, qw, qh, vp(x, 16 * (sw + 32) + 16), sw + 32, 2)
        oracle.oracle_generate_padding(p(x), sw + 32, sw, sh, 16, 16)
        return [x, q, full]
    lv = [decimate(pc) for pc in pics]
    if opt.get("me_early_exit_th"):  # the static first SB column really takes the early exits, the moving one does not
        zz = lambda x0: 2 * int(np.abs(pics[3][PAD:PAD + 64:2, PAD + x0:PAD + x0 + 64].astype(np.int32) - pics[2][PAD:PAD + 64:2, PAD + x0:PAD + x0 + 64]).sum())
        assert zz(0) < opt["me_early_exit_th"] // 4 < zz(128), (zz(0), zz(128))  # (with the 64*64*8 threshold also below th / 6: single-point ME)

    def picture(i):
        P_ = RefPicture()
        for k, (o, sh_) in enumerate(((16, 2), (32, 1), (PAD, 0))):
            a = lv[i][k]
            P_.lvl[k] = RefPlane(a.ctypes.data, a.shape[1], o, o, W >> sh_, H >> sh_)
        P_.picture_number = numbers[i]
        return P_
    refs = (RefPicture * 8)()
    refs[0], refs[1], refs[4] = picture(2), picture(1), picture(0)
    srcp = picture(3)
    O = RefMeStageOptions()
    O.num_hme_sa_w, O.num_hme_sa_h = nw, nh
    O.hme_l0_min_w, O.hme_l0_min_h, O.hme_l0_max_w, O.hme_l0_max_h = l0
    O.me_early_exit_th = opt.get("me_early_exit_th", 0)
    if "hme_prune" in opt:
        O.hme_prune_enabled, O.prune_ref_if_hme_sad_dev_bigger_than_th = 1, opt["hme_prune"]
    if "sr" in opt:
        (O.sr_adjustment, O.reduce_me_sr_based_on_mv_length_th, O.stationary_hme_sad_abs_th, O.stationary_me_sr_divisor, O.reduce_me_sr_based_on_hme_sad_abs_th,
         O.me_sr_divisor_for_low_hme_sad) = opt["sr"]
    O.hme_l1_w, O.hme_l1_h, O.hme_l2_w, O.hme_l2_h = 8, 3, 8, 3
    O.me_min_w, O.me_min_h, O.me_max_w, O.me_max_h = me_sa
    O.mv_adj_enabled, O.mv_adj_nearest_ref_only, O.mv_adj_mv_size_th, O.mv_adj_sa_multiplier = 1, 1, 4, 2
    O.temporal_layer_index, O.is_ref = opt.get("tl", 1), opt.get("is_ref", 0)
    O.hme_level2_off = 3 - opt.get("levels", 3)  # 0: three levels, 1: levels 0 and 1, 2: level 0 only
    O.distance_based_hme_resizing = opt.get("dbr", 0)
    if "l0_th" in opt:
        O.reduce_hme_l0_sr_th_min, O.reduce_hme_l0_sr_th_max = opt["l0_th"]
    O.hme_sub_sampled = O.me_sub_sad = opt.get("sub", 0)
    if "zz" in opt:
        O.zz_sad_th, O.zz_sad_pct = opt["zz"]
    if "prehme" in opt:
        ph = opt["prehme"]
        O.prehme_enabled, O.prehme_skip_search_line, O.prehme_l1_early_exit = 1, ph["skip"], ph["l1"]
        for k in range(2):
            O.prehme_sa_min_width[k], O.prehme_sa_min_height[k], O.prehme_sa_max_width[k], O.prehme_sa_max_height[k] = ph["sa"][k]
        if "phme" in ph:
            O.phme_sad_th, O.phme_sad_pct = ph["phme"]
    O.prev_me_stage_based_exit_th = opt.get("prev_stage", 0)
    if "mctf" in opt:
        O.me_type_mctf, O.tf_me_exit_th = 1, opt["mctf"]
    O.me_safe_limit_zz_th = opt.get("safe_zz", 0)
    if "var" in opt:
        O.me_8x8_var_enabled, (O.me_sr_div4_th, O.me_sr_div2_th, O.me_sr_mult2_th) = 1, opt["var"]
    n_exit = 0
    for sb in range(n_sb):
        tot, mvs, cands = np.zeros(85, np.uint8), np.zeros(85 * R.max_refs, np.uint32), np.zeros(85 * R.max_cand, np.uint8)
        st, bs, bm, dr = np.zeros(1, pkg.MeSbStats), np.zeros((2, 4, 85), np.uint32), np.zeros((2, 4, 85), np.uint32), np.zeros((2, 4), np.uint8)
        refme.ref_motion_estimation_b64(C.byref(O), C.byref(R), C.byref(srcp), C.byref(refs), W, H, (sb % sbs_x) * 64, (sb // sbs_x) * 64, p(tot), p(mvs), p(cands),
                                        p(st), p(bs), p(bm), p(dr))
        h_sc, h_sad, tf = np.zeros((2, 4, 2), np.int16), np.zeros((2, 4), np.uint32), np.zeros(3, np.uint32)
        refme.ref_me_last_hme(p(h_sc), p(h_sad), p(tf))
        for r, (l, ri) in enumerate(((0, 0), (0, 1), (1, 0))):  # search_results[list][ref].hme_sc_x / hme_sc_y / hme_sad
            assert np.array_equal(out["hme_sc"][r, sb], h_sc[l, ri]), ("hme centre", opt["name"], sb, r, out["hme_sc"][r, sb], h_sc[l, ri])
            if "mctf" in opt:  # (otherwise me_prune_ref has reused the field as its accumulator, :1529-1560)
                assert int(out["hme_sad"][r, sb]) == int(h_sad[l, ri]), ("hme sad", opt["name"], sb, r, out["hme_sad"][r, sb], h_sad[l, ri])
        if "mctf" in opt:
            exited = opt["mctf"] != 0 and int(h_sad[0, 0]) < opt["mctf"]
            assert (tf[0] == 255) == exited, ("tf exit", sb, tf, h_sad[0, 0])
            n_exit += exited
            horz = abs(int(h_sc[0, 0, 0])) > abs(int(h_sc[0, 0, 1]))  # tf_tot_horz_blks / tf_tot_vert_blks (hme_b64 :2469-2474) follow from the returned centres
            assert (int(tf[1]), int(tf[2])) == ((1, 0) if horz else (0, 1))
            if not exited:
                for r, (l, ri) in enumerate(((0, 0), (0, 1), (1, 0))):
                    assert np.array_equal(out["bs"][r, sb], bs[l, ri]) and np.array_equal(out["bm"][r, sb], bm[l, ri]), ("mctf tables", opt["name"], sb, r)
            continue
        assert np.array_equal(out["do_ref"][sb, 0, :2], dr[0, :2]) and out["do_ref"][sb, 1, 0] == dr[1, 0], ("do_ref", opt["name"], sb, out["do_ref"][sb], dr)
        for r, (l, ri) in enumerate(((0, 0), (0, 1), (1, 0))):
            if dr[l, ri]:  # a pruned reference is not searched by the reference: its tables are don't-care
                assert np.array_equal(out["bs"][r, sb], bs[l, ri]) and np.array_equal(out["bm"][r, sb], bm[l, ri]), ("tables", opt["name"], sb, r)
        assert np.array_equal(out["total"][sb], tot) and np.array_equal(out["cand"][sb], cands) and np.array_equal(out["mv"][sb], mvs), ("MeSbResults", sb)
        assert out["stats"][sb] == st[0], ("stats", sb, out["stats"][sb], st[0])
    if opt.get("mctf"):
        assert 0 < n_exit < n_sb, (n_exit, n_sb)  # both outcomes of the temporal filter's early exit were exercised


@pytest.mark.parametrize("qp", [20, 35, 50, 63])
@pytest.mark.parametrize("tl", [0, 1, 3])
def test_m8_me_settings_vs_reference(ref, qp, tl):
    """The preset-8 ME settings bench.py and the 1080p stage test use (pkg.m8_me_settings) == what the reference's own svt_aom_sig_deriv_me derives
    (enc_mode_config.c:681-815) for ENC_M8 at 1080p and 4K, every field ('preset 8' must be the reference's derivation, not a reading)."""
    if not os.path.exists(REF_ME_LIB):
        pytest.skip("oracle/_ref/libsvtref_me.so not available")
    refme = C.CDLL(REF_ME_LIB)
    pkg = load_pkg()
    for res in (4, 5):  # INPUT_SIZE_1080p_RANGE, INPUT_SIZE_4K_RANGE
        O, ex = RefMeStageOptions(), (C.c_int32 * 6)()
        refme.ref_sig_deriv_me(8, res, qp, 0, tl, 4, 0, C.byref(O), ex)
        m = pkg.m8_me_settings(qp, tl)
        if res == 5:  # 4K only differs in nothing set_me_search_params reads above M6 with 5 hierarchical levels (:311-318)
            pass
        assert (O.num_hme_sa_w, O.num_hme_sa_h) == m["num_hme_sa"] and (3 - O.hme_level2_off) == m["hme_levels"] and ex[3] == 0 and ex[4] == 1
        assert (O.hme_l0_min_w, O.hme_l0_min_h, O.hme_l0_max_w, O.hme_l0_max_h) == m["hme_l0"]
        assert (O.hme_l1_w, O.hme_l1_h) == m["hme_l1"] and (O.hme_l2_w, O.hme_l2_h) == m["hme_l2"]
        assert (O.me_min_w, O.me_min_h, O.me_max_w, O.me_max_h) == m["me"], (qp, (O.me_min_w, O.me_min_h, O.me_max_w, O.me_max_h), m["me"])
        assert O.hme_sub_sampled == O.me_sub_sad == m["sub_sampled"]
        ph = m["prehme"]
        assert (O.prehme_enabled, O.prehme_skip_search_line, O.prehme_l1_early_exit) == (1, ph["skip"], ph["l1"])
        for k in range(2):
            assert (O.prehme_sa_min_width[k], O.prehme_sa_min_height[k], O.prehme_sa_max_width[k], O.prehme_sa_max_height[k]) == ph["sa"][k]
        assert (O.hme_prune_enabled, O.prune_ref_if_hme_sad_dev_bigger_than_th, ex[0] & 0xffff, ex[2]) == (1, m["hme_prune"], m["me_prune"], 1)
        assert (O.zz_sad_th, O.zz_sad_pct) == m["zz"] and (O.phme_sad_th, O.phme_sad_pct) == m["phme"]
        sr = m["sr"]
        assert (O.sr_adjustment, O.reduce_me_sr_based_on_mv_length_th, O.stationary_hme_sad_abs_th, O.stationary_me_sr_divisor, O.reduce_me_sr_based_on_hme_sad_abs_th,
                O.me_sr_divisor_for_low_hme_sad, O.distance_based_hme_resizing) == (sr["level"], sr["mv_length_th"], sr["stationary_th"], sr["stationary_div"],
                                                                                  sr["low_sad_th"], sr["low_sad_div"], sr["distance_based"])
        assert O.mv_adj_enabled == m["mv_adj"] and (O.me_8x8_var_enabled, O.me_sr_div4_th, O.me_sr_div2_th, O.me_sr_mult2_th) == (1,) + m["var"]
        assert ex[1] == m["prune_me_candidates_th"] and O.me_early_exit_th == m["me_early_exit_th"] and O.prev_me_stage_based_exit_th == m["prev_me_stage_based_exit_th"]
