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
    be.lib.svt_hip_hme_level_batch(C.addressof(P), be.ptr(d_pl), be.ptr(d_pl), be.ptr(d_prev), be.ptr(d_sad), be.ptr(d_sc), be.ptr(d_ws), be.stream)
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
        be.lib.svt_hip_hme_level_batch(C.addressof(P), be.ptr(d_pl), be.ptr(d_pl), be.ptr(d_prev), be.ptr(d_sad), be.ptr(d_sc), be.ptr(d_ws), be.stream)
        assert np.array_equal(be.host(d_sad), sad_o) and np.array_equal(be.host(d_sc), prev_o), lv
        d_prev = d_sc
    assert np.abs(prev_o).max() > 0
