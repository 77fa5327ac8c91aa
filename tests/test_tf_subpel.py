"""Temporal filter sub-pel refinement (temporal_filtering.c:1560-2250): the luma motion compensation (8-tap regular / bilinear,
svt_aom_simple_luma_unipred) + block variance + the half / quarter / eighth-pel rings with their early exits.
  * oracle (oracle/oracle_tf_subpel.c) pinned against the reference's own svt_aom_simple_luma_unipred and (static) tf_subpel_search, compiled where they lie
    through oracle/ref_wrap/ref_tf_subpel.c;
  * device: svt_hip_tf_subpel_search_batch vs the oracle (emulator here, MI355X with -m gpu)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import REF_LIB, ROOT, load_pkg, p, rng

REF_ME_LIB = os.path.join(ROOT, "oracle", "_ref", "libsvtref_me.so")


class SubpelParams(C.Structure):
    _fields_ = [("half_pel_mode", C.c_uint8), ("quarter_pel_mode", C.c_uint8), ("eight_pel_mode", C.c_uint8), ("subsampling_shift", C.c_uint8),
                ("bit_depth", C.c_uint8), ("pad", C.c_uint8 * 3), ("early_exit_th", C.c_uint32), ("mi_rows", C.c_uint32), ("mi_cols", C.c_uint32),
                ("ref_org_x", C.c_uint32), ("ref_org_y", C.c_uint32), ("ref_stride", C.c_uint32)]


def make_pictures(g, W, H, PAD, bd, shift=(3, 2)):
    """source picture + padded reference = the source displaced by a fractional amount (smooth content so that sub-pel positions matter) + noise"""
    yy, xx = np.mgrid[0:H + 2 * PAD, 0:W + 2 * PAD].astype(np.float64)
    amp = (1 << bd) - 1
    tex = lambda x, y: 0.5 + 0.22 * np.sin(x / 3.1) * np.cos(y / 4.3) + 0.2 * np.sin((x + 2 * y) / 7.7) + 0.05 * np.sin(x * 1.9)  # noqa: E731
    ref = np.clip(tex(xx, yy) * amp + g.normal(0, amp / 200, xx.shape), 0, amp)
    src = np.clip(tex(xx[PAD:PAD + H, PAD:PAD + W] + shift[0] + 0.375, yy[PAD:PAD + H, PAD:PAD + W] + shift[1] - 0.25) * amp + g.normal(0, amp / 200, (H, W)), 0, amp)
    dt = np.uint16 if bd > 8 else np.uint8
    return np.ascontiguousarray(src.astype(dt)), np.ascontiguousarray(ref.astype(dt))


def params(mode, ss, bd, th, W, H, PAD, stride):
    P = SubpelParams()
    P.half_pel_mode, P.quarter_pel_mode, P.eight_pel_mode = mode
    P.subsampling_shift, P.bit_depth, P.early_exit_th = ss, bd, th
    P.mi_rows, P.mi_cols, P.ref_org_x, P.ref_org_y, P.ref_stride = H // 4, W // 4, PAD, PAD, stride
    return P


CASES = [((1, 1, 1), 0, 0), ((1, 1, 0), 1, 0), ((2, 2, 2), 0, 0), ((1, 2, 0), 1, 35), ((1, 0, 0), 0, 3), ((2, 1, 1), 1, 0)]


@pytest.mark.parametrize("bd", [8, 10])
def test_tf_subpel_oracle_vs_reference(oracle, ref, bd):
    if not os.path.exists(REF_ME_LIB):
        pytest.skip("oracle/_ref/libsvtref_me.so not available")
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    C.CDLL(REF_LIB, mode=C.RTLD_GLOBAL)
    refme = C.CDLL(REF_ME_LIB)
    g = rng(600 + bd)
    W, H, PAD = 192, 128, 80
    stride = W + 2 * PAD
    src, refp = make_pictures(g, W, H, PAD, bd)
    dt = src.dtype
    # 1. the prediction alone: every phase pair, block sizes, MVs that reach far outside the picture (clamped), bilinear and regular
    P = params((1, 1, 1), 0, bd, 0, W, H, PAD, stride)
    for it in range(160):
        bsize = (8, 16, 32, 64)[it % 4]
        pu_x, pu_y = int(g.integers(0, (W - bsize) // 4 + 1)) * 4, int(g.integers(0, (H - bsize) // 4 + 1)) * 4
        mvx, mvy = (int(g.integers(-40, 41)), int(g.integers(-40, 41))) if it % 5 else (int(g.integers(-3000, 3001)), int(g.integers(-2000, 2001)))
        if it < 16:
            mvx, mvy = it % 8 + 8 * (it // 8), 8 - it % 8
        bil, ss = (it // 3) % 2, (it // 7) % 2
        a = np.zeros(64 * 64, np.uint16)
        oracle.oracle_tf_luma_pred(C.byref(P), p(refp), pu_x, pu_y, bsize, mvx, mvy, bil, ss, p(a))
        b = np.zeros((64, 64), dt)
        refme.ref_tf_luma_pred(C.byref(P), p(refp), W, H, pu_x, pu_y, bsize, mvx, mvy, bil, ss, p(b))
        want = b[0:bsize:(1 << ss), :bsize].astype(np.uint16)
        assert np.array_equal(a[:bsize * (bsize >> ss)].reshape(bsize >> ss, bsize), want), (it, bsize, mvx, mvy, bil, ss)
    # 2. the search: every ring mode, sub-sampling, early exits, all block sizes, starting MVs near and far from the true displacement
    n_moved = 0
    for (mode, ss, th) in CASES:
        P = params(mode, ss, bd, th, W, H, PAD, stride)
        for it in range(40):
            bsize = (64, 32, 16, 8)[it % 4]
            nb = 64 // bsize
            sbx, sby = int(g.integers(0, W // 64)) * 64, int(g.integers(0, H // 64)) * 64
            ix, iy = int(g.integers(0, nb)), int(g.integers(0, nb))
            start = ((3 + int(g.integers(-1, 2))) * 8, (2 + int(g.integers(-1, 2))) * 8) if it % 3 else (int(g.integers(-6, 7)) * 8, int(g.integers(-6, 7)) * 8)
            bil = it % 2
            blk = src[sby:, sbx:]
            mx1, my1, d1 = C.c_int16(start[0]), C.c_int16(start[1]), C.c_uint64(0x7fffffff)
            mx2, my2, d2 = C.c_int16(start[0]), C.c_int16(start[1]), C.c_uint64(0x7fffffff)
            oracle.oracle_tf_subpel_search(C.byref(P), C.c_void_p(blk.ctypes.data + (iy * bsize * W + ix * bsize) * src.itemsize), W, p(refp), sbx + ix * bsize,
                                           sby + iy * bsize, bsize, bil, C.byref(mx1), C.byref(my1), C.byref(d1))
            refme.ref_tf_subpel_search(C.byref(P), C.c_void_p(blk.ctypes.data), W, p(refp), W, H, sbx, sby, bsize, ix, iy, bil, C.byref(mx2), C.byref(my2), C.byref(d2))
            assert (mx1.value, my1.value, d1.value) == (mx2.value, my2.value, d2.value), (mode, ss, th, it, bsize, start)
            n_moved += (mx1.value, my1.value) != start
    assert n_moved > 40  # the refinement really moves the vectors


def subpel_batch(g, pkg, src, W, H, n, far_every=3):
    """n random block descriptors over one source picture (all four block sizes, both filters, starting MVs near / far from the true displacement)"""
    d = np.zeros(n, pkg.TfSubpelDesc)
    for i in range(n):
        bsize = (64, 32, 16, 8)[i % 4]
        x, y = int(g.integers(0, W // bsize)) * bsize, int(g.integers(0, H // bsize)) * bsize
        d[i]["src_off"], d[i]["src_stride"], d[i]["pu_x"], d[i]["pu_y"], d[i]["bsize"], d[i]["bilinear"] = y * W + x, W, x, y, bsize, (i // 4) % 2
        near = ((3 + int(g.integers(-1, 2))) * 8, (2 + int(g.integers(-1, 2))) * 8)
        d[i]["mv_x"], d[i]["mv_y"] = near if i % far_every else (int(g.integers(-40, 41)) * 8, int(g.integers(-30, 31)) * 8)
    return d


def oracle_subpel(oracle, P, src, refp, descs):
    out = np.zeros(len(descs), [("dist", "<u8"), ("mv_x", "<i2"), ("mv_y", "<i2")])
    for i, d in enumerate(descs):
        mx, my, dist = C.c_int16(int(d["mv_x"])), C.c_int16(int(d["mv_y"])), C.c_uint64(0x7fffffff)
        oracle.oracle_tf_subpel_search(C.byref(P), C.c_void_p(src.ctypes.data + int(d["src_off"]) * src.itemsize), int(d["src_stride"]),
                                       C.c_void_p(refp.ctypes.data + int(d["ref_off"]) * refp.itemsize), int(d["pu_x"]), int(d["pu_y"]), int(d["bsize"]),
                                       int(d["bilinear"]), C.byref(mx), C.byref(my), C.byref(dist))
        out[i] = (dist.value, mx.value, my.value)
    return out


@pytest.mark.parametrize("bd", [8, 10])
def test_tf_subpel_search_batch_hip(be, oracle, bd):
    """svt_hip_tf_subpel_search_batch == oracle_tf_subpel_search for every ring mode / sub-sampling / early-exit setting, two reference pictures in one batch"""
    pkg = be.pkg
    g = rng(640 + bd)
    W, H, PAD = 192, 128, 80
    stride = W + 2 * PAD
    src, ref0 = make_pictures(g, W, H, PAD, bd)
    _, ref1 = make_pictures(g, W, H, PAD, bd, shift=(-5, 1))
    refs = np.ascontiguousarray(np.stack([ref0, ref1]))
    d_src, d_ref = be.dev(src), be.dev(refs)
    n = 24 if not be.is_gpu else 400
    moved = 0
    for (mode, ss, th) in CASES:
        P = params(mode, ss, bd, th, W, H, PAD, stride)
        descs = subpel_batch(g, pkg, src, W, H, n)
        descs["ref_off"] = (np.arange(n) % 2) * ref0.size
        want = oracle_subpel(oracle, P, src, refs.reshape(-1), descs)
        d_out = be.empty((n,), pkg.TfSubpelResult)
        PP = pkg.TfSubpelParams.from_buffer_copy(bytes(P))
        be.lib.svt_hip_tf_subpel_search_batch(C.byref(PP), be.ptr(d_src), be.ptr(d_ref), be.ptr(be.dev(descs)), n, be.ptr(d_out), be.stream)
        got = be.host(d_out)
        for k in ("dist", "mv_x", "mv_y"):
            bad = np.nonzero(got[k] != want[k])[0]
            assert bad.size == 0, (mode, ss, th, k, bad[:5], got[bad[:5]], want[bad[:5]], descs[bad[:5]])
        moved += int(np.count_nonzero((got["mv_x"] != descs["mv_x"]) | (got["mv_y"] != descs["mv_y"])))
    assert moved > n  # the refinement really moves vectors


@pytest.mark.parametrize("bd", [8, 10])
def test_tf_subpel_full_picture_properties(be, oracle, bd):
    """1080p, every 64/32/16/8 block of the picture against one reference (43 350 blocks): (1) a reference equal to the source -> distortion 0 at the starting
    MV (0, 0) for every block (best == 0 stops the search); (2) determinism over two launches; (3) a sample of blocks against the oracle."""
    if not be.is_gpu:
        pytest.skip("full-picture sizes run on the GPU only")
    pkg = be.pkg
    g = rng(660 + bd)
    W, H, PAD = 1920, 1088, 80
    stride = W + 2 * PAD
    amp = (1 << bd) - 1
    dt = np.uint16 if bd > 8 else np.uint8
    yy, xx = np.mgrid[0:H + 2 * PAD, 0:W + 2 * PAD].astype(np.float32)
    refp = np.clip((0.5 + 0.25 * np.sin(xx / 3.3) * np.cos(yy / 4.1) + 0.2 * np.sin((xx + 2 * yy) / 9.1)) * amp + g.normal(0, amp / 150, xx.shape), 0, amp).astype(dt)
    same = np.ascontiguousarray(refp[PAD:PAD + H, PAD:PAD + W])
    moved = np.ascontiguousarray(refp[PAD + 2:PAD + 2 + H, PAD - 3:PAD - 3 + W])  # true displacement (-3, +2) full-pel
    blocks = [(x, y, b) for b in (64, 32, 16, 8) for y in range(0, H, b) for x in range(0, W, b)]
    n = len(blocks)
    descs = np.zeros(n, pkg.TfSubpelDesc)
    a = np.array(blocks)
    descs["pu_x"], descs["pu_y"], descs["bsize"], descs["src_stride"] = a[:, 0], a[:, 1], a[:, 2], W
    descs["src_off"] = a[:, 1].astype(np.uint64) * W + a[:, 0].astype(np.uint64)
    P = params((1, 1, 1), 1, bd, 0, W, H, PAD, stride)
    PP = pkg.TfSubpelParams.from_buffer_copy(bytes(P))
    d_ref, d_descs, d_out = be.dev(refp), be.dev(descs), be.empty((n,), pkg.TfSubpelResult)
    be.lib.svt_hip_tf_subpel_search_batch(C.byref(PP), be.ptr(be.dev(same)), be.ptr(d_ref), be.ptr(d_descs), n, be.ptr(d_out), be.stream)
    got = be.host(d_out)
    assert not got["dist"].any() and not got["mv_x"].any() and not got["mv_y"].any()
    descs["mv_x"], descs["mv_y"] = -3 * 8 + 8 * g.integers(-1, 2, n), 2 * 8 + 8 * g.integers(-1, 2, n)
    d_descs, d_moved = be.dev(descs), be.dev(moved)
    outs = []
    for _ in range(2):
        d_out = be.empty((n,), pkg.TfSubpelResult)
        be.lib.svt_hip_tf_subpel_search_batch(C.byref(PP), be.ptr(d_moved), be.ptr(d_ref), be.ptr(d_descs), n, be.ptr(d_out), be.stream)
        outs.append(be.host(d_out))
    assert np.array_equal(outs[0], outs[1])
    pick = g.choice(n, 300, replace=False)
    want = oracle_subpel(oracle, P, moved, refp.reshape(-1), descs[pick])
    for k in ("dist", "mv_x", "mv_y"):
        assert np.array_equal(outs[0][k][pick], want[k]), k
    # blocks that started one pel off found the true displacement (interior blocks, exact copy -> distortion 0)
    inner = (a[:, 0] >= 64) & (a[:, 1] >= 64) & (a[:, 0] + a[:, 2] <= W - 64) & (a[:, 1] + a[:, 2] <= H - 64) & (descs["mv_x"] == -24) & (descs["mv_y"] == 16)
    assert inner.sum() > 1000 and not outs[0]["dist"][inner].any()


def make_yuv(g, W, H, PAD, bd):
    """a padded 4:2:0 reference picture: luma like make_pictures, smooth chroma + noise (chroma padding PAD / 2)"""
    _, y = make_pictures(g, W, H, PAD, bd)
    amp = (1 << bd) - 1
    dt = y.dtype
    yy, xx = np.mgrid[0:H // 2 + PAD, 0:W // 2 + PAD].astype(np.float64)
    u = np.clip((0.5 + 0.3 * np.sin(xx / 4.7) * np.cos(yy / 3.9)) * amp + g.normal(0, amp / 100, xx.shape), 0, amp).astype(dt)
    v = np.clip((0.5 + 0.3 * np.cos((xx + yy) / 5.3)) * amp + g.normal(0, amp / 100, xx.shape), 0, amp).astype(dt)
    return [y, np.ascontiguousarray(u), np.ascontiguousarray(v)]


def oracle_mc(oracle, P, planes, pu_x, pu_y, bsize, mvx, mvy, chroma):
    outs = [np.zeros((bsize, bsize), np.uint16), np.zeros((bsize // 2, bsize // 2), np.uint16), np.zeros((bsize // 2, bsize // 2), np.uint16)]
    pl = (C.c_void_p * 3)(*[x.ctypes.data for x in planes])
    st = (C.c_uint32 * 3)(*[x.shape[1] for x in planes])
    op = (C.c_void_p * 3)(*[o.ctypes.data for o in outs])
    pit = (C.c_int * 3)(bsize, bsize // 2, bsize // 2)
    oracle.oracle_tf_inter_pred(C.byref(P), pl, st, pu_x, pu_y, bsize, mvx, mvy, int(chroma), op, pit)
    return outs


@pytest.mark.parametrize("bd", [8, 10])
def test_tf_inter_pred_oracle_vs_reference(oracle, ref, bd):
    """oracle_tf_inter_pred == svt_aom_inter_prediction as the temporal filter calls it (MULTITAP_SHARP, luma + 4:2:0 chroma, 8x8 .. 64x64 blocks, MVs that reach far
    outside the picture)"""
    if not os.path.exists(REF_ME_LIB):
        pytest.skip("oracle/_ref/libsvtref_me.so not available")
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    C.CDLL(REF_LIB, mode=C.RTLD_GLOBAL)
    refme = C.CDLL(REF_ME_LIB)
    g = rng(900 + bd)
    W, H, PAD = 192, 128, 80
    planes = make_yuv(g, W, H, PAD, bd)
    P = params((1, 1, 1), 0, bd, 0, W, H, PAD, planes[0].shape[1])
    dt = planes[0].dtype
    pl = (C.c_void_p * 3)(*[x.ctypes.data for x in planes])
    st = (C.c_uint32 * 3)(*[x.shape[1] for x in planes])
    for it in range(120):
        bsize = (64, 32, 16, 8)[it % 4]
        nb = 64 // bsize
        sbx, sby = int(g.integers(0, W // 64)) * 64, int(g.integers(0, H // 64)) * 64
        ix, iy = int(g.integers(0, nb)), int(g.integers(0, nb))
        mvx, mvy = (int(g.integers(-60, 61)), int(g.integers(-60, 61))) if it % 5 else (int(g.integers(-3000, 3001)), int(g.integers(-2000, 2001)))
        if it < 16:
            mvx, mvy = it % 8, 8 - it % 8 - (it // 8) * 8
        chroma = it % 3 != 2
        want = oracle_mc(oracle, P, planes, sbx + ix * bsize, sby + iy * bsize, bsize, mvx, mvy, chroma)
        pred = [np.zeros((64, 64), dt), np.zeros((32, 32), dt), np.zeros((32, 32), dt)]
        pp = (C.c_void_p * 3)(*[x.ctypes.data for x in pred])
        refme.ref_tf_inter_pred(C.byref(P), pl, st, W, H, sbx, sby, bsize, ix, iy, mvx, mvy, int(chroma), pp)
        got_y = pred[0][iy * bsize:(iy + 1) * bsize, ix * bsize:(ix + 1) * bsize]
        assert np.array_equal(got_y, want[0]), (it, bsize, mvx, mvy)
        if chroma:
            cb = bsize // 2
            cy, cx = (((iy * bsize) >> 3) << 3) // 2, (((ix * bsize) >> 3) << 3) // 2
            for k in (1, 2):
                assert np.array_equal(pred[k][cy:cy + cb, cx:cx + cb], want[k]), (it, k, bsize, mvx, mvy)


@pytest.mark.parametrize("bd", [8, 10])
def test_tf_inter_pred_batch_hip(be, oracle, bd):
    """svt_hip_tf_inter_pred_batch == oracle_tf_inter_pred: non-overlapping blocks of all four sizes from two reference pictures into picture-sized prediction planes"""
    pkg = be.pkg
    g = rng(920 + bd)
    W, H, PAD = (448, 256, 80) if be.is_gpu else (256, 192, 80)
    refs = [make_yuv(g, W, H, PAD, bd), make_yuv(g, W, H, PAD, bd)]
    dt = refs[0][0].dtype
    ref_all = [np.ascontiguousarray(np.stack([r[pl] for r in refs])) for pl in range(3)]
    P = params((1, 1, 1), 0, bd, 0, W, H, PAD, refs[0][0].shape[1])
    # one block per 64x64 cell, random size / position inside it (so blocks never overlap), random reference
    blocks = []
    for sy in range(0, H, 64):
        for sx in range(0, W, 64):
            bsize = (64, 32, 16, 8)[int(g.integers(0, 4))]
            nb = 64 // bsize
            blocks.append((sx + int(g.integers(0, nb)) * bsize, sy + int(g.integers(0, nb)) * bsize, bsize, int(g.integers(0, 2))))
    n = len(blocks)
    d = np.zeros(n, pkg.TfMcDesc)
    pred_shape = [(H, W), (H // 2, W // 2), (H // 2, W // 2)]
    want = [np.zeros((2,) + s, dt) for s in pred_shape]
    for i, (x, y, b, r) in enumerate(blocks):
        far = i % 7 == 3
        mvx, mvy = (int(g.integers(-2500, 2501)), int(g.integers(-1500, 1501))) if far else (int(g.integers(-70, 71)), int(g.integers(-70, 71)))
        d[i]["pu_x"], d[i]["pu_y"], d[i]["bsize"], d[i]["mv_x"], d[i]["mv_y"] = x, y, b, mvx, mvy
        for pl in range(3):
            d[i]["ref_off"][pl] = r * refs[0][pl].size
            d[i]["pred_off"][pl] = r * pred_shape[pl][0] * pred_shape[pl][1]
        o = oracle_mc(oracle, P, refs[r], x, y, b, mvx, mvy, True)
        want[0][r, y:y + b, x:x + b] = o[0]
        cx, cy = ((x >> 3) << 3) // 2, ((y >> 3) << 3) // 2
        for pl in (1, 2):
            want[pl][r, cy:cy + b // 2, cx:cx + b // 2] = o[pl]
    d_ref = [be.dev(a) for a in ref_all]
    d_pred = [be.empty((2,) + s, dt) for s in pred_shape]
    PL = pkg.TfMcPlanes()
    for pl in range(3):
        PL.ref[pl], PL.pred[pl], PL.ref_stride[pl], PL.pred_stride[pl] = be.ptr(d_ref[pl]), be.ptr(d_pred[pl]), refs[0][pl].shape[1], pred_shape[pl][1]
    PP = pkg.TfSubpelParams.from_buffer_copy(bytes(P))
    be.lib.svt_hip_tf_inter_pred_batch(C.byref(PP), C.byref(PL), be.ptr(be.dev(d)), n, 1, be.stream)
    for pl in range(3):
        got = be.host(d_pred[pl])
        assert np.array_equal(got, want[pl]), (pl, np.argwhere(got != want[pl])[:5])
