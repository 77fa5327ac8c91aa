"""The temporal filter of one central picture as a device stage (svt_hip_tf_picture_host, csrc/tf_picture.hip): sub-pel refinement of every block size, the reference's
64x64 / 32x32 / 16x16 / 8x8 decision tree (temporal_filtering.c:3183-3340), the final motion compensation, the 32x32 errors of the 64x64 predictions and the filter --
against oracle/oracle_tf_picture.c, which walks the blocks in the reference's own order and searches a size only where the reference does.  Emulator here, MI355X with
-m gpu.  The stage against the reference itself: tests/test_encoder_identity.py (SVT_HIP_TF_SEAM=1, bitstream unchanged)."""
import ctypes as C

import numpy as np
import pytest

from conftest import p, rng
from test_tf_subpel import make_yuv


def make_case(g, pkg, W, H, PAD, bd, n_refs, th64, exit_th, th32, with8, two_tap, ss, chroma, zz, mode=(1, 1, 1), sp8=False):
    nsx, nsy = (W + 63) // 64, (H + 63) // 64
    n_sb = nsx * nsy
    pics = []
    for r in range(n_refs + 1):  # central + references: the same texture displaced by a fraction, + noise (make_yuv draws new noise every call)
        y, u, v = make_yuv(rng(77), W, H, PAD, bd)
        amp = (1 << bd) - 1
        if r:
            y = np.clip(np.roll(y, (r, -r), (0, 1)).astype(np.int32) + g.integers(-3, 4, y.shape) * (1 << (bd - 8)), 0, amp).astype(y.dtype)
            u = np.clip(np.roll(u, r, 1).astype(np.int32) + g.integers(-2, 3, u.shape) * (1 << (bd - 8)), 0, amp).astype(u.dtype)
            v = np.clip(np.roll(v, -r, 0).astype(np.int32) + g.integers(-2, 3, v.shape) * (1 << (bd - 8)), 0, amp).astype(v.dtype)
        pics.append([np.ascontiguousarray(y), np.ascontiguousarray(u), np.ascontiguousarray(v)])
    P = pkg.TfPictureParams()
    P.sp.half_pel_mode, P.sp.quarter_pel_mode, P.sp.eight_pel_mode = mode
    P.sp.subsampling_shift, P.sp.bit_depth, P.sp.early_exit_th = ss, bd, 0
    P.sp.mi_rows, P.sp.mi_cols, P.sp.ref_org_x, P.sp.ref_org_y, P.sp.ref_stride = H // 4, W // 4, PAD, PAD, pics[0][0].shape[1]
    P.tf.tf_decay_factor_fp16[0], P.tf.tf_decay_factor_fp16[1], P.tf.tf_decay_factor_fp16[2] = 2400000, 5200000, 4800000
    P.tf.tf_mv_dist_th, P.tf.tf_chroma, P.tf.use_zz_based_filter, P.tf.encoder_bit_depth, P.tf.ss_x, P.tf.ss_y = 135, int(chroma), int(zz), bd, 1, 1
    P.pic_w_sb, P.pic_h_sb, P.uv_stride, P.me_exit_th, P.pred_error_32x32_th = nsx, nsy, pics[0][1].shape[1], exit_th, th32
    P.use_2tap, P.enable_8x8_pred, P.use_pred_64x64_only_th, P.subpel_8bit = int(two_tap), int(with8), th64, int(sp8)
    tabs = []
    for r in range(n_refs):
        mvx, mvy = g.integers(-6, 7, (n_sb, 85)) - (r + 1), g.integers(-6, 7, (n_sb, 85)) + (r + 1)  # near the true displacement (-r, +r), full pel
        best_mv = ((mvy.astype(np.int16).astype(np.uint16).astype(np.uint32) << 16) | mvx.astype(np.int16).astype(np.uint16)).astype(np.uint32)
        sad32 = g.integers(200, 4000, (n_sb, 4))
        best_sad = g.integers(50, 60000, (n_sb, 85)).astype(np.uint32)
        best_sad[:, 1:5] = sad32
        best_sad[:, 0] = (sad32.sum(1) * g.uniform(0.9, 1.5, n_sb)).astype(np.uint32)  # around tf_use_64x64_pred's threshold
        hme_sc = np.stack([-(r + 1) + g.integers(-1, 2, n_sb), (r + 1) + g.integers(-1, 2, n_sb)], 1).astype(np.int16)
        hme_sad = g.integers(0, 2 * max(exit_th, 1), n_sb).astype(np.uint64)
        tabs.append([np.ascontiguousarray(x) for x in (best_sad, best_mv, hme_sc, hme_sad)])
    return P, pics, tabs


def run_oracle(oracle, P, pics, tabs):
    n_refs = len(tabs)
    out = [x.copy() for x in pics[0]]
    cen = (C.c_void_p * 3)(*[x.ctypes.data for x in pics[0]])
    refs = (C.c_void_p * (3 * n_refs))(*[x.ctypes.data for pic in pics[1:] for x in pic])
    arr = lambda k: (C.c_void_p * n_refs)(*[t[k].ctypes.data for t in tabs])  # noqa: E731
    o = (C.c_void_p * 3)(*[x.ctypes.data for x in out])
    stats = np.zeros(5, np.uint32)
    oracle.oracle_tf_picture.restype = C.c_int
    y8 = [np.ascontiguousarray((pic[0] >> 2).astype(np.uint8)) for pic in pics] if P.subpel_8bit else None  # the 8 MSBs of a 10-bit picture = its 8-bit luma buffer
    r8 = (C.c_void_p * n_refs)(*[x.ctypes.data for x in y8[1:]]) if y8 else None
    assert oracle.oracle_tf_picture(C.byref(P), cen, refs, arr(0), arr(1), arr(2), arr(3), n_refs, o, p(stats), C.c_void_p(y8[0].ctypes.data) if y8 else None, r8) == 0
    return out, stats


def run_device(be, P, pics, tabs):
    pkg = be.pkg
    n_refs = len(tabs)
    y8 = {id(pic[0]): np.ascontiguousarray((pic[0] >> 2).astype(np.uint8)) for pic in pics} if P.subpel_8bit else {}
    hp = lambda pic, k=None: pkg.TfHostPicture(pic[0].ctypes.data, pic[1].ctypes.data, pic[2].ctypes.data, pic[0].size, pic[1].size,  # noqa: E731
                                               y8[id(k if k is not None else pic[0])].ctypes.data if y8 else None)
    cen = hp(pics[0])
    refs = (pkg.TfHostPicture * n_refs)(*[hp(x) for x in pics[1:]])
    me = (pkg.TfMeTables * n_refs)(*[pkg.TfMeTables(*[x.ctypes.data for x in t]) for t in tabs])
    out = [x.copy() for x in pics[0]]  # in place, like the reference: the output buffers start as the central picture
    cen_inplace = hp(out, pics[0][0])
    st = pkg.TfPictureStats()
    rc = be.lib.svt_hip_tf_picture_host(C.byref(P), C.byref(cen_inplace), refs, me, n_refs, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, C.byref(st))
    assert rc == 0
    del cen
    return out, np.array([st.blocks_64x64, st.blocks_32x32, st.blocks_16x16, st.blocks_8x8, st.early_exit_blocks], np.uint32)


def run_resident(be, P, pics, tabs):
    """svt_hip_tf_picture: pictures and ME tables resident on the device, reference frames back to back, filtered in place"""
    pkg, n_refs = be.pkg, len(tabs)
    d_c = [be.dev(x) for x in pics[0]]
    d_r = [be.dev(np.concatenate([pics[1 + r][pl].reshape(-1) for r in range(n_refs)])) for pl in range(3)]
    d_t = [be.dev(np.concatenate([tabs[r][k].reshape(-1).view(np.uint8) for r in range(n_refs)])) for k in range(4)]
    D = pkg.TfDevicePictures()
    for pl in range(3):
        D.central[pl], D.refs[pl] = be.ptr(d_c[pl]), be.ptr(d_r[pl])
    D.ref_pitch, D.ref_uv_pitch = pics[0][0].size, pics[0][1].size
    if P.subpel_8bit:
        d_c8 = be.dev(np.ascontiguousarray((pics[0][0] >> 2).astype(np.uint8)))
        d_r8 = be.dev(np.concatenate([(pics[1 + r][0] >> 2).astype(np.uint8).reshape(-1) for r in range(n_refs)]))
        D.central_y8, D.refs_y8, D.ref_y8_pitch = be.ptr(d_c8), be.ptr(d_r8), pics[0][0].size
    M = pkg.TfMeTables(*[be.ptr(x) for x in d_t])
    ws = be.empty(be.lib.svt_hip_tf_picture_workspace(C.byref(P), n_refs), np.uint8)
    d_st = be.empty(32, np.uint8)
    assert be.lib.svt_hip_tf_picture(C.byref(P), C.byref(D), C.byref(M), n_refs, be.ptr(ws), be.ptr(d_st), be.stream) == 0
    be.sync()
    return [be.host(x) for x in d_c], be.host(d_st).view(np.uint32)[:5]


#        bd  n_refs th64 exit_th th32    8x8 2tap ss chroma zz
CASES = [(8, 2, 0, 0, 0, True, False, 0, True, False),          # every 32x32 goes through derive_tf_32x32_block_split_flag, with 8x8
         (8, 3, 20, 900, 3000, False, True, 1, True, False),    # tf_use_64x64_pred, early exits, bilinear searches, sub-sampled distortions; odd reference count
         (8, 1, 255, 0, 1 << 20, False, False, 0, False, True), # 64x64 only; luma only; the zero-motion filter
         (10, 2, 35, 500, 20000, True, False, 1, True, False),  # 10 bit
         (10, 2, 30, 300, 2500, False, True, 1, True, False),   # 10 bit with the searches on the 8-bit luma (tf_ctrls.use_8bit_subpel)
         (8, 14, 25, 400, 3000, False, False, 0, True, False)]  # more frames than one launch of the filter takes (chunked accumulation)


@pytest.mark.parametrize("case", range(len(CASES)))
def test_tf_picture_stage(be, oracle, case):
    bd, n_refs, th64, exit_th, th32, with8, two_tap, ss, chroma, zz = CASES[case]
    if not be.is_gpu and case in (3, 5):
        pytest.skip("emulator: the 10-bit case runs on the GPU (the u16 paths of every piece are covered by their own emulator tests)")
    W, H, PAD = (320, 200, 80) if be.is_gpu else ((64, 72, 80) if case in (0, 4) else (128, 72, 80))  # (a partial last block row: H is not a multiple of 64)
    g = rng(500 + case)
    P, pics, tabs = make_case(g, be.pkg, W, H, PAD, bd, n_refs, th64, exit_th, th32, with8, two_tap, ss, chroma, zz, sp8=(case == 4))
    want, wstats = run_oracle(oracle, P, pics, tabs)
    got, gstats = run_device(be, P, pics, tabs)
    assert np.array_equal(wstats, gstats), (wstats, gstats)
    for pl in range(3):
        assert np.array_equal(want[pl], got[pl]), (case, pl, int((want[pl] != got[pl]).sum()))
    res, rstats = run_resident(be, P, pics, tabs)  # the device-resident form: same result, in place
    assert np.array_equal(rstats, gstats), (rstats, gstats)
    for pl in range(3 if chroma else 1):
        assert np.array_equal(res[pl], got[pl]), (case, pl, "resident form", int((res[pl] != got[pl]).sum()))
    assert not np.array_equal(want[0], pics[0][0])  # the filter changed the picture
    if not chroma:
        assert np.array_equal(got[1], pics[0][1]) and np.array_equal(got[2], pics[0][2])
    # every decision branch the case is meant to reach was taken
    if case == 0: assert gstats[3] > 0 and gstats[2] > 0  # noqa: E701
    if case == 1: assert gstats[4] > 0 and gstats[0] > 0 and gstats[1] > 0  # noqa: E701
    if case == 2: assert gstats[0] == n_refs * P.pic_w_sb * P.pic_h_sb and not gstats[1:4].any()  # noqa: E701


@pytest.mark.parametrize("bd,zz,ss", [(8, False, 0), (8, True, 1), (10, False, 1)])
def test_tf_picture_zero_motion(be, oracle, bd, zz, ss):
    """the low-delay form (produce_temporally_filtered_pic_ld, temporal_filtering.c:3415-3846): no ME tables at all, every block is the co-located 64x64 prediction
    of every frame, its 32x32 errors are variances, then the filter"""
    if not be.is_gpu and bd == 10:
        pytest.skip("emulator: the 10-bit case runs on the GPU")
    W, H, PAD, n_refs = (320, 200, 80, 3) if be.is_gpu else (128, 72, 80, 3)
    g = rng(560 + bd + ss)
    P, pics, tabs = make_case(g, be.pkg, W, H, PAD, bd, n_refs, 20, 900, 3000, False, False, ss, True, zz)
    P.zero_motion = 1
    for r in range(1, n_refs + 1):  # a static scene: the frames are the central picture + noise (make_case's displaced textures get weight 0 at vector (0, 0))
        pics[r] = [np.ascontiguousarray(np.clip(x.astype(np.int32) + g.integers(-4, 5, x.shape) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(x.dtype)) for x in pics[0]]
    out = [x.copy() for x in pics[0]]
    cen = (C.c_void_p * 3)(*[x.ctypes.data for x in pics[0]])
    refs = (C.c_void_p * (3 * n_refs))(*[x.ctypes.data for pic in pics[1:] for x in pic])
    o = (C.c_void_p * 3)(*[x.ctypes.data for x in out])
    wstats = np.zeros(5, np.uint32)
    oracle.oracle_tf_picture.restype = C.c_int
    assert oracle.oracle_tf_picture(C.byref(P), cen, refs, None, None, None, None, n_refs, o, p(wstats), None, None) == 0
    pkg = be.pkg
    hp = lambda pic: pkg.TfHostPicture(pic[0].ctypes.data, pic[1].ctypes.data, pic[2].ctypes.data, pic[0].size, pic[1].size, None)  # noqa: E731
    got = [x.copy() for x in pics[0]]
    st = pkg.TfPictureStats()
    hrefs = (pkg.TfHostPicture * n_refs)(*[hp(x) for x in pics[1:]])
    hcen = hp(got)
    assert be.lib.svt_hip_tf_picture_host(C.byref(P), C.byref(hcen), hrefs, None, n_refs, got[0].ctypes.data, got[1].ctypes.data, got[2].ctypes.data, C.byref(st)) == 0
    assert st.blocks_64x64 == n_refs * P.pic_w_sb * P.pic_h_sb == wstats[0] and st.blocks_32x32 == st.blocks_16x16 == st.blocks_8x8 == st.early_exit_blocks == 0
    for pl in range(3):
        assert np.array_equal(out[pl], got[pl]), (pl, int((out[pl] != got[pl]).sum()))
    assert not np.array_equal(out[0], pics[0][0])
    # the ordinary form on the same pictures (searched vectors) gives another picture: the mode switch is observed
    want2, _ = run_oracle(oracle, make_case(rng(560 + bd + ss), be.pkg, W, H, PAD, bd, n_refs, 20, 900, 3000, False, False, ss, True, zz)[0], pics, tabs)
    assert not np.array_equal(want2[0], out[0])
