"""Temporal-filter pixel kernels (DSP part): plane-wise non-local-means accumulation (with / without motion, 8-bit / high
bit depth), central initialisation, normalisation, noise estimate.  Oracle pinned against the reference's `_c` functions (called with a
MeContext built by oracle/ref_wrap/ref_tf.c from the plain parameter structs); HIP path checked against the oracle through the C-ABI."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, load_pkg, p, rng

REF_ME_LIB = os.path.join(ROOT, "oracle", "_ref", "libsvtref_me.so")


def ref_tf():
    if not os.path.exists(REF_ME_LIB):
        pytest.skip("oracle/_ref/libsvtref_me.so not available")
    return C.CDLL(REF_ME_LIB)


def make_params(pkg, g, bd, zz, chroma, ss):
    P = pkg.TfParams()
    for c in range(3):
        P.tf_decay_factor_fp16[c] = int(g.choice([1 << 10, 3000, 1 << 15, 90000, 1 << 18, 5 << 19, 1 << 22]))
    P.tf_mv_dist_th = int(g.choice([0, 3, 16, 40, 100]))
    P.tf_chroma, P.use_zz_based_filter, P.encoder_bit_depth, P.ss_x, P.ss_y = chroma, zz, bd, ss[0], ss[1]
    return P


def make_blocks(pkg, g, n, bd):
    B = np.zeros(n, pkg.TfBlock)
    B["split"] = g.integers(0, 2, n)
    B["block_error"] = (g.integers(0, 4096, (n, 4)) * g.choice([0, 1, 4, 16, 256], (n, 1))) << (0 if bd == 8 else 4)
    scale = g.choice([0, 1, 1, 2, 2, 8, 40], (n, 1))
    B["mv_x"] = g.integers(-6, 7, (n, 4)) * scale
    B["mv_y"] = g.integers(-4, 5, (n, 4)) * scale
    return B


def make_pair(g, bd, shape):
    dt = np.uint8 if bd == 8 else np.uint16
    a = g.integers(0, 1 << bd, shape)
    noise = int(g.choice([1, 2, 4, 12, 60])) << (bd - 8)
    b = np.clip(a + g.integers(-noise, noise + 1, shape), 0, (1 << bd) - 1)
    return a.astype(dt), b.astype(dt)


def vp(a, off=0):
    return C.c_void_p(a.ctypes.data + off * a.itemsize)


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("zz", [0, 1])
def test_tf_planewise_oracle_vs_reference(oracle, ref, bd, zz):
    refme, pkg, g = ref_tf(), load_pkg(), rng(1000 + bd + zz)
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))  # sqrt_fast goes through the svt_log2f pointer
    hbd = int(bd > 8)
    changed = 0
    for it in range(60):
        ss = [(1, 1), (1, 0), (0, 0)][it % 3]
        bw = bh = 32 if it % 5 else 16
        P = make_params(pkg, g, bd, zz, it % 4 != 3, ss)
        B = make_blocks(pkg, g, 1, bd)
        SS, PS = 80, 40  # source stride, prediction (= accumulator) stride
        ys, yp = make_pair(g, bd, (bh, SS))
        us, up = make_pair(g, bd, (bh, SS))
        vs, vq = make_pair(g, bd, (bh, SS))
        yp, up, vq = (np.ascontiguousarray(x[:, :PS]) for x in (yp, up, vq))
        acc = [g.integers(0, 1 << 24, (bh, PS)).astype(np.uint32) for _ in range(3)]
        cnt = [g.integers(0, 5000, (bh, PS)).astype(np.uint16) for _ in range(3)]
        a0, c0 = [x.copy() for x in acc], [x.copy() for x in cnt]
        a1, c1 = [x.copy() for x in acc], [x.copy() for x in cnt]
        oracle.oracle_tf_planewise(C.byref(P), vp(B), vp(ys), SS, vp(yp), PS, vp(us), vp(vs), SS, vp(up), vp(vq), PS, bw, bh, ss[0], ss[1], vp(a0[0]), vp(c0[0]),
                                   vp(a0[1]), vp(c0[1]), vp(a0[2]), vp(c0[2]), zz, hbd)
        refme.ref_tf_planewise(C.byref(P), vp(B), 0, 0, vp(ys), SS, vp(yp), PS, vp(us), vp(vs), SS, vp(up), vp(vq), PS, bw, bh, ss[0], ss[1], vp(a1[0]), vp(c1[0]),
                               vp(a1[1]), vp(c1[1]), vp(a1[2]), vp(c1[2]), zz, hbd)
        for c in range(3):
            assert np.array_equal(a0[c], a1[c]) and np.array_equal(c0[c], c1[c]), (bd, zz, it, c)
        changed += not np.array_equal(a0[0], acc[0])
    assert changed >= 12, changed  # the remaining cases legitimately get weight 0


def ref_frame_chain(refme, pkg, P, central, cstride, preds, blocks, n_refs, hbd):
    """produce_temporally_filtered_pic's sequence for ONE 64x64 block with the reference's own kernels: central -> per reference and 32x32
    block the plane-wise filter at the offsets of apply_filtering_block_plane_wise (:1412-1440) -> get_final_filtered_pixels."""
    ssx, ssy = P.ss_x, P.ss_y
    cw, chh = 64 >> ssx, 64 >> ssy
    acc = [np.zeros(64 * 64, np.uint32), np.zeros(cw * chh, np.uint32), np.zeros(cw * chh, np.uint32)]
    cnt = [np.zeros(64 * 64, np.uint16), np.zeros(cw * chh, np.uint16), np.zeros(cw * chh, np.uint16)]
    A, K = (C.c_void_p * 3)(*[x.ctypes.data for x in acc]), (C.c_void_p * 3)(*[x.ctypes.data for x in cnt])
    S = (C.c_void_p * 3)(*[x.ctypes.data for x in central])
    assert cstride[1] == cstride[0] >> ssx  # apply_filtering_central derives the chroma stride from the luma one
    refme.ref_tf_central(C.byref(P), S, cstride[0], A, K, hbd)
    stride_pred = [64, cw]
    for r in range(n_refs):
        for br in range(2):
            for bc in range(2):
                oy = br * 32 * cstride[0] + bc * 32
                oc = br * (32 >> ssy) * cstride[1] + bc * (32 >> ssx)
                py = br * 32 * 64 + bc * 32
                pc = br * (32 >> ssy) * cw + bc * (32 >> ssx)
                pr = preds[r]
                refme.ref_tf_planewise(C.byref(P), vp(blocks[r, br, bc:bc + 1]), br, bc, vp(central[0], oy), cstride[0], vp(pr[0], py), 64, vp(central[1], oc),
                                       vp(central[2], oc), cstride[1], vp(pr[1], pc), vp(pr[2], pc), cw, 32, 32, ssx, ssy, vp(acc[0], py), vp(cnt[0], py),
                                       vp(acc[1], pc), vp(cnt[1], pc), vp(acc[2], pc), vp(cnt[2], pc), int(P.use_zz_based_filter), hbd)
    out = [x.copy() for x in central]
    D = (C.c_void_p * 3)(*[x.ctypes.data for x in out])
    st = (C.c_uint32 * 3)(cstride[0], cstride[1], cstride[1])
    refme.ref_tf_final(C.byref(P), D, A, K, st, 0, 0, hbd)
    return out


def oracle_frame(oracle, P, central, cstride, preds, pstrides, blocks, n_refs, nbx, nby):
    out = [x.copy() for x in central]
    cen = (C.c_void_p * 3)(*[x.ctypes.data for x in central])
    o = (C.c_void_p * 3)(*[x.ctypes.data for x in out])
    pr = (C.c_void_p * (3 * n_refs))(*[x.ctypes.data for r in range(n_refs) for x in preds[r]])
    ps = (C.c_int * (2 * n_refs))(*[v for r in range(n_refs) for v in pstrides[r]])
    cs = (C.c_int * 2)(*cstride)
    oracle.oracle_tf_filter_frame(C.byref(P), cen, cs, pr, ps, vp(blocks), n_refs, nbx, nby, o, cs)
    return out


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("zz", [0, 1])
def test_tf_frame_oracle_vs_reference_chain(oracle, ref, bd, zz):
    refme, pkg, g = ref_tf(), load_pkg(), rng(1100 + bd + zz)
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
    hbd, changed = int(bd > 8), 0
    for it, ss in enumerate([(1, 1), (1, 0), (0, 0), (1, 1)]):
        chroma = it != 3
        P = make_params(pkg, g, bd, zz, chroma, ss)
        n_refs = 3
        cw, chh = 64 >> ss[0], 64 >> ss[1]
        cstride = [96, 96 >> ss[0]]
        central = [make_pair(g, bd, (64, cstride[0]))[0], make_pair(g, bd, (chh, cstride[1]))[0], make_pair(g, bd, (chh, cstride[1]))[0]]
        preds = []
        for r in range(n_refs):
            pl = []
            for c in range(3):
                w, h = (64, 64) if c == 0 else (cw, chh)
                noise = int(g.choice([2, 8, 30])) << (bd - 8)
                pl.append(np.clip(central[c][:h, :w].astype(np.int32) + g.integers(-noise, noise + 1, (h, w)), 0, (1 << bd) - 1).astype(central[c].dtype))
            preds.append(pl)
        blocks = make_blocks(pkg, g, n_refs * 4, bd).reshape(n_refs, 2, 2)
        want = ref_frame_chain(refme, pkg, P, central, cstride, preds, blocks, n_refs, hbd)
        got = oracle_frame(oracle, P, central, cstride, preds, [[64, cw]] * n_refs, blocks, n_refs, 2, 2)
        for c in range(3 if chroma else 1):
            assert np.array_equal(want[c], got[c]), (bd, zz, it, c)
        changed += not np.array_equal(got[0], central[0])
    assert changed >= 2


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_noise_estimate_oracle_vs_reference(oracle, ref, bd):
    g = rng(1200 + bd)
    for (w, h, stride, kind) in ((64, 48, 80, 0), (130, 33, 136, 1), (40, 40, 40, 2), (16, 6, 24, 3)):
        dt = np.uint8 if bd == 8 else np.uint16
        base = (np.add.outer(np.arange(h), np.arange(stride)) * (2 if kind else 0)) << (bd - 8)
        amp = [2, 6, 200, 1][kind] << (bd - 8)
        a = np.clip(base + (100 << (bd - 8)) + g.integers(-amp, amp + 1, (h, stride)), 0, (1 << bd) - 1).astype(dt)
        got = oracle.oracle_estimate_noise_fp16(vp(a), w, h, stride, bd)
        want = ref.svt_estimate_noise_fp16_c(vp(a), w, h, stride) if bd == 8 else ref.svt_estimate_noise_highbd_fp16_c(vp(a), w, h, stride, bd)
        assert got == want, (bd, w, h, kind, got, want)


# ------------------------------------------------------------------------------------------------ HIP path vs oracle (C-ABI)
@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("zz", [0, 1])
def test_tf_planewise_symbols_hip(be, oracle, bd, zz):
    pkg, g, hbd = load_pkg(), rng(1300 + bd + zz), int(bd > 8)
    name = "svt_av1_apply_%stemporal_filter_planewise_medium%s_hip" % ("zz_based_" if zz else "", "_hbd" if hbd else "")
    fn = getattr(be.lib, name)
    for it in range(6 if not be.is_gpu else 16):
        ss = [(1, 1), (1, 0), (0, 0)][it % 3]
        bw = bh = 32 if it % 4 else 16
        P = make_params(pkg, g, bd, zz, it % 4 != 3, ss)
        B = make_blocks(pkg, g, 1, bd)
        SS, PS = 80, 40
        ys, yp = make_pair(g, bd, (bh, SS))
        us, up = make_pair(g, bd, (bh, SS))
        vs, vq = make_pair(g, bd, (bh, SS))
        yp, up, vq = (np.ascontiguousarray(x[:, :PS]) for x in (yp, up, vq))
        acc = [g.integers(0, 1 << 24, (bh, PS)).astype(np.uint32) for _ in range(3)]
        cnt = [g.integers(0, 5000, (bh, PS)).astype(np.uint16) for _ in range(3)]
        a0, c0 = [x.copy() for x in acc], [x.copy() for x in cnt]
        a1, c1 = [x.copy() for x in acc], [x.copy() for x in cnt]
        oracle.oracle_tf_planewise(C.byref(P), vp(B), vp(ys), SS, vp(yp), PS, vp(us), vp(vs), SS, vp(up), vp(vq), PS, bw, bh, ss[0], ss[1], vp(a0[0]), vp(c0[0]),
                                   vp(a0[1]), vp(c0[1]), vp(a0[2]), vp(c0[2]), zz, hbd)
        tail = [bw, bh, ss[0], ss[1], vp(a1[0]), vp(c1[0]), vp(a1[1]), vp(c1[1]), vp(a1[2]), vp(c1[2])] + ([bd] if hbd else [])
        if zz:
            fn(C.addressof(P), vp(B), vp(yp), PS, vp(up), vp(vq), PS, *tail)
        else:
            fn(C.addressof(P), vp(B), vp(ys), SS, vp(yp), PS, vp(us), vp(vs), SS, vp(up), vp(vq), PS, *tail)
        for c in range(3):
            assert np.array_equal(a0[c], a1[c]) and np.array_equal(c0[c], c1[c]), (bd, zz, it, c)


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("zz", [0, 1])
def test_tf_filter_frame_hip(be, oracle, bd, zz):
    """Whole-picture form: central + n references + normalisation in one launch vs the oracle's per-block chain; in place and out of place."""
    pkg, g = load_pkg(), rng(1400 + bd + zz)
    for it, ss in enumerate([(1, 1), (1, 0), (0, 0), (1, 1), (0, 1), (1, 1), (1, 1), (1, 1), (1, 1)]):
        chroma = it != 3
        nbx, nby = (3, 2) if not be.is_gpu else (9, 5)
        if not be.is_gpu and it == 6:
            continue  # (the emulator keeps one of the two large reference counts)
        n_refs = [3, 1, 6, 0, 2, 8, 9, 12, 27][it] if be.is_gpu or it != 2 else 2  # (27 > SVT_HIP_TF_MAX_REFS: the chunked form, three launches; the reference's filter takes up to 32)
        P = make_params(pkg, g, bd, zz, chroma, ss)
        W, H = nbx * 32, nby * 32
        cw, chh = W >> ss[0], H >> ss[1]
        cstride = [W + 24, cw + 9]
        central = [make_pair(g, bd, (H, cstride[0]))[0], make_pair(g, bd, (chh, cstride[1]))[0], make_pair(g, bd, (chh, cstride[1]))[0]]
        preds, pstrides = [], []
        for r in range(n_refs):
            st = [W + 3 * r + 1, cw + 2 * r + 5]
            pl = []
            for c in range(3):
                w, h = (W, H) if c == 0 else (cw, chh)
                noise = int(g.choice([2, 8, 30])) << (bd - 8)
                a = np.zeros((h, st[c > 0]), central[c].dtype)
                a[:, :w] = np.clip(central[c][:h, :w].astype(np.int32) + g.integers(-noise, noise + 1, (h, w)), 0, (1 << bd) - 1)
                pl.append(a)
            preds.append(pl)
            pstrides.append(st)
        blocks = make_blocks(pkg, g, max(n_refs, 1) * nby * nbx, bd)
        want = oracle_frame(oracle, P, central, cstride, preds, pstrides, blocks, n_refs, nbx, nby)
        d_c = [be.dev(x) for x in central]
        d_o = [be.dev(x) for x in central]
        d_p = [[be.dev(x) for x in pl] for pl in preds]
        d_b = be.dev(blocks)
        PL = pkg.TfPlanes
        cen = PL(be.ptr(d_c[0]), be.ptr(d_c[1]), be.ptr(d_c[2]), cstride[0], cstride[1])
        out = PL(be.ptr(d_o[0]), be.ptr(d_o[1]), be.ptr(d_o[2]), cstride[0], cstride[1])
        prs = (PL * max(n_refs, 1))(*[PL(be.ptr(d_p[r][0]), be.ptr(d_p[r][1]), be.ptr(d_p[r][2]), pstrides[r][0], pstrides[r][1]) for r in range(n_refs)])
        def frame(dst):
            if n_refs > 12:
                ws = be.empty(be.lib.svt_hip_tf_filter_frame_workspace(C.addressof(P), nbx, nby), np.uint8)
                be.lib.svt_hip_tf_filter_frame_chunked(C.addressof(P), C.addressof(cen), C.addressof(prs), n_refs, be.ptr(d_b), nbx, nby, C.addressof(dst), be.ptr(ws), be.stream)
            else:
                be.lib.svt_hip_tf_filter_frame(C.addressof(P), C.addressof(cen), C.addressof(prs), n_refs, be.ptr(d_b), nbx, nby, C.addressof(dst), be.stream)
        frame(out)
        for c in range(3 if chroma else 1):
            assert np.array_equal(be.host(d_o[c]), want[c]), (bd, zz, it, c, "out of place")
        frame(cen)
        for c in range(3 if chroma else 1):
            assert np.array_equal(be.host(d_c[c]), want[c]), (bd, zz, it, c, "in place")


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_noise_estimate_hip(be, oracle, bd):
    g = rng(1500 + bd)
    cases = [(64, 48, 80, 0), (130, 33, 136, 1), (40, 40, 40, 2), (16, 6, 24, 3), (3, 3, 8, 0)] + ([(1920, 1080, 2056, 1)] if be.is_gpu else [])
    for (w, h, stride, kind) in cases:
        dt = np.uint8 if bd == 8 else np.uint16
        base = (np.add.outer(np.arange(h), np.arange(stride)) * (2 if kind else 0)) << (bd - 8)
        amp = [2, 6, 200, 1][kind] << (bd - 8)
        a = np.clip(base % (200 << (bd - 8)) + (20 << (bd - 8)) + g.integers(-amp, amp + 1, (h, stride)), 0, (1 << bd) - 1).astype(dt)
        want = oracle.oracle_estimate_noise_fp16(vp(a), w, h, stride, bd)
        got = be.lib.svt_estimate_noise_fp16_hip(vp(a), w, h, stride) if bd == 8 else be.lib.svt_estimate_noise_highbd_fp16_hip(vp(a), w, h, stride, bd)
        assert got == want, (bd, w, h, kind, got, want)
        d_a, d_out = be.dev(a), be.empty(2, np.int32)
        d_ws = be.dev(np.full(be.lib.svt_hip_estimate_noise_workspace(w, h), 0x5a, np.uint8))  # any content
        for _ in range(2):
            be.lib.svt_hip_estimate_noise_batch(be.ptr(d_a), w, h, stride, bd, be.ptr(d_out), be.ptr(d_ws), be.stream)
            assert int(be.host(d_out)[0]) == want
