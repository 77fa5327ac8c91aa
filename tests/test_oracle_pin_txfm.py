"""Pins the generated 1-D flow graphs (tools/gen_txfm.py) and the 2-D drivers of oracle/oracle_txfm.c against the REAL
reference (oracle/_ref) and against golden vectors (tests/golden/txfm.npz).  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, p, rng

TX_SIZES = ["4x4", "8x8", "16x16", "32x32", "64x64", "4x8", "8x4", "8x16", "16x8", "16x32", "32x16", "32x64", "64x32", "4x16", "16x4",
            "8x32", "32x8", "16x64", "64x16"]
TXW = [int(s.split("x")[0]) for s in TX_SIZES]
TXH = [int(s.split("x")[1]) for s in TX_SIZES]
DCT_DCT, IDTX, V_DCT, H_DCT = 0, 9, 10, 11


def allowed_types(ts):  # test/TxfmCommon.h:160-209
    name = TX_SIZES[ts]
    if name == "32x32":
        return [DCT_DCT, IDTX, V_DCT, H_DCT]
    if name in ("32x64", "64x32", "16x64", "64x16"):
        return [DCT_DCT]
    if name in ("16x32", "32x16", "64x64", "8x32", "32x8"):
        return [DCT_DCT, IDTX]
    return list(range(16))


def fwd_ref_name(ts, pf=0):
    w, h = TXW[ts], TXH[ts]
    sfx = ["", "_N2", "_N4"][pf]
    if w == h:
        return ("svt_av1_transform_two_d_%dx%d%s_c" if pf == 0 else "svt_aom_transform_two_d_%dx%d%s_c") % (w, h, sfx)
    return "svt_av1_fwd_txfm2d_%dx%d%s_c" % (w, h, sfx)


def call_ref_inv(ref, ts, coeff, pred, stride, out, tx_type, bd):
    w, h = TXW[ts], TXH[ts]
    f = getattr(ref, "svt_av1_inv_txfm2d_add_%dx%d_c" % (w, h))
    if w == h:
        f(p(coeff), p(pred), stride, p(out), stride, tx_type, bd)
    elif (w, h) in ((4, 8), (8, 4), (4, 16), (16, 4)):  # no eob argument (inv_transforms.c:2583,2590,2689,2696)
        f(p(coeff), p(pred), stride, p(out), stride, tx_type, ts, bd)
    else:
        f(p(coeff), p(pred), stride, p(out), stride, tx_type, ts, w * h, bd)


def test_trig_tables(oracle, ref):
    cos = (C.c_int32 * (7 * 64)).in_dll(ref, "svt_aom_eb_av1_cospi_arr_data")
    sin = (C.c_int32 * (7 * 5)).in_dll(ref, "svt_aom_eb_av1_sinpi_arr_data")
    oracle.oracle_cospi_table.restype = C.POINTER(C.c_int32)
    oracle.oracle_sinpi_table.restype = C.POINTER(C.c_int32)
    for bit in range(10, 17):
        a, b = oracle.oracle_cospi_table(bit), oracle.oracle_sinpi_table(bit)
        assert [a[i] for i in range(64)] == list(cos)[(bit - 10) * 64:(bit - 9) * 64]
        assert [b[i] for i in range(5)] == list(sin)[(bit - 10) * 5:(bit - 9) * 5]


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64])
def test_1d_kernels_vs_reference(oracle, ref, n):
    g = rng(n)
    sr = (C.c_int8 * 16)(*([18] * 16))
    cases = [(0, "dct")] + ([(1, "adst")] if n <= 16 else [])
    for kind, nm in cases:
        for cos_bit in (10, 11, 12, 13):
            for amp in (1 << 10, 1 << 15, 1 << 17):
                for _ in range(50):
                    x = g.integers(-amp, amp + 1, n).astype(np.int32)
                    a, b = np.zeros(n, np.int32), np.zeros(n, np.int32)
                    oracle.oracle_fwd_txfm1d(kind, n, p(x), p(a), cos_bit)
                    getattr(ref, "svt_av1_f%s%d_new" % (nm, n))(p(x), p(b), cos_bit, sr)
                    assert np.array_equal(a, b), ("fwd", nm, n, cos_bit, amp)
                    for clamp in (16, 18, 20):
                        srr = (C.c_int8 * 16)(*([clamp] * 16))
                        oracle.oracle_inv_txfm1d(kind, n, p(x), p(a), cos_bit, clamp)
                        getattr(ref, "svt_av1_i%s%d_new" % (nm, n))(p(x), p(b), cos_bit, srr)
                        assert np.array_equal(a, b), ("inv", nm, n, cos_bit, amp, clamp)


def ref_fwd(ref, ts, res, stride, tx_type, bd, pf=0):
    out = np.zeros(TXW[ts] * TXH[ts], np.int32)
    getattr(ref, fwd_ref_name(ts, pf))(p(res), p(out), stride, tx_type, bd)
    return out


@pytest.mark.parametrize("ts", range(19))
def test_2d_fwd_inv_vs_reference(oracle, ref, ts):
    """Same generators as test/FwdTxfm2dAsmTest.cc:272-330 (residual in +-(2^bd - 1)) and test/InvTxfm2dAsmTest.cc:92-145
    (inverse input = C forward transform of a random residual)."""
    g = rng(100 + ts)
    w, h = TXW[ts], TXH[ts]
    stride = w + 5
    for bd in (8, 10):
        for tx_type in allowed_types(ts):
            for it in range(6):
                amp = (1 << bd) - 1
                res = g.integers(-amp, amp + 1, h * stride).astype(np.int16)
                if it == 0:
                    res[:] = amp
                if it == 1:
                    res[:] = -amp
                for pf in (0, 1, 2):
                    a = np.zeros(w * h, np.int32)
                    oracle.oracle_fwd_txfm2d(p(res), p(a), stride, tx_type, ts, bd, pf)
                    b = ref_fwd(ref, ts, res, stride, tx_type, bd, pf)
                    if pf:  # the reference's N2/N4 kernels leave the dropped area unwritten; the test zeroes it (FwdTxfm2dAsmTest.cc:333-356)
                        bb = b.reshape(h, w)
                        bb[h >> pf:, :] = 0
                        bb[:, w >> pf:] = 0
                    assert np.array_equal(a, b), ("fwd", TX_SIZES[ts], tx_type, bd, pf, it)
                coeff = ref_fwd(ref, ts, res, stride, tx_type, bd)
                iw, ih = min(w, 32), min(h, 32)
                packed = np.ascontiguousarray(coeff.reshape(h, w)[:ih, :iw]).reshape(-1)
                pred = g.integers(0, 1 << bd, h * stride).astype(np.uint16)
                o1, o2 = np.zeros(h * stride, np.uint16), np.zeros(h * stride, np.uint16)
                oracle.oracle_inv_txfm2d_add(p(packed), p(pred), stride, p(o1), stride, tx_type, ts, bd)
                call_ref_inv(ref, ts, packed, pred, stride, o2, tx_type, bd)
                assert np.array_equal(o1, o2), ("inv", TX_SIZES[ts], tx_type, bd, it)


COL_ADST = {1, 3, 4, 6, 7, 8, 12, 14}  # transform types whose column (vertical) 1-D kernel is an ADST / flipped ADST (inv_transforms.h:147-186)
ROW_ADST = {2, 3, 5, 6, 7, 8, 13, 15}


def c_defined_types(ts):
    """Every transform type the reference's `_c` inverse computes for a size -- a superset of what AV1 allows (allowed_types): a 32-point ADST exists
    (av1_iadst32_new, inv_transforms.c:1119-1552), a 64-point one does not (inv_transforms.h:190-195: TXFM_TYPE_INVALID)."""
    w, h = TXW[ts], TXH[ts]
    return [t for t in range(16) if not (h == 64 and t in COL_ADST) and not (w == 64 and t in ROW_ADST)]


@pytest.mark.parametrize("ts", [t for t in range(19) if 32 in (TXW[t], TXH[t])])
def test_2d_inverse_legacy_types_vs_reference(oracle, ref, ts):
    """The types outside AV1's allowed set that the reference's own InvTxfm2dAddTest feeds its `_c` functions (test/InvTxfm2dAsmTest.cc:755-775, rows of
    txfm_support_matrix with a 1): inverse input = the reference's C forward transform with that type, as the fixture makes it (:92-135), and raw random blocks."""
    g = rng(300 + ts)
    w, h = TXW[ts], TXH[ts]
    iw, ih = min(w, 32), min(h, 32)
    stride = w + 3
    legacy = [t for t in c_defined_types(ts) if t not in allowed_types(ts)]
    assert legacy
    for bd in (8, 10):
        amp = (1 << bd) - 1
        for tx_type in legacy:
            for it in range(4):
                if it < 2:
                    res = g.integers(-amp, amp + 1, h * stride).astype(np.int16)
                    coeff = ref_fwd(ref, ts, res, stride, tx_type, bd)
                    packed = np.ascontiguousarray(coeff.reshape(h, w)[:ih, :iw]).reshape(-1)
                else:
                    packed = g.integers(-(1 << (bd + 8)), 1 << (bd + 8), iw * ih).astype(np.int32)
                pred = g.integers(0, 1 << bd, h * stride).astype(np.uint16)
                o1, o2 = np.zeros(h * stride, np.uint16), np.zeros(h * stride, np.uint16)
                oracle.oracle_inv_txfm2d_add(p(packed), p(pred), stride, p(o1), stride, tx_type, ts, bd)
                call_ref_inv(ref, ts, packed, pred, stride, o2, tx_type, bd)
                assert np.array_equal(o1, o2), ("inv legacy", TX_SIZES[ts], tx_type, bd, it)


def test_txfm_oracle_vs_golden(oracle):
    path = os.path.join(GOLDEN, "txfm.npz")
    assert os.path.exists(path), "run tools/gen_golden.py txfm"
    z = np.load(path)
    for i, (ts, tx_type, bd) in enumerate(z["cfg"]):
        ts, tx_type, bd = int(ts), int(tx_type), int(bd)
        w, h = TXW[ts], TXH[ts]
        res = z["res_%d" % i]
        a = np.zeros(w * h, np.int32)
        oracle.oracle_fwd_txfm2d(p(res), p(a), w, tx_type, ts, bd, 0)
        assert np.array_equal(a, z["coeff_%d" % i])
        pred = z["pred_%d" % i]
        o = np.zeros(w * h, np.uint16)
        iw, ih = min(w, 32), min(h, 32)
        packed = np.ascontiguousarray(a.reshape(h, w)[:ih, :iw]).reshape(-1)
        oracle.oracle_inv_txfm2d_add(p(packed), p(pred), w, p(o), w, tx_type, ts, bd)
        assert np.array_equal(o, z["recon_%d" % i])


def test_wht4x4_vs_reference(oracle, ref):
    """Lossless-mode Walsh-Hadamard: svt_av1_fwht4x4_c, svt_av1_highbd_iwht4x4_16_add_c / _1_add_c and the 8-bit wrapper
    svt_av1_inv_txfm_add_c with TxfmParam.lossless (inv_transforms.c:2833-2848, 3177-3192)."""
    from conftest import load_pkg
    pkg = load_pkg()
    g = rng(77)
    for it in range(200):
        bd = (8, 10, 12)[it % 3]
        amp = (1 << bd) - 1
        stride = 4 + it % 5
        res = g.integers(-amp, amp + 1, 4 * stride).astype(np.int16)
        if it < 4:
            res[:] = (amp, -amp, amp, -amp)[it]
        a, b = np.zeros(16, np.int32), np.zeros(16, np.int32)
        oracle.oracle_fwht4x4(p(res), p(a), stride)
        ref.svt_av1_fwht4x4_c(p(res), p(b), stride)
        assert np.array_equal(a, b)
        coeff = a if it % 2 else g.integers(-(amp << 4), (amp << 4) + 1, 16).astype(np.int32)
        pred = g.integers(0, amp + 1, 4 * stride).astype(np.uint16)
        for eob in (1, 16):
            w1, w2 = pred.copy(), pred.copy()
            oracle.oracle_iwht4x4_add(p(coeff), p(pred), stride, p(w1), stride, eob, bd)
            # CONVERT_TO_BYTEPTR(x) = (uint8_t*)((uintptr_t)x >> 1)  (definitions.h)
            f = ref.svt_av1_highbd_iwht4x4_16_add_c if eob > 1 else ref.svt_av1_highbd_iwht4x4_1_add_c
            f(p(coeff), C.c_void_p(pred.ctypes.data >> 1), stride, C.c_void_p(w2.ctypes.data >> 1), stride, bd)
            assert np.array_equal(w1, w2), (it, eob, bd)
            if bd == 8:  # the 8-bit pointer form with the whole TxfmParam
                tp = np.zeros(1, pkg.TxfmParam)
                tp[0] = (0, 0, 1, 8, 0, 0, eob)
                p8 = pred.astype(np.uint8)
                w8 = p8.copy()
                ref.svt_av1_inv_txfm_add_c(p(coeff), p(p8), stride, p(w8), stride, p(tp))
                assert np.array_equal(w8.astype(np.uint16), w1), (it, eob)
