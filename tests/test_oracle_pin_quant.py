"""Pins oracle/oracle_quant.c against the real reference's `*_c` quantizers and svt_handle_transform*_c.  CPU only."""
import ctypes as C

import numpy as np
import pytest

from conftest import p, rng
from quant_common import DEQUANTS, make_qparams, make_scan


def run_ref(ref, mode, qm, coeff, n, P, scan, iscan, qmv, iqmv, ls):
    q, dq, eob = np.zeros(n, np.int32), np.zeros(n, np.int32), C.c_uint16(0)
    a = [p(coeff), C.c_ssize_t(n), p(P["zbin"]), p(P["round"]), p(P["quant"]), p(P["quant_shift"]), p(q), p(dq), p(P["dequant"]),
         C.byref(eob), p(scan), p(iscan)]
    Q = [p(qmv) if qm else None, p(iqmv) if qm else None]
    if mode == 0:
        ref.svt_aom_quantize_b_c_ii(*a, *Q, ls)
    elif mode == 1:
        ref.svt_aom_highbd_quantize_b_c(*a, *Q, ls)
    elif mode == 2:
        if qm:
            ref.svt_av1_quantize_fp_qm_c(*a, *Q, C.c_int16(ls))
        else:
            [ref.svt_av1_quantize_fp_c, ref.svt_av1_quantize_fp_32x32_c, ref.svt_av1_quantize_fp_64x64_c][ls](*a)
    else:
        if qm:
            ref.svt_av1_highbd_quantize_fp_qm_c(*a, *Q, C.c_int16(ls))
        else:
            ref.svt_av1_highbd_quantize_fp_c(*a, C.c_int16(ls))
    return q, dq, eob.value


def run_oracle(oracle, mode, qm, coeff, n, P, scan, qmv, iqmv, ls):
    q, dq, eob = np.zeros(n, np.int32), np.zeros(n, np.int32), C.c_uint16(0)
    oracle.oracle_quantize(mode, p(coeff), n, p(P["zbin"]), p(P["round"]), p(P["quant"]), p(P["quant_shift"]), p(q), p(dq), p(P["dequant"]),
                           C.byref(eob), p(scan), p(qmv) if qm else None, p(iqmv) if qm else None, ls)
    return q, dq, eob.value


def gen_coeff(g, n, amp, kind):
    c = g.integers(-amp, amp + 1, n).astype(np.int32)
    if kind == 1:
        c[g.random(n) < 0.8] = 0
    if kind == 2:
        c[n // 3:] = g.integers(-3, 4, n - n // 3)
    if kind == 3:
        c[:] = 0
    return c


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_quantizers_vs_reference(oracle, ref, mode):
    g = rng(40 + mode)
    for n, ls in ((16, 0), (64, 0), (256, 0), (1024, 1), (1024, 2)):
        scan, iscan = make_scan(n, g)
        for (dc, ac) in DEQUANTS:
            if mode in (0, 2) and ac > 1836:
                continue  # 8-bit quantizers see 8-bit tables only
            P = make_qparams(dc, ac, fp=mode >= 2)
            for qm in (False, True):
                qmv = g.integers(16, 255, n).astype(np.uint8) if qm else None
                iqmv = g.integers(16, 64, n).astype(np.uint8) if qm else None
                for kind in range(4):
                    for amp in (40, 1 << 11, 1 << 15, (1 << 18) if mode in (1, 3) else (1 << 15)):
                        c = gen_coeff(g, n, amp, kind)
                        assert_same(run_oracle(oracle, mode, qm, c, n, P, scan, qmv, iqmv, ls), run_ref(ref, mode, qm, c, n, P, scan, iscan, qmv, iqmv, ls),
                                    (mode, n, ls, dc, ac, qm, kind, amp))


def assert_same(a, b, tag):
    assert np.array_equal(a[0], b[0]), ("qcoeff", tag)
    assert np.array_equal(a[1], b[1]), ("dqcoeff", tag)
    assert a[2] == b[2], ("eob", tag, a[2], b[2])


def test_handle_transform_vs_reference(oracle, ref):
    g = rng(8)
    oracle.oracle_handle_transform.restype = C.c_uint64
    for (w, h) in ((64, 64), (32, 64), (64, 32), (16, 64), (64, 16)):
        for n2n4 in (0, 1):
            x = g.integers(-(1 << 20), 1 << 20, w * h).astype(np.int32)
            a, b = x.copy(), x.copy()
            f = getattr(ref, "svt_handle_transform%dx%d%s_c" % (w, h, "_N2_N4" if n2n4 else ""))
            f.restype = C.c_uint64
            ea, eb = oracle.oracle_handle_transform(p(a), w, h, n2n4), f(p(b))
            kept = min(w, 32) * min(h, 32)
            assert ea == eb and np.array_equal(a[:kept], b[:kept]), (w, h, n2n4)


def test_reference_tables_golden_and_quantizers(oracle, ref):
    """tests/golden/quant_tables.npz == what the reference's svt_av1_build_quantizer / av1_scan_orders produce here (when the wrapper library exists),
    and the oracle's quantizers == the reference's `_c` quantizers on exactly those tables and scans."""
    import os
    from conftest import REF_LIB
    from quant_common import _qt, real_qparams, real_scans
    me_path = os.path.join(os.path.dirname(REF_LIB), "libsvtref_me.so")
    T = _qt()
    if os.path.exists(me_path):
        me = C.CDLL(me_path)
        if hasattr(me, "ref_build_quantizer_y"):
            for b, bd in enumerate((8, 10)):
                for i, q in enumerate(T["q"]):
                    t = np.zeros((7, 2), np.int16)
                    me.ref_build_quantizer_y(bd, 0, 0, int(q), p(t))
                    assert np.array_equal(t, T["tables"][b, i]), (bd, int(q))
            for ts in range(19):
                for tt in range(16):
                    sc, isc = np.zeros(1024, np.int16), np.zeros(1024, np.int16)
                    n = me.ref_scan_order(ts, tt, p(sc), p(isc))
                    assert np.array_equal(sc[:n], T["scan_%d" % ts][tt]) and np.array_equal(isc[:n], T["iscan_%d" % ts][tt]), (ts, tt)
    g = rng(77)
    for ts in (0, 1, 2, 3, 4, 9, 12, 13, 17):
        scans, iscans = real_scans(ts)
        n = scans.shape[1]
        pels = [16, 64, 256, 1024, 4096, 32, 32, 128, 128, 512, 512, 2048, 2048, 64, 64, 256, 256, 1024, 1024][ts]
        ls = int(pels > 256) + int(pels > 1024)
        for mode in range(4):
            bd = 10 if mode in (1, 3) else 8
            for P in real_qparams(bd, mode >= 2):
                for tt in (0, 3, 9, 10, 11):
                    c = gen_coeff(g, n, 1 << (bd + 5), int(g.integers(0, 3)))
                    scan, iscan = np.ascontiguousarray(scans[tt]), np.ascontiguousarray(iscans[tt])
                    assert_same(run_oracle(oracle, mode, False, c, n, P, scan, None, None, ls), run_ref(ref, mode, False, c, n, P, scan, iscan, None, None, ls),
                                (ts, mode, tt))
