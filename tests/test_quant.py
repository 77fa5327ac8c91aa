"""Quantize / dequantize and svt_handle_transform* (a11): HIP path vs the oracle, bit-exact.
Parameter sets span q_index 0..255 at 8 and 10 bit (QuantAsmTest.cc:77-330 sweeps the same range), with and without
quantization matrices, log_scale 0/1/2, dense / sparse / all-zero coefficient blocks."""
import ctypes as C

import numpy as np
import pytest

from conftest import p, rng
from quant_common import DEQUANTS, make_qparams, make_scan
from test_oracle_pin_quant import assert_same, gen_coeff, run_oracle


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("qm", [False, True])
def test_quantize_batch(be, oracle, mode, qm):
    g = rng(70 + mode)
    for n_coeffs, ls in ((16, 0), (64, 0), (256, 0), (1024, 1), (1024, 2)):
        deqs = [d for d in DEQUANTS if not (mode in (0, 2) and d[1] > 1836)]
        if not be.is_gpu:
            deqs = deqs[::3]
        scans = [make_scan(n_coeffs, g) for _ in range(2)]
        nqm = 2
        qmt = g.integers(16, 255, (nqm, n_coeffs)).astype(np.uint8)
        iqmt = g.integers(16, 64, (nqm, n_coeffs)).astype(np.uint8)
        params = np.zeros(len(deqs), dtype=be.pkg.QuantParams)
        plist = []
        for i, (dc, ac) in enumerate(deqs):
            P = make_qparams(dc, ac, fp=mode >= 2)
            plist.append(P)
            params[i] = (P["zbin"], P["round"], P["quant"], P["quant_shift"], P["dequant"], ls)
        nblk = len(deqs) * 8
        coeff = np.zeros((nblk, n_coeffs), np.int32)
        descs = np.zeros(nblk, dtype=be.pkg.QuantDesc)
        for b in range(nblk):
            amp = [40, 1 << 11, 1 << 15, (1 << 18) if mode in (1, 3) else (1 << 15)][b % 4]
            coeff[b] = gen_coeff(g, n_coeffs, amp, (b // 4) % 4)
            descs[b] = (b % len(deqs), b % 2, (b // 2) % nqm, 0)
        iscans = np.stack([s[1] for s in scans])
        dco, dpa, dis, dde = be.dev(coeff), be.dev(params), be.dev(iscans), be.dev(descs)
        dqm, diq = be.dev(qmt), be.dev(iqmt)
        q, dq, eob = be.empty((nblk, n_coeffs), np.int32), be.empty((nblk, n_coeffs), np.int32), be.empty(nblk, np.uint16)
        be.lib.svt_hip_quantize_batch(mode, be.ptr(dco), nblk, n_coeffs, be.ptr(dpa), be.ptr(dis), be.ptr(dqm) if qm else None,
                                      be.ptr(diq) if qm else None, be.ptr(dde), be.ptr(q), be.ptr(dq), be.ptr(eob), be.stream)
        gq, gdq, geob = be.host(q), be.host(dq), be.host(eob)
        for b in range(nblk):
            d = descs[b]
            want = run_oracle(oracle, mode, qm, coeff[b], n_coeffs, plist[int(d["qparam_idx"])], scans[int(d["iscan_idx"])][0],
                              qmt[int(d["qm_idx"])], iqmt[int(d["qm_idx"])], ls)
            assert_same((gq[b], gdq[b], int(geob[b])), want, (mode, qm, n_coeffs, ls, b))


def test_quantize_single_call_symbols(be, oracle):
    g = rng(5)
    n = 256
    scan, iscan = make_scan(n, g)
    c = gen_coeff(g, n, 1 << 12, 2)
    qmv, iqmv = g.integers(16, 255, n).astype(np.uint8), g.integers(16, 64, n).astype(np.uint8)

    def call(name, mode, qm, ls, extra):
        P = make_qparams(88, 112, fp=mode >= 2)
        q, dq, eob = np.zeros(n, np.int32), np.zeros(n, np.int32), C.c_uint16(0)
        getattr(be.lib, name)(p(c), n, p(P["zbin"]), p(P["round"]), p(P["quant"]), p(P["quant_shift"]), p(q), p(dq), p(P["dequant"]),
                              C.cast(C.byref(eob), C.c_void_p), p(scan), p(iscan), *extra)
        assert_same((q, dq, eob.value), run_oracle(oracle, mode, qm, c, n, P, scan, qmv, iqmv, ls), name)
    call("svt_aom_quantize_b_hip", 0, False, 0, [None, None, 0])
    call("svt_aom_quantize_b_hip", 0, True, 1, [p(qmv), p(iqmv), 1])
    call("svt_aom_highbd_quantize_b_hip", 1, False, 2, [None, None, 2])
    call("svt_av1_quantize_fp_hip", 2, False, 0, [])
    call("svt_av1_quantize_fp_32x32_hip", 2, False, 1, [])
    call("svt_av1_quantize_fp_64x64_hip", 2, False, 2, [])
    call("svt_av1_quantize_fp_qm_hip", 2, True, 1, [p(qmv), p(iqmv), 1])
    call("svt_av1_highbd_quantize_fp_hip", 3, False, 1, [1])
    call("svt_av1_highbd_quantize_fp_qm_hip", 3, True, 0, [p(qmv), p(iqmv), 0])


def test_handle_transform(be, oracle):
    g = rng(6)
    oracle.oracle_handle_transform.restype = C.c_uint64
    for ts, (w, h) in ((4, (64, 64)), (11, (32, 64)), (12, (64, 32)), (17, (16, 64)), (18, (64, 16))):
        for n2n4 in (0, 1):
            n = 3
            x = g.integers(-(1 << 20), 1 << 20, (n, w * h)).astype(np.int32)
            d = be.dev(x)
            e = be.empty(n, np.uint64)
            be.lib.svt_hip_handle_transform_batch(be.ptr(d), n, ts, n2n4, be.ptr(e), be.stream)
            got, ge = be.host(d), be.host(e)
            kept = min(w, 32) * min(h, 32)
            for i in range(n):
                a = x[i].copy()
                ea = oracle.oracle_handle_transform(p(a), w, h, n2n4)
                assert ea == int(ge[i]) and np.array_equal(a[:kept], got[i][:kept]), (w, h, n2n4, i)
            y = x[0].copy()
            f = getattr(be.lib, "svt_handle_transform%dx%d%s_hip" % (w, h, "_N2_N4" if n2n4 else ""))
            a = x[0].copy()
            assert f(p(y)) == oracle.oracle_handle_transform(p(a), w, h, n2n4) and np.array_equal(a[:kept], y[:kept])
