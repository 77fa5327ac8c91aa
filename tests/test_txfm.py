"""Forward / inverse 2-D transforms: HIP path vs the oracle, bit-exact.
Generators follow the reference's tests: residual uniform in +-(2^bd - 1) (test/FwdTxfm2dAsmTest.cc:281); the inverse's
input is the forward transform of a random residual (test/InvTxfm2dAsmTest.cc:92-145) so coefficients stay conformant."""
import numpy as np
import pytest

from conftest import p, rng
from test_oracle_pin_txfm import TX_SIZES, TXH, TXW, allowed_types


def oracle_fwd(oracle, res, stride, tx_type, ts, bd, pf=0):
    a = np.zeros(TXW[ts] * TXH[ts], np.int32)
    oracle.oracle_fwd_txfm2d(p(res), p(a), stride, tx_type, ts, bd, pf)
    return a


@pytest.mark.parametrize("ts", range(19))
def test_fwd_txfm2d_batch(be, oracle, ts):
    g = rng(300 + ts)
    w, h = TXW[ts], TXH[ts]
    types = allowed_types(ts)
    per_type = 2 if (be.is_gpu or w * h <= 256) else 1
    n = len(types) * per_type
    if not be.is_gpu and w * h >= 2048:
        types, n = types[:2], min(n, 2)
    stride = w + 3
    for bd in ((8, 10, 12) if be.is_gpu else (10, 12)):
        amp = (1 << bd) - 1
        res = g.integers(-amp, amp + 1, (n, h * stride)).astype(np.int16)
        res[0, :] = amp  # extreme block
        descs = np.zeros(n, dtype=be.pkg.FwdTxfmDesc)
        for i in range(n):
            descs[i] = (i * h * stride, stride, types[i % len(types)], (0, 0, 0))
        for pf in (0, 1, 2):
            dres, dd = be.dev(res), be.dev(descs)
            out = be.empty(n * w * h, np.int32)
            be.lib.svt_hip_fwd_txfm2d_batch(be.ptr(dres), be.ptr(dd), n, ts, bd, pf, be.ptr(out), be.stream)
            got = be.host(out).reshape(n, w * h)
            for i in range(n):
                want = oracle_fwd(oracle, res[i], stride, int(descs[i]["tx_type"]), ts, bd, pf)
                assert np.array_equal(got[i], want), (TX_SIZES[ts], int(descs[i]["tx_type"]), bd, pf, i)


@pytest.mark.parametrize("ts", range(19))
def test_inv_txfm2d_add_batch(be, oracle, ts):
    g = rng(500 + ts)
    w, h = TXW[ts], TXH[ts]
    iw, ih = min(w, 32), min(h, 32)
    types = allowed_types(ts)
    n = len(types) * (2 if be.is_gpu else 1)
    if not be.is_gpu and w * h >= 2048:
        types, n = types[:2], 2
    stride = w + 5
    for bd in ((8, 10, 12) if be.is_gpu else (10, 12)):
        amp = (1 << bd) - 1
        coeffs = np.zeros((n, iw * ih), np.int32)
        pred = g.integers(0, 1 << bd, (n, h * stride)).astype(np.uint16)
        descs = np.zeros(n, dtype=be.pkg.InvTxfmDesc)
        for i in range(n):
            tt = types[i % len(types)]
            res = g.integers(-amp, amp + 1, h * w).astype(np.int16)
            full = oracle_fwd(oracle, res, w, tt, ts, bd)
            coeffs[i] = np.ascontiguousarray(full.reshape(h, w)[:ih, :iw]).reshape(-1)
            descs[i] = (i * iw * ih, i * h * stride, i * h * stride, stride, stride, tt, 0, (0,) * 6)
        dco, dpr, dd = be.dev(coeffs), be.dev(pred), be.dev(descs)
        drc = be.empty(n * h * stride, np.uint16)
        be.lib.svt_hip_inv_txfm2d_add_batch(be.ptr(dco), be.ptr(dpr), be.ptr(drc), be.ptr(dd), n, ts, bd, be.stream)
        got = be.host(drc).reshape(n, h, stride)
        for i in range(n):
            want = np.zeros(h * stride, np.uint16)
            oracle.oracle_inv_txfm2d_add(p(coeffs[i]), p(pred[i]), stride, p(want), stride, int(descs[i]["tx_type"]), ts, bd)
            assert np.array_equal(got[i][:, :w], want.reshape(h, stride)[:, :w]), (TX_SIZES[ts], int(descs[i]["tx_type"]), bd, i)


def test_txfm_single_call_symbols(be, oracle):
    """RTCD-signature forms: svt_av1_fwd_txfm2d_WxH[_N2|_N4]_hip, svt_av1_inv_txfm2d_add_WxH_hip (3 signature shapes), 8-bit dst."""
    g = rng(9)
    for ts in ([0, 3, 5, 8, 13, 17] if be.is_gpu else [0, 5, 8]):
        w, h = TXW[ts], TXH[ts]
        iw, ih = min(w, 32), min(h, 32)
        tt = allowed_types(ts)[-1]
        res = g.integers(-255, 256, h * (w + 2)).astype(np.int16)
        for pf, sfx in ((0, ""), (1, "_N2"), (2, "_N4")):
            out = np.zeros(w * h, np.int32)
            getattr(be.lib, "svt_av1_fwd_txfm2d_%dx%d%s_hip" % (w, h, sfx))(p(res), p(out), w + 2, tt, 8)
            assert np.array_equal(out, oracle_fwd(oracle, res, w + 2, tt, ts, 8, pf))
        full = oracle_fwd(oracle, res, w + 2, tt, ts, 8)
        packed = np.ascontiguousarray(full.reshape(h, w)[:ih, :iw]).reshape(-1)
        pred = g.integers(0, 256, h * (w + 4)).astype(np.uint16)
        got, want = pred.copy(), pred.copy()  # in place: output_r == output_w
        f = getattr(be.lib, "svt_av1_inv_txfm2d_add_%dx%d_hip" % (w, h))
        if w == h:
            f(p(packed), p(got), w + 4, p(got), w + 4, tt, 8)
        elif (w, h) in ((4, 8), (8, 4), (4, 16), (16, 4)):
            f(p(packed), p(got), w + 4, p(got), w + 4, tt, ts, 8)
        else:
            f(p(packed), p(got), w + 4, p(got), w + 4, tt, ts, w * h, 8)
        oracle.oracle_inv_txfm2d_add(p(packed), p(pred), w + 4, p(want), w + 4, tt, ts, 8)
        assert np.array_equal(got, want)
        pred8 = pred.astype(np.uint8)
        got8 = pred8.copy()
        be.lib.svt_av1_inv_txfm_add_u8_hip(p(packed), p(got8), w + 4, p(got8), w + 4, tt, ts, 0, 64)
        assert np.array_equal(got8.astype(np.uint16), want)


def test_wht4x4(be, oracle):
    """Lossless mode: svt_av1_fwht4x4_hip, the batched forward / inverse WHT and svt_av1_inv_txfm_add_hip (exact pointer prototype,
    TxfmParam.lossless / eob honoured: eob <= 1 is a different function in the reference, not a shortcut)."""
    g = rng(91)
    n = 300 if be.is_gpu else 40
    for bd in (8, 10, 12):
        amp = (1 << bd) - 1
        stride = 7
        res = g.integers(-amp, amp + 1, (n, 4 * stride)).astype(np.int16)
        res[0, :], res[1, :] = amp, -amp
        fd = np.zeros(n, dtype=be.pkg.FwdTxfmDesc)
        for i in range(n):
            fd[i] = (i * 4 * stride, stride, 0, (0, 0, 0))
        out = be.empty(n * 16, np.int32)
        dres, dfd = be.dev(res), be.dev(fd)
        be.lib.svt_hip_fwht4x4_batch(be.ptr(dres), be.ptr(dfd), n, be.ptr(out), be.stream)
        got = be.host(out).reshape(n, 16)
        want = np.zeros((n, 16), np.int32)
        for i in range(n):
            oracle.oracle_fwht4x4(p(res[i]), p(want[i]), stride)
        assert np.array_equal(got, want)
        single = np.zeros(16, np.int32)
        be.lib.svt_av1_fwht4x4_hip(p(res[2]), p(single), stride)
        assert np.array_equal(single, want[2])
        coeffs = want.copy()
        coeffs[n // 2:] = g.integers(-(amp << 4), (amp << 4) + 1, (n - n // 2, 16))
        pred = g.integers(0, amp + 1, (n, 4 * stride)).astype(np.uint16)
        idesc = np.zeros(n, dtype=be.pkg.InvTxfmDesc)
        for i in range(n):
            idesc[i] = (i * 16, i * 4 * stride, i * 4 * stride, stride, stride, 0, i & 1, (0,) * 6)
        dco, dpr, dd = be.dev(coeffs), be.dev(pred), be.dev(idesc)
        drc = be.empty(n * 4 * stride, np.uint16)
        be.lib.svt_hip_iwht4x4_add_batch(be.ptr(dco), be.ptr(dpr), be.ptr(drc), be.ptr(dd), n, bd, be.stream)
        rec = be.host(drc).reshape(n, 4, stride)
        for i in range(n):
            w = np.zeros(4 * stride, np.uint16)
            oracle.oracle_iwht4x4_add(p(coeffs[i]), p(pred[i]), stride, p(w), stride, 16 if i & 1 else 1, bd)
            assert np.array_equal(rec[i][:, :4], w.reshape(4, stride)[:, :4]), (bd, i)
        if bd == 8:
            for i in range(6):
                for eob in (0, 1, 2, 16):
                    tp = np.zeros(1, be.pkg.TxfmParam)
                    tp[0] = (0, 0, 1, 8, 0, 0, eob)
                    p8 = pred[i].astype(np.uint8)
                    w8, w16 = p8.copy(), np.zeros(4 * stride, np.uint16)
                    be.lib.svt_av1_inv_txfm_add_hip(p(coeffs[i]), p(p8), stride, p(w8), stride, p(tp))
                    oracle.oracle_iwht4x4_add(p(coeffs[i]), p(pred[i]), stride, p(w16), stride, eob, 8)
                    assert np.array_equal(w8.reshape(4, stride)[:, :4].astype(np.uint16), w16.reshape(4, stride)[:, :4]), (i, eob)
