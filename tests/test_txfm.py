"""Forward / inverse 2-D transforms: HIP path vs the oracle, bit-exact.
Generators follow the reference's tests: residual uniform in +-(2^bd - 1) (test/FwdTxfm2dAsmTest.cc:281); the inverse's
input is the forward transform of a random residual (test/InvTxfm2dAsmTest.cc:92-145) so coefficients stay conformant."""
import numpy as np
import pytest

from conftest import p, rng
from test_oracle_pin_txfm import TX_SIZES, TXH, TXW, allowed_types, c_defined_types


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


@pytest.mark.parametrize("ts", [t for t in range(19) if 32 in (TXW[t], TXH[t])])
def test_inv_txfm2d_add_any_type(be, oracle, ts):
    """The transform types outside AV1's allowed set that the reference's `_c` inverse still computes (32-point ADST, av1_iadst32_new, inv_transforms.c:1119-1552) and
    its own InvTxfm2dAddTest feeds (test/InvTxfm2dAsmTest.cc:755-775): the `_any_type` batch forms and the single-call symbol, u16 and u8 destinations, against the
    oracle (pinned on the reference for exactly these types: test_oracle_pin_txfm.py::test_2d_inverse_legacy_types_vs_reference)."""
    g = rng(700 + ts)
    w, h = TXW[ts], TXH[ts]
    iw, ih = min(w, 32), min(h, 32)
    types = c_defined_types(ts)
    if not be.is_gpu:
        types = [t for t in types if t not in allowed_types(ts)][:3] + allowed_types(ts)[:1]
    n, stride = len(types), w + 3
    for bd in ((8, 10, 12) if be.is_gpu else (10,)):
        coeffs = g.integers(-(1 << (bd + 8)), 1 << (bd + 8), (n, iw * ih)).astype(np.int32)
        coeffs[:, iw * ih // 2:] //= 64  # (most of the energy in the low frequencies, as a forward transform leaves it)
        pred = g.integers(0, 1 << bd, (n, h * stride)).astype(np.uint16)
        descs = np.zeros(n, dtype=be.pkg.InvTxfmDesc)
        for i in range(n):
            descs[i] = (i * iw * ih, i * h * stride, i * h * stride, stride, stride, types[i], 0, (0,) * 6)
        dco, dpr, dd = be.dev(coeffs), be.dev(pred), be.dev(descs)
        drc = be.empty(n * h * stride, np.uint16)
        be.lib.svt_hip_inv_txfm2d_add_batch_any_type(be.ptr(dco), be.ptr(dpr), be.ptr(drc), be.ptr(dd), n, ts, bd, be.stream)
        got = be.host(drc).reshape(n, h, stride)
        want = np.zeros((n, h * stride), np.uint16)
        for i in range(n):
            oracle.oracle_inv_txfm2d_add(p(coeffs[i]), p(pred[i]), stride, p(want[i]), stride, types[i], ts, bd)
            assert np.array_equal(got[i][:, :w], want[i].reshape(h, stride)[:, :w]), (TX_SIZES[ts], types[i], bd)
        if bd == 8:
            pred8 = pred.astype(np.uint8)
            dpr8 = be.dev(pred8)
            drc8 = be.empty(n * h * stride, np.uint8)
            be.lib.svt_hip_inv_txfm2d_add_batch_any_type_u8(be.ptr(dco), be.ptr(dpr8), be.ptr(drc8), be.ptr(dd), n, ts, be.stream)
            got8 = be.host(drc8).reshape(n, h, stride)
            for i in range(n):
                assert np.array_equal(got8[i][:, :w].astype(np.uint16), want[i].reshape(h, stride)[:, :w]), (TX_SIZES[ts], types[i], "u8")
        # the single-call symbol (what the dispatch pointer receives) on the first legacy type
        i = next(k for k in range(n) if types[k] not in allowed_types(ts))
        one = pred[i].copy()
        f = getattr(be.lib, "svt_av1_inv_txfm2d_add_%dx%d_hip" % (w, h))
        if w == h:
            f(p(coeffs[i]), p(one), stride, p(one), stride, types[i], bd)
        else:
            f(p(coeffs[i]), p(one), stride, p(one), stride, types[i], ts, w * h, bd)
        assert np.array_equal(one.reshape(h, stride)[:, :w], want[i].reshape(h, stride)[:, :w]), (TX_SIZES[ts], types[i], bd, "single call")


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
        # the pointer's own prototype with TxfmParam.bd = 10 on the 8-bit destination (no caller does this; svt_av1_inv_txfm_add_c defines it as widen ->
        # high-bit-depth inverse clipping at 1023 -> truncating narrow)
        tp = np.zeros(1, be.pkg.TxfmParam)
        tp[0] = (tt, ts, 0, 10, 0, 0, 64)
        g10, w10 = pred8.copy(), np.zeros(h * (w + 4), np.uint16)
        be.lib.svt_av1_inv_txfm_add_hip(p(packed), p(g10), w + 4, p(g10), w + 4, p(tp))
        pred16 = pred8.astype(np.uint16)  # (named: p() of a temporary would hand the oracle freed memory)
        oracle.oracle_inv_txfm2d_add(p(packed), p(pred16), w + 4, p(w10), w + 4, tt, ts, 10)
        assert np.array_equal(g10[:h * (w + 4)].reshape(h, w + 4)[:, :w], w10.reshape(h, w + 4)[:, :w].astype(np.uint8)), (ts, "bd 10 on u8")


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


def _roundtrip_case(be, oracle, ts, g, bd, fp, types, plist, scans, iscans, qmt, iqmt, n, check_chain):
    """One svt_hip_txfm_quant_roundtrip_batch launch checked against the CPU checker's composition oracle_fwd_txfm2d -> oracle_handle_transform ->
    oracle_quantize -> oracle_inv_txfm2d_add block by block (qcoeff, dqcoeff, eob, recon), and -- check_chain -- against the HIP four-launch chain."""
    from quant_common import oracle_roundtrip
    w, h = TXW[ts], TXH[ts]
    iw, ih = min(w, 32), min(h, 32)
    ncoef, pels = iw * ih, w * h
    ls = int(pels > 256) + int(pels > 1024)
    qm = qmt is not None
    qmode = (1 if bd > 8 else 0) + 2 * fp
    amp = (1 << bd) - 1
    dt = np.uint16 if bd > 8 else np.uint8
    stride = w + 3
    res = g.integers(-amp, amp + 1, (n, h * stride)).astype(np.int16)
    res[0, :] = amp  # extreme block
    if n > 2:
        res[2, :] = g.integers(-3, 4, h * stride)  # nearly flat residual: mostly-zero qcoeff, small eob
    pred = g.integers(0, amp + 1, (n, h * stride)).astype(dt)
    params = np.zeros(len(plist), dtype=be.pkg.QuantParams)
    for i, P in enumerate(plist):
        params[i] = (P["zbin"], P["round"], P["quant"], P["quant_shift"], P["dequant"], ls)
    nqm = 1 if not qm else qmt.shape[0]
    rd = np.zeros(n, dtype=be.pkg.RoundtripDesc)
    for i in range(n):
        tt = types[i % len(types)]
        rd[i] = (i * h * stride, i * h * stride, i * h * stride, stride, stride, stride, i % len(plist), tt if scans.shape[0] == 16 else i % scans.shape[0],
                 (i // 2) % nqm, tt, (0,) * 7)
    d_res, d_pred, d_rd, d_par, d_is = be.dev(res), be.dev(pred), be.dev(rd), be.dev(params), be.dev(iscans)
    d_qm, d_iqm = (be.dev(qmt), be.dev(iqmt)) if qm else (None, None)
    q1, dq1, e1, rec1 = be.empty((n, ncoef), np.int32), be.empty((n, ncoef), np.int32), be.empty(n, np.uint16), be.empty((n, h * stride), dt)
    be.lib.svt_hip_txfm_quant_roundtrip_batch(be.ptr(d_res), be.ptr(d_pred), be.ptr(rec1), be.ptr(d_rd), n, ts, bd, qmode, be.ptr(d_par), be.ptr(d_is),
                                              be.ptr(d_qm) if qm else None, be.ptr(d_iqm) if qm else None, be.ptr(q1), be.ptr(dq1), be.ptr(e1), be.stream)
    gq, gdq, ge, grec = be.host(q1), be.host(dq1), be.host(e1), be.host(rec1).reshape(n, h, stride)[:, :, :w]
    for i in range(n):
        d = rd[i]
        si, qi = int(d["iscan_idx"]), int(d["qm_idx"])
        wq, wdq, weob, wrec = oracle_roundtrip(oracle, res[i], stride, pred[i].astype(np.uint16), stride, w, h, int(d["tx_type"]), ts, bd, qmode,
                                               plist[int(d["qparam_idx"])], scans[si], qmt[qi] if qm else None, iqmt[qi] if qm else None, ls)
        tag = (TX_SIZES[ts], bd, qmode, int(d["tx_type"]), int(d["qparam_idx"]), i)
        assert np.array_equal(gq[i], wq), ("qcoeff",) + tag
        assert np.array_equal(gdq[i], wdq), ("dqcoeff",) + tag
        assert int(ge[i]) == weob, ("eob",) + tag
        assert np.array_equal(grec[i].astype(np.uint16), wrec), ("recon",) + tag
    assert np.count_nonzero(gq) > 0
    if not check_chain:
        return
    fd, idesc, qd = np.zeros(n, dtype=be.pkg.FwdTxfmDesc), np.zeros(n, dtype=be.pkg.InvTxfmDesc), np.zeros(n, dtype=be.pkg.QuantDesc)
    for i in range(n):
        tt = int(rd[i]["tx_type"])
        fd[i] = (i * h * stride, stride, tt, (0, 0, 0))
        idesc[i] = (i * ncoef, i * h * stride, i * h * stride, stride, stride, tt, 0, (0,) * 6)
        qd[i] = (int(rd[i]["qparam_idx"]), int(rd[i]["iscan_idx"]), int(rd[i]["qm_idx"]), 0)
    d_fd, d_id, d_qd = be.dev(fd), be.dev(idesc), be.dev(qd)
    co = be.empty((n, pels), np.int32)
    be.lib.svt_hip_fwd_txfm2d_batch(be.ptr(d_res), be.ptr(d_fd), n, ts, bd, 0, be.ptr(co), be.stream)
    if max(w, h) == 64:
        en = be.empty(n, np.uint64)
        be.lib.svt_hip_handle_transform_batch(be.ptr(co), n, ts, 0, be.ptr(en), be.stream)
        co = be.dev(be.host(co).reshape(n, pels)[:, :ncoef].copy())  # packed in place to 32-wide rows; the blocks still start W*H apart
    q2, dq2, e2, rec2 = be.empty((n, ncoef), np.int32), be.empty((n, ncoef), np.int32), be.empty(n, np.uint16), be.empty((n, h * stride), dt)
    be.lib.svt_hip_quantize_batch(qmode, be.ptr(co), n, ncoef, be.ptr(d_par), be.ptr(d_is), be.ptr(d_qm) if qm else None, be.ptr(d_iqm) if qm else None,
                                  be.ptr(d_qd), be.ptr(q2), be.ptr(dq2), be.ptr(e2), be.stream)
    if bd > 8:
        be.lib.svt_hip_inv_txfm2d_add_batch(be.ptr(dq2), be.ptr(d_pred), be.ptr(rec2), be.ptr(d_id), n, ts, bd, be.stream)
    else:
        be.lib.svt_hip_inv_txfm2d_add_batch_u8(be.ptr(dq2), be.ptr(d_pred), be.ptr(rec2), be.ptr(d_id), n, ts, be.stream)
    assert np.array_equal(gq, be.host(q2)) and np.array_equal(gdq, be.host(dq2)) and np.array_equal(ge, be.host(e2)), (TX_SIZES[ts], bd, qmode, "chain")
    assert np.array_equal(grec, be.host(rec2).reshape(n, h, stride)[:, :, :w]), (TX_SIZES[ts], bd, qmode, "chain recon")


@pytest.mark.parametrize("ts", range(19))
def test_txfm_quant_roundtrip_fused(be, oracle, ts):
    """BASELINE config 3 in one launch (svt_hip_txfm_quant_roundtrip_batch) against the ORACLE composition (fwd -> svt_handle_transform -> quantize ->
    inverse + reconstruction): qcoeff, dqcoeff, eob and recon bit-identical for every TX size, 8 and 10 bit, both quantizer families, five
    quantizer steps spanning q_index 0..255, with and without quantization matrices, shuffled scans (any permutation must work); on the GPU also
    against the HIP four-launch chain."""
    from quant_common import make_qparams, make_scan
    g = rng(800 + ts)
    w, h = TXW[ts], TXH[ts]
    ncoef, pels = min(w, 32) * min(h, 32), w * h
    types = allowed_types(ts)
    if not be.is_gpu and pels >= 2048:
        types = types[:1]
    steps = [(4, 4), (20, 22), (88, 112), (336, 460), (1336, 1828)]  # (dc, ac) dequant steps from q_index 0 up to 255 (8-bit tables)
    n = len(types) * len(steps) * (2 if be.is_gpu else 1)
    if not be.is_gpu:
        n = min(n, 4 if pels >= 1024 else 12)
    for bd, fp, qm in ((8, 0, False), (10, 0, False), (10, 1, True), (8, 1, False)) if be.is_gpu else ((10, 0, False), (8, 1, True)):
        plist = [make_qparams(dc * (4 if bd > 8 else 1), ac * (4 if bd > 8 else 1), fp=bool(fp)) for (dc, ac) in steps]
        sc = [make_scan(ncoef, g) for _ in range(2)]
        scans, iscans = np.stack([s[0] for s in sc]), np.stack([s[1] for s in sc])
        qmt = g.integers(16, 255, (2, ncoef)).astype(np.uint8) if qm else None
        iqmt = g.integers(16, 64, (2, ncoef)).astype(np.uint8) if qm else None
        _roundtrip_case(be, oracle, ts, g, bd, fp, types, plist, scans, iscans, qmt, iqmt, n, check_chain=be.is_gpu)


@pytest.mark.parametrize("ts", range(19))
def test_config3_reference_tables(be, oracle, ts):
    """SURVEY 8(d) config 3 on the reference's own data: quantizer tables from svt_av1_build_quantizer (md_config_process.c:111-189) at
    q in {0, 60, 120, 180, 255} (QuantAsmTest.cc:86-96), scans from av1_scan_orders[tx_size][tx_type] (coefficients.h:2197) -- both frozen from the
    reference in tests/golden/quant_tables.npz --, tx types cycling through the allowed set (TxfmCommon.h:160-209), residuals uniform in
    +-(2^bd - 1), 8 and 10 bit, quantize_b and quantize_fp.  fwd coeffs -> qcoeff, dqcoeff, eob, recon all equal to the oracle composition."""
    from quant_common import real_qparams, real_scans
    g = rng(13596 + ts)
    w, h = TXW[ts], TXH[ts]
    pels = w * h
    types = allowed_types(ts)
    scans, iscans = real_scans(ts)
    n = len(types) * 5 * (2 if be.is_gpu else 1)
    if not be.is_gpu:
        n = min(n, 5 if pels >= 1024 else 10)
    for bd, fp in ((8, 0), (10, 0), (8, 1), (10, 1)) if be.is_gpu else ((8, 0), (10, 1)):
        _roundtrip_case(be, oracle, ts, g, bd, fp, types, real_qparams(bd, fp), scans, iscans, None, None, n, check_chain=False)
