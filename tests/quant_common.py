"""Shared quantizer test data: parameter sets built with libaom's formulas (the reference derives its tables the same way
in svt_av1_build_quantizer, md_config_process.c:111-189) and a zig-zag-like scan with its inverse."""
import numpy as np


def msb(v):
    return int(v).bit_length() - 1


def make_qparams(dq_dc, dq_ac, fp):
    """(zbin, round, quant, quant_shift, dequant) DC/AC pairs for a given dequant step."""
    out = {k: [0, 0] for k in ("zbin", "round", "quant", "quant_shift", "dequant")}
    for k, d in enumerate((dq_dc, dq_ac)):
        out["dequant"][k] = d
        if fp:  # quant_fp = (1 << 16) / dequant, round_fp = (64 * dequant) >> 7
            out["quant"][k] = min((1 << 16) // d, 32767)
            out["round"][k] = (64 * d) >> 7
        else:   # invert_quant(): m = 1 + (1 << (16 + msb)) / d ; quant = m - (1 << 16) ; shift = 1 << (16 - msb)
            t = msb(d)
            m = 1 + (1 << (16 + t)) // d
            out["quant"][k] = np.int16(np.uint16((m - (1 << 16)) & 0xffff))
            out["quant_shift"][k] = min(1 << (16 - t), 32767)
            out["zbin"][k] = (84 * d + 64) >> 7
            out["round"][k] = (48 * d) >> 7
    return {k: np.array(v, np.int16) for k, v in out.items()}


def make_scan(n, g):
    """A diagonal-ish permutation with scan[0] == 0 (DC first, as every AV1 scan) and its inverse."""
    rest = np.arange(1, n)
    g.shuffle(rest[n // 4:])  # keep a low-frequency-first prefix, shuffle the tail
    scan = np.concatenate([[0], rest]).astype(np.int16)
    iscan = np.zeros(n, np.int16)
    iscan[scan] = np.arange(n, dtype=np.int16)
    return scan, iscan


DEQUANTS = [(4, 4), (8, 9), (20, 22), (88, 112), (336, 460), (1336, 1828), (5347, 21387 // 4)]


# ---- SURVEY 8(d) config 3 on the reference's OWN data: tests/golden/quant_tables.npz holds the luma tables svt_av1_build_quantizer
# (md_config_process.c:111-189) produces at q in {0, 60, 120, 180, 255} for 8 / 10 bit and av1_scan_orders (coefficients.h:2197) for every TX
# size and type, frozen by tools/gen_golden.py through oracle/ref_wrap/ref_quant_tables.c.
_QT = None


def _qt():
    global _QT
    if _QT is None:
        import os
        _QT = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quant_tables.npz"))
    return _QT


def real_qparams(bd, fp):
    """[(zbin, round, quant, quant_shift, dequant) dict] for the five q indices; fp selects quant_fp / round_fp (quantize_fp ignores zbin / quant_shift)."""
    t = _qt()["tables"][0 if bd == 8 else 1]
    out = []
    for i in range(t.shape[0]):
        zbin, rnd, quant, qshift, deq, quant_fp, round_fp = (t[i, k].copy() for k in range(7))
        out.append({"zbin": zbin, "round": round_fp if fp else rnd, "quant": quant_fp if fp else quant, "quant_shift": qshift, "dequant": deq})
    return out


def real_scans(ts):
    """(scan[16][n], iscan[16][n]) of av1_scan_orders[ts][*]"""
    return _qt()["scan_%d" % ts], _qt()["iscan_%d" % ts]


def oracle_roundtrip(oracle, res, in_stride, pred, pred_stride, w, h, tt, ts, bd, qmode, P, scan, qmv, iqmv, ls):
    """The config-3 chain on the CPU checker: oracle_fwd_txfm2d -> oracle_handle_transform (64-point sizes) -> oracle_quantize ->
    oracle_inv_txfm2d_add.  pred: u16 view of the prediction rows (pred_stride); returns qcoeff, dqcoeff, eob, recon[h][w] (u16)."""
    import ctypes as C
    vp = lambda a: C.c_void_p(a.ctypes.data)
    iw, ih = min(w, 32), min(h, 32)
    n = iw * ih
    co = np.zeros(w * h, np.int32)
    oracle.oracle_fwd_txfm2d(vp(res), vp(co), in_stride, tt, ts, bd, 0)
    if max(w, h) == 64:
        oracle.oracle_handle_transform.restype = C.c_uint64
        oracle.oracle_handle_transform(vp(co), w, h, 0)
    co = np.ascontiguousarray(co[:n])
    q, dq, eob = np.zeros(n, np.int32), np.zeros(n, np.int32), C.c_uint16(0)
    oracle.oracle_quantize(qmode, vp(co), n, vp(P["zbin"]), vp(P["round"]), vp(P["quant"]), vp(P["quant_shift"]), vp(q), vp(dq), vp(P["dequant"]),
                           C.byref(eob), vp(scan), vp(qmv) if qmv is not None else None, vp(iqmv) if iqmv is not None else None, ls)
    pr = np.ascontiguousarray(pred, dtype=np.uint16)
    rec = np.zeros(h * pred_stride, np.uint16)
    oracle.oracle_inv_txfm2d_add(vp(dq), vp(pr), pred_stride, vp(rec), pred_stride, tt, ts, bd)
    return q, dq, eob.value, rec.reshape(h, pred_stride)[:, :w].copy()
