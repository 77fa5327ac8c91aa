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
