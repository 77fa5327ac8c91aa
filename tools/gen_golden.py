#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libsvtref.so, built from /root/reference by
`make -C oracle ref`).  Run in the build container only; the .npz files are committed so that the oracle can be
checked on machines where the reference sources do not exist (the GPU box).  Usage: python tools/gen_golden.py [family...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN, ORACLE_LIB, REF_LIB, p  # noqa: E402


def gen_sad(ref, oracle):
    from test_oracle_pin_sad import ref_me
    g = np.random.default_rng(20250925)
    ss, rs = 96, 160
    src = g.integers(0, 256, 64 * ss, dtype=np.uint8)
    r = g.integers(0, 256, 160 * rs, dtype=np.uint8)
    r[: 40 * rs] = np.resize(src, 40 * rs)  # some exact matches -> zero SADs and ties
    ref.svt_nxm_sad_kernel_helper_c.restype = C.c_uint32
    nxm_sizes = np.array([(64, 64), (32, 16), (24, 10), (6, 4), (4, 4), (16, 64), (48, 48)], np.int32)
    nxm_out = np.array([ref.svt_nxm_sad_kernel_helper_c(p(src), ss, p(r), rs, int(h), int(w)) for (w, h) in nxm_sizes], np.uint32)
    loop_cfg = np.array([(16, 16, 24, 9, 0), (16, 8, 24, 9, 1), (32, 32, 8, 3, 0), (64, 64, 8, 3, 0), (24, 10, 15, 6, 0), (6, 4, 33, 7, 0)], np.int32)
    loop_out = []
    for (w, h, aw, ah, skip) in loop_cfg:
        a = [C.c_uint64(0), C.c_int16(0), C.c_int16(0)]
        ref.svt_sad_loop_kernel_c(p(src), ss, p(r), rs, int(h), int(w), C.byref(a[0]), C.byref(a[1]), C.byref(a[2]), rs, int(skip), int(aw), int(ah))
        loop_out.append([v.value for v in a])
    me_cfg = np.array([(16, 9, 0), (16, 9, 1), (15, 6, 0), (8, 3, 1), (21, 5, 0), (40, 12, 0)], np.int32)
    me_sad, me_mv = [], []
    for (aw, ah, sub) in me_cfg:
        bs, bm = ref_me(oracle, ref, p(src), ss, p(r), rs, -5, -2, int(aw), int(ah), int(sub))
        me_sad.append(bs)
        me_mv.append(bm)
    np.savez_compressed(os.path.join(GOLDEN, "sad.npz"), src=src, ref=r, src_stride=ss, ref_stride=rs, nxm_sizes=nxm_sizes, nxm_out=nxm_out,
                        loop_cfg=loop_cfg, loop_out=np.array(loop_out, np.int64), me_cfg=me_cfg, me_sad=np.array(me_sad), me_mv=np.array(me_mv))


def gen_txfm(ref, oracle):
    from test_oracle_pin_txfm import TXW, TXH, allowed_types, ref_fwd, call_ref_inv
    g = np.random.default_rng(77)
    cfg, arrays = [], {}
    for ts in range(19):
        types = allowed_types(ts)
        for bd in (8, 10):
            for tx_type in sorted(set([types[0], types[-1], types[len(types) // 2]])):
                w, h = TXW[ts], TXH[ts]
                amp = (1 << bd) - 1
                res = g.integers(-amp, amp + 1, h * w).astype(np.int16)
                coeff = ref_fwd(ref, ts, res, w, tx_type, bd)
                pred = g.integers(0, 1 << bd, h * w).astype(np.uint16)
                recon = np.zeros(h * w, np.uint16)
                iw, ih = min(w, 32), min(h, 32)
                packed = np.ascontiguousarray(coeff.reshape(h, w)[:ih, :iw]).reshape(-1)
                call_ref_inv(ref, ts, packed, pred, w, recon, tx_type, bd)
                i = len(cfg)
                cfg.append((ts, tx_type, bd))
                arrays["res_%d" % i], arrays["coeff_%d" % i], arrays["pred_%d" % i], arrays["recon_%d" % i] = res, coeff, pred, recon
    np.savez_compressed(os.path.join(GOLDEN, "txfm.npz"), cfg=np.array(cfg, np.int32), **arrays)


FAMILIES = {"sad": gen_sad, "txfm": gen_txfm}

if __name__ == "__main__":
    os.system("make -s -C %s oracle ref" % os.path.join(ROOT, "oracle"))
    ref, oracle = C.CDLL(REF_LIB), C.CDLL(ORACLE_LIB)
    for name in (sys.argv[1:] or FAMILIES):
        FAMILIES[name](ref, oracle)
        print("golden:", name)
