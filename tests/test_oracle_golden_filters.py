"""Oracle restatements vs the committed golden vectors of the filter / statistics families (tests/golden/filters.npz: inputs and the
REAL reference's outputs, written by tools/gen_golden.py in the build container).  Runs wherever the oracle builds; the reference
sources are not needed, so this is the pin that also holds on the GPU box."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, p
from quant_common import make_qparams
from test_oracle_pin_cdef import BSTRIDE, in_ptr
from test_oracle_pin_quant import run_oracle
from test_oracle_pin_restoration import aligned_i16


@pytest.fixture(scope="module")
def z():
    path = os.path.join(GOLDEN, "filters.npz")
    assert os.path.exists(path), "run tools/gen_golden.py in the build container"
    return np.load(path)


def test_cdef_golden(oracle, z):
    tiles = {bd: np.ascontiguousarray(z["cdef_tiles"][i]) for i, bd in enumerate((8, 10, 12))}
    for (bd, by, bx, d, var) in z["cdef_dir"]:
        v = C.c_int32(0)
        got = oracle.oracle_cdef_find_dir(in_ptr(tiles[int(bd)], int(by) * 8 * BSTRIDE + int(bx) * 8), BSTRIDE, C.byref(v), int(bd) - 8)
        assert (got & 255, v.value) == (d, var)
    for cfg, want in zip(z["cdef_cfg"], z["cdef_out"]):
        bd, by, bx, pri, sec, damp, dirn, sub = (int(v) for v in cfg)
        cs = bd - 8
        o = np.zeros(64, np.uint16)
        oracle.oracle_cdef_filter_block(None, p(o), 8, in_ptr(tiles[bd], by * 8 * BSTRIDE + bx * 8), pri << cs, sec << cs, dirn, damp + cs, damp + cs - 1, 8, 8, cs, sub)
        assert np.array_equal(o, want), tuple(cfg)


@pytest.mark.parametrize("bd", [8, 10])
def test_restoration_golden(oracle, z, bd):
    S, w, h = 96, 48, 20
    hb = int(bd > 8)
    src = np.ascontiguousarray(z["lr_src_%d" % bd])
    sp = C.c_void_p(src.ctypes.data + (6 * S + 8) * src.itemsize)
    fx, fy = aligned_i16(z["lr_taps_%d" % bd][0]), aligned_i16(z["lr_taps_%d" % bd][1])
    d = np.zeros((h, S), src.dtype)
    oracle.oracle_wiener_convolve_add_src(sp, S, p(d), S, p(fx), p(fy), w, h, bd, hb)
    assert np.array_equal(d[:, :w], z["lr_wiener_%d" % bd])
    for k, idx in enumerate((0, 7, 10, 14)):
        a0, a1 = np.full((h, w), -7, np.int32), np.full((h, w), -7, np.int32)
        oracle.oracle_selfguided_restoration(sp, w, h, S, p(a0), p(a1), w, idx, bd, hb)
        assert np.array_equal(np.stack([a0, a1]), z["lr_sgr_%d" % bd][k]), idx


def test_deblock_golden(oracle, z):
    for cfg, a, want in zip(z["lpf_cfg"], z["lpf_in"], z["lpf_out"]):
        bd, ln, vert, bl, li, th = (int(v) for v in cfg)
        b = np.ascontiguousarray(a.astype(np.uint8 if bd == 8 else np.uint16))
        oracle.oracle_lpf(C.c_void_p(b.ctypes.data + (16 * 32 + 16) * b.itemsize), 32, int(bd > 8), vert, ln, bl, li, th, bd)
        assert np.array_equal(b.astype(np.uint16), want), tuple(cfg)


def test_downsample_golden(oracle, z):
    src = np.ascontiguousarray(z["down_src"])
    for step in (2, 4):
        o = np.zeros((24, 40), np.uint8)
        oracle.oracle_downsample_2d(p(src), 70, 66, 40, p(o), 40, step)
        assert np.array_equal(o, z["down_%d" % step])
        assert o.any()


def test_cdef_strength_pick_golden(oracle, z):
    oracle.oracle_search_one_dual.restype = C.c_uint64
    m0, m1 = np.ascontiguousarray(z["pick_m0"]), np.ascontiguousarray(z["pick_m1"])
    la, lb = np.zeros(9, np.int32), np.zeros(9, np.int32)
    tot = [oracle.oracle_search_one_dual(p(la), p(lb), nb, p(m0), p(m1), 57, 0, 64) for nb in range(4)]
    assert tot == [int(v) for v in z["pick_tot"]]
    assert np.array_equal(la, z["pick_lev0"]) and np.array_equal(lb, z["pick_lev1"])


def test_wiener_stats_golden(oracle, z):
    dgd, src = np.ascontiguousarray(z["stats_dgd"]), np.ascontiguousarray(z["stats_src"])
    M, H = np.zeros(49, np.int64), np.zeros(49 * 49, np.int64)
    oracle.oracle_compute_stats(7, p(dgd), p(src), 5, 69, 4, 52, 80, 80, p(M), p(H), 10)
    assert np.array_equal(M, z["stats_M"]) and np.array_equal(H, z["stats_H"])


def test_quantize_b_golden(oracle, z):
    P = make_qparams(88, 112, fp=False)
    scan = np.ascontiguousarray(z["quant_scan"])
    for mode in (0, 1):
        coeff = np.ascontiguousarray(z["quant_coeff"][mode])
        q, dq, eob = run_oracle(oracle, mode, False, coeff, 256, P, scan, None, None, 0)
        assert np.array_equal(np.concatenate([q, dq, [eob]]), z["quant_out"][mode])
        assert eob > 0
