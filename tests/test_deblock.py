"""Deblocking edge filters: svt_aom_lpf_{horizontal,vertical}_{4,6,8,14} and the highbd family.  The oracle restatement
(one routine for 8/10/12 bit) is pinned against the reference's sixteen `_c` functions; the HIP single-call symbols and the batched edge-list
form (vertical pass, then horizontal pass over a whole plane, as deblocking_filter.c orders them) are compared with the oracle, bit-exact.
Inputs follow test/DeblockTest.cc: random samples, smooth ramps that trigger the flat / flat2 branches, random limit bytes."""
import ctypes as C

import numpy as np
import pytest

from conftest import p, rng

LENS = (4, 6, 8, 14)


def make_patch(g, bd, kind, shape=(32, 32)):
    mx = (1 << bd) - 1
    if kind == 0:
        a = g.integers(0, mx + 1, shape)
    elif kind == 1:  # almost flat: exercises the 7 / 13-tap branches
        a = int(g.integers(8, mx - 8)) + g.integers(-1, 2, shape) * (1 << (bd - 8))
    else:            # a step across the diagonal plus small noise
        a = np.where(np.arange(shape[1])[None, :] + np.arange(shape[0])[:, None] < shape[0], mx // 3, mx // 3 + (6 << (bd - 8))) + g.integers(-1, 2, shape)
    return np.clip(a, 0, mx)


def limits(g, kind):
    if kind == 0:
        return int(g.integers(0, 256)), int(g.integers(0, 64)), int(g.integers(0, 16))
    return int(g.integers(40, 256)), int(g.integers(8, 64)), int(g.integers(0, 4))


def lim_arrays(bl, li, th):
    """the RTCD signatures take the three limit bytes by pointer; the arrays must outlive the call"""
    return [np.array([v] * 16, np.uint8) for v in (bl, li, th)]


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_lpf_oracle_vs_reference(oracle, ref, bd):
    g = rng(200 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    for ln in LENS:
        for vert in (0, 1):
            name = "svt_aom_%slpf_%s_%d_c" % ("" if bd == 8 else "highbd_", "vertical" if vert else "horizontal", ln)
            f = getattr(ref, name)
            for it in range(60):
                a = make_patch(g, bd, it % 3).astype(dt)
                b = a.copy()
                bl, li, th = limits(g, it % 2)
                off = (16 * 32 + 16) * a.itemsize
                oracle.oracle_lpf(C.c_void_p(a.ctypes.data + off), 32, int(bd > 8), vert, ln, bl, li, th, bd)
                keep = lim_arrays(bl, li, th)
                args = [C.c_void_p(b.ctypes.data + off), C.c_int32(32)] + [p(k) for k in keep]
                if bd > 8:
                    args.append(C.c_int32(bd))
                f(*args)
                assert np.array_equal(a, b), (name, it, bl, li, th)


@pytest.mark.parametrize("bd", [8, 10])
def test_lpf_single_call_symbols(be, oracle, bd):
    g = rng(210 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    for ln in LENS:
        for vert in (0, 1):
            f = getattr(be.lib, "svt_aom_%slpf_%s_%d_hip" % ("" if bd == 8 else "highbd_", "vertical" if vert else "horizontal", ln))
            for it in range(12 if be.is_gpu else 3):
                a = make_patch(g, bd, it % 3).astype(dt)
                b = a.copy()
                bl, li, th = limits(g, it % 2)
                off = (16 * 32 + 16) * a.itemsize
                oracle.oracle_lpf(C.c_void_p(a.ctypes.data + off), 32, int(bd > 8), vert, ln, bl, li, th, bd)
                keep = lim_arrays(bl, li, th)
                args = [C.c_void_p(b.ctypes.data + off), 32] + [p(k) for k in keep]
                if bd > 8:
                    args.append(bd)
                f(*args)
                assert np.array_equal(a, b), (ln, vert, it)


@pytest.mark.parametrize("bd", [8, 10])
def test_lpf_edges_batch_plane(be, oracle, bd):
    """A whole plane: every 8th column boundary (vertical pass), then every 8th row boundary (horizontal pass), 4-sample segments with a
    random filter length that fits the spacing."""
    g = rng(220 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    w, h = (640, 360) if be.is_gpu else (96, 64)
    plane = np.empty((h, w), dt)
    for y0 in range(0, h, 32):
        for x0 in range(0, w, 32):
            plane[y0:y0 + 32, x0:x0 + 32] = make_patch(g, bd, int(g.integers(0, 3)), (min(32, h - y0), min(32, w - x0)))
    want = plane.copy()
    d_plane = be.dev(plane)
    # Edge layouts the standard allows: a 14-tap filter needs 16-sample blocks on both sides, so segments alternate between a 16-sample grid
    # (any length) and an 8-sample grid (lengths up to 8); anything denser would make neighbouring segments overlap (6 + 3 > 8).
    for vert in (1, 0):
        edges = []
        along_n, across_n = (h, w) if vert else (w, h)
        for a in range(0, along_n - 3, 4):
            coarse = (a // 4) % 2 == 0
            for c in range(16 if coarse else 8, across_n, 16 if coarse else 8):
                ln = int(g.choice([4, 6, 8, 14] if coarse and c + 7 <= across_n else [4, 6, 8]))
                x, y = (c, a) if vert else (a, c)
                edges.append((x, y, vert, ln) + limits(g, int(g.integers(0, 2))) + ((0, 0, 0),))
        e = np.array(edges, dtype=be.pkg.LpfEdge)
        for r in e:
            oracle.oracle_lpf(C.c_void_p(want.ctypes.data + (int(r["y"]) * w + int(r["x"])) * want.itemsize), w, int(bd > 8), int(r["vertical"]), int(r["length"]),
                              int(r["blimit"]), int(r["limit"]), int(r["thresh"]), bd)
        d_e = be.dev(e)
        be.lib.svt_hip_lpf_edges_batch(be.ptr(d_plane), w, int(bd > 8), bd, be.ptr(d_e), len(e), be.stream)
    got = be.host(d_plane).reshape(h, w)
    assert np.array_equal(got, want), np.argwhere(got != want)[:8]
