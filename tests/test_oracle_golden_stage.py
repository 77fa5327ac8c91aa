"""Oracle restatements of the ME result formatting and the temporal-filter kernels vs the committed golden vectors (tests/golden/stage.npz:
inputs and the REAL reference's outputs, written by tools/gen_golden.py in the build container).  The reference is not needed to run this."""
import ctypes as C
import os

import numpy as np
import pytest

import test_me_results as M
import test_tf as T
from conftest import GOLDEN, load_pkg, p


@pytest.fixture(scope="module")
def z():
    path = os.path.join(GOLDEN, "stage.npz")
    assert os.path.exists(path), "run tools/gen_golden.py in the build container"
    return np.load(path)


def test_me_results_golden(oracle, z):
    pkg = load_pkg()
    for k in range(4):
        cfg = tuple(int(v) for v in z["me%d_cfg" % k])
        P = pkg.MeResultsParams.from_buffer_copy(z["me%d_params" % k].tobytes())
        got = M.run_cpu(oracle.oracle_me_results_sb, pkg, P, cfg, np.ascontiguousarray(z["me%d_sad" % k]), np.ascontiguousarray(z["me%d_mv" % k]),
                        np.ascontiguousarray(z["me%d_do_ref" % k]), z["me%d_sb_size" % k], 0xaa)
        assert np.array_equal(got[0], z["me%d_total" % k]) and np.array_equal(got[1], z["me%d_mvs" % k]) and np.array_equal(got[2], z["me%d_cands" % k]), k
        assert np.array_equal(got[3].view(np.uint8).reshape(len(got[3]), -1), z["me%d_stats" % k]) and np.array_equal(got[4], z["me%d_do_ref_out" % k]), k
        assert got[0].max() > 1 or cfg[1] + cfg[2] == 1


def test_tf_golden(oracle, z):
    pkg = load_pkg()
    for k in range(4):
        P = pkg.TfParams.from_buffer_copy(z["tf%d_params" % k].tobytes())
        blocks = np.ascontiguousarray(z["tf%d_blocks" % k]).view(pkg.TfBlock).reshape(3, 2, 2)
        central = [np.ascontiguousarray(z["tf%d_central%d" % (k, c)]) for c in range(3)]
        preds = [[np.ascontiguousarray(z["tf%d_pred%d_%d" % (k, r, c)]) for c in range(3)] for r in range(3)]
        got = T.oracle_frame(oracle, P, central, [96, 48], preds, [[64, 32]] * 3, blocks, 3, 2, 2)
        for c in range(3):
            assert np.array_equal(got[c], z["tf%d_out%d" % (k, c)]), (k, c)
        changed = locals().get("changed", 0) + (not np.array_equal(got[0], central[0]))
    assert changed >= 2


def test_noise_golden(oracle, z):
    a8, a10 = np.ascontiguousarray(z["noise_a8"]), np.ascontiguousarray(z["noise_a10"])
    assert oracle.oracle_estimate_noise_fp16(p(a8), 64, 40, 72, 8) == int(z["noise_out"][0]) > 0
    assert oracle.oracle_estimate_noise_fp16(p(a10), 64, 40, 72, 10) == int(z["noise_out"][1]) > 0


def test_hme_levels_golden(oracle, z):
    import test_hme as Hm
    W, H = 200, 136
    for k in range(3):
        level, sub, nw, nh, sa_w, sa_h = (int(v) for v in z["hme%d_case" % k])
        src, refs = np.ascontiguousarray(z["hme%d_src" % k]), [np.ascontiguousarray(z["hme%d_ref0" % k]), np.ascontiguousarray(z["hme%d_ref1" % k])]
        sh, org = {0: 2, 1: 1, 2: 0}[level], Hm.ORG[level]
        sad, sc = Hm.cpu_level(oracle.oracle_hme_level, level, sub, nw, nh, src, refs, W >> sh, H >> sh, org, src.shape[1], W, H, sa_w, sa_h, z["hme%d_prev" % k])
        assert np.array_equal(sad, z["hme%d_sad" % k]) and np.array_equal(sc, z["hme%d_sc" % k]), k


def test_integer_search_golden(oracle, z):
    import test_hme as Hm
    W, H, org = 200, 136, Hm.ORG[2]
    for k in range(2):
        P = Hm.int_params(Hm.INT_CASES[int(z["int%d_case" % k][0])])
        src, ref = np.ascontiguousarray(z["int%d_src" % k]), np.ascontiguousarray(z["int%d_ref" % k])
        sad, sc = np.ascontiguousarray(z["int%d_hme_sad" % k]), np.ascontiguousarray(z["int%d_hme_sc" % k])
        stride = src.shape[1]
        for sb in range(12):
            o_sc, o_sad, area = np.zeros(2, np.int16), C.c_uint64(0), np.zeros(4, np.int16)
            bs, bm = np.zeros(85, np.uint32), np.zeros(85, np.uint32)
            oracle.oracle_me_integer_search(C.byref(P), 4, p(sad[sb]), p(sc[sb]), p(src), stride, org, org, p(ref), stride, org, org, (sb % 4) * 64, (sb // 4) * 64,
                                            (W + 7) & ~7, (H + 7) & ~7, p(o_sc), C.byref(o_sad), p(area), p(bs), p(bm))
            assert np.array_equal(o_sc, z["int%d_sc" % k][sb]) and o_sad.value == int(z["int%d_sad" % k][sb]), (k, sb)
            assert np.array_equal(bs, z["int%d_bs" % k][sb]) and np.array_equal(bm, z["int%d_bm" % k][sb]), (k, sb)
