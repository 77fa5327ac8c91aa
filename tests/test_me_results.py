"""ME result formatting: reference pruning on ME SADs, MeSbResults candidate lists, per-SB distortion statistics and
GM-detection flags.  The oracle restatement is pinned against the reference's own static functions (compiled where they lie into
oracle/_ref/libsvtref_me.so), then the HIP path is checked against the oracle through the C-ABI."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, load_pkg, p, rng

REF_ME_LIB = os.path.join(ROOT, "oracle", "_ref", "libsvtref_me.so")
NPU = 85

# (lists, refs L0, refs L1, 16x16, 8x8, only_l_bwd, best-unipred-only, prune_ref, ref th, cand th, low-res, gm, gm distance th)
CONFIGS = [
    (1, 1, 0, 1, 1, 0, 0, 0, 0xffff, 0, 0, 0, 0),
    (1, 1, 0, 1, 0, 0, 0, 1, 30, 0, 0, 1, 0),
    (2, 1, 1, 1, 1, 0, 0, 0, 0xffff, 0, 0, 1, 1),
    (2, 1, 1, 1, 0, 0, 1, 1, 80, 50, 0, 1, 0),
    (2, 1, 1, 0, 0, 0, 1, 0, 80, 30, 1, 1, 1),
    (1, 1, 1, 1, 1, 0, 0, 1, 40, 65, 0, 0, 0),
    (2, 4, 3, 1, 1, 0, 0, 1, 30, 65, 0, 1, 1),
    (2, 4, 3, 1, 0, 0, 0, 1, 0xffff, 0, 1, 1, 0),
    (2, 2, 2, 1, 1, 1, 0, 1, 10, 20, 1, 1, 1),
    (2, 3, 1, 0, 0, 0, 0, 1, 60, 100, 0, 1, 0),
    (1, 3, 0, 1, 1, 0, 0, 1, 25, 40, 0, 1, 1),
    (2, 2, 3, 1, 1, 0, 0, 0, 30, 5, 1, 1, 0),
]


def make_params(pkg, cfg, n_sb, g):
    nl, r0, r1, e16, e8, olb, ubu, pr, rth, cth, low, gm, gmd = cfg
    P = pkg.MeResultsParams()
    P.n_sb, P.num_of_list_to_search = n_sb, nl
    P.num_of_ref_pic_to_search[0], P.num_of_ref_pic_to_search[1] = r0, r1
    P.max_refs, P.max_cand = pkg.me_max_allocated_refs(r0, r1)
    P.max_l0 = r0
    P.enable_me_16x16, P.enable_me_8x8, P.only_l_bwd, P.use_best_unipred_cand_only = e16, e8, olb, ubu
    P.prune_ref, P.low_resolution, P.gm_enabled, P.gm_use_distance_based_active_th = pr, low, gm, gmd
    P.prune_ref_if_me_sad_dev_bigger_than_th, P.prune_me_candidates_th = rth, cth
    P.picture_number = 40
    for l in range(2):
        for r in range(4):
            P.ref_picture_number[l][r] = int(40 + (1 if l else -1) * (1 + 2 * r) * int(g.integers(1, 4)))
    return P


def make_tables(g, cfg, n_sb):
    """Search tables with the statistics that make the pruning rules bite: per-reference quality offsets, a few exact ties, small MVs."""
    nl, r0, r1 = cfg[:3]
    slots = r0 + r1
    base = g.integers(200, 3000, (1, n_sb, NPU)).astype(np.int64)
    scale = np.array([1] + [4] * 4 + [16] * 16 + [64] * 64)[::-1].copy()  # bigger blocks, bigger SADs
    scale = np.array([64] + [16] * 4 + [4] * 16 + [1] * 64)
    qual = g.choice([1.0, 1.02, 1.1, 1.4, 2.5], (slots, n_sb, 1))
    sad = (base * scale * qual + g.integers(0, 40, (slots, n_sb, NPU))).astype(np.uint32)
    sad[:, ::7, 30:40] = sad[:1, ::7, 30:40]  # ties between references
    kind = g.integers(0, 3, (slots, n_sb, 1))
    mvx = np.where(kind == 0, g.integers(-1, 2, (slots, n_sb, NPU)), g.integers(-40, 41, (slots, n_sb, NPU))) + np.where(kind == 2, 30, 0)
    mvy = np.where(kind == 0, g.integers(-1, 2, (slots, n_sb, NPU)), g.integers(-20, 21, (slots, n_sb, NPU))) - np.where(kind == 2, 25, 0)
    mv = ((mvy.astype(np.int64) & 0xffff) << 16 | (mvx.astype(np.int64) & 0xffff)).astype(np.uint32)
    do_ref = (g.random((n_sb, 2, 4)) < 0.85).astype(np.uint8)
    do_ref[::5] = 1
    do_ref[:, 0, 0] |= (g.random(n_sb) < 0.9).astype(np.uint8)
    sb_size = np.full((n_sb, 2), 64, np.uint8)
    sb_size[-3:, 1] = 56
    sb_size[1::4, 0] = 48
    return sad, mv, do_ref, sb_size


def expand(tab, cfg, sb):
    """[slot][sb][85] -> the reference's p_sb_best_*[2][4][85] for one SB (untouched slots zero, as init_me_hme_data leaves the MVs)."""
    r0, r1 = cfg[1:3]
    out = np.zeros((2, 4, NPU), np.uint32)
    out[0, :r0] = tab[:r0, sb]
    out[1, :r1] = tab[r0:r0 + r1, sb]
    return out


def run_cpu(fn, pkg, P, cfg, sad, mv, do_ref, sb_size, fill):
    n_sb = sad.shape[1]
    n_pus = 85 if (cfg[3] and cfg[4]) else 21 if cfg[3] else 5
    total = np.full((n_sb, n_pus), fill, np.uint8)
    mvs = np.full((n_sb, n_pus * P.max_refs), 0x01010101 * fill, np.uint32)
    cands = np.full((n_sb, n_pus * P.max_cand), fill, np.uint8)
    stats = np.zeros(n_sb, pkg.MeSbStats)
    dr = do_ref.copy()
    for sb in range(n_sb):
        s, m = expand(sad, cfg, sb), expand(mv, cfg, sb)
        fn(C.byref(P), p(s), p(m), C.c_void_p(dr[sb].ctypes.data), int(sb_size[sb, 0]), int(sb_size[sb, 1]), C.c_void_p(total[sb].ctypes.data),
           C.c_void_p(mvs[sb].ctypes.data), C.c_void_p(cands[sb].ctypes.data), C.c_void_p(stats[sb:sb + 1].ctypes.data))
    return total, mvs, cands, stats, dr


def same(a, b, tag):
    for x, y, name in zip(a, b, ("total", "mv", "cand", "stats", "do_ref")):
        if name == "do_ref":  # slots beyond the searched references are bookkeeping only
            continue
        assert np.array_equal(x, y), (tag, name, np.argwhere(np.atleast_1d(x != y))[:5])


@pytest.mark.parametrize("ci", range(len(CONFIGS)))
def test_me_results_oracle_vs_reference(oracle, ref, ci):
    if not os.path.exists(REF_ME_LIB):
        pytest.skip("oracle/_ref/libsvtref_me.so not available")
    refme = C.CDLL(REF_ME_LIB)
    pkg, cfg, g = load_pkg(), CONFIGS[ci], rng(900 + ci)
    n_sb = 24
    P = make_params(pkg, cfg, n_sb, g)
    sad, mv, do_ref, sb_size = make_tables(g, cfg, n_sb)
    for fill in (0, 0xaa, 0xff):  # stale output bytes are part of the contract (GM detection reads the first candidate slot)
        a = run_cpu(oracle.oracle_me_results_sb, pkg, P, cfg, sad, mv, do_ref, sb_size, fill)
        b = run_cpu(refme.ref_me_results_sb, pkg, P, cfg, sad, mv, do_ref, sb_size, fill)
        same(a, b, (cfg, fill))
        assert np.array_equal(a[4], b[4]), "do_ref"


@pytest.mark.parametrize("ci", range(len(CONFIGS)))
def test_me_results_hip(be, oracle, ci):
    pkg, cfg, g = load_pkg(), CONFIGS[ci], rng(900 + ci)
    n_sb = 24 if not be.is_gpu else 510
    P = make_params(pkg, cfg, n_sb, g)
    sad, mv, do_ref, sb_size = make_tables(g, cfg, n_sb)
    n_pus = 85 if (cfg[3] and cfg[4]) else 21 if cfg[3] else 5
    for fill in (0, 0xaa):
        want = run_cpu(oracle.oracle_me_results_sb, pkg, P, cfg, sad, mv, do_ref, sb_size, fill)
        d_sad, d_mv, d_do, d_sz = be.dev(sad), be.dev(mv), be.dev(do_ref), be.dev(sb_size)
        d_tot = be.dev(np.full((n_sb, n_pus), fill, np.uint8))
        d_mvs = be.dev(np.full((n_sb, n_pus * P.max_refs), 0x01010101 * fill, np.uint32))
        d_cand = be.dev(np.full((n_sb, n_pus * P.max_cand), fill, np.uint8))
        d_st = be.empty(n_sb, pkg.MeSbStats)
        be.lib.svt_hip_me_results_batch(C.addressof(P), be.ptr(d_sad), be.ptr(d_mv), be.ptr(d_do), be.ptr(d_sz), be.ptr(d_tot), be.ptr(d_mvs), be.ptr(d_cand),
                                        be.ptr(d_st), be.stream)
        got = (be.host(d_tot), be.host(d_mvs), be.host(d_cand), be.host(d_st), be.host(d_do))
        same(want, got, (cfg, fill))
        searched = np.zeros((2, 4), bool)
        searched[0, :cfg[1]] = cfg[0] >= 1
        searched[1, :cfg[2]] = cfg[0] >= 2
        assert np.array_equal(want[4][:, searched], got[4][:, searched])


def test_me_session_results(be, oracle):
    """The ME session returning the stage's final product: search + formatting on the device; compared with the oracle's formatting of the
    session's own raw tables (which test_sad.py pins against the oracle search)."""
    pkg, g = load_pkg(), rng(78)
    W, H, PAD = (384, 200, 68) if be.is_gpu else (128, 72, 20)
    stride, rows = W + 2 * PAD, H + 2 * PAD + 64
    aw, ah, lib = 16, 9, be.lib
    sess = lib.svt_hip_me_session_create(W, H, stride, PAD, PAD, rows, 4, 3, aw, ah, 2)
    base = g.integers(0, 256, (rows + 8, stride + 8), dtype=np.uint8)
    pics = [np.ascontiguousarray(base[k:k + rows, 2 * k:2 * k + stride] // 2 + g.integers(0, 20, (rows, stride), dtype=np.uint8)) for k in range(4)]
    sbs_x, sbs_y = (W + 63) // 64, (H + 63) // 64
    sbs = sbs_x * sbs_y
    for k in range(3):
        assert lib.svt_hip_me_session_submit(sess, k, p(pics[k]), None, 0, aw, ah, 0, None, None) >= 0
    cfg = (2, 2, 1, 1, 1, 0, 0, 1, 30, 40, 0, 1, 1)
    P = make_params(pkg, cfg, 0, g)
    refs = np.array([2, 1, 0], np.int64)  # list 0: pictures 2, 1; list 1: picture 0
    n_pus = 85
    do_ref = np.ones((sbs, 2, 4), np.uint8)
    do_ref[1, 0, 1] = 0
    out = dict(do_ref=do_ref.copy(), total=np.full((sbs, n_pus), 7, np.uint8), mv=np.full((sbs, n_pus * P.max_refs), 7, np.uint32),
               cand=np.full((sbs, n_pus * P.max_cand), 7, np.uint8), stats=np.zeros(sbs, pkg.MeSbStats), bs=np.zeros((3, sbs, 85), np.uint32),
               bm=np.zeros((3, sbs, 85), np.uint32))
    Hst = pkg.MeResultsHost(p(out["do_ref"]), p(out["total"]), p(out["mv"]), p(out["cand"]), p(out["stats"]), p(out["bs"]), p(out["bm"]))
    bad = make_params(pkg, (2, 1, 1) + cfg[3:], 0, g)
    assert lib.svt_hip_me_session_submit_results(sess, 3, p(pics[3]), p(refs), 3, aw, ah, 0, C.addressof(bad), C.addressof(Hst)) == -4
    slot = lib.svt_hip_me_session_submit_results(sess, 3, p(pics[3]), p(refs), 3, aw, ah, 0, C.addressof(P), C.addressof(Hst))
    assert slot >= 0
    lib.svt_hip_me_session_wait(sess, slot)
    lib.svt_hip_me_session_destroy(sess)
    assert out["bs"].any() and out["total"].max() < 7
    sb_size = np.array([[min(64, ((W + 7) & ~7) - 64 * (i % sbs_x)), min(64, ((H + 7) & ~7) - 64 * (i // sbs_x))] for i in range(sbs)], np.uint8)
    P.n_sb = sbs
    want = run_cpu(oracle.oracle_me_results_sb, pkg, P, cfg, out["bs"], out["bm"], do_ref, sb_size, 0)
    same(want, (out["total"], out["mv"], out["cand"], out["stats"], out["do_ref"]), "session")
    assert np.array_equal(want[4][:, :, :2], out["do_ref"][:, :, :2])
