"""TPL dispenser with the option set of tpl levels 0-3 (set_tpl_params, initial_rc_process.c:301-342; level 1 = presets M0-M2): csrc/tpl_full.hip against the
REFERENCE ITSELF -- tpl_mc_flow_dispenser_sb_generic compiled where it lies (oracle/ref_wrap/ref_tpl.c -> oracle/_ref/libsvtref_me.so, which travels to the GPU box):
all thirteen intra modes with the directional edge filter, transform + SATD costs, half- / quarter-pel vectors (svt_av1_find_best_sub_pixel_tree_pruned with the
bilinear sub-pixel variance), the regular 8-tap prediction of fractional vectors, the coefficient-rate estimate, and the reconstruction half with intra blocks of every
mode predicted from the reconstruction (above-right samples of the first block column included).  Bit-exact: the source-based statistics of every block, the TplStats
grid result_model_store leaves, and the reconstructed TPL picture."""
import ctypes as C

import numpy as np
import pytest

import test_tpl as T
from conftest import p, rng

PAD = T.PAD
# search_flags: bit 0 SATD costs, bit 1 rate, bits 2-3 sub-pel rounds, bit 4 no diagonal refinement
LEVELS = {0: dict(ime=0, flags=1, pf=0), 1: dict(ime=12, flags=1 | 2 | (2 << 2), pf=0), 2: dict(ime=12, flags=(2 << 2), pf=1), 3: dict(ime=0, flags=(2 << 2) | 16, pf=2)}
CASES = [  # tpl level, picture, references, quantizer index
    dict(tl=1, W=200, H=136, noi=0, isl=0, me16=1, me8=0, l0=2, l1=1, q=120),
    dict(tl=1, W=195, H=131, noi=0, isl=0, me16=1, me8=1, l0=1, l1=1, q=40),   # blocks less than half inside are skipped; 85-PU ME tables
    dict(tl=1, W=136, H=72, noi=0, isl=1, me16=1, me8=0, l0=1, l1=0, q=200),   # I slice: intra only
    dict(tl=1, W=264, H=152, noi=0, isl=0, me16=0, me8=0, l0=3, l1=2, q=255),  # 5-PU ME tables, an excluded reference, the coarsest quantizer
    dict(tl=0, W=200, H=136, noi=0, isl=0, me16=1, me8=0, l0=2, l1=1, q=90),
    dict(tl=2, W=200, H=136, noi=1, isl=0, me16=1, me8=0, l0=1, l1=1, q=120),  # intra prediction off for this picture (disable_intra_pred_nref at the top layer)
    dict(tl=2, W=200, H=136, noi=0, isl=0, me16=1, me8=0, l0=2, l1=2, q=60),
    dict(tl=3, W=200, H=136, noi=0, isl=0, me16=1, me8=0, l0=2, l1=1, q=120),
]
GPU_CASES = [dict(tl=1, W=1920, H=1080, noi=0, isl=0, me16=1, me8=0, l0=2, l1=2, q=140)]


def make_case(c, seed):
    """T.make_case's geometry and ME tables with pictures on which the inter path matters: smooth content, references = the source displaced by whole and half
    samples plus a little noise, and most ME vectors near the true displacement"""
    lv = LEVELS[c["tl"]]
    base = dict(c, level=0, ss=0, pf=lv["pf"])
    P, planes, tot, mvs, cand, n_pus, cells = T.make_case(base, seed)
    g = rng(seed + 77)
    rows, stride = planes.shape[1], planes.shape[2]
    yy, xx = np.mgrid[0:rows, 0:stride]
    img = 128 + 60 * np.sin(xx / 9.0 + yy / 23.0) + 40 * np.sin(yy / 7.0 - xx / 31.0) + 25 * ((xx // 24 + yy // 20) % 2)
    planes[0] = np.clip(img + g.integers(-4, 5, img.shape), 0, 255)
    planes[0, PAD + 32:PAD + 64, PAD + 16:PAD + 80] = 77
    planes[0, PAD + 64:PAD + 96, PAD + 96:PAD + 160] = np.clip(90 + (xx[:32, :64] + yy[:32, :64]) * 2, 0, 255)  # a diagonal ramp: directional modes win
    n_ref = c["l0"] + c["l1"]
    shifts = []
    for r in range(n_ref):
        dy, dx = r + 1, -2 * r - 1
        a = np.roll(planes[0].astype(np.int32), (dy, dx), (0, 1))
        b = np.roll(planes[0].astype(np.int32), (dy + (r & 1), dx + 1), (0, 1))  # + half a sample
        planes[1 + r] = np.clip(((a + b + 1) >> 1) + g.integers(-2 - r, 3 + r, a.shape), 0, 255)
        shifts.append((dx, dy))
    mv = mvs.reshape(P.n_sb, n_pus, n_ref).copy()
    near = g.random(mv.shape) < 0.7
    for r in range(n_ref):
        dx, dy = shifts[r]
        vx = (dx + g.integers(-1, 2, mv.shape[:2])).astype(np.int16)
        vy = (dy + g.integers(-1, 2, mv.shape[:2])).astype(np.int16)
        v = (vy.astype(np.uint16).astype(np.uint32) << 16) | vx.astype(np.uint16).astype(np.uint32)
        mv[:, :, r] = np.where(near[:, :, r], v, mv[:, :, r])
    P.intra_mode_end, P.search_flags = lv["ime"], lv["flags"]
    return P, planes, tot, np.ascontiguousarray(mv.reshape(mvs.shape)), cand, n_pus, cells


def run_reference(me, c, P, planes, tot, mvs, cand, n_pus, cells):
    src = np.zeros(cells, T.SrcStats)
    stats = np.zeros((cells, 8), np.int64)
    rec = np.zeros((P.height, P.width), np.uint8)
    me.ref_tpl_dispenser_picture(C.byref(P), c["q"], p(planes), PAD, PAD, p(planes), p(tot), p(mvs), p(cand), n_pus, p(src), p(stats), p(rec))
    return src, stats, rec


@pytest.mark.parametrize("ci", range(len(CASES) + len(GPU_CASES)))
def test_tpl_full_vs_reference(be, ci):
    if ci >= len(CASES) and not be.is_gpu:
        pytest.skip("full-size pictures run on the GPU only")
    if ci in (3, 6) and not be.is_gpu:
        pytest.skip("the two largest small cases run on the GPU only (the CPU suite's time budget); their option sets are covered by cases 0, 1 and 5")
    c = CASES[ci] if ci < len(CASES) else GPU_CASES[ci - len(CASES)]
    me = T.ref_lib()
    pkg = be.pkg
    P, planes, tot, mvs, cand, n_pus, cells = make_case(c, 8000 + ci)
    want_src, want_stats, want_rec = run_reference(me, c, P, planes, tot, mvs, cand, n_pus, cells)  # (fills P's quantizer row)
    w = want_src["written"] == 1
    assert w.sum() > 0
    # ---- the source-based half, device arrays ----
    d_pl, d_tot, d_mv, d_cand = be.dev(planes), be.dev(tot), be.dev(mvs), be.dev(cand)
    d_out = be.dev(np.zeros(cells * T.SrcStats.itemsize, np.uint8))
    be.lib.svt_hip_tpl_src_stage(C.addressof(P), be.ptr(d_pl), be.ptr(d_pl), be.ptr(d_tot), be.ptr(d_mv), be.ptr(d_cand), be.ptr(d_out), be.stream)
    got = be.host(d_out).view(T.SrcStats)
    T.same_stats(got, want_src, ("device, source-based half", ci))
    # what the case exercises
    lv = LEVELS[c["tl"]]
    if not c["isl"]:
        assert (want_src["best_mode"][w] == 16).sum() >= 5, "too few NEWMV blocks"
        if lv["flags"] & 12:
            assert (((want_src["mv_row"][w] | want_src["mv_col"][w]) & 7) != 0).any(), "no fractional vector"
    if not c["noi"]:
        modes = np.unique(want_src["best_intra_mode"][w])
        assert lv["ime"] == 0 or len(modes) >= 5, modes
    if lv["flags"] & 2 and not c["isl"]:
        assert (want_src["srcrf_rate"][w] > 0).any()
    # ---- the reconstruction half from the device's own statistics ----
    R = pkg.TplReconParams()
    C.memmove(C.addressof(R.src), C.addressof(P), C.sizeof(P))
    for i in range(8):
        C.memmove(C.addressof(R.rec_refs[i]), C.addressof(P.refs[i]), C.sizeof(T.TplRef))
    rec0, off, stride = T.recon_geometry(P, planes, fill=0)
    R.recon_off, R.recon_stride, R.is_ref = off, stride, 1
    d_rec = be.dev(rec0)
    d_rs = be.dev(np.zeros(cells * T.ReconStats.itemsize, np.uint8))
    be.lib.svt_hip_tpl_recon_stage(C.addressof(R), be.ptr(d_pl), be.ptr(d_pl), be.ptr(d_out), be.ptr(d_rec), be.ptr(d_rs), be.stream)
    be.sync()
    rs = be.host(d_rs).view(T.ReconStats)
    assert rs["pad"][0][0] != 0xEE, "a block gave up waiting for its neighbours"
    cols16 = (P.aligned_width + 15) // 16
    grid, seen = T.expand_like_result_model_store(P, rs, cols16)
    assert np.array_equal(seen, w)
    bad = np.nonzero((grid != want_stats[:, :4]).any(1) & seen)[0]
    assert bad.size == 0, (ci, bad[:5], grid[bad[:5]], want_stats[bad[:5], :4], want_src[bad[:5]])
    got_rec = be.host(d_rec).reshape(rec0.shape)[PAD:PAD + P.height, PAD:PAD + P.width]
    assert np.array_equal(got_rec, want_rec), (ci, int((got_rec != want_rec).sum()), np.argwhere(got_rec != want_rec)[:4])
    assert rs["coded"][w].any()
    # ---- both halves in one host call ----
    SP, RP = T.TplHostPlanes(), T.TplHostPlanes()
    rows, psize = planes.shape[1], planes.shape[1] * planes.shape[2]
    RF = pkg.TplReconParams.from_buffer_copy(R)
    SP.src_buf, SP.src_rows = planes[0].ctypes.data, rows
    copies = {}
    for r in range(8):
        if P.refs[r].valid:
            k = P.refs[r].plane_off // psize
            copies[k] = copies.get(k, planes[k].copy())
            SP.ref_buf[r], SP.ref_rows[r] = planes[k].ctypes.data, rows
            RP.ref_buf[r], RP.ref_rows[r] = copies[k].ctypes.data, rows
            RF.rec_refs[r].plane_off = 0
            RF.src.refs[r].plane_off = 0
    rec_f, out_f, src_f = rec0.copy(), np.zeros(cells, T.ReconStats), np.zeros(cells, T.SrcStats)
    assert be.lib.svt_hip_tpl_stage_host(C.addressof(RF), C.addressof(SP), C.addressof(RP), p(tot), p(mvs), p(cand), p(src_f), p(rec_f), rows, p(out_f)) == 0
    T.same_stats(src_f, want_src, ("fused host form", ci))
    for f in ("srcrf_dist", "recrf_dist", "srcrf_rate", "recrf_rate", "written", "coded"):
        assert np.array_equal(out_f[f], rs[f]), ("fused host form", ci, f)
    assert np.array_equal(rec_f[PAD:PAD + P.height, PAD:PAD + P.width], want_rec)


def test_tpl_full_option_set_limits(be):
    """32x32 blocks / subsampled transforms do not exist at tpl levels 0-3: the host forms decline them (-1) instead of guessing"""
    c = CASES[0]
    P, planes, tot, mvs, cand, n_pus, cells = make_case(c, 8100)
    P.quant_fp[0], P.quant_fp[1], P.round_fp[0], P.round_fp[1], P.dequant[0], P.dequant[1] = 532, 431, 61, 76, 123, 152
    HP = T.TplHostPlanes()
    rows, psize = planes.shape[1], planes.shape[1] * planes.shape[2]
    HP.src_buf, HP.src_rows = planes[0].ctypes.data, rows
    for r in range(8):
        if P.refs[r].valid:
            k = P.refs[r].plane_off // psize
            HP.ref_buf[r], HP.ref_rows[r] = planes[k].ctypes.data, rows
    out = np.zeros(cells, T.SrcStats)
    for field, value in (("dispenser_search_level", 1), ("subsample_tx", 2), ("intra_mode_end", 13)):
        Q = T.TplParams.from_buffer_copy(P)
        setattr(Q, field, value)
        assert be.lib.svt_hip_tpl_src_stage_host(C.addressof(Q), C.addressof(HP), p(tot), p(mvs), p(cand), p(out)) == -1, field
