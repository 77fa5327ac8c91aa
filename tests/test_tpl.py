"""TPL dispenser, source-based half (Codec/src_ops_process.c:519-969): the oracle restatement against the reference's own static
function (compiled where it lies through oracle/ref_wrap/ref_tpl.c), and the device stage svt_hip_tpl_src_stage against the oracle -- bit-exact TplSrcStats
for every 16x16 / 32x32 block of a picture, incl. partial superblocks, picture borders, clamped far vectors, excluded references, I slices and pictures with
intra prediction disabled."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import REF_LIB, load_pkg, p, rng

PAD = 96  # luma border of every test plane (the reference pads 68 + and TPL vectors are clamped to the picture + 32)


class TplRef(C.Structure):
    _fields_ = [("plane_off", C.c_uint64), ("picture_number", C.c_uint64), ("stride", C.c_uint32), ("org_x", C.c_uint32), ("org_y", C.c_uint32),
                ("max_width", C.c_uint16), ("max_height", C.c_uint16), ("valid", C.c_uint8), ("pad", C.c_uint8 * 3)]


class TplParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("aligned_width", C.c_uint32), ("sbs_x", C.c_uint32), ("n_sb", C.c_uint32), ("src_stride", C.c_uint32),
                ("src_off", C.c_uint64), ("dispenser_search_level", C.c_uint8), ("subsample_tx", C.c_uint8), ("pf_shape", C.c_uint8), ("disable_intra_pred", C.c_uint8),
                ("i_slice", C.c_uint8), ("enable_me_16x16", C.c_uint8), ("enable_me_8x8", C.c_uint8), ("max_cand", C.c_uint8), ("max_refs", C.c_uint8),
                ("max_l0", C.c_uint8), ("intra_mode_end", C.c_uint8), ("search_flags", C.c_uint8), ("quant_fp", C.c_int16 * 2), ("round_fp", C.c_int16 * 2), ("dequant", C.c_int16 * 2),
                ("refs", TplRef * 8)]


SrcStats = np.dtype([("srcrf_dist", "<i8"), ("srcrf_rate", "<i8"), ("ref_frame_poc", "<u8"), ("mv_row", "<i2"), ("mv_col", "<i2"), ("best_rf_idx", "<i4"),
                     ("best_mode", "u1"), ("best_intra_mode", "u1"), ("written", "u1"), ("pad", "u1", (5,))])
assert SrcStats.itemsize == 40 and C.sizeof(TplParams) == 376 and C.sizeof(TplRef) == 40

CASES = [  # (W, H, level, subsample_tx, pf_shape, disable_intra, i_slice, me16, me8, n_l0, n_l1, q_index)
    dict(W=200, H=136, level=0, ss=0, pf=2, noi=0, isl=0, me16=1, me8=0, l0=2, l1=1, q=120),
    dict(W=200, H=136, level=0, ss=0, pf=1, noi=1, isl=0, me16=1, me8=1, l0=1, l1=1, q=60),
    dict(W=264, H=152, level=1, ss=2, pf=2, noi=0, isl=0, me16=1, me8=0, l0=2, l1=2, q=180),
    dict(W=136, H=72, level=0, ss=0, pf=0, noi=0, isl=1, me16=1, me8=0, l0=1, l1=0, q=30),
    dict(W=200, H=136, level=0, ss=0, pf=2, noi=0, isl=0, me16=0, me8=0, l0=3, l1=2, q=255),
    dict(W=328, H=200, level=1, ss=2, pf=1, noi=1, isl=0, me16=1, me8=0, l0=1, l1=1, q=0),
    dict(W=256, H=152, level=1, ss=2, pf=2, noi=0, isl=0, me16=1, me8=0, l0=1, l1=1, q=90),  # no SB column cut by the right edge: 32x32 rows above 16x16 rows
    # picture sizes with (size % 16) in 1..7: the last block column / row is less than half inside and the reference skips it (src_ops_process.c:580) -- its cells stay
    # unwritten and its pixels of the reconstruction plane keep the caller's content
    dict(W=195, H=131, level=0, ss=0, pf=2, noi=0, isl=0, me16=1, me8=0, l0=2, l1=1, q=120),
    dict(W=261, H=149, level=1, ss=2, pf=2, noi=0, isl=0, me16=1, me8=0, l0=1, l1=1, q=90),
]


def make_case(c, seed):
    g = rng(seed)
    W, H = c["W"], c["H"]
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    sbs_x, sbs_y = (aw + 63) // 64, (ah + 63) // 64
    stride, rows = aw + 2 * PAD + 24, ah + 2 * PAD + 16
    yy, xx = np.mgrid[0:rows, 0:stride]
    base = ((xx * 3 + yy * 2) & 255).astype(np.int32) + (((xx // 16 + yy // 16) % 7) << 3)
    n_ref = c["l0"] + c["l1"]
    planes = np.zeros((1 + n_ref, rows, stride), np.uint8)
    planes[0] = np.clip(base + g.integers(-12, 13, base.shape), 0, 255)
    for r in range(n_ref):  # shifted + noisy copies: some candidates beat the DC prediction, some do not
        planes[1 + r] = np.clip(np.roll(planes[0].astype(np.int32), (r + 1, -2 * r - 1), (0, 1)) + g.integers(-6 - 8 * r, 7 + 8 * r, base.shape), 0, 255)
    planes[0, PAD + 32:PAD + 64, PAD + 16:PAD + 80] = 77  # flat area: DC prediction exact, intra wins
    P = TplParams()
    P.width, P.height, P.aligned_width, P.sbs_x, P.n_sb, P.src_stride = W, H, aw, sbs_x, sbs_x * sbs_y, stride
    P.src_off = PAD * stride + PAD
    P.dispenser_search_level, P.subsample_tx, P.pf_shape, P.disable_intra_pred, P.i_slice = c["level"], c["ss"], c["pf"], c["noi"], c["isl"]
    P.enable_me_16x16, P.enable_me_8x8 = c["me16"], c["me8"]
    n_pus = 85 if c["me8"] else (21 if c["me16"] else 5)
    P.max_refs, P.max_l0 = n_ref, c["l0"]
    P.max_cand = max_cand = min(3 + n_ref, 9)
    for l in range(2):
        for r in range(4):
            R = P.refs[l * 4 + r]
            have = r < (c["l0"] if l == 0 else c["l1"])
            slot = r if l == 0 else c["l0"] + r
            R.plane_off = (1 + slot) * rows * stride if have else 0
            R.picture_number = 100 + 10 * l + r
            R.stride, R.org_x, R.org_y, R.max_width, R.max_height = stride, PAD, PAD, W, H
            R.valid = 1 if have else 0
    if n_ref >= 3:
        P.refs[1].valid = 0  # an excluded reference (:779-781): its candidates are skipped
    n_sb = P.n_sb
    tot = g.integers(0, max_cand + 1, (n_sb, n_pus)).astype(np.uint8)
    cand = np.zeros((n_sb, n_pus, max_cand), np.uint8)
    d = g.integers(0, 3, cand.shape)  # 0 / 1: uni-directional from list 0 / 1, 2: bi-directional (ignored by TPL)
    if c["l1"] == 0:
        d[:] = np.where(d == 1, 0, d)
    r0 = g.integers(0, max(c["l0"], 1), cand.shape)
    r1 = g.integers(0, max(c["l1"], 1), cand.shape)
    cand[:] = d | (r0 << 2) | (r1 << 4) | (g.integers(0, 2, cand.shape) << 6)
    mvx = g.integers(-24, 25, (n_sb, n_pus, n_ref)).astype(np.int16)
    mvy = g.integers(-16, 17, (n_sb, n_pus, n_ref)).astype(np.int16)
    far = g.random(mvx.shape) < 0.08  # vectors far outside the picture: clamped to the picture + 32 (:791-801)
    mvx[far] = g.integers(-400, 401, int(far.sum())).astype(np.int16)
    mvy[far] = g.integers(-300, 301, int(far.sum())).astype(np.int16)
    mvs = (mvy.astype(np.uint16).astype(np.uint32) << 16) | mvx.astype(np.uint16).astype(np.uint32)
    return P, planes, tot, np.ascontiguousarray(mvs), cand, n_pus, (aw + 15) // 16 * ((ah + 15) // 16)


def ref_lib():
    path = os.path.join(os.path.dirname(REF_LIB), "libsvtref_me.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libsvtref_me.so not available")
    lib = C.CDLL(path)
    if not hasattr(lib, "ref_tpl_dispenser_picture"):
        pytest.skip("libsvtref_me.so predates ref_tpl.c")
    return lib


def run_oracle(oracle, P, planes, tot, mvs, cand, cells):
    out = np.zeros(cells, SrcStats)
    oracle.oracle_tpl_src_picture(C.byref(P), p(planes), p(planes), p(tot), p(mvs), p(cand), p(out))
    return out


def same_stats(a, b, tag):
    for f in SrcStats.names:
        if f != "pad":
            bad = np.nonzero(a[f] != b[f])[0]
            assert bad.size == 0, (tag, f, bad[:6], a[f][bad[:6]], b[f][bad[:6]])


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_tpl_src_oracle_vs_reference(oracle, ref, ci):
    """oracle_tpl_src_picture == the TplSrcStats the reference's tpl_mc_flow_dispenser_sb_generic stores, every block of the picture."""
    me = ref_lib()
    P, planes, tot, mvs, cand, n_pus, cells = make_case(CASES[ci], 4000 + ci)
    want = np.zeros(cells, SrcStats)
    me.ref_tpl_dispenser_picture(C.byref(P), CASES[ci]["q"], p(planes), PAD, PAD, p(planes), p(tot), p(mvs), p(cand), n_pus, p(want), None, None)
    got = run_oracle(oracle, P, planes, tot, mvs, cand, cells)
    same_stats(got, want, ci)
    assert want["written"].sum() > 0
    if not CASES[ci]["isl"]:
        assert (want["best_mode"] == 16).any()
    if not CASES[ci]["noi"]:
        assert (want["best_mode"][want["written"] == 1] == 0).any()


class TplHostPlanes(C.Structure):
    _fields_ = [("src_buf", C.c_void_p), ("src_rows", C.c_uint32), ("ref_rows", C.c_uint32 * 8), ("ref_buf", C.c_void_p * 8)]


GPU_CASES = [dict(W=1920, H=1080, level=0, ss=0, pf=2, noi=0, isl=0, me16=1, me8=0, l0=2, l1=2, q=140),
             dict(W=1920, H=1080, level=1, ss=2, pf=2, noi=1, isl=0, me16=1, me8=0, l0=1, l1=1, q=140),
             dict(W=3840, H=2160, level=0, ss=0, pf=2, noi=0, isl=0, me16=1, me8=1, l0=2, l1=1, q=100)]


@pytest.mark.parametrize("ci", range(len(CASES) + len(GPU_CASES)))
def test_tpl_src_stage_device(be, oracle, ci):
    """svt_hip_tpl_src_stage (device arrays) and svt_hip_tpl_src_stage_host (host pictures) == the oracle, every cell of the statistics grid; on the GPU also
    whole 1080p / 4K pictures."""
    if ci >= len(CASES) and not be.is_gpu:
        pytest.skip("full-size pictures run on the GPU only")
    c = CASES[ci] if ci < len(CASES) else GPU_CASES[ci - len(CASES)]
    P, planes, tot, mvs, cand, n_pus, cells = make_case(c, 5000 + ci)
    path = os.path.join(os.path.dirname(REF_LIB), "libsvtref_me.so")
    if os.path.exists(path) and hasattr(C.CDLL(path), "ref_tpl_dispenser_picture") and ci < len(CASES):  # the quantizer row of the reference's own tables
        scratch = np.zeros(cells, SrcStats)
        C.CDLL(path).ref_tpl_dispenser_picture(C.byref(P), c["q"], p(planes), PAD, PAD, p(planes), p(tot), p(mvs), p(cand), n_pus, p(scratch), None, None)
    else:  # no reference at hand, or a full-size picture: a legal quantizer row (q index 120 of the 8-bit tables)
        P.quant_fp[0], P.quant_fp[1], P.round_fp[0], P.round_fp[1], P.dequant[0], P.dequant[1] = 532, 431, 61, 76, 123, 152
    want = run_oracle(oracle, P, planes, tot, mvs, cand, cells)
    d_pl, d_tot, d_mv, d_cand = be.dev(planes), be.dev(tot), be.dev(mvs), be.dev(cand)
    d_out = be.empty(cells * SrcStats.itemsize, np.uint8)
    be.lib.svt_hip_tpl_src_stage(C.addressof(P), be.ptr(d_pl), be.ptr(d_pl), be.ptr(d_tot), be.ptr(d_mv), be.ptr(d_cand), be.ptr(d_out), be.stream)
    got = be.host(d_out).view(SrcStats)
    same_stats(got, want, ("device", ci))
    # host form: the reference pictures as separate host buffers (two references sharing one buffer are uploaded once)
    HP = TplHostPlanes()
    rows = planes.shape[1]
    psize = planes.shape[1] * planes.shape[2]
    PH = TplParams.from_buffer_copy(P)
    HP.src_buf, HP.src_rows = planes[0].ctypes.data, rows
    for r in range(8):
        if P.refs[r].valid:
            k = P.refs[r].plane_off // psize
            HP.ref_buf[r], HP.ref_rows[r] = planes[k].ctypes.data, rows
            PH.refs[r].plane_off = 0
    got_h = np.zeros(cells, SrcStats)
    assert be.lib.svt_hip_tpl_src_stage_host(C.addressof(PH), C.addressof(HP), p(tot), p(mvs), p(cand), p(got_h)) == 0
    same_stats(got_h, want, ("host form", ci))
    assert want["written"].sum() > 0


# ---- the reconstruction half (src_ops_process.c:979-1198) ----------------------------------------------------------------------------------------------------------
ReconStats = np.dtype([("srcrf_dist", "<i8"), ("recrf_dist", "<i8"), ("srcrf_rate", "<i8"), ("recrf_rate", "<i8"), ("written", "u1"), ("coded", "u1"), ("pad", "u1", (2,)), ("reserved", "<u4")])
assert ReconStats.itemsize == 40


def recon_geometry(P, planes, fill=0):
    """a reconstruction plane of the test pictures' geometry (border PAD), zero-filled like the reference wrapper's (fill = 0) or holding earlier content (the device
    tests: pixels no block writes must survive the stage)"""
    rows, stride = planes.shape[1], planes.shape[2]
    if not fill:
        return np.zeros((rows, stride), np.uint8), PAD * stride + PAD, stride
    yy, xx = np.mgrid[0:rows, 0:stride]
    return ((xx * 7 + yy * 13 + fill) & 255).astype(np.uint8), PAD * stride + PAD, stride


def run_recon_oracle(oracle, P, planes, src_stats, is_ref, fill=0):
    rec, off, stride = recon_geometry(P, planes, fill)
    cells = src_stats.size
    out = np.zeros(cells, ReconStats)
    refs = (TplRef * 8)(*[P.refs[i] for i in range(8)])
    oracle.oracle_tpl_recon_picture(C.byref(P), refs, is_ref, p(planes), p(planes), p(src_stats), C.c_void_p(rec.ctypes.data + off), stride, p(out))
    return rec, out


def expand_like_result_model_store(P, st, cols16):
    """result_model_store (:266-341) for synth_blk_size 16: every statistic at least 1; a 32x32 block writes its four cells with the values / 4 (at least 1)"""
    grid = np.zeros((st.size, 4), np.int64)
    seen = np.zeros(st.size, bool)
    aligned_h = (P.height + 7) & ~7
    for i in np.nonzero(st["written"])[0]:
        cy, cx = divmod(int(i), cols16)
        complete = (P.aligned_width - (cx * 16 & ~63) >= 64) and (aligned_h - (cy * 16 & ~63) >= 64)
        v = np.maximum(1, np.array([st["srcrf_dist"][i], st["recrf_dist"][i], st["srcrf_rate"][i], st["recrf_rate"][i]], np.int64))
        if complete and P.dispenser_search_level:
            v = np.maximum(1, v // 4)
            for dy in (0, 1):
                for dx in (0, 1):
                    grid[i + dy * cols16 + dx] = v
                    seen[i + dy * cols16 + dx] = True
        else:
            grid[i] = v
            seen[i] = True
    return grid, seen


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_tpl_recon_oracle_vs_reference(oracle, ref, ci):
    """oracle_tpl_recon_picture == the reference's whole dispenser: the TplStats grid result_model_store leaves behind and the reconstructed TPL picture."""
    me = ref_lib()
    c = CASES[ci]
    P, planes, tot, mvs, cand, n_pus, cells = make_case(c, 4000 + ci)
    src = np.zeros(cells, SrcStats)
    want_stats = np.zeros((cells, 8), np.int64)
    want_rec = np.zeros((P.height, P.width), np.uint8)
    me.ref_tpl_dispenser_picture(C.byref(P), c["q"], p(planes), PAD, PAD, p(planes), p(tot), p(mvs), p(cand), n_pus, p(src), p(want_stats), p(want_rec))
    rec, got = run_recon_oracle(oracle, P, planes, src, 1)
    cols16 = (P.aligned_width + 15) // 16
    grid, seen = expand_like_result_model_store(P, got, cols16)
    assert seen.sum() > 0
    bad = np.nonzero((grid != want_stats[:, :4]).any(1) & seen)[0]
    assert bad.size == 0, (ci, bad[:5], grid[bad[:5]], want_stats[bad[:5], :4])
    got_rec = rec[PAD:PAD + P.height, PAD:PAD + P.width]
    assert np.array_equal(got_rec, want_rec), (ci, int((got_rec != want_rec).sum()), np.argwhere(got_rec != want_rec)[:4])
    w = got["written"] == 1
    assert got["coded"][w].any()                                       # the inverse transform ran for some blocks ...
    if c["q"] == 255: assert (got["coded"][w] == 0).any()              # noqa: E701  ... and, at the coarsest quantizer, not for others
    assert (src["best_mode"][w] == 0).any()                            # intra blocks (DC from the reconstruction) exist in every case


@pytest.mark.parametrize("ci", range(len(CASES) + len(GPU_CASES)))
@pytest.mark.parametrize("is_ref,form", [(1, 5), (0, 5), (1, 4), (1, 0), (0, 0), (1, 1), (1, 2), (1, 3), (1, 6), (1, 7), (0, 7)])
def test_tpl_recon_stage_device(be, oracle, ci, is_ref, form, monkeypatch):
    """svt_hip_tpl_recon_stage (device arrays, one launch per anti-diagonal) and svt_hip_tpl_recon_stage_host == the oracle: statistics of every block and the whole
    reconstruction plane; is_ref = 0 with intra prediction off leaves the prediction in place (:1135)."""
    if ci >= len(CASES) and not be.is_gpu:
        pytest.skip("full-size pictures run on the GPU only")
    c = CASES[ci] if ci < len(CASES) else GPU_CASES[ci - len(CASES)]
    if not is_ref and not c["noi"]:
        pytest.skip("is_ref only matters with intra prediction disabled")
    # form 5 (the default) = one launch, one wave per block, DC blocks wait for their neighbours' cells, release / acquire fences (4: sequentially-consistent fences);
    # form 1 = the row wavefront in one launch (SVT_HIP_TPL_RECON_FORM=1; csrc/tpl.hip: tpl_recon_rows_kernel), which falls back to the diagonal launches when an SB
    # column is cut by the right picture edge at dispenser level 1
    if form == 2 and not be.is_gpu:
        pytest.skip("form 2 orders the rows for the device's XCDs; the emulator runs workgroups one after the other in id order (a row would wait for one not yet run)")
    monkeypatch.setenv("SVT_HIP_TPL_RECON_FORM", str(form))
    pkg = be.pkg
    P, planes, tot, mvs, cand, n_pus, cells = make_case(c, 5000 + ci)
    P.quant_fp[0], P.quant_fp[1], P.round_fp[0], P.round_fp[1], P.dequant[0], P.dequant[1] = 532, 431, 61, 76, 123, 152  # q index 120 of the 8-bit tables
    src = run_oracle(oracle, P, planes, tot, mvs, cand, cells)
    want_rec, want = run_recon_oracle(oracle, P, planes, src, is_ref, fill=0x5A)
    R = pkg.TplReconParams()
    C.memmove(C.addressof(R.src), C.addressof(P), C.sizeof(P))
    for i in range(8):
        C.memmove(C.addressof(R.rec_refs[i]), C.addressof(P.refs[i]), C.sizeof(TplRef))
    rec0, off, stride = recon_geometry(P, planes, fill=0x5A)
    R.recon_off, R.recon_stride, R.is_ref = off, stride, is_ref
    d_pl, d_src, d_rec = be.dev(planes), be.dev(src.view(np.uint8)), be.dev(rec0)
    d_out = be.dev(np.zeros(cells * ReconStats.itemsize, np.uint8))
    be.lib.svt_hip_tpl_recon_stage(C.addressof(R), be.ptr(d_pl), be.ptr(d_pl), be.ptr(d_src), be.ptr(d_rec), be.ptr(d_out), be.stream)
    be.sync()
    got = be.host(d_out).view(ReconStats)
    for f in ("srcrf_dist", "recrf_dist", "srcrf_rate", "recrf_rate", "written", "coded"):
        bad = np.nonzero(got[f] != want[f])[0]
        assert bad.size == 0, ("device", ci, f, bad[:6], got[f][bad[:6]], want[f][bad[:6]])
    got_rec = be.host(d_rec).reshape(rec0.shape)
    assert np.array_equal(got_rec, want_rec), ("device recon", ci, int((got_rec != want_rec).sum()), np.argwhere(got_rec != want_rec)[:4])
    assert want["written"].sum() > 0 and want["coded"].any()
    # host form: pictures as separate host buffers, the reconstruction buffer updated in place (picture rows)
    HP = TplHostPlanes()
    rows, psize = planes.shape[1], planes.shape[1] * planes.shape[2]
    RH = pkg.TplReconParams.from_buffer_copy(R)
    HP.src_buf, HP.src_rows = planes[0].ctypes.data, rows
    for r in range(8):
        if P.refs[r].valid:
            k = P.refs[r].plane_off // psize
            HP.ref_buf[r], HP.ref_rows[r] = planes[k].ctypes.data, rows
            RH.rec_refs[r].plane_off = 0
    rec_h = rec0.copy()
    out_h = np.zeros(cells, ReconStats)
    assert be.lib.svt_hip_tpl_recon_stage_host(C.addressof(RH), C.addressof(HP), p(src), p(rec_h), rows, p(out_h)) == 0
    for f in ("srcrf_dist", "recrf_dist", "written", "coded"):
        assert np.array_equal(out_h[f], want[f]), ("host form", ci, f)
    # the WHOLE plane: the written rectangle equals the oracle's, every other pixel (borders, skipped blocks) still holds what the caller had there
    assert np.array_equal(rec_h, want_rec), ("host form recon", ci, int((rec_h != want_rec).sum()), np.argwhere(rec_h != want_rec)[:4])
    # both halves in one host call (svt_hip_tpl_stage_host): the source pictures as the source-based half sees them, the reconstruction references as separate buffers
    # (here: copies of the same test planes), statistics of both halves and the reconstruction as the two separate calls give them
    SP, RP = TplHostPlanes(), TplHostPlanes()
    rec_copies = {}
    RF = pkg.TplReconParams.from_buffer_copy(R)
    SP.src_buf, SP.src_rows = planes[0].ctypes.data, rows
    for r in range(8):
        if P.refs[r].valid:
            k = P.refs[r].plane_off // psize
            rec_copies[k] = rec_copies.get(k, planes[k].copy())
            SP.ref_buf[r], SP.ref_rows[r] = planes[k].ctypes.data, rows
            RP.ref_buf[r], RP.ref_rows[r] = rec_copies[k].ctypes.data, rows
            RF.rec_refs[r].plane_off = 0
            RF.src.refs[r].plane_off = 0
    rec_f, out_f, src_f = rec0.copy(), np.zeros(cells, ReconStats), np.zeros(cells, SrcStats)
    assert be.lib.svt_hip_tpl_stage_host(C.addressof(RF), C.addressof(SP), C.addressof(RP), p(tot), p(mvs), p(cand), p(src_f), p(rec_f), rows, p(out_f)) == 0
    same_stats(src_f, src, ("fused host form, source-based statistics", ci))
    for f in ("srcrf_dist", "recrf_dist", "srcrf_rate", "recrf_rate", "written", "coded"):
        assert np.array_equal(out_f[f], want[f]), ("fused host form", ci, f)
    assert np.array_equal(rec_f, want_rec), ("fused host form recon", ci, int((rec_f != want_rec).sum()))


class TplPlaneIds(C.Structure):
    _fields_ = [("src", C.c_uint64), ("src_ref", C.c_uint64 * 8), ("rec_ref", C.c_uint64 * 8), ("recon", C.c_uint64), ("recon_width", C.c_uint32),
                ("recon_height", C.c_uint32), ("recon_org_x", C.c_uint32), ("recon_org_y", C.c_uint32)]


_id_ranges = [0]


def _fresh_id_range():
    import os
    import time
    _id_ranges[0] += 1
    return (int(time.time() * 1000) % (1 << 30) + os.getpid()) * 100000 + _id_ranges[0] * 10000


@pytest.mark.parametrize("ci", [0, 2, 6])
def test_tpl_stage_host_resident(be, oracle, ci):
    """svt_hip_tpl_stage_host_resident: planes kept on the device across calls.  Picture 1 is processed with ids (every plane uploaded once, its reconstruction stays on
    the device with replicated borders); picture 2 takes picture 1's reconstruction as the TPL reconstruction of its first reference and far vectors reach into that
    plane's border -- served from the device copy (look-ups hit), it must give what the call without ids gives from the host's copy padded like svt_aom_generate_padding
    pads it; a third call with only the reconstruction half (stored statistics) goes through the same path; a dropped buffer is uploaded again."""
    pkg, lib = be.pkg, be.lib
    c = CASES[ci]
    P, planes, tot, mvs, cand, n_pus, cells = make_case(c, 6100 + ci)
    P.quant_fp[0], P.quant_fp[1], P.round_fp[0], P.round_fp[1], P.dequant[0], P.dequant[1] = 532, 431, 61, 76, 123, 152
    rows, stride = planes.shape[1], planes.shape[2]
    W, H = P.width, P.height

    def params(planes_of_refs):
        R = pkg.TplReconParams()
        C.memmove(C.addressof(R.src), C.addressof(P), C.sizeof(P))
        for i in range(8):
            C.memmove(C.addressof(R.rec_refs[i]), C.addressof(P.refs[i]), C.sizeof(TplRef))
            R.rec_refs[i].plane_off = 0
            R.src.refs[i].plane_off = 0
        R.recon_off, R.recon_stride, R.is_ref = PAD * stride + PAD, stride, 1
        SP, RP = TplHostPlanes(), TplHostPlanes()
        SP.src_buf, SP.src_rows = planes[0].ctypes.data, rows
        for r in range(8):
            if P.refs[r].valid:
                k = P.refs[r].plane_off // (rows * stride)
                SP.ref_buf[r], SP.ref_rows[r] = planes[k].ctypes.data, rows
                RP.ref_buf[r], RP.ref_rows[r] = planes_of_refs[k].ctypes.data, rows
        return R, SP, RP

    # (ids name CONTENT: they must never repeat for other content at the same host address -- numpy hands freed addresses out again, so every run of this test takes
    # its ids from a fresh range)
    base = _fresh_id_range()

    def ids_for(rec_id_of, recon_id):
        I = TplPlaneIds()
        I.src, I.recon = base + 100, base + recon_id
        for r in range(8):
            if P.refs[r].valid:
                k = P.refs[r].plane_off // (rows * stride)
                I.src_ref[r], I.rec_ref[r] = base + 200 + k, base + rec_id_of(k)
        I.recon_width, I.recon_height, I.recon_org_x, I.recon_org_y = W, H, PAD, PAD
        return I

    def call(R, SP, RP, I, rec, stored=None):
        out, src = np.zeros(cells, ReconStats), (np.zeros(cells, SrcStats) if stored is None else stored.copy())
        f = lib.svt_hip_tpl_stage_host_resident
        a = (p(tot), p(mvs), p(cand)) if stored is None else (None, None, None)
        assert f(C.addressof(R), C.addressof(SP), C.addressof(RP), C.addressof(I) if I is not None else None, a[0], a[1], a[2], p(src), p(rec), rows, p(out)) == 0
        return src, out

    def counts():
        h, m = C.c_uint64(0), C.c_uint64(0)
        lib.svt_hip_tpl_plane_counts(C.byref(h), C.byref(m))
        return h.value, m.value
    rec_copies = [planes[k].copy() for k in range(planes.shape[0])]  # the references' TPL reconstructions: separate host buffers
    # picture 1, with ids and without: the same results; nothing was resident yet
    R, SP, RP = params(rec_copies)
    rec1, rec1_plain = np.zeros((rows, stride), np.uint8), np.zeros((rows, stride), np.uint8)
    h0, m0 = counts()
    s1, o1 = call(R, SP, RP, ids_for(lambda k: 300 + k, 999), rec1)
    s1p, o1p = call(R, SP, RP, None, rec1_plain)
    same_stats(s1, s1p, "resident vs plain, picture 1")
    assert np.array_equal(o1["recrf_dist"], o1p["recrf_dist"]) and np.array_equal(rec1, rec1_plain)
    h1, m1 = counts()
    assert m1 > m0 and h1 == h0
    # what svt_aom_generate_padding makes of the host's reconstruction after the dispenser
    rec1_padded = rec1.copy()
    rec1_padded[0:H + 2 * PAD, 0:W + 2 * PAD] = np.pad(rec1[PAD:PAD + H, PAD:PAD + W], PAD, mode="edge")
    # picture 2: the first valid reference's TPL reconstruction IS picture 1's reconstruction
    first = next(r for r in range(8) if P.refs[r].valid) if not c["isl"] else None
    if first is not None:
        k0 = P.refs[first].plane_off // (rows * stride)
        recs2 = list(rec_copies)
        recs2[k0] = rec1_padded
        R2, SP2, RP2 = params(recs2)
        rec2, rec2_plain = np.zeros((rows, stride), np.uint8), np.zeros((rows, stride), np.uint8)
        # (the device copy lives under the HOST buffer's address: the padded copy is a different buffer, so name picture 1's own buffer for the resident call)
        RP2r = TplHostPlanes.from_buffer_copy(RP2)
        RP2r.ref_buf[first] = rec1.ctypes.data
        s2, o2 = call(R2, SP2, RP2r, ids_for(lambda k: 999 if k == k0 else 300 + k, 1000), rec2)
        s2p, o2p = call(R2, SP2, RP2, None, rec2_plain)
        h2, m2 = counts()
        assert h2 > h1, "picture 2 found nothing resident"
        same_stats(s2, s2p, "resident vs plain, picture 2")
        for f in ("srcrf_dist", "recrf_dist", "written", "coded"):
            assert np.array_equal(o2[f], o2p[f]), ("picture 2", f)
        assert np.array_equal(rec2, rec2_plain)
        # the reconstruction half alone with the caller's statistics (a later TPL group), still from resident planes
        rec3 = np.zeros((rows, stride), np.uint8)
        _, o3 = call(R2, SP2, RP2r, ids_for(lambda k: 999 if k == k0 else 300 + k, 1001), rec3, stored=s2)
        assert np.array_equal(o3["recrf_dist"], o2["recrf_dist"]) and np.array_equal(rec3, rec2)
        # a dropped buffer is uploaded again (a miss), with the same result
        lib.svt_hip_tpl_plane_drop(C.c_void_p(planes[0].ctypes.data))
        h3, m3 = counts()
        rec4 = np.zeros((rows, stride), np.uint8)
        call(R2, SP2, RP2r, ids_for(lambda k: 999 if k == k0 else 300 + k, 1002), rec4)
        h4, m4 = counts()
        assert m4 > m3 and np.array_equal(rec4, rec2)
