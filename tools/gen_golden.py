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


def gen_filters(ref, oracle):
    """CDEF (find_dir, filter_block), Wiener, self-guided, deblocking edge filters, 2x2 decimation, svt_search_one_dual, svt_compute_stats,
    quantize_b: inputs + the REAL reference's outputs for a handful of seeded cases per family."""
    from test_oracle_pin_cdef import BLOCK, BSTRIDE, in_ptr, make_tile
    from test_oracle_pin_restoration import aligned_i16, byteptr, conv_params, wiener_taps
    from test_deblock import lim_arrays, make_patch
    from test_cdef_pick import ptr_tables, tables
    from test_oracle_pin_quant import gen_coeff, run_ref
    from quant_common import make_qparams, make_scan
    g = np.random.default_rng(424242)
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    out = {}
    # ---- CDEF
    cd_cfg, cd_dir, cd_out = [], [], []
    tiles = []
    for bd in (8, 10, 12):
        cs = bd - 8
        t = make_tile(g, bd, 5)
        tiles.append(t.copy())
        for (by, bx) in ((0, 0), (3, 4), (7, 7)):
            off = by * 8 * BSTRIDE + bx * 8
            v = C.c_int32(0)
            d = ref.svt_aom_cdef_find_dir_c(in_ptr(t, off), BSTRIDE, C.byref(v), cs)
            cd_dir.append((bd, by, bx, d & 255, v.value))
            for (pri, sec, damp, dirn, sub) in ((4, 2, 5, d & 7, 1), (15, 4, 3, 3, 1), (0, 1, 6, 0, 2), (1, 0, 4, 7, 1)):
                o = np.zeros(64, np.uint16)
                ref.svt_cdef_filter_block_c(None, p(o), 8, in_ptr(t, off), pri << cs, sec << cs, dirn, damp + cs, damp + cs - 1, BLOCK[(8, 8)], cs, sub)
                cd_cfg.append((bd, by, bx, pri, sec, damp, dirn, sub))
                cd_out.append(o)
    out.update(cdef_tiles=np.stack(tiles), cdef_dir=np.array(cd_dir, np.int32), cdef_cfg=np.array(cd_cfg, np.int32), cdef_out=np.stack(cd_out))
    # ---- Wiener / self-guided (10-bit and 8-bit)
    S, w, h = 96, 48, 20
    for bd in (8, 10):
        hb = bd > 8
        dt = np.uint16 if hb else np.uint8
        src = g.integers(0, 1 << bd, (h + 12, S)).astype(dt)
        org = 6 * S + 8
        taps = np.stack([wiener_taps(g, 0)[:8], wiener_taps(g, 2)[:8]])
        fx, fy = aligned_i16(taps[0]), aligned_i16(taps[1])
        d1 = np.zeros((h, S), dt)
        cp = conv_params(bd)
        if hb:
            ref.svt_av1_highbd_wiener_convolve_add_src_c(byteptr(src, org), C.c_ssize_t(S), byteptr(d1), C.c_ssize_t(S), p(fx), p(fy), w, h, C.byref(cp), bd)
        else:
            ref.svt_av1_wiener_convolve_add_src_c(C.c_void_p(src.ctypes.data + org), C.c_ssize_t(S), p(d1), C.c_ssize_t(S), p(fx), p(fy), w, h, C.byref(cp))
        sg = []
        sp = C.c_void_p(src.ctypes.data + org * src.itemsize)
        for idx in (0, 7, 10, 14):
            b0, b1 = np.full((h, w), -7, np.int32), np.full((h, w), -7, np.int32)
            ref.svt_av1_selfguided_restoration_c(byteptr(src, org) if hb else sp, w, h, S, p(b0), p(b1), w, idx, bd, int(hb))
            sg.append(np.stack([b0, b1]))
        out["lr_src_%d" % bd], out["lr_taps_%d" % bd], out["lr_wiener_%d" % bd], out["lr_sgr_%d" % bd] = src, taps, d1[:, :w].copy(), np.stack(sg)
    # ---- deblocking
    lp_cfg, lp_in, lp_out = [], [], []
    for bd in (8, 10):
        for ln in (4, 6, 8, 14):
            for vert in (0, 1):
                for kind in (0, 1, 2):
                    a = make_patch(g, bd, kind).astype(np.uint16)
                    b = a.astype(np.uint8 if bd == 8 else np.uint16)
                    bl, li, th = (60 + 40 * kind, 10 + 8 * kind, kind)
                    keep = lim_arrays(bl, li, th)
                    f = getattr(ref, "svt_aom_%slpf_%s_%d_c" % ("" if bd == 8 else "highbd_", "vertical" if vert else "horizontal", ln))
                    args = [C.c_void_p(b.ctypes.data + (16 * 32 + 16) * b.itemsize), C.c_int32(32)] + [p(k) for k in keep]
                    if bd > 8:
                        args.append(C.c_int32(bd))
                    f(*args)
                    lp_cfg.append((bd, ln, vert, bl, li, th))
                    lp_in.append(a)
                    lp_out.append(b.astype(np.uint16))
    out.update(lpf_cfg=np.array(lp_cfg, np.int32), lpf_in=np.stack(lp_in), lpf_out=np.stack(lp_out))
    # ---- 2x2 decimation
    dsrc = g.integers(0, 256, (41, 70), dtype=np.uint8)
    for step in (2, 4):
        o = np.zeros((24, 40), np.uint8)
        ref.svt_aom_downsample_2d_c(p(dsrc), C.c_uint32(70), C.c_uint32(66), C.c_uint32(40), p(o), C.c_uint32(40), C.c_uint32(step))
        out["down_%d" % step] = o
    out["down_src"] = dsrc
    # ---- CDEF strength selection
    m0, m1 = tables(g, 57)
    arr, keep = ptr_tables(m0, m1)
    ref.svt_search_one_dual_c.restype = C.c_uint64
    la, lb = np.zeros(9, np.int32), np.zeros(9, np.int32)
    tot = [ref.svt_search_one_dual_c(p(la), p(lb), nb, arr, 57, 0, 64) for nb in range(4)]
    out.update(pick_m0=m0, pick_m1=m1, pick_lev0=la, pick_lev1=lb, pick_tot=np.array(tot, np.uint64))
    # ---- Wiener statistics
    dgd = g.integers(0, 1024, (60, 80)).astype(np.uint16)
    ssrc = np.clip(dgd.astype(np.int32) + g.integers(-9, 10, dgd.shape), 0, 1023).astype(np.uint16)
    M, H = np.zeros(49, np.int64), np.zeros(49 * 49, np.int64)
    ref.svt_av1_compute_stats_highbd_c(7, byteptr(dgd), byteptr(ssrc), 5, 69, 4, 52, 80, 80, p(M), p(H), 10)
    out.update(stats_dgd=dgd, stats_src=ssrc, stats_M=M, stats_H=H)
    # ---- quantize_b (low and high bit depth), 16x16
    P = make_qparams(88, 112, fp=False)
    scan, iscan = make_scan(256, g)
    qz = []
    coeffs = []
    for mode in (0, 1):
        coeff = gen_coeff(g, 256, 1 << 11, 1)
        q, dq, eob = run_ref(ref, mode, False, coeff, 256, P, scan, iscan, None, None, 0)
        coeffs.append(coeff)
        qz.append(np.concatenate([q, dq, [eob]]))
    out.update(quant_coeff=np.stack(coeffs), quant_scan=scan, quant_iscan=iscan, quant_out=np.stack(qz))
    np.savez_compressed(os.path.join(GOLDEN, "filters.npz"), **out)


def gen_stage(ref, oracle):
    """ME result formatting and temporal-filter kernels: inputs + outputs of the reference's own functions (static ones through
    oracle/_ref/libsvtref_me.so, see oracle/ref_wrap/)."""
    import test_me_results as M
    import test_tf as T
    from conftest import load_pkg
    pkg = load_pkg()
    refme = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsvtref_me.so"))
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
    out = {}
    # ---- MeSbResults
    for k, ci in enumerate((3, 6, 8, 10)):
        cfg, g = M.CONFIGS[ci], np.random.default_rng(7000 + ci)
        P = M.make_params(pkg, cfg, 12, g)
        sad, mv, do_ref, sb_size = M.make_tables(g, cfg, 12)
        tot, mvs, cands, stats, dr = M.run_cpu(refme.ref_me_results_sb, pkg, P, cfg, sad, mv, do_ref, sb_size, 0xaa)
        out.update({"me%d_cfg" % k: np.array(cfg, np.int32), "me%d_params" % k: np.frombuffer(bytes(P), np.uint8).copy(), "me%d_sad" % k: sad, "me%d_mv" % k: mv,
                    "me%d_do_ref" % k: do_ref, "me%d_sb_size" % k: sb_size, "me%d_total" % k: tot, "me%d_mvs" % k: mvs, "me%d_cands" % k: cands,
                    "me%d_stats" % k: stats.view(np.uint8).reshape(len(stats), -1), "me%d_do_ref_out" % k: dr})
    # ---- temporal filter, one 64x64 block through the reference's chain (central -> plane-wise per reference -> normalisation)
    k = 0
    for bd in (8, 10):
        for zz in (0, 1):
            g = np.random.default_rng(7100 + bd + zz)
            ss = (1, 1)
            P = T.make_params(pkg, g, bd, zz, 1, ss)
            cstride = [96, 48]
            central = [T.make_pair(g, bd, (64, 96))[0], T.make_pair(g, bd, (32, 48))[0], T.make_pair(g, bd, (32, 48))[0]]
            preds = []
            for r in range(3):
                pl = []
                for c in range(3):
                    w, h = (64, 64) if c == 0 else (32, 32)
                    noise = int(g.choice([2, 8, 30])) << (bd - 8)
                    pl.append(np.clip(central[c][:h, :w].astype(np.int32) + g.integers(-noise, noise + 1, (h, w)), 0, (1 << bd) - 1).astype(central[c].dtype))
                preds.append(pl)
            blocks = T.make_blocks(pkg, g, 12, bd).reshape(3, 2, 2)
            want = T.ref_frame_chain(refme, pkg, P, central, cstride, preds, blocks, 3, int(bd > 8))
            out.update({"tf%d_params" % k: np.frombuffer(bytes(P), np.uint8).copy(), "tf%d_blocks" % k: blocks.view(np.uint8).reshape(3, 2, 2, -1)})
            for c in range(3):
                out["tf%d_central%d" % (k, c)] = central[c]
                out["tf%d_out%d" % (k, c)] = want[c]
                for r in range(3):
                    out["tf%d_pred%d_%d" % (k, r, c)] = preds[r][c]
            k += 1
    # ---- noise estimate
    g = np.random.default_rng(7200)
    a8 = np.clip(100 + np.add.outer(np.arange(40), np.arange(72)) + g.integers(-5, 6, (40, 72)), 0, 255).astype(np.uint8)
    a10 = (a8.astype(np.uint16) << 2) + g.integers(0, 4, a8.shape).astype(np.uint16)
    out.update(noise_a8=a8, noise_a10=a10, noise_out=np.array([ref.svt_estimate_noise_fp16_c(p(a8), 64, 40, 72),
                                                              ref.svt_estimate_noise_highbd_fp16_c(p(a10), 64, 40, 72, 10)], np.int32))
    # ---- HME leaf drivers and integer_search_b64 (static in the reference)
    import test_hme as Hm
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    W, H = 200, 136
    for k, case in enumerate(((0, 0, 2, 2, 16, 8), (1, 1, 2, 2, 16, 8), (2, 0, 2, 2, 8, 3))):
        level, sub, nw, nh, sa_w, sa_h = case
        g = np.random.default_rng(7300 + k)
        src, refs, w, h, org, stride = Hm.make_planes(g, W, H, level, 2)
        n = 2 * 4 * 3 * nw * nh
        prev = np.stack([g.integers(-3 * w // 4, 3 * w // 4, n), g.integers(-3 * h // 4, 3 * h // 4, n)], 1).astype(np.int16)
        prev[::3] //= 8
        sad, sc = Hm.cpu_level(refme.ref_hme_level, level, sub, nw, nh, src, refs, w, h, org, stride, W, H, sa_w, sa_h, prev)
        out.update({"hme%d_case" % k: np.array(case, np.int32), "hme%d_src" % k: src, "hme%d_ref0" % k: refs[0], "hme%d_ref1" % k: refs[1],
                    "hme%d_prev" % k: prev, "hme%d_sad" % k: sad, "hme%d_sc" % k: sc})
    for k, ci in enumerate((1, 3)):
        c, g = Hm.INT_CASES[ci], np.random.default_rng(7400 + k)
        P = Hm.int_params(c)
        src, refs, w, h, org, stride = Hm.make_planes(g, W, H, 2, 1)
        aw, ah = (W + 7) & ~7, (H + 7) & ~7
        sad, sc = Hm.make_hme_results(g, 12, 4, W, H)
        bs, bm, fsc, fsad = np.zeros((12, 85), np.uint32), np.zeros((12, 85), np.uint32), np.zeros((12, 2), np.int16), np.zeros(12, np.uint64)
        for sb in range(12):
            o_sad = C.c_uint64(0)
            refme.ref_me_integer_search(C.byref(P), 2, 2, p(sad[sb]), p(sc[sb]), p(src), stride, org, org, p(refs[0]), stride, org, org, W, H, (sb % 4) * 64,
                                        (sb // 4) * 64, aw, ah, C.c_void_p(fsc[sb].ctypes.data), C.byref(o_sad), C.c_void_p(bs[sb].ctypes.data),
                                        C.c_void_p(bm[sb].ctypes.data))
            fsad[sb] = o_sad.value
        out.update({"int%d_case" % k: np.array([ci], np.int32), "int%d_src" % k: src, "int%d_ref" % k: refs[0], "int%d_hme_sad" % k: sad, "int%d_hme_sc" % k: sc,
                    "int%d_bs" % k: bs, "int%d_bm" % k: bm, "int%d_sc" % k: fsc, "int%d_sad" % k: fsad})
    np.savez_compressed(os.path.join(GOLDEN, "stage.npz"), **out)


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


def gen_quant_tables(ref, oracle):
    """SURVEY 8(d) config 3 data: the reference's own luma quantizer tables (svt_av1_build_quantizer, md_config_process.c:111-189, base_q_idx 0,
    sharpness 0, as QuantAsmTest.cc:86-96 builds them) at q in {0, 60, 120, 180, 255} for 8 and 10 bit, and av1_scan_orders (coefficients.h:2197)
    for every TX size and type, through oracle/ref_wrap/ref_quant_tables.c."""
    me = C.CDLL(os.path.join(os.path.dirname(REF_LIB), "libsvtref_me.so"))
    qs = np.array([0, 60, 120, 180, 255], np.int32)
    tabs = np.zeros((2, len(qs), 7, 2), np.int16)  # [bd 8|10][q][zbin, round, quant, quant_shift, dequant, quant_fp, round_fp][dc, ac]
    for b, bd in enumerate((8, 10)):
        for i, q in enumerate(qs):
            me.ref_build_quantizer_y(bd, 0, 0, int(q), p(tabs[b, i]))
    out = {"q": qs, "tables": tabs}
    for ts in range(19):
        scans, iscans = [], []
        for tt in range(16):
            sc, isc = np.zeros(1024, np.int16), np.zeros(1024, np.int16)
            n = me.ref_scan_order(ts, tt, p(sc), p(isc))
            scans.append(sc[:n].copy())
            iscans.append(isc[:n].copy())
        out["scan_%d" % ts], out["iscan_%d" % ts] = np.stack(scans), np.stack(iscans)
    np.savez_compressed(os.path.join(GOLDEN, "quant_tables.npz"), **out)


FAMILIES = {"sad": gen_sad, "txfm": gen_txfm, "filters": gen_filters, "stage": gen_stage, "quant_tables": gen_quant_tables}

if __name__ == "__main__":
    os.system("make -s -C %s oracle ref" % os.path.join(ROOT, "oracle"))
    ref, oracle = C.CDLL(REF_LIB), C.CDLL(ORACLE_LIB)
    for name in (sys.argv[1:] or FAMILIES):
        FAMILIES[name](ref, oracle)
        print("golden:", name)
