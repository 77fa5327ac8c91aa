"""Extra measurement legs of bench.py (`--extra`): BASELINE configs 3 and 4 beyond the slices the headline metric names, and the
HME levels of config 2.  Same rules as bench.py: inputs resident in HBM before the timed region, HIP events on the launch stream,
algorithmic bytes / kernel time against the 8 TB/s HBM peak.  Nothing here touches oracle/ (parity lives in tests/)."""
import ctypes as C

import numpy as np

import bench_regions as regions

HBM_PEAK_GBS = 8000.0


def roof(alg_bytes, seconds, kernel=None, bound="hbm", **extra):
    """roofline object of a leg: ALGORITHMIC bytes per call / event-timed seconds per call against the 8 TB/s HBM peak; `region` = the timed region it was
    measured in, through which bench.py attaches this run's counter passes (HBM bytes moved, VALU instructions, VALU-active cycles)"""
    r = {"bound": bound, "achieved": alg_bytes / seconds / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / seconds / 1e9 / HBM_PEAK_GBS,
         "algorithmic_bytes_per_launch": alg_bytes, "kernel_us": seconds * 1e6, "region": regions.LAST}
    if kernel:
        r["kernel"] = kernel
    r.update(extra)
    return r


MIN_S = 0.3  # bench.py sets this from --min-leg-s: every leg is timed over at least this much device time


def _time(torch, fn, steps, warmup, batches=5):
    """seconds per call: median over `batches` event-timed runs of back-to-back launches; `steps` per run is raised until the runs together last >= MIN_S
    (short kernels see clock ramps and the ~5 us dispatch gap otherwise)"""
    import math
    idx = regions.open_region()
    if regions.PMC_CHILD:
        return regions.pmc_run(fn, idx)
    for _ in range(max(warmup, 1)):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    t1 = max(e0.elapsed_time(e1) / 1e3, 1e-7)
    steps = max(steps, int(math.ceil(MIN_S / batches / t1))) if MIN_S > 0 else steps
    ts = []
    for _ in range(batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 1e3 / steps)
    return sorted(ts)[len(ts) // 2]


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()


def _qparams(pkg, dq_dc, dq_ac, log_scale):
    """invert_quant() / zbin / round as svt_av1_build_quantizer derives them (md_config_process.c:111-189)."""
    P = np.zeros(1, dtype=pkg.QuantParams)
    for k, d in enumerate((dq_dc, dq_ac)):
        t = int(d).bit_length() - 1
        m = 1 + (1 << (16 + t)) // d
        P["quant"][0][k] = np.int16(np.uint16((m - (1 << 16)) & 0xffff))
        P["quant_shift"][0][k] = min(1 << (16 - t), 32767)
        P["zbin"][0][k] = (84 * d + 64) >> 7
        P["round"][0][k] = (48 * d) >> 7
        P["dequant"][0][k] = d
    P["log_scale"] = log_scale
    return P


def txfm_roundtrip(torch, lib, pkg, stream, steps, warmup):
    """config 3: FwdTxfm2d (+ svt_handle_transform for 64-point sizes) -> quantize_b -> InvTxfm2d_add, all 19 TX sizes, 8 and 10 bit.
    N per launch as SURVEY 8(d): 65536 (<= 16-point), 16384 (32-point), 4096 (64-point); tx_type cycles through the allowed set."""
    out = {}
    g = np.random.default_rng(13596)
    for ts, (w, h) in enumerate(pkg.TX_SIZES):
        big = max(w, h)
        n = 65536 if big <= 16 else (16384 if big == 32 else 4096)
        iw, ih = min(w, 32), min(h, 32)
        ncoef = iw * ih
        pels = w * h
        ls = int(pels > 256) + int(pels > 1024)
        types = pkg.allowed_tx_types(ts)
        for bd in (8, 10):
            amp = (1 << bd) - 1
            res = g.integers(-amp, amp + 1, n * pels, dtype=np.int16)
            fd = np.zeros(n, dtype=pkg.FwdTxfmDesc)
            fd["in_off"] = np.arange(n, dtype=np.uint64) * pels
            fd["in_stride"] = w
            fd["tx_type"] = np.array(types, np.uint8)[np.arange(n) % len(types)]
            idesc = np.zeros(n, dtype=pkg.InvTxfmDesc)
            idesc["coeff_off"] = np.arange(n, dtype=np.uint64) * ncoef
            idesc["pred_off"] = idesc["recon_off"] = np.arange(n, dtype=np.uint64) * pels
            idesc["pred_stride"] = idesc["recon_stride"] = w
            idesc["tx_type"] = fd["tx_type"]
            qd = np.zeros(n, dtype=pkg.QuantDesc)
            d_res, d_fd, d_id, d_qd = _dev(torch, res), _dev(torch, fd), _dev(torch, idesc), _dev(torch, qd)
            d_qp = _dev(torch, _qparams(pkg, 88, 112, ls))
            d_iscan = _dev(torch, np.arange(ncoef, dtype=np.int16))
            d_pred = _dev(torch, g.integers(0, 1 << bd, n * pels, dtype=np.uint16))
            d_coef = torch.zeros(n * pels, dtype=torch.int32, device="cuda")
            d_q = torch.zeros(n * ncoef, dtype=torch.int32, device="cuda")
            d_dq = torch.zeros(n * ncoef, dtype=torch.int32, device="cuda")
            d_eob = torch.zeros(n, dtype=torch.int16, device="cuda")
            d_en = torch.zeros(n, dtype=torch.int64, device="cuda")
            d_rec = torch.zeros(n * pels, dtype=torch.int16, device="cuda")
            mode = 1 if bd > 8 else 0

            def fwd():
                lib.svt_hip_fwd_txfm2d_batch(d_res.data_ptr(), d_fd.data_ptr(), n, ts, bd, 0, d_coef.data_ptr(), stream)
                if big == 64:
                    lib.svt_hip_handle_transform_batch(d_coef.data_ptr(), n, ts, 0, d_en.data_ptr(), stream)

            def quant():
                lib.svt_hip_quantize_batch(mode, d_coef.data_ptr(), n, ncoef, d_qp.data_ptr(), d_iscan.data_ptr(), None, None, d_qd.data_ptr(),
                                           d_q.data_ptr(), d_dq.data_ptr(), d_eob.data_ptr(), stream)

            def inv():
                lib.svt_hip_inv_txfm2d_add_batch(d_dq.data_ptr(), d_pred.data_ptr(), d_rec.data_ptr(), d_id.data_ptr(), n, ts, bd, stream)

            def chain():
                fwd()
                quant()
                inv()
            rd = np.zeros(n, dtype=pkg.RoundtripDesc)
            rd["in_off"] = rd["pred_off"] = rd["recon_off"] = np.arange(n, dtype=np.uint64) * pels
            rd["in_stride"] = rd["pred_stride"] = rd["recon_stride"] = w
            rd["tx_type"] = fd["tx_type"]
            d_rd = _dev(torch, rd)

            def fused():  # the same round trip in ONE launch, coefficients kept in LDS (qcoeff out, no dqcoeff)
                lib.svt_hip_txfm_quant_roundtrip_batch(d_res.data_ptr(), d_pred.data_ptr(), d_rec.data_ptr(), d_rd.data_ptr(), n, ts, bd, mode, d_qp.data_ptr(),
                                                       d_iscan.data_ptr(), None, None, d_q.data_ptr(), None, d_eob.data_ptr(), stream)
            # note: for 64-point sizes handle_transform repacks in place, so quant reads the packed layout; chain order keeps that valid
            tf, tc = _time(torch, fwd, steps, warmup), None
            if big == 64:  # re-run fwd once so that d_coef holds a packed block set for the isolated quant timing
                fwd()
            tq = _time(torch, quant, steps, warmup)
            ti = _time(torch, inv, steps, warmup)
            tc = _time(torch, chain, steps, warmup)
            tfu = _time(torch, fused, steps, warmup)
            b_fused = 2 * pels + 4 * ncoef + 2 + 4 * pels  # residual in, qcoeff + eob out, prediction in, reconstruction out (u16 pixels: SURVEY 8d's 10 B/px)
            # algorithmic bytes per block: fwd 2*pels in + 4*pels out; quant 4*ncoef in + 8*ncoef out + 2; inv 4*ncoef in + 2*pels pred + 2*pels recon
            b_f, b_q, b_i = 6 * pels, 12 * ncoef + 2, 4 * ncoef + 4 * pels
            out["%dx%d_bd%d" % (w, h, bd)] = {
                "blocks": n, "chain_Mblocks_s": n / tc / 1e6, "fwd_Mblocks_s": n / tf / 1e6, "quant_Mblocks_s": n / tq / 1e6, "inv_Mblocks_s": n / ti / 1e6,
                "chain_GBs": n * (b_f + b_q + b_i) / tc / 1e9, "fwd_GBs": n * b_f / tf / 1e9, "quant_GBs": n * b_q / tq / 1e9, "inv_GBs": n * b_i / ti / 1e9,
                "chain_hbm_frac": n * (b_f + b_q + b_i) / tc / 1e9 / HBM_PEAK_GBS,
                "fused_Mblocks_s": n / tfu / 1e6, "fused_GBs": n * b_fused / tfu / 1e9, "fused_hbm_frac": n * b_fused / tfu / 1e9 / HBM_PEAK_GBS,
                "fused_speedup_vs_chain": tc / tfu}
            del d_res, d_coef, d_q, d_dq, d_rec, d_pred
    return out


def lr_frames(torch, lib, pkg, stream, steps, warmup):
    """config 4, restoration half: one 3840x2160 10-bit luma plane, 256x256 units, every unit Wiener / every unit self-guided / mixed."""
    import os
    Wc, Hc, bd, us = 3840, int(os.environ.get("SVT_LR_H", "2160")), 10, 256  # (SVT_LR_H: plane-height sweep for the fixed-cost / throughput split)
    g = np.random.default_rng(44)
    yy, xx = np.mgrid[0:Hc, 0:Wc]
    plane = np.clip(((xx * 3 + yy * 2) % 1024) // 2 + ((xx // 8 + yy // 8) % 2) * 24 + g.integers(0, 64, (Hc, Wc)), 0, 1023).astype(np.uint16)
    nstripes = (Hc + 8 + 63) // 64
    above = g.integers(0, 1024, (2 * nstripes, Wc)).astype(np.uint16)
    below = g.integers(0, 1024, (2 * nstripes, Wc)).astype(np.uint16)
    nvu, nhu = max((Hc + us // 2) // us, 1), max((Wc + us // 2) // us, 1)
    d_pl, d_ab, d_bl = _dev(torch, plane), _dev(torch, above), _dev(torch, below)
    d_out = torch.zeros(Hc * Wc, dtype=torch.int16, device="cuda")
    out = {}
    import os
    only = os.environ.get("SVT_LR_ONLY")  # profiling passes: one unit type per process, so that per-kernel counters are not a mix
    for name, tsel in (("wiener", [1]), ("sgrproj", [2]), ("mixed", [1, 2, 0, 2, 1]), ("copy", [0])):  # "copy": every unit RESTORE_NONE = staging + stores only
        if (only and name != only) or (name == "copy" and only != "copy"):
            continue
        units = np.zeros(nvu * nhu, dtype=pkg.LrUnit)
        for i in range(len(units)):
            f = [int(g.integers(-5, 11)), int(g.integers(-23, 9)), int(g.integers(-17, 47))]
            taps = [f[0], f[1], f[2], -2 * sum(f), f[2], f[1], f[0], 0]
            units[i] = (tsel[i % len(tsel)], taps, taps, int(g.integers(0, 16)), (int(g.integers(-96, 32)), int(g.integers(-32, 96))))
        d_un = _dev(torch, units)
        sh = int(os.environ.get("SVT_LR_SHIFT", "0"))  # alignment experiment: the plane starts `sh` samples into the buffer (width reduced accordingly)
        P = pkg.LrParams(d_pl.data_ptr() + 2 * sh, d_ab.data_ptr() + 2 * sh, d_bl.data_ptr() + 2 * sh, d_out.data_ptr(), Wc, Wc, Wc, Wc - 2 * sh, Hc, us, 0, 0, 1, bd,
                         d_un.data_ptr())
        t = _time(torch, lambda: lib.svt_hip_lr_filter_frame(C.byref(P), stream), steps, warmup)
        nbytes = Wc * Hc * 4 + 4 * nstripes * Wc * 2
        out["lr_%s_4k10" % name] = {"frames_per_s": 1 / t, "Mpx_s": Wc * Hc / t / 1e6, "ms": t * 1e3,
                                    "roofline": roof(nbytes, t, "lr_frame_kernel<%s>" % (os.environ.get("SVT_HIP_LR_UR") or "32"), algorithmic_bytes_per_frame=nbytes)}
    if only:
        return out
    # ---- the whole 4:2:0 PICTURE: Y with 256-sample units + U, V at half size with 128-sample units (chroma units cover the same picture area,
    # restoration.c:1146-1160; 64-row stripes become 32 chroma rows offset by 4), mixed unit types, one launch per plane
    cw, ch, cus = Wc // 2, Hc // 2, us // 2
    cstripes = (ch + 4 + 31) // 32
    cplane = [np.ascontiguousarray(plane[::2, ::2]), np.ascontiguousarray(plane[1::2, 1::2])]
    cnvu, cnhu = max((ch + cus // 2) // cus, 1), max((cw + cus // 2) // cus, 1)
    Ps = [pkg.LrParams(d_pl.data_ptr(), d_ab.data_ptr(), d_bl.data_ptr(), d_out.data_ptr(), Wc, Wc, Wc, Wc, Hc, us, 0, 0, 1, bd, d_un.data_ptr())]
    keepalive = []
    for k in range(2):
        cun = np.zeros(cnvu * cnhu, dtype=pkg.LrUnit)
        for i in range(len(cun)):
            f = [int(g.integers(-5, 11)), int(g.integers(-23, 9)), int(g.integers(-17, 47))]
            taps = [0, f[1], f[2], -2 * (f[1] + f[2]), f[2], f[1], 0, 0]  # chroma: 5-tap Wiener (WIENER_WIN_CHROMA)
            cun[i] = ([1, 2, 0, 2, 1][i % 5], taps, taps, int(g.integers(0, 16)), (int(g.integers(-96, 32)), int(g.integers(-32, 96))))
        dcp, dcu = _dev(torch, cplane[k]), _dev(torch, cun)
        dca, dcb = _dev(torch, g.integers(0, 1024, (2 * cstripes, cw)).astype(np.uint16)), _dev(torch, g.integers(0, 1024, (2 * cstripes, cw)).astype(np.uint16))
        dco = torch.zeros(ch * cw, dtype=torch.int16, device="cuda")
        keepalive += [dcp, dcu, dca, dcb, dco]
        Ps.append(pkg.LrParams(dcp.data_ptr(), dca.data_ptr(), dcb.data_ptr(), dco.data_ptr(), cw, cw, cw, cw, ch, cus, 1, 1, 1, bd, dcu.data_ptr()))

    def frame():
        for q in Ps:
            lib.svt_hip_lr_filter_frame(C.byref(q), stream)
    t = _time(torch, frame, steps, warmup)
    nbytes = (Wc * Hc * 4 + 4 * nstripes * Wc * 2) + 2 * (cw * ch * 4 + 4 * cstripes * cw * 2)
    out["lr_mixed_4k10_420"] = {"frames_per_s": 1 / t, "frame_us": t * 1e6, "planes": 3, "Mpx_s": Wc * Hc * 1.5 / t / 1e6,
                                "roofline": roof(nbytes, t, "lr_frame_kernel<%s> x 3 planes" % (os.environ.get("SVT_HIP_LR_UR") or "32"), algorithmic_bytes_per_frame=nbytes,
                                                 note="one 4:2:0 picture = three launches (Y 256-sample units, U / V 128-sample units, sub-sampled stripes)")}
    return out


def hme_sad_loop(torch, lib, pkg, stream, steps, warmup, nframes=32):
    """config 2, hierarchical levels: svt_sad_loop_kernel over the 1/16 plane (16x16 block per SB, 32x16 area) and the 1/4 plane
    (32x32 block, 16x16 area) of 1080p, all 510 SBs x nframes frames per launch."""
    out = {}
    g = np.random.default_rng(7)
    for name, scale, area in (("hme_l0_sixteenth", 4, (32, 16)), ("hme_l1_quarter", 2, (16, 16))):
        pw, ph, pad = 1920 // scale, 1080 // scale, 64 // scale * 2
        stride, rows = pw + 2 * pad, ph + 2 * pad
        planes = g.integers(0, 256, (nframes + 1, rows, stride), dtype=np.uint8)
        bs = 64 // scale
        aw, ah = area
        descs = []
        for f in range(nframes):
            for sy in range((ph + bs - 1) // bs):
                for sx in range((pw + bs - 1) // bs):
                    x0, y0 = pad + sx * bs, pad + sy * bs
                    rx = min(max(x0 - aw // 2, 0), stride - bs - aw)
                    ry = min(max(y0 - ah // 2, 0), rows - bs - ah)
                    bh = min(bs, ph - sy * bs)
                    descs.append((f * rows * stride + y0 * stride + x0, (f + 1) * rows * stride + ry * stride + rx, stride, stride, stride, bs, bh,
                                  aw, ah, 0, (0, 0, 0)))
        d = np.array(descs, dtype=pkg.SadLoopDesc)
        n = len(d)
        d_pl, d_d = _dev(torch, planes), _dev(torch, d)
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
        d_keys = torch.zeros(n, dtype=torch.int64, device="cuda")
        t = _time(torch, lambda: lib.svt_hip_sad_loop_batch(d_pl.data_ptr(), d_pl.data_ptr(), d_d.data_ptr(), n, aw, ah, bs, bs, 1, d_res.data_ptr(), d_keys.data_ptr(), stream),
                  steps, warmup)
        out[name] = {"value": n * aw * ah / t / 1e6, "unit": "M(block x position)/s", "searches": n, "block": "%dx%d" % (bs, bs), "area": "%dx%d" % area,
                     "ms": t * 1e3, "sad_ops_per_s": float(np.sum(d["block_width"].astype(np.int64) * d["block_height"])) * aw * ah / t}
    return out


def picprep(torch, lib, pkg, stream, steps, warmup, nframes=32):
    """per 1080p frame, border replication of the full-resolution luma plane (pad 68) + 1/4 (pad 32) + 1/16 (pad 16)
    planes as svt_aom_downsample_filtering_input_picture; three launches per frame, nframes frames per step."""
    w, h, pad = 1920, 1080, 68
    stride, rows = w + 2 * pad, h + 2 * pad
    g = np.random.default_rng(5)
    full = torch.from_numpy(g.integers(0, 256, (nframes, rows, stride), dtype=np.uint8)).cuda()
    qw, qh, sw, sh = w // 2, h // 2, w // 4, h // 4
    qs, ss = qw + 64, sw + 32
    quarter = torch.zeros((nframes, qh + 64, qs), dtype=torch.uint8, device="cuda")
    sixteenth = torch.zeros((nframes, sh + 32, ss), dtype=torch.uint8, device="cuda")

    def fn():
        for f in range(nframes):
            fb = full.data_ptr() + f * rows * stride
            qb = quarter.data_ptr() + f * (qh + 64) * qs
            sb = sixteenth.data_ptr() + f * (sh + 32) * ss
            lib.svt_hip_generate_padding(fb, stride, w, h, pad, pad, stream)
            lib.svt_hip_downsample_2d_padded(fb + pad * stride + pad, stride, w, h, qb, qs, 32, 32, 2, stream)
            lib.svt_hip_downsample_2d_padded(qb + 32 * qs + 32, qs, qw, qh, sb, ss, 16, 16, 2, stream)
    t = _time(torch, fn, steps, warmup)
    # the same 3 x nframes launches captured once into a HIP graph (launch-bound sequence: one graph launch per step instead of 96 kernel launches)
    st = lib.svt_hip_stream_create()
    stream_eager, stream = stream, st
    lib.svt_hip_graph_capture_begin(st)
    fn()
    gexec = lib.svt_hip_graph_capture_end(st)
    stream = stream_eager
    lib.svt_hip_graph_launch(gexec, st)
    lib.svt_hip_stream_synchronize(st)
    import time as _t
    t0 = _t.perf_counter()
    for _ in range(steps):
        lib.svt_hip_graph_launch(gexec, st)
    lib.svt_hip_stream_synchronize(st)
    tg = (_t.perf_counter() - t0) / steps
    lib.svt_hip_graph_destroy(gexec)
    lib.svt_hip_stream_destroy(st)
    nbytes = nframes * (w * h + (rows * stride - w * h) + (qh + 64) * qs + qw * qh + (sh + 32) * ss)  # reads + writes per frame
    return {"picprep_1080p": {"frames_per_s": nframes / t, "us_per_frame": t / nframes * 1e6, "hip_graph_us_per_frame": tg / nframes * 1e6,
                              "roofline": {"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / t / 1e9 / HBM_PEAK_GBS,
                                           "algorithmic_bytes_per_frame": nbytes // nframes, "note": "three launches per frame; launch-latency bound at 1080p"}}}


def deblock(torch, lib, pkg, stream, steps, warmup):
    """deblocking of a 3840x2160 10-bit luma plane, edges on a 16-sample grid with 14-tap filters in even 4-sample segments and on an
    8-sample grid with 8-tap filters in odd ones, vertical pass then horizontal pass = two launches per frame."""
    w, h, bd = 3840, 2160, 10
    g = np.random.default_rng(6)
    yy, xx = np.mgrid[0:h, 0:w]
    plane = np.clip(((xx * 3 + yy * 2) % 1024) // 2 + ((xx // 8 + yy // 8) % 2) * 6 + g.integers(0, 3, (h, w)), 0, 1023).astype(np.uint16)
    passes = []
    for vert in (1, 0):
        # even 4-sample segments: 16-sample grid with 14-tap filters; odd segments: 8-sample grid with 8-tap filters (a legal, overlap-free layout)
        along_n, across_n = (h, w) if vert else (w, h)
        parts = []
        for par, grid, length in ((0, 16, 14), (1, 8, 8)):
            # raster order (x fastest) in both passes: consecutive threads then touch consecutive addresses
            if vert:
                cs, as_ = np.meshgrid(np.arange(grid, across_n, grid), np.arange(4 * par, along_n, 8))
            else:
                as_, cs = np.meshgrid(np.arange(4 * par, along_n, 8), np.arange(grid, across_n, grid))
            parts.append((cs.ravel(), as_.ravel(), np.full(cs.size, length)))
        cs, as_, ln = (np.concatenate([q[i] for q in parts]) for i in range(3))
        xs, ys = (cs, as_) if vert else (as_, cs)
        e = np.zeros(xs.size, dtype=pkg.LpfEdge)
        e["x"], e["y"], e["vertical"], e["length"] = xs.ravel(), ys.ravel(), vert, ln.ravel()
        e["blimit"], e["limit"], e["thresh"] = 60, 12, 2
        passes.append((_dev(torch, e), len(e)))
    d_plane = _dev(torch, plane)

    def fn():
        for d_e, n in passes:
            lib.svt_hip_lpf_edges_batch(d_plane.data_ptr(), w, 1, bd, d_e.data_ptr(), n, stream)
    t = _time(torch, fn, steps, warmup)
    nedges = sum(n for _, n in passes)
    nbytes = 2 * (w * h * 2 * 2) + nedges * 16  # each pass reads and writes (nearly) every sample once + the edge list
    return {"deblock_4k10": {"frames_per_s": 1 / t, "ms": t * 1e3, "Medges_s": nedges / t / 1e6,
                             "roofline": {"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / t / 1e9 / HBM_PEAK_GBS,
                                          "algorithmic_bytes_per_frame": nbytes}}}


def lr_stats(torch, lib, pkg, stream, steps, warmup):
    """a24: Wiener search statistics (svt_av1_compute_stats_highbd) of every 256x256 restoration unit of a 3840x2160 10-bit luma plane, win 7:
    H = Z^T Z on the matrix cores.  Algorithmic integer MACs = pixels x 50 x 51 / 2 (upper triangle + M); the int8 digit split issues 3 x 64 x 64
    MACs per pixel to the MFMA pipe (padding and the two-digit products included)."""
    w, h, bd, us = 3840, 2160, 10, 256
    g = np.random.default_rng(8)
    pad = 4
    dgd = g.integers(0, 1 << bd, (h + 2 * pad, w + 2 * pad), dtype=np.uint16)
    src = np.clip(dgd.astype(np.int32) + g.integers(-9, 10, dgd.shape), 0, (1 << bd) - 1).astype(np.uint16)
    rects = []
    nvu, nhu = max((h + us // 2) // us, 1), max((w + us // 2) // us, 1)
    for r in range(nvu):
        for c in range(nhu):
            rects.append((pad + c * us, pad + (w if c == nhu - 1 else (c + 1) * us), pad + r * us, pad + (h if r == nvu - 1 else (r + 1) * us)))
    rr = np.array(rects, np.int32)
    d_dgd, d_src, d_r = _dev(torch, dgd), _dev(torch, src), _dev(torch, rr)
    n = len(rects)
    d_M = torch.zeros(n * 49, dtype=torch.int64, device="cuda")
    d_H = torch.zeros(n * 49 * 49, dtype=torch.int64, device="cuda")
    mw, mh = int((rr[:, 1] - rr[:, 0]).max()), int((rr[:, 3] - rr[:, 2]).max())
    stride = w + 2 * pad
    t = _time(torch, lambda: lib.svt_hip_lr_compute_stats_batch(d_dgd.data_ptr(), d_src.data_ptr(), d_r.data_ptr(), n, mw, mh, stride, stride, 7, bd, d_M.data_ptr(),
                                                                 d_H.data_ptr(), stream), steps, warmup)
    issued = 3 * 64 * 64 * w * h * 2  # int8 ops (multiply + add) handed to the matrix cores
    return {"lr_compute_stats_4k10_win7": {"frames_per_s": 1 / t, "ms": t * 1e3, "units": n, "algorithmic_GMAC_s": w * h * 50 * 51 / 2 / t / 1e9,
                                           "roofline": {"bound": "mfma", "achieved": issued / t / 1e12, "peak": 5000.0, "unit": "TOP/s (int8, dense)",
                                                        "frac": issued / t / 1e12 / 5000.0, "kernel": "stats_mfma_kernel", "kernel_us": t * 1e6, "region": regions.LAST,
                                                        "algorithmic_bytes_per_launch": w * h * 2 * 2 + n * (49 + 49 * 49) * 8,
                                                        "note": "issued int8 ops incl. 64-column padding and the three digit products; peak = 2x the 2.5 PFLOP/s bf16 dense figure"}}}


def cdef_chain(torch, lib, pkg, stream, steps, warmup):
    """config 4, the whole CDEF stage of a 3840x2160 10-bit 4:2:0 picture, device resident: strength search over all 64 strengths on Y, U and V
    (cdef_process.c:106) -> joint_strength_search_dual with 8 pairs (enc_cdef.c:697) -> per-filter-block assignment (enc_cdef.c:916) -> apply on
    the three planes (enc_cdef.c:284).  torch only glues tables between the calls (U + V distortion, strength index -> (pri, sec))."""
    Wc, Hc, bd = 3840, 2160, 10
    g = np.random.default_rng(14)

    def plane(w, h):
        yy, xx = np.mgrid[0:h, 0:w]
        rec = np.clip(((xx * 2 + yy * 3) % 1024) // 2 + (((xx // 8 + yy // 8) % 5) << 5) + g.integers(-16, 17, (h, w)), 0, 1023).astype(np.uint16)
        return rec, np.clip(rec.astype(np.int32) + g.integers(-6, 7, rec.shape), 0, 1023).astype(np.uint16)
    nhfb, nvfb = Wc // 64, (Hc + 63) // 64
    nfb = nhfb * nvfb
    planes = [plane(Wc, Hc), plane(Wc // 2, Hc // 2), plane(Wc // 2, Hc // 2)]
    d_rec = [_dev(torch, r) for r, _ in planes]
    d_src = [_dev(torch, s_) for _, s_ in planes]
    d_out = [_dev(torch, r) for r, _ in planes]
    d_skip = _dev(torch, np.zeros((nvfb * 8, nhfb * 8), np.uint8))
    cands = [(pr, sc) for pr in range(16) for sc in (0, 1, 2, 4)]
    d_pri, d_sec = _dev(torch, np.array([c[0] for c in cands], np.int32)), _dev(torch, np.array([c[1] for c in cands], np.int32))
    d_dir, d_var = torch.zeros(nfb * 64, dtype=torch.uint8, device="cuda"), torch.zeros(nfb * 64, dtype=torch.int32, device="cuda")
    d_mse = [torch.zeros(nfb * 64, dtype=torch.int64, device="cuda") for _ in range(3)]
    d_lev0, d_lev1 = torch.zeros(65, dtype=torch.int32, device="cuda"), torch.zeros(65, dtype=torch.int32, device="cuda")
    d_best, d_ws = torch.zeros(1, dtype=torch.int64, device="cuda"), torch.zeros(4096 + nfb, dtype=torch.int64, device="cuda")
    d_gi = torch.zeros(nfb, dtype=torch.int8, device="cuda")
    sec_map = torch.tensor([0, 1, 2, 4], dtype=torch.int32, device="cuda")
    a_pri = [torch.zeros(nfb, dtype=torch.int32, device="cuda") for _ in range(2)]
    a_sec = [torch.zeros(nfb, dtype=torch.int32, device="cuda") for _ in range(2)]

    def params(mode, pl, pri_t, sec_t):
        w, h, dec = (Wc, Hc, 0) if pl == 0 else (Wc // 2, Hc // 2, 1)
        return pkg.CdefParams(d_rec[pl].data_ptr(), d_src[pl].data_ptr(), d_out[pl].data_ptr(), w, w, w, w, h, dec, dec, pl, 1, bd - 8, 4, 4, 1, 64 if mode else 0,
                              d_skip.data_ptr(), pri_t.data_ptr(), sec_t.data_ptr(), d_dir.data_ptr(), d_var.data_ptr(), d_mse[pl].data_ptr())
    P_search = [params(1, pl, d_pri, d_sec) for pl in range(3)]
    P_apply = [params(0, pl, a_pri[min(pl, 1)], a_sec[min(pl, 1)]) for pl in range(3)]

    def fn():
        for pl in range(3):
            lib.svt_hip_cdef_frame(1, C.byref(P_search[pl]), stream)
        d_mse[1].add_(d_mse[2])
        lib.svt_hip_cdef_joint_strength_search(d_mse[0].data_ptr(), d_mse[1].data_ptr(), d_lev0.data_ptr(), d_lev1.data_ptr(), 8, nfb, 0, 64, d_best.data_ptr(),
                                               d_ws.data_ptr(), stream)
        lib.svt_hip_cdef_assign_fb_strengths(d_mse[0].data_ptr(), d_mse[1].data_ptr(), d_lev0.data_ptr(), d_lev1.data_ptr(), 8, nfb, d_gi.data_ptr(), stream)
        gi = d_gi.long()
        for k, lev in enumerate((d_lev0, d_lev1)):
            st = lev[gi]
            a_pri[k].copy_(st // 4)
            a_sec[k].copy_(sec_map[(st % 4).long()])
        for pl in range(3):  # (the luma directions / variances are the ones the search pass just wrote: mode 2 does not search them again)
            lib.svt_hip_cdef_frame(2 if pl == 0 else 0, C.byref(P_apply[pl]), stream)
    t = _time(torch, fn, steps, warmup)
    return {"cdef_stage_4k10_420": {"frames_per_s": 1 / t, "ms": t * 1e3, "filter_blocks": nfb, "strengths_searched": 64, "pairs_selected": 8,
                                    "note": "Y+U+V search, joint strength search (8 greedy + 32 refinement rounds), per-block assignment, Y+U+V apply"}}


def me_session(torch, lib, pkg, stream, steps, warmup, npics=32):
    """PCIe-inclusive ME stage: 1080p pictures in pinned HOST memory go through svt_hip_me_session (upload once, reference by id, 4 references per
    picture, 16x9 area, results downloaded to pinned host memory), two submissions in flight.  This is the rate a caller that owns host buffers
    sees; it is never bench.py's `value` (inputs resident in HBM)."""
    import time as _t
    W, H, PAD = 1920, 1080, 68
    stride, rows = W + 2 * PAD, H + 2 * PAD
    nbytes = stride * rows
    g = np.random.default_rng(3)
    sbs = ((W + 63) // 64) * ((H + 63) // 64)
    hp = [lib.svt_hip_host_alloc(nbytes) for _ in range(8)]
    for q in hp:
        a = g.integers(0, 256, nbytes, dtype=np.uint8)  # (kept alive across the copy)
        C.memmove(q, a.ctypes.data, nbytes)
    SLOTS = 3  # pictures in flight (each on its own stream)
    res = [(lib.svt_hip_host_alloc(4 * sbs * 85 * 4), lib.svt_hip_host_alloc(4 * sbs * 85 * 4)) for _ in range(SLOTS)]
    sess = lib.svt_hip_me_session_create(W, H, stride, PAD, PAD, rows, 8, 4, 16, 9, SLOTS)

    def run(n):
        pending = []
        for k in range(n):
            nref = min(k, 4)
            refs = np.array([k - 1 - r for r in range(nref)], np.int64)
            out = res[k % SLOTS]
            slot = lib.svt_hip_me_session_submit(sess, k, hp[k % 8], refs.ctypes.data if nref else None, nref, 16, 9, 0, out[0], out[1])
            assert slot >= 0, slot
            pending.append(slot)
            if len(pending) == SLOTS:
                lib.svt_hip_me_session_wait(sess, pending.pop(0))
        for slot in pending:
            lib.svt_hip_me_session_wait(sess, slot)
    lib.svt_hip_me_session_destroy(sess)
    sess = lib.svt_hip_me_session_create(W, H, stride, PAD, PAD, rows, 8, 4, 16, 9, SLOTS)
    run(8)
    lib.svt_hip_me_session_destroy(sess)
    ts = []
    for _ in range(max(steps // 4, 2)):
        sess = lib.svt_hip_me_session_create(W, H, stride, PAD, PAD, rows, 8, 4, 16, 9, SLOTS)
        t0 = _t.perf_counter()
        run(npics)
        ts.append(_t.perf_counter() - t0)
        lib.svt_hip_me_session_destroy(sess)
    t = min(ts)
    for q in hp:
        lib.svt_hip_host_free(q)
    for a, b in res:
        lib.svt_hip_host_free(a)
        lib.svt_hip_host_free(b)
    sbrefs = sbs * sum(min(k, 4) for k in range(npics))
    return {"me_session_1080p_host": {"pictures_per_s": npics / t, "us_per_picture": t / npics * 1e6, "value": sbrefs * 144 / t / 1e6,
                                      "unit": "M(SB x position)/s, PCIe inclusive", "h2d_MB_per_picture": nbytes / 1e6, "d2h_MB_per_picture": 4 * sbs * 85 * 8 / 1e6}}


def me_results(torch, lib, pkg, stream, steps, warmup, npics=32):
    """ME result formatting.  (1) the formatting kernel alone over one 1080p picture's search tables resident in HBM, 4 + 3
    references; (2) the ME session returning the final product (MeSbResults + per-SB statistics) to pinned host memory instead of the raw tables,
    2 + 2 references, 16x9 area -- the PCIe-inclusive rate of the whole stage."""
    import time as _t
    g = np.random.default_rng(5)
    sbs = 510
    P = pkg.MeResultsParams()
    P.n_sb, P.num_of_list_to_search = sbs, 2
    P.num_of_ref_pic_to_search[0], P.num_of_ref_pic_to_search[1] = 4, 3
    P.max_refs, P.max_cand = pkg.me_max_allocated_refs(4, 3)
    P.max_l0, P.enable_me_16x16, P.enable_me_8x8, P.prune_ref, P.gm_enabled = 4, 1, 1, 1, 1
    P.prune_ref_if_me_sad_dev_bigger_than_th, P.prune_me_candidates_th, P.picture_number = 30, 65, 16
    for l in range(2):
        for r in range(4):
            P.ref_picture_number[l][r] = 16 + (1 if l else -1) * (r + 1)
    dev = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()
    sad = dev(g.integers(100, 60000, (7, sbs, 85)).astype(np.uint32))
    mv = dev(g.integers(0, 1 << 32, (7, sbs, 85), dtype=np.uint64).astype(np.uint32))
    do_ref = dev(np.ones((sbs, 8), np.uint8))
    sz = dev(np.full((sbs, 2), 64, np.uint8))
    tot, mvs = torch.zeros(sbs * 85, dtype=torch.uint8, device="cuda"), torch.zeros(sbs * 85 * P.max_refs * 4, dtype=torch.uint8, device="cuda")
    cands, st = torch.zeros(sbs * 85 * P.max_cand, dtype=torch.uint8, device="cuda"), torch.zeros(sbs * 28, dtype=torch.uint8, device="cuda")
    t = _time(torch, lambda: lib.svt_hip_me_results_batch(C.addressof(P), sad.data_ptr(), mv.data_ptr(), do_ref.data_ptr(), sz.data_ptr(), tot.data_ptr(),
                                                          mvs.data_ptr(), cands.data_ptr(), st.data_ptr(), stream), steps, warmup)
    out = {"me_results_1080p_7refs": {"us": t * 1e6, "Msb_per_s": sbs / t / 1e6, "GBps_algorithmic": (7 * sbs * 85 * 8 + sbs * 85 * (1 + P.max_cand + 4 * P.max_refs)) / t / 1e9}}

    W, H, PAD = 1920, 1080, 68
    stride, rows = W + 2 * PAD, H + 2 * PAD
    nbytes = stride * rows
    hp = [lib.svt_hip_host_alloc(nbytes) for _ in range(8)]
    for q in hp:
        a = g.integers(0, 256, nbytes, dtype=np.uint8)  # (kept alive across the copy)
        C.memmove(q, a.ctypes.data, nbytes)
    Q = pkg.MeResultsParams()
    C.memmove(C.addressof(Q), C.addressof(P), C.sizeof(P))
    Q.num_of_ref_pic_to_search[0], Q.num_of_ref_pic_to_search[1] = 2, 2
    Q.max_refs, Q.max_cand = pkg.me_max_allocated_refs(2, 2)
    Q.max_l0 = 2
    sizes = [sbs * 85, sbs * 85 * Q.max_refs * 4, sbs * 85 * Q.max_cand, sbs * 28]
    hosts = []
    for _ in range(2):
        b = [lib.svt_hip_host_alloc(n) for n in sizes]
        hosts.append((b, pkg.MeResultsHost(None, b[0], b[1], b[2], b[3], None, None)))

    def run(sess, n):
        pending = []
        for k in range(n):
            slot = (lib.svt_hip_me_session_submit(sess, k, hp[k % 8], None, 0, 16, 9, 0, None, None) if k < 4 else
                    lib.svt_hip_me_session_submit_results(sess, k, hp[k % 8], np.array([k - 1, k - 2, k - 3, k - 4], np.int64).ctypes.data, 4, 16, 9, 0,
                                                          C.addressof(Q), C.addressof(hosts[k & 1][1])))
            assert slot >= 0, slot
            pending.append(slot)
            if len(pending) == 2:
                lib.svt_hip_me_session_wait(sess, pending.pop(0))
        for slot in pending:
            lib.svt_hip_me_session_wait(sess, slot)
    ts = []
    for it in range(max(steps // 4, 2) + 1):
        sess = lib.svt_hip_me_session_create(W, H, stride, PAD, PAD, rows, 8, 4, 16, 9, 2)
        t0 = _t.perf_counter()
        run(sess, npics if it else 8)
        if it:
            ts.append(_t.perf_counter() - t0)
        lib.svt_hip_me_session_destroy(sess)
    for q in hp:
        lib.svt_hip_host_free(q)
    for b, _ in hosts:
        for q in b:
            lib.svt_hip_host_free(q)
    t = min(ts)
    out["me_session_1080p_host_formatted"] = {"pictures_per_s": npics / t, "us_per_picture": t / npics * 1e6, "d2h_MB_per_picture": sum(sizes) / 1e6,
                                              "note": "4 references (2 + 2), MeSbResults + per-SB statistics returned instead of the raw tables"}
    return out


def tf_frames(torch, lib, pkg, stream, steps, warmup):
    """Temporal filter, whole picture in one launch (central + 6 motion-compensated references + normalisation): 1080p 8-bit and 4K 10-bit 4:2:0.
    HBM-bound: (n_refs + 2) samples per pixel (read central and every prediction once, write the filtered picture)."""
    out = {}
    g = np.random.default_rng(11)
    for (name, W, H, bd) in (("tf_1080p8_420_6refs", 1920, 1088, 8), ("tf_4k10_420_6refs", 3840, 2176, 10)):
        n_refs, nbx, nby = 6, W // 32, H // 32
        dt, tdt = (np.uint8, torch.uint8) if bd == 8 else (np.uint16, torch.int16)
        P = pkg.TfParams()
        for c in range(3):
            P.tf_decay_factor_fp16[c] = 1 << 19
        P.tf_mv_dist_th, P.tf_chroma, P.use_zz_based_filter, P.encoder_bit_depth, P.ss_x, P.ss_y = 16, 1, 0, bd, 1, 1
        ys, cs = W + 64, W // 2 + 32

        def planes():
            t = [torch.from_numpy(g.integers(0, 1 << bd, (H, ys)).astype(dt).view(np.int16 if bd > 8 else np.uint8)).cuda(),
                 torch.from_numpy(g.integers(0, 1 << bd, (H // 2, cs)).astype(dt).view(np.int16 if bd > 8 else np.uint8)).cuda(),
                 torch.from_numpy(g.integers(0, 1 << bd, (H // 2, cs)).astype(dt).view(np.int16 if bd > 8 else np.uint8)).cuda()]
            return t, pkg.TfPlanes(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), ys, cs)
        cen_t, cen = planes()
        out_t, outp = planes()
        refs = [planes() for _ in range(n_refs)]
        prs = (pkg.TfPlanes * n_refs)(*[r[1] for r in refs])
        B = np.zeros(n_refs * nbx * nby, pkg.TfBlock)
        B["split"] = g.integers(0, 2, len(B))
        B["block_error"] = g.integers(0, 4096, (len(B), 4)) << (0 if bd == 8 else 4)
        B["mv_x"], B["mv_y"] = g.integers(-8, 9, (len(B), 4)), g.integers(-8, 9, (len(B), 4))
        d_b = torch.from_numpy(B.view(np.uint8).reshape(-1)).cuda()
        t = _time(torch, lambda: lib.svt_hip_tf_filter_frame(C.addressof(P), C.addressof(cen), C.addressof(prs), n_refs, d_b.data_ptr(), nbx, nby,
                                                              C.addressof(outp), stream), steps, warmup)
        nbytes = W * H * 1.5 * (n_refs + 2) * (1 if bd == 8 else 2)
        out[name] = {"us": t * 1e6, "frames_per_s": 1 / t, "GBps_algorithmic": nbytes / t / 1e9, "hbm_frac": nbytes / t / 8e12}
    # noise estimate of a 1080p luma plane (one launch + finalize)
    a = torch.from_numpy(g.integers(90, 110, (1080, 2056), dtype=np.uint8)).cuda()
    res, ws = torch.zeros(4, dtype=torch.int32, device="cuda"), torch.zeros(lib.svt_hip_estimate_noise_workspace(1920, 1080), dtype=torch.uint8, device="cuda")
    t = _time(torch, lambda: lib.svt_hip_estimate_noise_batch(a.data_ptr(), 1920, 1080, 2056, 8, res.data_ptr(), ws.data_ptr(), stream), steps, warmup)
    out["noise_estimate_1080p8"] = {"us": t * 1e6, "GBps_algorithmic": 1920 * 1080 / t / 1e9}
    return out


def tf_subpel(torch, lib, pkg, stream, steps, warmup, keep=None):
    """SURVEY 8f rank 4, the temporal filter's sub-pel refinement: one 1080p 8-bit central picture against 6 reference pictures, the preset-8 search shape
    (64x64 and 32x32 blocks bilinear with tf_ctrls.use_2tap, 16x16 regular; half + quarter pel, no eighth, sub_sampling_shift 1) = 6 x (510 + 2040 + 8160)
    blocks in one launch.  Compute bound (8-tap separable interpolation): reported as blocks/s and candidate samples/s, not against HBM."""
    g = np.random.default_rng(17)
    W, H, PAD, n_refs = 1920, 1088, 80, 6
    stride, rows = W + 2 * PAD, H + 2 * PAD
    yy, xx = np.mgrid[0:rows, 0:stride].astype(np.float32)
    base = (0.5 + 0.25 * np.sin(xx / 3.3) * np.cos(yy / 4.1) + 0.2 * np.sin((xx + 2 * yy) / 9.1)) * 255
    refs = np.stack([np.clip(np.roll(base, (r - 3, 2 * r - 5), (0, 1)) + g.normal(0, 2, base.shape), 0, 255).astype(np.uint8) for r in range(n_refs)])
    src = np.ascontiguousarray(np.clip(base[PAD:PAD + H, PAD:PAD + W] + g.normal(0, 2, (H, W)), 0, 255).astype(np.uint8))
    blocks = [(x, y, b) for b in (64, 32, 16) for y in range(0, H, b) for x in range(0, W, b)]
    a = np.array(blocks * n_refs)
    n = len(a)
    d = np.zeros(n, pkg.TfSubpelDesc)
    d["pu_x"], d["pu_y"], d["bsize"], d["src_stride"], d["bilinear"] = a[:, 0], a[:, 1], a[:, 2], W, a[:, 2] >= 32
    d["src_off"] = a[:, 1].astype(np.uint64) * W + a[:, 0].astype(np.uint64)
    r = np.repeat(np.arange(n_refs), len(blocks))
    d["ref_off"] = r.astype(np.uint64) * (rows * stride)
    d["mv_x"], d["mv_y"] = 8 * ((2 * r - 5) + g.integers(-1, 2, n)), 8 * ((r - 3) + g.integers(-1, 2, n))
    P = pkg.TfSubpelParams()
    P.half_pel_mode, P.quarter_pel_mode, P.eight_pel_mode, P.subsampling_shift, P.bit_depth = 1, 1, 0, 1, 8
    P.mi_rows, P.mi_cols, P.ref_org_x, P.ref_org_y, P.ref_stride = H // 4, W // 4, PAD, PAD, stride
    d_src, d_ref, d_d = _dev(torch, src), _dev(torch, refs), torch.from_numpy(d.view(np.uint8).reshape(-1)).cuda()
    d_out = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
    t = _time(torch, lambda: lib.svt_hip_tf_subpel_search_batch(C.addressof(P), d_src.data_ptr(), d_ref.data_ptr(), d_d.data_ptr(), n, d_out.data_ptr(), stream),
              steps, warmup, batches=3)
    res = d_out.cpu().numpy().view(pkg.TfSubpelResult)
    cand_px = float(np.sum(a[:, 2].astype(np.float64) ** 2 / 2)) * 17  # 1 + 8 + 8 candidates, every other row
    if keep is not None:  # bench.py's checker / CPU baseline leg works on the same inputs
        keep.update(P=P, src=src, refs=refs, descs=d, results=res.copy(), W=W, H=H)
    # algorithmic bytes per block: the b x b source block + the (b + 8)^2 reference window an 8-tap sub-pel search around the full-pel vector touches + the 16-byte result
    alg = int(np.sum(a[:, 2].astype(np.int64) ** 2 + (a[:, 2].astype(np.int64) + 8) ** 2 + 16))
    return {"tf_subpel_1080p8_6refs": {"us": t * 1e6, "blocks_per_s": n / t, "pictures_per_s": 1 / t, "candidate_Gsamples_per_s": cand_px / t / 1e9,
                                        "roofline": roof(alg, t, "tf_subpel_kernel<unsigned char>"),
                                        "moved_frac": float(np.mean((res["mv_x"] != d["mv_x"]) | (res["mv_y"] != d["mv_y"])))}}


def lr_search(torch, lib, pkg, stream, steps, warmup, keep=None):
    """the per-unit half of the loop-restoration search (restoration_seg_search) of one 3840x2160 10-bit luma plane, 256x256 units (15 x 8):
    `full` = sg_filter level 1 (all 16 self-guided parameter sets, refinement) + 7-tap Wiener with refinement; `fast` = 2 parameter sets + 5-tap Wiener with one
    refinement step.  Latency-bound lock-step search (tens of dependent trials per unit): reported as ms per plane, with the trial kernel's share."""
    import time as _t
    g = np.random.default_rng(19)
    W, H, PAD, bd = 3840, 2160, 8, 10
    amp = (1 << bd) - 1
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    tex = 0.5 + 0.2 * np.sin(xx / 2.3) * np.cos(yy / 3.1) + 0.15 * np.sin((xx + 2 * yy) / 6.7) + 0.1 * np.sign(np.sin(xx / 9.0) * np.sin(yy / 7.0))
    src = np.clip(tex * amp + g.normal(0, amp / 120, tex.shape), 0, amp).astype(np.uint16)
    blur = (src.astype(np.float32) * 4 + np.roll(src, 1, 0) + np.roll(src, -1, 0) + np.roll(src, 1, 1) + np.roll(src, -1, 1)) / 8
    dgd = np.pad(np.clip(np.round(blur / 6) * 6 + g.normal(0, 4, blur.shape), 0, amp).astype(np.uint16), PAD, mode="edge")
    d_src, d_dgd = _dev(torch, src), _dev(torch, dgd)
    out = {}
    for name, wn, sg in (("lr_search_4k10_full", (1, 7, 1, 0), (1, 0, 16, 1, 1)), ("lr_search_4k10_fast", (1, 5, 1, 1), (1, 0, 16, 8, 1))):
        P = pkg.LrSearchParams()
        P.src, P.dgd = d_src.data_ptr(), d_dgd.data_ptr() + (PAD * dgd.shape[1] + PAD) * 2
        P.dgd_stride, P.src_stride, P.width, P.height, P.unit_size, P.ss_y, P.highbd, P.bit_depth = dgd.shape[1], W, W, H, 256, 0, 1, bd
        P.wn_enabled, P.wiener_win, P.wn_use_refinement, P.wn_max_one_refinement_step = wn
        P.sg_enabled, P.sg_start_ep, P.sg_end_ep, P.sg_ep_inc, P.sg_refine = sg
        n = ((H + 128) // 256) * ((W + 128) // 256)
        ws = torch.zeros(lib.svt_hip_lr_search_workspace(C.addressof(P)), dtype=torch.uint8, device="cuda")
        d_out = torch.zeros(n * 72, dtype=torch.uint8, device="cuda")
        ts = []

        def one():
            assert lib.svt_hip_lr_search_plane(C.addressof(P), None, d_out.data_ptr(), ws.data_ptr(), stream) == 0
            torch.cuda.synchronize()
        tp = regions.hook(one, name="lr_search_" + name.rsplit("_", 1)[1])
        for it in range(0 if tp is not None else warmup + max(steps, 2)):
            torch.cuda.synchronize()
            t0 = _t.perf_counter()
            one()
            if it >= warmup:
                ts.append(_t.perf_counter() - t0)
        if tp is not None:
            ts = [tp]
        res = d_out.cpu().numpy().view(pkg.LrSearchUnit)
        t = float(np.median(ts))
        # algorithmic bytes: the source and the degraded plane once (2 B / sample each) + the 72-byte result per unit -- the search re-reads them per trial by design
        out[name] = {"ms": t * 1e3, "planes_per_s": 1 / t, "units": n, "workspace_MB": ws.numel() / 1e6, "roofline": roof(2 * W * H * 2 + n * 72, t),
                     "wiener_units": int(np.count_nonzero(res["sse"][:, 1] != np.iinfo(np.int64).max)),
                     "sse_gain_wiener": float(1 - res["sse"][:, 1][res["sse"][:, 1] != np.iinfo(np.int64).max].sum() / max(1, res["sse"][:, 0][res["sse"][:, 1] != np.iinfo(np.int64).max].sum())),
                     "sse_gain_sgrproj": float(1 - res["sse"][:, 2].sum() / max(1, res["sse"][:, 0].sum()))}
        if keep is not None:
            keep[name] = dict(P=P, src=src, dgd=dgd, pad=PAD, results=res.copy())
    return out


def hadamard_satd(torch, lib, pkg, stream, steps, warmup, oracle=None):
    """SURVEY a8 / a9 (north_star: "SAD/SATD search"): hadamard_path_c's body -- residual of an 8-bit block against its prediction -> svt_aom_hadamard_32x32 ->
    svt_aom_satd -- for every 32x32 block of a 1080p picture x 8 candidate predictions (16 320 blocks per launch).  Bytes: 2 x 1 KB in + 4 B out per block."""
    g = np.random.default_rng(41)
    W, H, n_pred = 1920, 1088, 8
    src = g.integers(0, 256, (H, W), dtype=np.uint8)
    preds = np.clip(src[None].astype(np.int16) + g.integers(-24, 25, (n_pred, H, W)), 0, 255).astype(np.uint8)
    blocks = [(x, y) for y in range(0, H, 32) for x in range(0, W, 32)]
    n = len(blocks) * n_pred
    d = np.zeros(n, dtype=pkg.SatdDesc)
    xy = np.array(blocks * n_pred)
    d["in_off"] = xy[:, 1].astype(np.uint64) * W + xy[:, 0].astype(np.uint64)
    d["pred_off"] = np.repeat(np.arange(n_pred, dtype=np.uint64), len(blocks)) * (H * W) + d["in_off"]
    d["in_stride"] = d["pred_stride"] = W
    d_src, d_pred, d_d = _dev(torch, src), _dev(torch, preds), _dev(torch, d)
    d_out = torch.zeros(n, dtype=torch.int32, device="cuda")
    fn = lambda: lib.svt_hip_hadamard_satd_batch(d_src.data_ptr(), d_pred.data_ptr(), d_d.data_ptr(), n, 32, d_out.data_ptr(), None, stream)  # noqa: E731
    fn()
    torch.cuda.synchronize()
    got, checked = d_out.cpu().numpy().view(np.uint32), 0
    if oracle is not None:
        oracle.oracle_hadamard_satd.restype = C.c_uint32
        for i in np.linspace(0, n - 1, 48).astype(int):
            want = oracle.oracle_hadamard_satd(C.c_void_p(src.ctypes.data + int(d["in_off"][i])), W, C.c_void_p(preds.ctypes.data + int(d["pred_off"][i])), W, 32)
            assert int(got[i]) == int(want), ("hadamard_satd_32x32", i, int(got[i]), int(want))
            checked += 1
    t = _time(torch, fn, steps, warmup)
    return {"hadamard_satd_32x32": {"value": n / t / 1e6, "unit": "Mblocks/s (32x32: residual + Hadamard + SATD)", "blocks_per_launch": n, "parity_checked_values": checked,
                                    "roofline": roof(n * (2 * 1024 + 4), t)}}


def tf_inter_pred(torch, lib, pkg, stream, steps, warmup):
    """The temporal filter's final motion compensation: one 1080p 8-bit 4:2:0 central picture against 6 reference pictures, every 16x16 block (luma + chroma,
    MULTITAP_SHARP) = 6 x 8 160 blocks in one launch, predictions written as picture-sized planes.  Bytes: 1.5 B per sample out + the interpolation window in."""
    g = np.random.default_rng(23)
    W, H, PAD, n_refs = 1920, 1088, 80, 6
    shapes = [(H + 2 * PAD, W + 2 * PAD), (H // 2 + PAD, W // 2 + PAD), (H // 2 + PAD, W // 2 + PAD)]
    refs = [torch.randint(0, 256, (n_refs,) + s, dtype=torch.uint8, device="cuda") for s in shapes]
    pshape = [(H, W), (H // 2, W // 2), (H // 2, W // 2)]
    preds = [torch.zeros((n_refs,) + s, dtype=torch.uint8, device="cuda") for s in pshape]
    blocks = [(x, y) for y in range(0, H, 16) for x in range(0, W, 16)]
    n = len(blocks) * n_refs
    d = np.zeros(n, pkg.TfMcDesc)
    a = np.array(blocks * n_refs)
    r = np.repeat(np.arange(n_refs), len(blocks))
    d["pu_x"], d["pu_y"], d["bsize"] = a[:, 0], a[:, 1], 16
    d["mv_x"], d["mv_y"] = g.integers(-40, 41, n), g.integers(-40, 41, n)
    for pl in range(3):
        d["ref_off"][:, pl] = r.astype(np.uint64) * (shapes[pl][0] * shapes[pl][1])
        d["pred_off"][:, pl] = r.astype(np.uint64) * (pshape[pl][0] * pshape[pl][1])
    P = pkg.TfSubpelParams()
    P.bit_depth, P.mi_rows, P.mi_cols, P.ref_org_x, P.ref_org_y, P.ref_stride = 8, H // 4, W // 4, PAD, PAD, shapes[0][1]
    PL = pkg.TfMcPlanes()
    for pl in range(3):
        PL.ref[pl], PL.pred[pl], PL.ref_stride[pl], PL.pred_stride[pl] = refs[pl].data_ptr(), preds[pl].data_ptr(), shapes[pl][1], pshape[pl][1]
    d_d = torch.from_numpy(d.view(np.uint8).reshape(-1)).cuda()
    t = _time(torch, lambda: lib.svt_hip_tf_inter_pred_batch(C.addressof(P), C.addressof(PL), d_d.data_ptr(), n, 1, stream), steps, warmup, batches=3)
    out_bytes = n_refs * W * H * 1.5
    return {"tf_inter_pred_1080p8_6refs": {"us": t * 1e6, "blocks_per_s": n / t, "GBps_out": out_bytes / t / 1e9, "hbm_frac_out_plus_in": 2 * out_bytes / t / 8e12}}


def hme_chain(torch, lib, pkg, stream, steps, warmup):
    """The three HME levels of a 1080p picture against 4 references, chained on the device (svt_hip_hme_level_batch x 3: descriptor kernel ->
    svt_hip_sad_loop_batch -> rescale kernel per level): level 0 on the 1/16-area planes (2 x 2 regions of 16x16), levels 1 and 2 with 8x3 areas
    around the previous level's centres -- the preset-8 shape (enc_mode_config.c:141-216)."""
    g = np.random.default_rng(12)
    W, H, n_refs, nw, nh = 1920, 1080, 4, 2, 2
    aw, ah = W, (H + 7) & ~7
    sbs_x, sbs_y = (aw + 63) // 64, (ah + 63) // 64
    n = n_refs * sbs_x * sbs_y * nw * nh
    org = {0: 16, 1: 32, 2: 68}
    sa = {0: (16, 16), 1: (8, 3), 2: (8, 3)}
    stages = []
    for lv in (0, 1, 2):
        sh = 2 - lv
        w, h, o = W >> sh, H >> sh, org[lv]
        stride, rows = w + 2 * o, h + 2 * o + (64 >> sh)
        planes = torch.from_numpy(g.integers(0, 256, (1 + n_refs, rows, stride), dtype=np.uint8)).cuda()
        P = pkg.HmeLevelParams()
        P.level, P.sub_sampled, P.num_hme_sa_w, P.num_hme_sa_h, P.sa_width, P.sa_height = lv, 0, nw, nh, sa[lv][0], sa[lv][1]
        P.sbs_x, P.sbs_y, P.n_refs, P.prev_shift, P.aligned_width, P.aligned_height = sbs_x, sbs_y, n_refs, int(lv == 1), aw, ah
        P.src_off, P.src_stride = o * stride + o, stride
        P.ref_stride, P.ref_org_x, P.ref_org_y, P.ref_width, P.ref_height = stride, o, o, w, h
        for r in range(n_refs):
            P.ref_off[r] = (1 + r) * rows * stride
        ws = torch.zeros(lib.svt_hip_hme_level_workspace(C.addressof(P)), dtype=torch.uint8, device="cuda")
        stages.append((P, planes, ws, torch.zeros(n, dtype=torch.int64, device="cuda"), torch.zeros(2 * n, dtype=torch.int16, device="cuda")))
    zero = torch.zeros(2 * n, dtype=torch.int16, device="cuda")

    def fn():
        prev = zero
        for (P, planes, ws, sad, sc) in stages:
            lib.svt_hip_hme_level_batch(C.addressof(P), planes.data_ptr(), planes.data_ptr(), prev.data_ptr(), None, sad.data_ptr(), sc.data_ptr(), ws.data_ptr(),
                                        stream)
            prev = sc
    t = _time(torch, fn, steps, warmup)
    PA = (pkg.HmeLevelParams * 3)(*[st[0] for st in stages])
    pl = (C.c_void_p * 3)(*[st[1].data_ptr() for st in stages])
    sp = (C.c_void_p * 3)(*[st[3].data_ptr() for st in stages])
    cp = (C.c_void_p * 3)(*[st[4].data_ptr() for st in stages])
    tf = _time(torch, lambda: lib.svt_hip_hme_chain_batch(C.addressof(PA), C.addressof(pl), C.addressof(pl), None, C.addressof(sp), C.addressof(cp), stream), steps, warmup)
    # algorithmic bytes: every level's source and reference planes once (picture sizes, no borders) + the per-search result (8 B SAD + 4 B centre) of each level
    alg = sum((1 + n_refs) * (W >> (2 - lv)) * (H >> (2 - lv)) for lv in (0, 1, 2)) + 3 * n * 12
    return {"hme_3level_1080p_4refs": {"us_per_picture": tf * 1e6, "pictures_per_s": 1 / tf, "searches_per_level": n, "launches": 1,
                                       "us_per_picture_level_by_level": t * 1e6, "roofline": roof(alg, tf, "hme_chain_kernel"),
                                       "note": "level 0: 2x2 regions of 16x16 on 1/16-area planes; levels 1, 2: 8x3; one fused launch vs three level calls"}}


def me_stage(torch, lib, pkg, stream, steps, warmup):
    """The open-loop ME stage of one 1080p picture against 4 references, resident in HBM, chained on the device with no host round trip:
    decimation of the source to the 1/4 and 1/16-area planes (2 launches) -> HME level 0 / 1 / 2 -> final search centre + integer full-pel search
    (8x3 .. 16x9 areas around the HME result) -> MeSbResults formatting.  Everything the reference's ME thread does per SB except the
    content-dependent search-range probes."""
    g = np.random.default_rng(13)
    W, H, n_refs, nw, nh = 1920, 1080, 4, 2, 2
    aw, ah = W, (H + 7) & ~7
    sbs_x, sbs_y = (aw + 63) // 64, (ah + 63) // 64
    n_sb = sbs_x * sbs_y
    org = {0: 16, 1: 32, 2: 68}
    geo, bufs = {}, {}
    for lv in (0, 1, 2):
        sh = 2 - lv
        w, h, o = W >> sh, H >> sh, org[lv]
        stride, rows = w + 2 * o, h + 2 * o + (64 >> sh)
        geo[lv] = (w, h, o, stride, rows)
        bufs[lv] = torch.from_numpy(g.integers(0, 256, (1 + n_refs, rows, stride), dtype=np.uint8)).cuda()
    n_items = n_refs * n_sb * nw * nh
    sa = {0: (16, 16), 1: (8, 3), 2: (8, 3)}
    hp = []
    for lv in (0, 1, 2):
        w, h, o, stride, rows = geo[lv]
        P = pkg.HmeLevelParams()
        P.level, P.sub_sampled, P.num_hme_sa_w, P.num_hme_sa_h, P.sa_width, P.sa_height = lv, 0, nw, nh, sa[lv][0], sa[lv][1]
        P.sbs_x, P.sbs_y, P.n_refs, P.prev_shift, P.aligned_width, P.aligned_height = sbs_x, sbs_y, n_refs, int(lv == 1), aw, ah
        P.src_off, P.src_stride = o * stride + o, stride
        P.ref_stride, P.ref_org_x, P.ref_org_y, P.ref_width, P.ref_height = stride, o, o, w, h
        for r in range(n_refs):
            P.ref_off[r] = (1 + r) * rows * stride
        ws = torch.zeros(lib.svt_hip_hme_level_workspace(C.addressof(P)), dtype=torch.uint8, device="cuda")
        hp.append((P, ws, torch.zeros(n_items, dtype=torch.int64, device="cuda"), torch.zeros(2 * n_items, dtype=torch.int16, device="cuda")))
    w, h, o, stride, rows = geo[2]
    Q = pkg.MeIntegerSearchParams()
    Q.sbs_x, Q.sbs_y, Q.n_refs, Q.regions, Q.aligned_width, Q.aligned_height = sbs_x, sbs_y, n_refs, nw * nh, aw, ah
    Q.sa_min_width, Q.sa_min_height, Q.sa_max_width, Q.sa_max_height = 8, 3, 16, 9  # preset-8 1080p (enc_mode_config.c:325-326)
    for r in range(n_refs):
        Q.dist[r], Q.ref_pic_index[r], Q.ref_off[r] = 1 + r, r % 2, (1 + r) * rows * stride
    Q.src_off, Q.src_stride, Q.ref_stride, Q.ref_org_x, Q.ref_org_y = o * stride + o, stride, stride, o, o
    ws_i = torch.zeros(lib.svt_hip_me_integer_search_workspace(C.addressof(Q)), dtype=torch.uint8, device="cuda")
    bs, bm = torch.zeros(n_refs * n_sb * 85, dtype=torch.int32, device="cuda"), torch.zeros(n_refs * n_sb * 85, dtype=torch.int32, device="cuda")
    sco, sado = torch.zeros(n_refs * n_sb * 2, dtype=torch.int16, device="cuda"), torch.zeros(n_refs * n_sb, dtype=torch.int64, device="cuda")
    R = pkg.MeResultsParams()
    R.n_sb, R.num_of_list_to_search = n_sb, 2
    R.num_of_ref_pic_to_search[0], R.num_of_ref_pic_to_search[1] = 2, 2
    R.max_refs, R.max_cand = pkg.me_max_allocated_refs(2, 2)
    R.max_l0, R.enable_me_16x16, R.enable_me_8x8, R.prune_ref, R.gm_enabled = 2, 1, 1, 1, 1
    R.prune_ref_if_me_sad_dev_bigger_than_th, R.prune_me_candidates_th, R.picture_number = 30, 65, 16
    do_ref = torch.ones(n_sb * 8, dtype=torch.uint8, device="cuda")
    sz = torch.full((n_sb * 2,), 64, dtype=torch.uint8, device="cuda")
    tot, mvs = torch.zeros(n_sb * 85, dtype=torch.uint8, device="cuda"), torch.zeros(n_sb * 85 * R.max_refs * 4, dtype=torch.uint8, device="cuda")
    cands, st = torch.zeros(n_sb * 85 * R.max_cand, dtype=torch.uint8, device="cuda"), torch.zeros(n_sb * 28, dtype=torch.uint8, device="cuda")
    PA = (pkg.HmeLevelParams * 3)(*[h[0] for h in hp])
    pl = (C.c_void_p * 3)(*[bufs[lv].data_ptr() for lv in (0, 1, 2)])
    sp = (C.c_void_p * 3)(*[h[2].data_ptr() for h in hp])
    cp = (C.c_void_p * 3)(*[h[3].data_ptr() for h in hp])
    full = bufs[2].data_ptr() + geo[2][2] * geo[2][3] + geo[2][2]

    def run(st_):
        lib.svt_hip_downsample_2d_padded(full, geo[2][3], W, H, bufs[1].data_ptr(), geo[1][3], geo[1][2], geo[1][2], 2, st_)
        lib.svt_hip_downsample_2d_padded(full, geo[2][3], W, H, bufs[0].data_ptr(), geo[0][3], geo[0][2], geo[0][2], 4, st_)
        lib.svt_hip_hme_chain_batch(C.addressof(PA), C.addressof(pl), C.addressof(pl), None, C.addressof(sp), C.addressof(cp), st_)
        lib.svt_hip_me_integer_search_batch(C.addressof(Q), bufs[2].data_ptr(), bufs[2].data_ptr(), hp[2][2].data_ptr(), hp[2][3].data_ptr(), None, None, None,
                                            bs.data_ptr(), bm.data_ptr(), sco.data_ptr(), sado.data_ptr(), ws_i.data_ptr(), st_)
        lib.svt_hip_me_results_batch(C.addressof(R), bs.data_ptr(), bm.data_ptr(), do_ref.data_ptr(), sz.data_ptr(), tot.data_ptr(), mvs.data_ptr(),
                                     cands.data_ptr(), st.data_ptr(), st_)
    t = _time(torch, lambda: run(stream), steps, warmup)
    # the same chain captured once into a HIP graph and replayed (every entry point only enqueues on the stream it is given)
    import time as _t
    gs = lib.svt_hip_stream_create()
    lib.svt_hip_graph_capture_begin(gs)
    run(gs)
    gexec = lib.svt_hip_graph_capture_end(gs)
    lib.svt_hip_graph_launch(gexec, gs)
    lib.svt_hip_stream_synchronize(gs)
    tgs = []
    for _ in range(5):
        t0 = _t.perf_counter()
        for _ in range(steps):
            lib.svt_hip_graph_launch(gexec, gs)
        lib.svt_hip_stream_synchronize(gs)
        tgs.append((_t.perf_counter() - t0) / steps)
    tg = sorted(tgs)[2]
    lib.svt_hip_graph_destroy(gexec)
    lib.svt_hip_stream_destroy(gs)
    return {"me_stage_1080p_4refs": {"us_per_picture": t * 1e6, "pictures_per_s": 1 / t, "hip_graph_us_per_picture": tg * 1e6, "sb_refs": n_refs * n_sb,
                                     "note": "decimate x2, HME L0-L2 (one fused launch), final centre + integer search (8x3..16x9), MeSbResults: device-resident chain"}}


def me_session_stage(torch, lib, pkg, stream, steps, warmup, npics=48, slots=None):
    """PCIe-inclusive WHOLE ME stage from pinned host pictures: upload once, quarter / sixteenth planes on the device, HME levels 0-2, final centre +
    integer search, MeSbResults returned to pinned host memory; 4 references (2 + 2), two submissions in flight."""
    import os, time as _t
    slots = slots or int(os.environ.get("SVT_BENCH_SLOTS", "3"))
    W, H, PAD = 1920, 1080, 68
    stride, rows = W + 2 * PAD, H + 2 * PAD
    nbytes = stride * rows
    g = np.random.default_rng(14)
    sbs = ((W + 63) // 64) * ((H + 63) // 64)
    hp = [lib.svt_hip_host_alloc(nbytes) for _ in range(8)]
    for q in hp:
        a = g.integers(0, 256, nbytes, dtype=np.uint8)  # (kept alive across the copy)
        C.memmove(q, a.ctypes.data, nbytes)
    S = pkg.MeStageParams()
    S.num_hme_sa_w, S.num_hme_sa_h = 2, 2
    for lv, (a, b) in enumerate(((16, 16), (8, 3), (8, 3))):
        S.hme_sa_width[lv], S.hme_sa_height[lv] = a, b
    S.me_sa_min_width, S.me_sa_min_height, S.me_sa_max_width, S.me_sa_max_height = 8, 3, 16, 9
    for r in range(4):
        S.dist[r], S.ref_pic_index[r] = 1 + r, r % 2
    S8 = pkg.MeStageParams()  # what the reference's svt_aom_sig_deriv_me gives for preset 8 at 1080p, CRF 35 (pkg.m8_me_settings, pinned in tests/test_hme.py)
    R = S.results
    R.num_of_list_to_search = 2
    R.num_of_ref_pic_to_search[0], R.num_of_ref_pic_to_search[1] = 2, 2
    R.max_refs, R.max_cand = pkg.me_max_allocated_refs(2, 2)
    R.max_l0, R.enable_me_16x16, R.enable_me_8x8, R.prune_ref, R.gm_enabled = 2, 1, 1, 1, 1
    R.prune_ref_if_me_sad_dev_bigger_than_th, R.prune_me_candidates_th, R.picture_number = 30, 65, 16
    sizes = [sbs * 85, sbs * 85 * R.max_refs * 4, sbs * 85 * R.max_cand, sbs * 28]
    hosts = []
    for _ in range(slots):
        b = [lib.svt_hip_host_alloc(n) for n in sizes]
        hosts.append((b, pkg.MeResultsHost(None, b[0], b[1], b[2], b[3], None, None)))

    C.memmove(C.addressof(S8.results), C.addressof(R), C.sizeof(R))
    m8 = pkg.fill_m8_stage_params(S8, [1, 2, 1, 2], [0, 1, 0, 1], qp=35, temporal_layer=1)

    sub = [0.0]

    def run(sess, n, S=S):
        pending = []
        sub[0] = 0.0
        for k in range(n):
            refs = np.array([k - 1, k - 2, k - 3, k - 4], np.int64)
            ts0 = _t.perf_counter()
            slot = lib.svt_hip_me_session_submit_stage(sess, k, hp[k % 8], refs.ctypes.data if k >= 4 else None, 4 if k >= 4 else 0, C.addressof(S),
                                                       C.addressof(hosts[k % slots][1]) if k >= 4 else None)
            sub[0] += _t.perf_counter() - ts0
            assert slot >= 0, slot
            pending.append(slot)
            if len(pending) == slots:
                lib.svt_hip_me_session_wait(sess, pending.pop(0))
        for slot in pending:
            lib.svt_hip_me_session_wait(sess, slot)
    ts, ts8 = [], []
    for it in range(max(steps // 4, 2) + 1):
        for cfg, acc in ((S, ts), (S8, ts8)):
            sess = lib.svt_hip_me_session_create(W, H, stride, PAD, PAD, rows, 8, 4, 16, 9, slots)
            assert lib.svt_hip_me_session_enable_stage(sess, 32, 16, 4, 32, 16) == 0
            t0 = _t.perf_counter()
            run(sess, npics if it else 8, cfg)
            if it:
                acc.append((_t.perf_counter() - t0, sub[0]))
            lib.svt_hip_me_session_destroy(sess)
    for q in hp:
        lib.svt_hip_host_free(q)
    for b, _ in hosts:
        for q in b:
            lib.svt_hip_host_free(q)
    (t, hs), (t8, hs8) = min(ts), min(ts8)
    return {"me_session_stage_1080p_host": {"pictures_per_s": npics / t, "us_per_picture": t / npics * 1e6, "host_submit_us_per_picture": hs / npics * 1e6,
                                            "pictures_in_flight": slots, "h2d_MB_per_picture": nbytes / 1e6,
                                            "roofline": {"bound": "pcie", "achieved": (nbytes + sum(sizes)) / (t / npics) / 1e9, "peak": 64.0, "unit": "GB/s",
                                                         "frac": (nbytes + sum(sizes)) / (t / npics) / 1e9 / 64.0, "kernel_us": t / npics * 1e6, "binds": "pcie",
                                                         "algorithmic_bytes_per_launch": nbytes + sum(sizes),
                                                         "note": "host form: one picture up, its MeSbResults down per call; peak = PCIe 5.0 x16 per direction"},
                                            "d2h_MB_per_picture": sum(sizes) / 1e6,
                                            "note": "decimation + HME 0-2 + integer search + MeSbResults per picture, 4 references, PCIe inclusive"},
            "me_session_stage_1080p_host_preset8": {"pictures_per_s": npics / t8, "us_per_picture": t8 / npics * 1e6, "host_submit_us_per_picture": hs8 / npics * 1e6,
                                                    "settings": {k: v for k, v in m8.items() if k in ("hme_levels", "hme_l0", "me", "hme_prune", "zz")},
                                                    "note": "the reference's own preset-8 derivation (svt_aom_sig_deriv_me at ENC_M8, 1080p, CRF 35): HME levels 0-1 only, "
                                                            "zero-motion gating, pre-HME (8x100..350 / 32..128x7), level-0 areas by reference index, HME pruning + "
                                                            "search-range divisors, 8x8-variance probe, sub-sampled SADs, ME area 8x3..8x4; synthetic noise pictures, so "
                                                            "no early exit fires"}}


def tpl_src_stage(torch, lib, pkg, stream, steps, warmup, keep=None):
    """the TPL dispenser's source-based half (src_ops_process.c:519-969) of one 1080p 8-bit picture as one launch: tpl level 0 (16x16 blocks, 8 160 of
    them), 2 + 2 reference pictures, up to 7 ME candidates per block from the 16x16 ME results, DC intra from source neighbours, the best mode's 16x16 forward DCT +
    quantize_fp error.  Algorithmic bytes per block: the source block + one reference block per candidate evaluated (256 B each) + the candidate / vector tables +
    the 40-byte statistics record."""
    g = np.random.default_rng(23)
    W, H, PAD, n_l0, n_l1 = 1920, 1080, 96, 2, 2
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    sbs_x, sbs_y, n_ref = (aw + 63) // 64, (ah + 63) // 64, n_l0 + n_l1
    stride, rows = aw + 2 * PAD + 24, ah + 2 * PAD + 16
    yy, xx = np.mgrid[0:rows, 0:stride]
    base = ((xx * 3 + yy * 2) & 255).astype(np.int32) + (((xx // 16 + yy // 16) % 7) << 3)
    planes = np.zeros((1 + n_ref, rows, stride), np.uint8)
    planes[0] = np.clip(base + g.integers(-12, 13, base.shape), 0, 255)
    for r in range(n_ref):
        planes[1 + r] = np.clip(np.roll(planes[0].astype(np.int32), (r + 1, -2 * r - 1), (0, 1)) + g.integers(-6 - 8 * r, 7 + 8 * r, base.shape), 0, 255)
    P = pkg.TplSrcParams()
    P.width, P.height, P.aligned_width, P.sbs_x, P.n_sb, P.src_stride = W, H, aw, sbs_x, sbs_x * sbs_y, stride
    P.src_off = PAD * stride + PAD
    P.dispenser_search_level, P.subsample_tx, P.pf_shape, P.disable_intra_pred, P.i_slice, P.enable_me_16x16, P.enable_me_8x8 = 0, 0, 2, 0, 0, 1, 0
    n_pus, max_cand = 21, 7
    P.max_refs, P.max_l0, P.max_cand = n_ref, n_l0, max_cand
    P.quant_fp[0], P.quant_fp[1], P.round_fp[0], P.round_fp[1], P.dequant[0], P.dequant[1] = 532, 431, 61, 76, 123, 152  # q index 120 of the 8-bit tables
    for l in range(2):
        for r in range(4):
            R, have = P.refs[l * 4 + r], r < (n_l0 if l == 0 else n_l1)
            slot = r if l == 0 else n_l0 + r
            R.plane_off = (1 + slot) * rows * stride if have else 0
            R.picture_number, R.stride, R.org_x, R.org_y, R.max_width, R.max_height, R.valid = 100 + 10 * l + r, stride, PAD, PAD, W, H, int(have)
    n_sb = P.n_sb
    tot = g.integers(2, max_cand + 1, (n_sb, n_pus)).astype(np.uint8)
    shp = (n_sb, n_pus, max_cand)
    cand = (g.integers(0, 2, shp) | (g.integers(0, n_l0, shp) << 2) | (g.integers(0, n_l1, shp) << 4)).astype(np.uint8)
    mvx, mvy = g.integers(-24, 25, (n_sb, n_pus, n_ref)).astype(np.int16), g.integers(-16, 17, (n_sb, n_pus, n_ref)).astype(np.int16)
    mvs = np.ascontiguousarray((mvy.astype(np.uint16).astype(np.uint32) << 16) | mvx.astype(np.uint16).astype(np.uint32))
    cells = (aw + 15) // 16 * ((ah + 15) // 16)
    d_pl, d_tot, d_mv, d_cand = _dev(torch, planes), _dev(torch, tot), _dev(torch, mvs), _dev(torch, cand)
    d_out = torch.zeros(cells * 40, dtype=torch.uint8, device="cuda")
    t = _time(torch, lambda: lib.svt_hip_tpl_src_stage(C.addressof(P), d_pl.data_ptr(), d_pl.data_ptr(), d_tot.data_ptr(), d_mv.data_ptr(), d_cand.data_ptr(), d_out.data_ptr(),
                                                       stream), steps, warmup, batches=3)
    out = d_out.cpu().numpy().view(pkg.TplSrcStats)
    n_blk = int(out["written"].sum())
    alg = n_blk * (256 + 40) + int(tot[:, 5:21].sum()) * 256 + tot.nbytes + mvs.nbytes + cand.nbytes
    if keep is not None:
        keep.update(P=P, planes=planes, tot=tot, mvs=mvs, cand=cand, out=out.copy(), cells=cells)
    return {"tpl_src_stage_1080p8": {"us": t * 1e6, "pictures_per_s": 1 / t, "blocks_16x16": n_blk, "Mblocks_per_s": n_blk / t / 1e6,
                                      "inter_wins_frac": float(np.mean(out["best_mode"][out["written"] > 0] != 0)) if n_blk else 0.0,
                                      "roofline": roof(alg, t, "tpl_src_kernel<16, 16>")}}


def tpl_level1_stage(torch, lib, pkg, stream, steps, warmup, keep, size=(1920, 1080)):
    """SURVEY 8f / VERDICT r3 item 9: both halves of the TPL dispenser with the option set of tpl level 1 (presets M0-M2: every intra mode DC..PAETH, transform + SATD
    costs, quarter-pel vectors, rate; csrc/tpl_full.hip) on a 1080p 8-bit picture, device-resident.  Smooth content and references displaced by whole and half samples,
    so that the inter path and its sub-pel search decide most blocks.  One wave per 16x16 block; VALU / latency bound (13 transforms per block for the intra modes, ~15
    bilinear variances per inter candidate)."""
    g = np.random.default_rng(29)
    W, H, PAD, n_l0, n_l1 = size[0], size[1], 96, 2, 2
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    sbs_x, sbs_y, n_ref = (aw + 63) // 64, (ah + 63) // 64, n_l0 + n_l1
    stride, rows = aw + 2 * PAD + 24, ah + 2 * PAD + 16
    yy, xx = np.mgrid[0:rows, 0:stride]
    img = 128 + 60 * np.sin(xx / 9.0 + yy / 23.0) + 40 * np.sin(yy / 7.0 - xx / 31.0) + 25 * ((xx // 24 + yy // 20) % 2)
    planes = np.zeros((1 + n_ref, rows, stride), np.uint8)
    planes[0] = np.clip(img + g.integers(-4, 5, img.shape), 0, 255)
    for r in range(n_ref):
        a = np.roll(planes[0].astype(np.int32), (r + 1, -2 * r - 1), (0, 1))
        b = np.roll(planes[0].astype(np.int32), (r + 1 + (r & 1), -2 * r), (0, 1))
        planes[1 + r] = np.clip(((a + b + 1) >> 1) + g.integers(-2 - r, 3 + r, a.shape), 0, 255)
    P = pkg.TplSrcParams()
    P.width, P.height, P.aligned_width, P.sbs_x, P.n_sb, P.src_stride = W, H, aw, sbs_x, sbs_x * sbs_y, stride
    P.src_off = PAD * stride + PAD
    P.dispenser_search_level, P.subsample_tx, P.pf_shape, P.disable_intra_pred, P.i_slice, P.enable_me_16x16, P.enable_me_8x8 = 0, 0, 0, 0, 0, 1, 0
    P.intra_mode_end, P.search_flags = 12, 1 | 2 | (2 << 2)
    n_pus, max_cand = 21, 7
    P.max_refs, P.max_l0, P.max_cand = n_ref, n_l0, max_cand
    P.quant_fp[0], P.quant_fp[1], P.round_fp[0], P.round_fp[1], P.dequant[0], P.dequant[1] = 532, 431, 61, 76, 123, 152  # q index 120 of the 8-bit tables
    for l in range(2):
        for r in range(4):
            R, have = P.refs[l * 4 + r], r < (n_l0 if l == 0 else n_l1)
            slot = r if l == 0 else n_l0 + r
            R.plane_off = (1 + slot) * rows * stride if have else 0
            R.picture_number, R.stride, R.org_x, R.org_y, R.max_width, R.max_height, R.valid = 100 + 10 * l + r, stride, PAD, PAD, W, H, int(have)
    n_sb = P.n_sb
    tot = g.integers(1, 4, (n_sb, n_pus)).astype(np.uint8)  # 1-3 candidates per block (what the 16x16 ME results of an M2 encode hold)
    shp = (n_sb, n_pus, max_cand)
    cand = (g.integers(0, 2, shp) | (g.integers(0, n_l0, shp) << 2) | (g.integers(0, n_l1, shp) << 4)).astype(np.uint8)
    mvx = np.zeros((n_sb, n_pus, n_ref), np.int16)
    mvy = np.zeros((n_sb, n_pus, n_ref), np.int16)
    for r in range(n_ref):
        mvx[:, :, r] = -2 * r - 1 + g.integers(-1, 2, (n_sb, n_pus))
        mvy[:, :, r] = r + 1 + g.integers(-1, 2, (n_sb, n_pus))
    bad = g.random((n_sb, n_pus)) < 0.35  # a third of the blocks get vectors that miss: the intra modes win there and the reconstruction half has its dependency chains
    mvx[bad] += 9
    mvy[bad] -= 7
    mvs = np.ascontiguousarray((mvy.astype(np.uint16).astype(np.uint32) << 16) | mvx.astype(np.uint16).astype(np.uint32))
    cells = (aw + 15) // 16 * ((ah + 15) // 16)
    d_pl, d_tot, d_mv, d_cand = _dev(torch, planes), _dev(torch, tot), _dev(torch, mvs), _dev(torch, cand)
    d_out = torch.zeros(cells * 40, dtype=torch.uint8, device="cuda")
    t_src = _time(torch, lambda: lib.svt_hip_tpl_src_stage(C.addressof(P), d_pl.data_ptr(), d_pl.data_ptr(), d_tot.data_ptr(), d_mv.data_ptr(), d_cand.data_ptr(),
                                                           d_out.data_ptr(), stream), steps, warmup, batches=2)
    out = d_out.cpu().numpy().view(pkg.TplSrcStats)
    n_blk, n_cand = int(out["written"].sum()), int(tot[:, 5:21].sum())
    alg_src = n_blk * (256 + 40) + n_cand * 17 * 17 * 15 + tot.nbytes + mvs.nbytes + cand.nbytes  # per candidate: ~15 bilinear variances of a 17x17 window
    roof_src = roof(alg_src, t_src, "tpl_full_src_kernel")  # (here: the roofline object remembers the region timed LAST)
    R = pkg.TplReconParams()
    C.memmove(C.addressof(R.src), C.addressof(P), C.sizeof(P))
    for i in range(8):
        C.memmove(C.addressof(R.rec_refs[i]), C.addressof(P.refs[i]), C.sizeof(pkg.TplRef))
    R.recon_off, R.recon_stride, R.is_ref = P.src_off, stride, 1
    d_rec = torch.zeros(rows * stride, dtype=torch.uint8, device="cuda")
    d_rs = torch.zeros(cells * 40, dtype=torch.uint8, device="cuda")

    def run():
        d_rec.zero_()
        lib.svt_hip_tpl_recon_stage(C.addressof(R), d_pl.data_ptr(), d_pl.data_ptr(), d_out.data_ptr(), d_rec.data_ptr(), d_rs.data_ptr(), stream)
    t_rec = _time(torch, run, steps, warmup, batches=2)
    rs = d_rs.cpu().numpy().view(pkg.TplReconStats)
    w = out["written"] > 0
    keep.update(P=P, planes=planes, tot=tot, mvs=mvs, cand=cand, out=out.copy(), cells=cells, recon=d_rec.cpu().numpy().reshape(rows, stride), recon_out=rs.copy(), pad=PAD, n_pus=n_pus)
    alg_rec = n_blk * (3 * 256 + 80)
    return {"tpl_l1_src_1080p8": {"us": t_src * 1e6, "pictures_per_s": 1 / t_src, "blocks_16x16": n_blk, "inter_wins_frac": float(np.mean(out["best_mode"][w] == 16)),
                                             "fractional_vectors_frac": float(np.mean(((out["mv_row"][w] | out["mv_col"][w]) & 7) != 0)),
                                             "intra_modes_chosen": int(len(np.unique(out["best_intra_mode"][w]))), "roofline": roof_src},
            "tpl_l1_recon_1080p8": {"us": t_rec * 1e6, "pictures_per_s": 1 / t_rec, "blocks_16x16": n_blk, "roofline": roof(alg_rec, t_rec, "tpl_full_recon_kernel")}}


def tpl_recon_stage(torch, lib, pkg, stream, steps, warmup, keep):
    """the TPL dispenser's reconstruction half (src_ops_process.c:979-1198) of the tpl_src_stage leg's 1080p picture, device-resident: one launch per
    anti-diagonal of the 16x16 block grid (187 at 1080p, tpl level 4).  Launch-latency bound by construction (csrc/tpl.hip); every run starts from a
    zeroed reconstruction plane (the memset is inside the timed region: 2.6 MB)."""
    P, planes, src = keep["P"], keep["planes"], keep["out"]
    rows, stride = planes.shape[1], planes.shape[2]
    R = pkg.TplReconParams()
    C.memmove(C.addressof(R.src), C.addressof(P), C.sizeof(P))
    for i in range(8):
        C.memmove(C.addressof(R.rec_refs[i]), C.addressof(P.refs[i]), C.sizeof(pkg.TplRef))
    R.recon_off, R.recon_stride, R.is_ref = P.src_off, stride, 1
    d_pl, d_src = _dev(torch, planes), _dev(torch, src.view(np.uint8))
    d_rec = torch.zeros(rows * stride, dtype=torch.uint8, device="cuda")
    d_out = torch.zeros(keep["cells"] * 40, dtype=torch.uint8, device="cuda")

    def run():
        d_rec.zero_()
        lib.svt_hip_tpl_recon_stage(C.addressof(R), d_pl.data_ptr(), d_pl.data_ptr(), d_src.data_ptr(), d_rec.data_ptr(), d_out.data_ptr(), stream)
    import os
    forms = {}
    # 7 = the default: ONE launch, every block in flight, DC blocks wait for the cells above / left of them, write-through stores + drained flag on the producer's side, one acquire on the
    # consumer's; 5 = release / acquire fences (round 4's default); 6 = 5 polling with loads; 4 = 5 with sequentially-consistent fences; 0 = one launch per
    # anti-diagonal; 1 = the row wavefront in one launch; 2 = 1 with the rows given to the XCDs in contiguous chunks; 3 = 1 with release / acquire fences.  All are kept
    # for the checker.
    for form in (4, 3, 2, 1, 0, 6, 5, 7):
        os.environ["SVT_HIP_TPL_RECON_FORM"] = str(form)
        t = _time(torch, run, steps, warmup, batches=3)
        out = d_out.cpu().numpy().view(pkg.TplReconStats)
        forms[form] = (t, d_rec.cpu().numpy().reshape(rows, stride), out.copy())
    os.environ.pop("SVT_HIP_TPL_RECON_FORM", None)
    n_blk = int(out["written"].sum())
    keep.update(R=R, recon=forms[7][1], recon_out=forms[7][2], recon_forms={f: (v[1], v[2]) for f, v in forms.items()}, recon_stride=stride)
    cols16, rows16 = (P.aligned_width + 15) // 16, (((P.height + 7) & ~7) + 15) // 16
    alg = n_blk * (3 * 256 + 80)  # per block: source, prediction (reference or neighbours), reconstruction, the two statistics records
    n_dc = int(np.sum((src["best_mode"] == 0) & (src["written"] > 0)))
    # both halves as ONE host call from page-locked host pictures (svt_hip_tpl_stage_host: what the encoder seam calls; the reference's pools are page-locked at init):
    # source + 4 source references + 4 reconstruction references uploaded, statistics of both halves and the written rectangle downloaded -- wall clock, PCIe inclusive
    import time as _t

    class HostPlanes(C.Structure):
        _fields_ = [("src_buf", C.c_void_p), ("src_rows", C.c_uint32), ("ref_rows", C.c_uint32 * 8), ("ref_buf", C.c_void_p * 8)]
    psize = rows * stride
    pinned = [lib.svt_hip_host_alloc(psize) for _ in range(planes.shape[0] * 2 - 1)]  # the source, its references, and copies standing for their TPL reconstructions
    for k in range(planes.shape[0]):
        C.memmove(pinned[k], planes[k].ctypes.data, psize)
        if k:
            C.memmove(pinned[planes.shape[0] + k - 1], planes[k].ctypes.data, psize)
    rec_pin = lib.svt_hip_host_alloc(psize)
    SP, RP = HostPlanes(), HostPlanes()
    RF = pkg.TplReconParams.from_buffer_copy(R)
    SP.src_buf, SP.src_rows = pinned[0], rows
    for r in range(8):
        if P.refs[r].valid:
            k = P.refs[r].plane_off // psize
            SP.ref_buf[r], SP.ref_rows[r] = pinned[k], rows
            RP.ref_buf[r], RP.ref_rows[r] = pinned[planes.shape[0] + k - 1], rows
            RF.rec_refs[r].plane_off = 0
            RF.src.refs[r].plane_off = 0
    tot, mvs, cand = keep["tot"], keep["mvs"], keep["cand"]
    h_src, h_out = np.zeros(keep["cells"], src.dtype), np.zeros(keep["cells"], pkg.TplReconStats)
    vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731

    def fused():
        assert lib.svt_hip_tpl_stage_host(C.addressof(RF), C.addressof(SP), C.addressof(RP), vp(tot), vp(mvs), vp(cand), vp(h_src), rec_pin, rows, vp(h_out)) == 0
    t_fused = regions.hook(fused, name="tpl_stage_host")
    if t_fused is None:
        for _ in range(2):
            fused()
        t0 = _t.perf_counter()
        for _ in range(10):
            fused()
        t_fused = (_t.perf_counter() - t0) / 10
    fused_ok = all(np.array_equal(h_out[f], forms[7][2][f]) for f in ("srcrf_dist", "recrf_dist", "written", "coded")) and np.array_equal(h_src["srcrf_dist"], src["srcrf_dist"])
    if not fused_ok:
        raise SystemExit("bench: svt_hip_tpl_stage_host differs from the two device stages -- no numbers recorded")
    # the same with the planes RESIDENT across calls (svt_hip_tpl_stage_host_resident, what the seam calls): every call brings a new source picture and makes a new
    # reconstruction; its references -- sources and reconstructions of pictures of earlier calls -- are found on the device
    class PlaneIds(C.Structure):
        _fields_ = [("src", C.c_uint64), ("src_ref", C.c_uint64 * 8), ("rec_ref", C.c_uint64 * 8), ("recon", C.c_uint64), ("recon_width", C.c_uint32),
                    ("recon_height", C.c_uint32), ("recon_org_x", C.c_uint32), ("recon_org_y", C.c_uint32)]
    I = PlaneIds()
    for r in range(8):
        if P.refs[r].valid:
            k = P.refs[r].plane_off // psize
            I.src_ref[r], I.rec_ref[r] = 1000 + k, 2000 + k
    I.recon_width, I.recon_height, I.recon_org_x, I.recon_org_y = P.width, P.height, int(P.src_off % stride), int(P.src_off // stride)
    serial = [10 ** 6]

    def resident():
        serial[0] += 1
        I.src, I.recon = serial[0], 10 ** 7 + serial[0]  # (new content every call: uploaded / produced anew)
        assert lib.svt_hip_tpl_stage_host_resident(C.addressof(RF), C.addressof(SP), C.addressof(RP), C.addressof(I), vp(tot), vp(mvs), vp(cand), vp(h_src), rec_pin, rows,
                                                   vp(h_out)) == 0
    t_res = regions.hook(resident, name="tpl_stage_host_resident")
    if t_res is None:
        for _ in range(3):
            resident()
        t0 = _t.perf_counter()
        for _ in range(10):
            resident()
        t_res = (_t.perf_counter() - t0) / 10
    if not all(np.array_equal(h_out[f], forms[7][2][f]) for f in ("srcrf_dist", "recrf_dist", "written", "coded")):
        raise SystemExit("bench: svt_hip_tpl_stage_host_resident differs from the two device stages -- no numbers recorded")
    for q in pinned + [rec_pin]:
        lib.svt_hip_tpl_plane_drop(q)
        lib.svt_hip_host_free(q)
    up_mb = (planes.shape[0] * 2 - 1) * psize / 1e6
    return {"tpl_stage_host_1080p8": {"ms": t_fused * 1e3, "pictures_per_s": 1 / t_fused, "uploaded_MB": up_mb, "pcie_inclusive": True, "equals_device_stages": True,
                                      "roofline": {"bound": "pcie", "achieved": up_mb * 1e6 / t_fused / 1e9, "peak": 64.0, "unit": "GB/s", "frac": up_mb * 1e6 / t_fused / 1e9 / 64.0,
                                                   "kernel_us": t_fused * 1e6, "binds": "pcie", "algorithmic_bytes_per_launch": up_mb * 1e6},
                                      "note": "both halves of the TPL dispenser in one host call from page-locked pictures (9 planes up), wall clock"},
            "tpl_stage_host_resident_1080p8": {"ms": t_res * 1e3, "pictures_per_s": 1 / t_res, "uploaded_MB": psize / 1e6, "pcie_inclusive": True, "equals_device_stages": True,
                                               "roofline": {"bound": "pcie", "achieved": 2 * psize / t_res / 1e9, "peak": 64.0, "unit": "GB/s", "frac": 2 * psize / t_res / 1e9 / 64.0,
                                                            "kernel_us": t_res * 1e6, "binds": "pcie", "algorithmic_bytes_per_launch": 2 * psize},
                                               "note": "the same call with the references' planes resident on the device: one new source picture up, the written rectangle and the statistics down"},
            "tpl_recon_stage_1080p8": {"us": t * 1e6, "pictures_per_s": 1 / t, "form": "7: one launch, one wave per block, dependencies as data, write-through stores + drained flag / one acquire (csrc/tpl.hip tpl_recon_dep_kernel)",
                                        "release_acquire_fences_form5_us": forms[5][0] * 1e6,
                                        "sequentially_consistent_fences_form_us": forms[4][0] * 1e6, "load_polling_form6_us": forms[6][0] * 1e6, "anti_diagonal_launches_form_us": forms[0][0] * 1e6,
                                        "row_wavefront_form_us": forms[1][0] * 1e6, "row_wavefront_xcd_chunks_form_us": forms[2][0] * 1e6,
                                        "row_wavefront_release_acquire_form_us": forms[3][0] * 1e6, "blocks_16x16": n_blk, "intra_blocks": n_dc,
                                        "anti_diagonals": cols16 + rows16 - 1, "coded_frac": float(np.mean(out["coded"][out["written"] > 0])) if n_blk else 0.0,
                                        "roofline": roof(alg, t, "tpl_recon_dep_kernel<16, 16>")}}


def tf_picture_stage(torch, lib, pkg, stream, steps, warmup, keep=None, size=(1920, 1080)):
    """The temporal filter of one 1080p 8-bit 4:2:0 central picture with 4 reference pictures as ONE stage call from HOST pictures (svt_hip_tf_picture_host: upload,
    sub-pel refinement of every block size, decisions, final motion compensation, 32x32 errors, filter, download) -- PCIe-inclusive wall time, the form the encoder
    seam calls.  Preset-8 controls (bilinear 64 / 32 searches, half + quarter pel, sub-sampled distortions, no 8x8)."""
    import time
    g = np.random.default_rng(29)
    (W, H), PAD, n_refs = size, 80, 4
    nsx, nsy = (W + 63) // 64, (H + 63) // 64
    n_sb = nsx * nsy
    stride, rows, cstride, crows = W + 2 * PAD, 64 * nsy + 2 * PAD, W // 2 + PAD, 32 * nsy + PAD
    yy, xx = np.mgrid[0:rows, 0:stride].astype(np.float32)
    base = (0.5 + 0.25 * np.sin(xx / 3.3) * np.cos(yy / 4.1) + 0.2 * np.sin((xx + 2 * yy) / 9.1)) * 255
    cy, cx = np.mgrid[0:crows, 0:cstride].astype(np.float32)
    cb = (0.5 + 0.3 * np.sin(cx / 4.7) * np.cos(cy / 3.9)) * 255
    pics = []
    for r in range(n_refs + 1):
        y = np.clip(np.roll(base, (r, -r), (0, 1)) + g.normal(0, 2, base.shape), 0, 255).astype(np.uint8)
        u = np.clip(np.roll(cb, r, 1) + g.normal(0, 2, cb.shape), 0, 255).astype(np.uint8)
        v = np.clip(np.roll(cb.T.copy().T, -r, 0) + g.normal(0, 2, cb.shape), 0, 255).astype(np.uint8)
        pics.append([np.ascontiguousarray(y), np.ascontiguousarray(u), np.ascontiguousarray(v)])
    P = pkg.TfPictureParams()
    P.sp.half_pel_mode, P.sp.quarter_pel_mode, P.sp.eight_pel_mode, P.sp.subsampling_shift, P.sp.bit_depth = 1, 1, 0, 1, 8
    P.sp.mi_rows, P.sp.mi_cols, P.sp.ref_org_x, P.sp.ref_org_y, P.sp.ref_stride = H // 4, W // 4, PAD, PAD, stride
    P.tf.tf_decay_factor_fp16[0], P.tf.tf_decay_factor_fp16[1], P.tf.tf_decay_factor_fp16[2] = 2400000, 5200000, 4800000
    P.tf.tf_mv_dist_th, P.tf.tf_chroma, P.tf.use_zz_based_filter, P.tf.encoder_bit_depth, P.tf.ss_x, P.tf.ss_y = 135, 1, 0, 8, 1, 1
    P.pic_w_sb, P.pic_h_sb, P.uv_stride, P.me_exit_th, P.pred_error_32x32_th = nsx, nsy, cstride, 0, 20 * 32 * 32
    P.use_2tap, P.enable_8x8_pred, P.use_pred_64x64_only_th = 1, 0, 35
    tabs = []
    for r in range(n_refs):
        mvx, mvy = g.integers(-1, 2, (n_sb, 85)) - (r + 1), g.integers(-1, 2, (n_sb, 85)) + (r + 1)
        best_mv = ((mvy.astype(np.int16).astype(np.uint16).astype(np.uint32) << 16) | mvx.astype(np.int16).astype(np.uint16)).astype(np.uint32)
        best_sad = g.integers(500, 4000, (n_sb, 85)).astype(np.uint32)
        best_sad[:, 0] = (best_sad[:, 1:5].sum(1) * g.uniform(0.9, 1.8, n_sb)).astype(np.uint32)
        hme_sc = np.stack([np.full(n_sb, -(r + 1)), np.full(n_sb, r + 1)], 1).astype(np.int16)
        tabs.append([np.ascontiguousarray(x) for x in (best_sad, best_mv, hme_sc, np.full(n_sb, 10 ** 6, np.uint64))])
    hp = lambda pic: pkg.TfHostPicture(pic[0].ctypes.data, pic[1].ctypes.data, pic[2].ctypes.data, pic[0].size, pic[1].size, None)  # noqa: E731
    refs = (pkg.TfHostPicture * n_refs)(*[hp(x) for x in pics[1:]])
    me = (pkg.TfMeTables * n_refs)(*[pkg.TfMeTables(*[x.ctypes.data for x in t]) for t in tabs])
    out = [x.copy() for x in pics[0]]
    st = pkg.TfPictureStats()

    def run():
        cen = hp(pics[0])
        rc = lib.svt_hip_tf_picture_host(C.byref(P), C.byref(cen), refs, me, n_refs, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, C.byref(st))
        assert rc == 0
    for _ in range(max(warmup, 1)):
        run()
    n = max(steps, 3)
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    t = (time.perf_counter() - t0) / n
    up = sum(x.nbytes for pic in pics for x in pic)
    if keep is not None:
        keep.update(P=P, pics=pics, tabs=tabs, out=[x.copy() for x in out], n_refs=n_refs)
    resident = {}
    if torch is not None:  # the same picture with everything resident in HBM: svt_hip_tf_picture (what a caller chains behind the ME stage), event-timed
        d_c0 = [_dev(torch, x) for x in pics[0]]
        d_c = [x.clone() for x in d_c0]
        d_r = [torch.cat([_dev(torch, pics[1 + r][pl]).reshape(-1) for r in range(n_refs)]) for pl in range(3)]
        d_t = [torch.cat([_dev(torch, tabs[r][k]).reshape(-1).view(torch.uint8) for r in range(n_refs)]) for k in range(4)]
        D = pkg.TfDevicePictures()
        for pl in range(3):
            D.central[pl], D.refs[pl] = d_c[pl].data_ptr(), d_r[pl].data_ptr()
        D.ref_pitch, D.ref_uv_pitch = pics[0][0].size, pics[0][1].size
        M = pkg.TfMeTables(*[x.data_ptr() for x in d_t])
        ws = torch.zeros(lib.svt_hip_tf_picture_workspace(C.byref(P), n_refs), dtype=torch.uint8, device="cuda")

        def dev_run():
            for pl in range(3):
                d_c[pl].copy_(d_c0[pl])  # (the stage filters in place: every timed call starts from the unfiltered picture)
            assert lib.svt_hip_tf_picture(C.byref(P), C.byref(D), C.byref(M), n_refs, ws.data_ptr(), None, stream) == 0
        td = _time(torch, dev_run, steps, warmup, batches=3)
        same = all(np.array_equal(d_c[pl].cpu().numpy().reshape(out[pl].shape), out[pl]) for pl in range(3))
        if not same:
            raise SystemExit("bench: svt_hip_tf_picture (resident form) differs from svt_hip_tf_picture_host -- no numbers recorded")
        alg_tf = (2 + n_refs) * sum(W * H // (1 if pl == 0 else 4) for pl in range(3)) + 4 * n_refs * n_sb * 85 * 8
        resident = {"tf_picture_stage_1080p8_4refs_resident": {"us": td * 1e6, "pictures_per_s": 1 / td, "equals_host_form": True, "roofline": roof(alg_tf, td),
                                                               "note": "pictures and ME tables resident in HBM, filtered in place; includes a 3-plane device copy that resets the central picture"}}
    moved = up + sum(x.nbytes for x in out)
    return {**resident, "tf_picture_stage_1080p8_4refs_host": {"ms": t * 1e3, "pictures_per_s": 1 / t, "uploaded_MB": up / 1e6, "pcie_inclusive": True,
                                                    "roofline": {"bound": "pcie", "achieved": moved / t / 1e9, "peak": 64.0, "unit": "GB/s", "frac": moved / t / 1e9 / 64.0,
                                                                 "kernel_us": t * 1e6, "binds": "pcie", "algorithmic_bytes_per_launch": moved},
                                                    "pred_64x64": st.blocks_64x64, "pred_32x32": st.blocks_32x32, "pred_16x16": st.blocks_16x16, "pred_8x8": st.blocks_8x8,
                                                    "note": "wall time of the synchronous host-picture call (what the encoder seam pays), not a kernel time"}}
