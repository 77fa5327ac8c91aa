"""The CPU-baseline column of bench.py for the legs that had none, or a one-core figure: the reference's OWN AVX2 (and, on an AVX-512 host,
AVX-512) kernels from oracle/_ref/libsvtref.so, driven by the time-bounded loops of oracle/ref_drivers2.c on every host core, on data of the legs' shapes.

    attach(kernels, H)      H = bench.py's helpers (cpu_pool, aligned_zeros, ref_libs, cpu_has, AVX512, host_cores)

Every entry is {"value", "unit", "cores", "kind": "reference", "single_thread_value", "sample"} under `cpu_baseline` (AVX2) / `cpu_baseline_avx512`; a leg whose
reference kernel has no AVX2 form in the library (NASM-only, or not built here) says so in `cpu_baseline_note` instead of quoting a C figure as if it were SIMD.
The oracle never runs inside a timed GPU region; nothing here touches the device."""
import ctypes as C

import numpy as np

SECONDS = 2.0  # per CPU leg


def _fn(lib, name):
    return C.cast(getattr(lib, name), C.c_void_p) if hasattr(lib, name) else None


def _entry(rate, one, cores, unit, sample, scale=1.0):
    return {"value": rate * scale, "unit": unit, "cores": cores, "kind": "reference", "single_thread_value": one * scale, "sample": sample}


def _q16(vals):
    """a quantizer table as the SIMD kernels load it: 8 int16 lanes (DC, AC, AC, ...), 32-byte aligned"""
    return vals


def quantize(kernels, H, coeff, qpar, iscan):
    """svt_aom_highbd_quantize_b_avx2 (highbd_quantize_intrin_avx2.c:388) and svt_av1_highbd_quantize_fp_avx2 (av1_quantize_avx2.c) on the 32x32 coefficient blocks the
    GPU leg quantises (log_scale 1)"""
    ref, ora = H["ref_libs"]()
    k = kernels.get("quantize_b_32x32")
    if ref is None or k is None or not H["cpu_has"]("avx2"):
        return
    tabs = {}
    for name in ("zbin", "round", "quant", "quant_shift", "dequant"):
        t = H["aligned_zeros"](16, np.int16)
        t[:] = qpar[name][0][1]
        t[0] = qpar[name][0][0]
        tabs[name] = t
    ns = 256
    cores = H["host_cores"]()
    ins = [H["aligned_zeros"](ns * 1024, np.int32) for _ in range(cores)]
    outs = [H["aligned_zeros"](2 * 1024, np.int32) for _ in range(cores)]
    for b in ins:
        b[:] = coeff[:ns * 1024]
    isc = H["aligned_zeros"](1024, np.int16)
    isc[:] = iscan
    for sym, drv, key, what in (("svt_aom_highbd_quantize_b_avx2", "oracle_time_quantize_b", "cpu_baseline", "quantize_b"),
                                ("svt_av1_highbd_quantize_fp_avx2", "oracle_time_quantize_fp", "cpu_baseline_quantize_fp", "quantize_fp")):
        fp = _fn(ref, sym)
        if fp is None:
            continue
        f = getattr(ora, drv)
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double]
        run = lambda i0, st, sec: f(fp, ins[i0].ctypes.data, ns, 1024, tabs["zbin"].ctypes.data, tabs["round"].ctypes.data, tabs["quant"].ctypes.data,  # noqa: E731
                                    tabs["quant_shift"].ctypes.data, tabs["dequant"].ctypes.data, isc.ctypes.data, isc.ctypes.data, 1, outs[i0].ctypes.data, 0, 1, sec)
        rate, one, nc = H["cpu_pool"](run, SECONDS)
        k[key] = _entry(rate, one, nc, "Mblocks/s (32x32, highbd %s, log_scale 1)" % what, "%s, 256 private blocks per thread, %g s" % (sym, SECONDS), 1e-6)


def config3(kernels):
    """the fused round trip's CPU figure = the three reference kernels in sequence on one block (forward + quantize_b + inverse): 1 / (1/f + 1/q + 1/i) of the legs above"""
    k = kernels.get("config3_roundtrip")
    f, q, i = (kernels.get(n) or {} for n in ("fwd_txfm2d_32x32", "quantize_b_32x32", "inv_txfm2d_add_32x32"))
    if k is None:
        return
    for key, ik in (("cpu_baseline", "cpu_baseline_sse4_1"), ("cpu_baseline_avx512", "cpu_baseline_avx512")):
        fv, qv, iv = (f.get(key) or {}).get("value"), (q.get("cpu_baseline") or {}).get("value"), (i.get(ik) or {}).get("value")
        if fv and qv and iv:
            k[key] = {"value": 1.0 / (1.0 / fv + 1.0 / qv + 1.0 / iv), "unit": "Mblocks/s (32x32 blocks through forward + quantize_b + inverse)", "cores": f[key]["cores"],
                      "kind": "reference", "sample": "harmonic composition of the fwd_txfm2d_32x32 / quantize_b_32x32 / inv_txfm2d_add_32x32 reference legs of this run (the "
                      "reference has no fused kernel; the GPU leg covers all 19 sizes, this figure the 32x32 size)"}


def restoration(kernels, H):
    """svt_av1_highbd_wiener_convolve_add_src_avx2 (wiener_convolve_avx2.c:512) and svt_apply_selfguided_restoration_avx2 (selfguided_avx2.c:742) over a 3840x2160
    10-bit plane in the reference's 64x64 processing units; svt_av1_compute_stats_highbd_avx2 / _avx512 (pickrst_avx2.c:3036) over its 256x256 units"""
    ref, ora = H["ref_libs"]()
    if ref is None or not H["cpu_has"]("avx2"):
        return
    Wc, Hc, B, bd = 3840, 2160, 16, 10
    g = np.random.default_rng(5)
    stride = Wc + 2 * B
    buf = H["aligned_zeros"]((Hc + 2 * B) * stride, np.uint16).reshape(Hc + 2 * B, stride)
    yy, xx = np.mgrid[0:Hc + 2 * B, 0:stride]
    buf[:] = np.clip(((xx * 2 + yy * 3) % 1024) // 2 + (((xx // 8 + yy // 8) % 5) << 5) + g.integers(-16, 17, buf.shape), 0, 1023).astype(np.uint16)
    src = H["aligned_zeros"](Hc * Wc, np.uint16).reshape(Hc, Wc)
    src[:] = np.clip(buf[B:B + Hc, B:B + Wc].astype(np.int32) + g.integers(-6, 7, (Hc, Wc)), 0, 1023)
    out = H["aligned_zeros"](Hc * Wc, np.uint16).reshape(Hc, Wc)
    p0 = buf.ctypes.data + (B * stride + B) * 2
    taps = H["aligned_zeros"](16, np.int16)
    taps[:8] = [3, -7, 15, 106, 15, -7, 3, 0]
    xqd = np.array([-20, 60], np.int32)
    f = ora.oracle_time_lr_plane
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32,
                  C.c_double]
    n64 = ((Wc + 63) // 64) * ((Hc + 63) // 64)
    for leg, sym, kind in (("lr_wiener_4k10", "svt_av1_highbd_wiener_convolve_add_src_avx2", 0), ("lr_sgrproj_4k10", "svt_apply_selfguided_restoration_avx2", 1)):
        fp, k = _fn(ref, sym), kernels.get(leg)
        if fp is None or k is None:
            continue
        run = lambda i0, st, sec: f(fp, kind, p0, stride, Wc, Hc, out.ctypes.data, Wc, bd, taps.ctypes.data, taps.ctypes.data, 4, xqd.ctypes.data, i0, st, sec)  # noqa: E731
        rate, one, nc = H["cpu_pool"](run, SECONDS)
        k["cpu_baseline"] = _entry(rate / n64, one / n64, nc, "planes/s (3840x2160 10-bit)", "%s over the plane in 64x64 processing units, %g s" % (sym, SECONDS))
    w, s, m = (kernels.get(n) or {} for n in ("lr_wiener_4k10", "lr_sgrproj_4k10", "lr_mixed_4k10"))
    if m and "cpu_baseline" in w and "cpu_baseline" in s:  # a third of the units each: Wiener, self-guided, none (a copy: not counted)
        wv, sv = w["cpu_baseline"]["value"], s["cpu_baseline"]["value"]
        m["cpu_baseline"] = {"value": 1.0 / (1.0 / (3 * wv) + 1.0 / (3 * sv)), "unit": "planes/s (3840x2160 10-bit)", "cores": w["cpu_baseline"]["cores"], "kind": "reference",
                             "sample": "a third of the units through each of the two reference kernels of this run (the unit types of the GPU leg), the rest copied"}
    # the Wiener statistics (the LR search's first step; the GPU side: lr_stats.hip inside lr_search_4k10_*)
    fs = ora.oracle_time_compute_stats
    fs.restype = C.c_uint64
    fs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_double]
    units = ((Wc + 128) // 256) * ((Hc + 128) // 256)
    for sym, key in (("svt_av1_compute_stats_highbd_avx2", "cpu_baseline_compute_stats"), ("svt_av1_compute_stats_highbd_avx512", "cpu_baseline_compute_stats_avx512")):
        fp = _fn(ref, sym)
        if fp is None or (key.endswith("avx512") and not H["cpu_has"](*H["AVX512"])):
            continue
        run = lambda i0, st, sec: fs(fp, 7, p0, stride, src.ctypes.data, Wc, Wc, Hc, 256, bd, i0, st, sec)  # noqa: E731
        rate, one, nc = H["cpu_pool"](run, SECONDS)
        for leg in ("lr_search_4k10_full", "lr_search_4k10_fast"):
            if leg in kernels:
                kernels[leg][key] = _entry(rate / units, one / units, nc, "planes/s (the 7-tap Wiener statistics of every 256x256 unit: the first step of the search only)",
                                           "%s, %g s" % (sym, SECONDS))


def hme(kernels, H):
    """svt_sad_loop_kernel_avx2_intrin / _avx512_intrin (compute_sad_intrin_avx2.c:390): the searches of the three HME levels of a 1080p picture x 4 references (the
    geometry of bench_legs.hme_chain: level 0 = 2 x 2 regions of 16 x 16 positions on the 1/16 planes with 16x16 blocks, levels 1 / 2 = 8 x 3 areas on the 1/4 and
    full planes with 32x32 / 64x64 blocks)"""
    ref, ora = H["ref_libs"]()
    k = kernels.get("hme_3level_1080p_4refs")
    if ref is None or k is None or not H["cpu_has"]("avx2"):
        return
    g = np.random.default_rng(9)
    f = ora.oracle_time_sad_loop
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_double,
                  C.c_void_p]
    sbs = 30 * 17
    levels = ((480, 270, 16, 16, 16, 16, 4 * 4), (960, 540, 32, 32, 8, 3, 4 * 4), (1920, 1080, 64, 64, 8, 3, 4 * 4))  # (w, h, bw, bh, area w, area h, searches per SB: refs x regions)
    for sym, key in (("svt_sad_loop_kernel_avx2_intrin", "cpu_baseline"), ("svt_sad_loop_kernel_avx512_intrin", "cpu_baseline_avx512")):
        fp = _fn(ref, sym)
        if fp is None or (key.endswith("avx512") and not H["cpu_has"](*H["AVX512"])):
            continue
        per_picture, nc, singles = 0.0, 0, 0.0
        for (w, h, bw, bh, aw, ah, per_sb) in levels:
            pad = 64
            stride = w + 2 * pad
            plane = g.integers(0, 256, (2, h + 2 * pad, stride), dtype=np.uint8)
            items = np.zeros(sbs, dtype=[("src_off", np.uint64), ("ref_off", np.uint64)])
            for i in range(sbs):
                x, y = (i % 30) * bw, (i // 30) * bh
                x, y = min(x, w - bw), min(y, h - bh)
                items[i] = ((y + pad) * stride + x + pad, plane[0].size + (y + pad - ah // 2) * stride + x + pad - aw // 2)
            chk = np.zeros(1, np.uint64)
            run = lambda i0, st, sec: f(fp, plane.ctypes.data, stride, plane.ctypes.data, stride, items.ctypes.data, sbs, bw, bh, aw, ah, i0, st, sec, chk.ctypes.data)  # noqa: E731
            rate, one, nc = H["cpu_pool"](run, SECONDS / 2)
            per_picture += sbs * per_sb / rate
            singles += sbs * per_sb / one
        k[key] = _entry(1.0 / per_picture, 1.0 / singles, nc, "pictures/s (1080p, 4 references, three HME levels)",
                        "%s: each level's searches timed %g s, composed per picture (the SAD loops only: the rescaling between levels is not counted)" % (sym, SECONDS / 2))


def cdef_search(kernels, H):
    """the CDEF strength search of a 4K 10-bit luma plane over 64 strengths as cdef_seg_search runs it (cdef_process.c:208-300): svt_cdef_filter_fb with the RTCD
    pointers at the AVX2 kernels + svt_aom_compute_cdef_dist_16bit_avx2 (cdef_avx2.c) per (filter block, strength)"""
    ref, ora = H["ref_libs"]()
    k = kernels.get("cdef_search_4k10_64strengths")
    if ref is None or k is None or not H["cpu_has"]("avx2") or not hasattr(ref, "svt_aom_compute_cdef_dist_16bit_avx2"):
        return
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    for ptr, fnn in (("svt_aom_cdef_find_dir", "svt_aom_cdef_find_dir_avx2"), ("svt_aom_cdef_find_dir_dual", "svt_aom_cdef_find_dir_dual_avx2"),
                     ("svt_cdef_filter_block", "svt_cdef_filter_block_avx2"), ("svt_cdef_filter_block_8xn_16", "svt_cdef_filter_block_8xn_16_avx2")):
        C.c_void_p.in_dll(ref, ptr).value = C.cast(getattr(ref, fnn), C.c_void_p).value
    Wc, Hc = 3840, 2160
    g = np.random.default_rng(11)
    yy, xx = np.mgrid[0:Hc, 0:Wc]
    rec = H["aligned_zeros"](Wc * Hc, np.uint16).reshape(Hc, Wc)
    rec[:] = np.clip(((xx * 2 + yy * 3) % 1024) // 2 + (((xx // 8 + yy // 8) % 5) << 5) + g.integers(-16, 17, (Hc, Wc)), 0, 1023)
    srcp = H["aligned_zeros"](Wc * Hc, np.uint16).reshape(Hc, Wc)
    srcp[:] = np.clip(rec.astype(np.int32) + g.integers(-5, 6, rec.shape), 0, 1023)
    pri = np.array([p for p in range(16) for _ in range(4)], np.int32)
    sec = np.array([s for _ in range(16) for s in (0, 1, 2, 4)], np.int32)
    f = ora.oracle_time_cdef_search
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_double,
                  C.c_void_p]
    fb, dist = _fn(ref, "svt_cdef_filter_fb"), _fn(ref, "svt_aom_compute_cdef_dist_16bit_avx2")
    chk = np.zeros(1, np.uint64)
    run = lambda i0, st, sec_: f(fb, dist, rec.ctypes.data, srcp.ctypes.data, Wc, Wc, Hc, pri.ctypes.data, sec.ctypes.data, 64, 5, 2, i0, st, sec_, chk.ctypes.data)  # noqa: E731
    rate, one, nc = H["cpu_pool"](run, SECONDS + 1.0)
    k["cpu_baseline"] = _entry(rate * 64, one * 64, nc, "M(8x8 block x strength)/s",
                               "svt_cdef_filter_fb (AVX2 kernels) + svt_aom_compute_cdef_dist_16bit_avx2 per (64x64 filter block, strength) of the same plane shape, %g s" % (SECONDS + 1.0),
                               1e-6)


def hadamard(kernels, H):
    """svt_aom_hadamard_32x32_avx2 / 16x16 (hadamard_avx2.c); svt_aom_satd's AVX2 form lives in convolve_avx2.c, which this library does not build"""
    ref, ora = H["ref_libs"]()
    k = kernels.get("hadamard_satd_32x32")
    if ref is None or k is None or not H["cpu_has"]("avx2"):
        return
    f = ora.oracle_time_hadamard
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double]
    ns = 512
    cores = H["host_cores"]()
    ins = [H["aligned_zeros"](ns * 1024, np.int16) for _ in range(cores)]
    outs = [H["aligned_zeros"](1024, np.int32) for _ in range(cores)]
    g = np.random.default_rng(3)
    for b in ins:
        b[:] = g.integers(-255, 256, ns * 1024)
    fp = _fn(ref, "svt_aom_hadamard_32x32_avx2")
    if fp is None:
        return
    run = lambda i0, st, sec: f(fp, ins[i0].ctypes.data, ns, 32, outs[i0].ctypes.data, 0, 1, sec)  # noqa: E731
    rate, one, nc = H["cpu_pool"](run, SECONDS)
    k["cpu_baseline"] = _entry(rate, one, nc, "Mblocks/s (32x32 Hadamard)", "svt_aom_hadamard_32x32_avx2, 512 private blocks per thread, %g s (the GPU leg adds the SATD sum: "
                               "svt_aom_satd_avx2 is in convolve_avx2.c, not built into oracle/_ref)" % SECONDS, 1e-6)


def attach(kernels, H):
    for fn in (restoration, hme, cdef_search, hadamard):
        try:
            fn(kernels, H)
        except Exception as e:  # noqa: BLE001  (a CPU leg must not cost the run its GPU figures)
            kernels.setdefault("_cpu_leg_errors", {})[fn.__name__] = "%s: %s" % (type(e).__name__, e)
    config3(kernels)
