#!/usr/bin/env python3
"""bench.py -- throughput of the block-DSP hot path on MI355X (contract: see the task statement / DESIGN.md section 6).

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: launched by torch.distributed.run)

Workload at N = 1 (BASELINE.json configs[1]): batched open-loop ME integer search, 1080p 8-bit, every 64x64 SB of
FRAMES source frames against REFS reference frames each, search area 16x9 (the preset-8 1080p maximum,
Source/Lib/Codec/enc_mode_config.c:325-326), inputs resident in HBM.  One "step" = one launch of
svt_hip_me_fullpel_search_batch over the whole batch.  `value` = M(SB x search position)/s, i.e. one "block" is one
candidate position of one 64x64 SB against one reference = the 85 block SADs of SURVEY 8(d).
N > 1: every rank processes its own batch of frames (frame-level sharding, no data-path collective) -> "weak".

Extra objects on the JSON line: `roofline` (dominant kernel vs the HBM roofline, algorithmic bytes of SURVEY 8(d)),
`cpu_baseline` (the reference's own AVX2 kernels from oracle/_ref timed on the host cores, bounded sample) and
`kernels` (the other primitives of the metric, each with its own algorithmic-bytes roofline).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
QSAD_PEAK = 1024 * 2.4e9 / 22.4 * 64 * 16  # |a-b| per second if every SIMD issued nothing but v_qsad_pk_u16_u8 (16 per lane)
W, H, PAD = 1920, 1080, 68  # luma plane padded 68 px each side (enc_handle.c:4084) -> stride 2056
STRIDE, ROWS = W + 2 * PAD, H + 2 * PAD
PLANE = STRIDE * ROWS


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_pmc_traffic.json, written by
    tools/pmc_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over THIS command's default workload).
    bench.py cannot read PMC counters itself; None when no summary is committed or the workload flags differ from the profiled ones."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    k = json.load(open(files[-1])).get("kernels", {}).get(kernel)
    return None if k is None else {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "read": k["hbm_read_bytes_per_launch"],
                                   "write": k["hbm_write_bytes_per_launch"], "source": os.path.relpath(files[-1], ROOT)}


def synth_planes(n, seed):
    """(x + y) & 255 gradient moving (2i, 3i) per frame + uniform +-8 noise (BASELINE.md section 2 generator)."""
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:ROWS, 0:STRIDE].astype(np.int32)
    out = np.empty((n, ROWS, STRIDE), np.uint8)
    for i in range(n):
        base = (xx + 2 * i + yy + 3 * i) & 255
        out[i] = np.clip(base + g.integers(-8, 9, base.shape), 0, 255).astype(np.uint8)
    return out


def time_steps(torch, fn, steps, warmup, dist=None):
    for _ in range(warmup):
        fn()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    return wall, ev0.elapsed_time(ev1) / 1e3  # seconds: host wall, device span on the launch stream


def cpu_me_baseline(descs, planes_src, planes_ref, area, budget_s=12.0):
    """Reference AVX2 kernels (svt_ext_all_sad_calculation_8x8_16x16_avx2 + svt_ext_eight_sad_calculation_32x32_64x64_avx2
    + the _c remainder kernels) driven like open_loop_me_fullpel_search_sblock, all host cores, bounded sample."""
    import concurrent.futures as cf
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libsvtref.so")
    ora_path = os.path.join(ROOT, "oracle", "liboracle.so")
    flags = open("/proc/cpuinfo").read()
    if not os.path.exists(ref_path):
        # the real reference is not available on this box: time our C restatement instead ("port")
        oracle, kind, label = C.CDLL(ora_path), "port", "oracle_me_fullpel_search (scalar C restatement)"
        cores = 1

        def one(i):
            d = descs[i]
            bs, bm = np.zeros(85, np.uint32), np.zeros(85, np.uint32)
            oracle.oracle_me_fullpel_search(C.c_void_p(planes_src.ctypes.data + int(d["src_off"])), int(d["src_stride"]),
                                            C.c_void_p(planes_ref.ctypes.data + int(d["ref_off"])), int(d["ref_stride"]), 0, 0,
                                            area[0], area[1], 0, C.c_void_p(bs.ctypes.data), C.c_void_p(bm.ctypes.data))
        t0, done = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s and done < len(descs):
            one(done)
            done += 1
        dt = time.perf_counter() - t0
        return {"value": done * area[0] * area[1] / dt / 1e6, "unit": "Mblocks/s", "cores": cores, "kind": kind,
                "sample": "%d SB-refs at %dx%d, %s" % (done, area[0], area[1], label)}
    ref, oracle = C.CDLL(ref_path), C.CDLL(ora_path)
    simd = "avx2" if " avx2 " in flags else "c"
    fn = lambda n: C.cast(getattr(ref, n), C.c_void_p)  # noqa: E731
    f_all = fn("svt_ext_all_sad_calculation_8x8_16x16_" + simd)
    f_eight = fn("svt_ext_eight_sad_calculation_32x32_64x64_" + simd)
    f_one, f_one2 = fn("svt_ext_sad_calculation_8x8_16x16_c"), fn("svt_ext_sad_calculation_32x32_64x64_c")
    drv = oracle.oracle_drive_ref_me_search_timed
    drv.restype = C.c_uint64
    drv.argtypes = [C.c_void_p] * 7 + [C.c_uint32] * 3 + [C.c_double, C.c_int]
    cores = len(os.sched_getaffinity(0))
    try:  # honour a cgroup CPU quota if the box has one
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    dd = np.ascontiguousarray(descs)

    def run(idx0, step, seconds):
        return drv(f_all, f_eight, f_one, f_one2, planes_src.ctypes.data, planes_ref.ctypes.data, dd.ctypes.data, len(dd), idx0, step, seconds, 0)
    t1 = time.perf_counter()
    n1 = run(0, 1, 2.0)  # single thread figure
    one_thread = n1 * area[0] * area[1] / (time.perf_counter() - t1) / 1e6
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL for the whole C loop
        done = sum(ex.map(lambda k: run(k, cores, budget_s), range(cores)))
    dt = time.perf_counter() - t0
    return {"value": done * area[0] * area[1] / dt / 1e6, "unit": "Mblocks/s", "cores": cores, "kind": "reference",
            "single_thread_value": one_thread,
            "sample": "%d SB-refs at %dx%d in %.1fs, reference %s kernels driven as open_loop_me_fullpel_search_sblock" % (done, area[0], area[1], dt, simd)}


def aligned_zeros(n, dtype, al=64):
    raw = np.zeros(n * np.dtype(dtype).itemsize + al, np.uint8)
    off = (-raw.ctypes.data) % al
    return raw[off:off + n * np.dtype(dtype).itemsize].view(dtype)


def host_cores():
    cores = len(os.sched_getaffinity(0))
    try:  # honour a cgroup CPU quota if the box has one
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    return cores


def cpu_pool(run, seconds):
    """run(idx0, step, seconds) -> units done, on every host core (ctypes drops the GIL inside the C loop)."""
    import concurrent.futures as cf
    cores = host_cores()
    t1 = time.perf_counter()
    n1 = run(0, 1, min(2.0, seconds))
    one = n1 / (time.perf_counter() - t1)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(lambda k: run(k, cores, seconds), range(cores)))
    return done / (time.perf_counter() - t0), one, cores


def ref_libs():
    ref_path, ora_path = os.path.join(ROOT, "oracle", "_ref", "libsvtref.so"), os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(ref_path) and os.path.exists(ora_path)):
        return None, None
    return C.CDLL(ref_path), C.CDLL(ora_path)


def bench_sad_pairs(torch, lib, pkg, stream, a):
    """config 1 on the GPU: 64x64 SAD of co-located SB pairs, 120 distinct planes (300 MB > the 256 MB Infinity Cache) so the
    byte rate is an HBM rate: algorithmic bytes = 2 * 64 * 64 per block."""
    nplanes = 120
    planes = torch.randint(0, 256, (nplanes * PLANE,), dtype=torch.uint8, device="cuda")
    pairs = np.zeros((nplanes - 1) * 510, dtype=pkg.SadPair)
    i = 0
    for f in range(nplanes - 1):
        for sy in range(17):
            for sx in range(30):
                o = (PAD + sy * 64) * STRIDE + PAD + sx * 64
                pairs[i] = (f * PLANE + o, (f + 1) * PLANE + o + 3 + 2 * STRIDE, STRIDE, STRIDE)
                i += 1
    d_pairs = torch.from_numpy(pairs.view(np.uint8)).cuda()
    d_out = torch.zeros(len(pairs), dtype=torch.int32, device="cuda")
    fn = lambda: lib.svt_hip_sad_nxm_batch(planes.data_ptr(), planes.data_ptr(), d_pairs.data_ptr(), len(pairs), 64, 64, d_out.data_ptr(), stream)  # noqa: E731
    _, dv = time_steps(torch, fn, a.steps, a.warmup)
    gbs = len(pairs) * 8192 / (dv / a.steps) / 1e9
    return {"value": len(pairs) / (dv / a.steps) / 1e6, "unit": "Mblocks/s (64x64 pairs)", "kernel": "sad_nxm_kernel", "footprint_MB": nplanes * PLANE / 1e6,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": (pmc_traffic("sad_nxm_kernel") or {}).get("hbm_bytes_per_launch"), "traffic_detail": pmc_traffic("sad_nxm_kernel"),
                         "algorithmic_bytes_per_launch": len(pairs) * 8192, "algorithmic_bytes_per_block": 8192}}


def bench_fwd_txfm(torch, lib, pkg, stream, a, cpu):
    """config 3 slice named by the metric: 32x32 forward transform, 10-bit residuals, DCT_DCT; 2 B/px in + 4 B/px out = 6144 B/block."""
    n, ts = 65536, 3
    g = np.random.default_rng(13596)
    res = g.integers(-1023, 1024, n * 1024).astype(np.int16)
    descs = np.zeros(n, dtype=pkg.FwdTxfmDesc)
    descs["in_off"] = np.arange(n, dtype=np.uint64) * 1024
    descs["in_stride"] = 32
    d_res, d_desc = torch.from_numpy(res).cuda(), torch.from_numpy(descs.view(np.uint8)).cuda()
    d_out = torch.zeros(n * 1024, dtype=torch.int32, device="cuda")
    fn = lambda: lib.svt_hip_fwd_txfm2d_batch(d_res.data_ptr(), d_desc.data_ptr(), n, ts, 10, 0, d_out.data_ptr(), stream)  # noqa: E731
    _, dv = time_steps(torch, fn, a.steps, a.warmup)
    gbs = n * 6144 / (dv / a.steps) / 1e9
    out = {"value": n / (dv / a.steps) / 1e6, "unit": "Mblocks/s (32x32)", "kernel": "fwd_txfm2d_kernel<32,32>", "blocks_per_step": n,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "traffic": (pmc_traffic("fwd_txfm2d_kernel<32, 32>") or {}).get("hbm_bytes_per_launch"),
                        "traffic_detail": pmc_traffic("fwd_txfm2d_kernel<32, 32>"), "algorithmic_bytes_per_launch": n * 6144,
                        "algorithmic_bytes_per_block": 6144, "note": "butterfly network: VALU/int32-multiply bound, not a dense contraction (DESIGN.md 4.2)"}}
    if cpu:
        ref, oracle = ref_libs()
        if ref is not None and " avx2 " in open("/proc/cpuinfo").read():
            f = oracle.oracle_time_fwd_txfm
            f.restype = C.c_uint64
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_uint8, C.c_uint8, C.c_uint32, C.c_uint32, C.c_double]
            fnp = C.cast(ref.svt_av1_fwd_txfm2d_32x32_avx2, C.c_void_p)
            ns = 512  # private input / output per thread (shared buffers serialise the cores through the cache hierarchy)
            bufs = [(aligned_zeros(ns * 1024, np.int16), aligned_zeros(2048, np.int32)) for _ in range(host_cores())]
            for b_in, _ in bufs:
                b_in[:] = res[:ns * 1024]
            run = lambda i0, st, sec: f(fnp, bufs[i0][0].ctypes.data, ns, 32, 32, bufs[i0][1].ctypes.data, 0, 10, 0, 1, sec)  # noqa: E731
            rate, one, cores = cpu_pool(run, 4.0)
            out["cpu_baseline"] = {"value": rate / 1e6, "unit": "Mblocks/s (32x32)", "cores": cores, "kind": "reference", "single_thread_value": one / 1e6,
                                   "sample": "svt_av1_fwd_txfm2d_32x32_avx2, 512 private blocks per thread, 4 s per leg"}
    return out


def bench_cdef(torch, lib, pkg, stream, a, cpu):
    """config 4: CDEF over a 4K 10-bit luma plane: strength search (all 64 luma strengths) and apply (pri 4, sec 2)."""
    Wc, Hc, bd = 3840, 2160, 10
    g = np.random.default_rng(4)
    yy, xx = np.mgrid[0:Hc, 0:Wc]
    plane = np.clip(((xx * 2 + yy * 3) % 1024) // 2 + (((xx // 8 + yy // 8) % 5) << 5) + g.integers(-16, 17, (Hc, Wc)), 0, 1023).astype(np.uint16)
    src = np.clip(plane.astype(np.int32) + g.integers(-6, 7, plane.shape), 0, 1023).astype(np.uint16)
    nhfb, nvfb = Wc // 64, (Hc + 63) // 64
    nfb = nhfb * nvfb
    skip = np.zeros((nvfb * 8, nhfb * 8), np.uint8)
    cands = [(pr, sc) for pr in range(16) for sc in (0, 1, 2, 4)]
    pri, sec = np.array([c[0] for c in cands], np.int32), np.array([c[1] for c in cands], np.int32)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).reshape(-1)).cuda()  # noqa: E731
    d_pl, d_src, d_out, d_skip, d_pri, d_sec = t(plane), t(src), t(plane), t(skip), t(pri), t(sec)
    d_dir, d_var = torch.zeros(nfb * 64, dtype=torch.uint8, device="cuda"), torch.zeros(nfb * 64, dtype=torch.int32, device="cuda")
    d_mse = torch.zeros(nfb * 64, dtype=torch.int64, device="cuda")
    apri, asec = t(np.full(nfb, 4, np.int32)), t(np.full(nfb, 2, np.int32))

    def params(mode):
        return pkg.CdefParams(d_pl.data_ptr(), d_src.data_ptr(), d_out.data_ptr(), Wc, Wc, Wc, Wc, Hc, 0, 0, 0, 1, bd - 8, 4, 4, 1, 64 if mode else 0,
                              d_skip.data_ptr(), (d_pri if mode else apri).data_ptr(), (d_sec if mode else asec).data_ptr(), d_dir.data_ptr(), d_var.data_ptr(),
                              d_mse.data_ptr())
    out = {}
    n8 = (Wc // 8) * (Hc // 8)
    for mode, name in ((1, "cdef_search_4k10_64strengths"), (0, "cdef_apply_4k10")):
        P = params(mode)
        fn = lambda: lib.svt_hip_cdef_frame(mode, C.byref(P), stream)  # noqa: E731
        st = max(3, a.steps // 4)
        _, dv = time_steps(torch, fn, st, 1)
        per = dv / st
        units = n8 * (64 if mode else 1)
        bytes_alg = Wc * Hc * 2 * (2 if mode == 0 else 2) + (nfb * 64 * 8 if mode else 0)  # apply: read + write; search: recon + source, 8 B per (fb, strength)
        gbs = bytes_alg / per / 1e9
        out[name] = {"value": units / per / 1e6, "unit": "M(8x8 block x strength)/s" if mode else "M(8x8 blocks)/s", "frames_per_s": 1 / per, "kernel": "cdef_frame_kernel",
                     "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                                  "algorithmic_bytes_per_frame": bytes_alg}}
    if cpu:
        ref, oracle = ref_libs()
        if ref is not None and " avx2 " in open("/proc/cpuinfo").read():
            ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
            ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
            for ptr, fnn in (("svt_aom_cdef_find_dir", "svt_aom_cdef_find_dir_avx2"), ("svt_aom_cdef_find_dir_dual", "svt_aom_cdef_find_dir_dual_avx2"),
                             ("svt_cdef_filter_block", "svt_cdef_filter_block_avx2"),
                             ("svt_cdef_filter_block_8xn_16", "svt_cdef_filter_block_8xn_16_avx2")):  # SIMD-internal pointer, NULL in a C-only setup
                C.c_void_p.in_dll(ref, ptr).value = C.cast(getattr(ref, fnn), C.c_void_p).value  # what RTCD would select with AVX2 detected
            f = oracle.oracle_time_cdef_apply
            f.restype = C.c_uint64
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_double]
            fb = C.cast(ref.svt_cdef_filter_fb, C.c_void_p)
            pl = aligned_zeros(Wc * Hc, np.uint16).reshape(Hc, Wc)  # the AVX2 kernels use aligned loads / stores
            pl[:] = plane
            cpu_out = aligned_zeros(Wc * Hc, np.uint16).reshape(Hc, Wc)
            run = lambda i0, stp, s: f(fb, pl.ctypes.data, Wc, Wc, Hc, cpu_out.ctypes.data, 4, 2, 4, 2, i0, stp, s)  # noqa: E731
            rate, one, cores = cpu_pool(run, 4.0)
            out["cdef_apply_4k10"]["cpu_baseline"] = {"value": rate * 64 / 1e6, "unit": "M(8x8 blocks)/s", "cores": cores, "kind": "reference",
                                                      "single_thread_value": one * 64 / 1e6,
                                                      "sample": "svt_cdef_filter_fb + svt_cdef_filter_block_avx2 / find_dir_dual_avx2 over the same 4K plane, 4 s per leg"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=32, help="source frames per step and per GPU")
    ap.add_argument("--refs", type=int, default=4)
    ap.add_argument("--area", type=str, default="16x9")
    ap.add_argument("--probe", action="store_true", help="print VALU issue rates and exit")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--extra", action="store_true", help="also sweep the other search areas / sub_sad (reported under kernels)")
    a = ap.parse_args()

    import torch
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if a.gpus > 1 or world > 1:
        import torch.distributed as dist  # RCCL; used for the barrier / max-over-ranks only: the path needs no exchange
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    pkg = entry._pkg()
    lib = pkg.load(init_device=local)
    stream = torch.cuda.current_stream().cuda_stream
    aw, ah = (int(v) for v in a.area.split("x"))

    if a.probe:
        sink = torch.zeros(4, dtype=torch.int32, device="cuda")
        names = ["v_sad_u8", "v_qsad_pk_u16_u8", "v_add+v_xor", "v_mul_lo_u32(+add)", "v_mad_i64_i32", "v_alignbyte_b32"]
        blocks, iters = 256 * 8, 4096
        for k, nm in enumerate(names):
            fn = lambda: lib.svt_hip_rate_probe(k, iters, blocks, sink.data_ptr(), stream)  # noqa: E731
            _, dev = time_steps(torch, fn, 5, 2)
            ops = 5 * blocks * 256 * iters * 8
            print("%-22s %8.1f Gop/s/lane-op  -> %.2f cycles per wave64 instruction per SIMD (2.4 GHz, 1024 SIMDs)" %
                  (nm, ops / dev / 1e9, 2.4e9 * 1024 * 64 / (ops / dev)))
        return

    # ---------------- config 2: batched ME full-pel search -------------------------------------------------------
    nplanes = a.frames + a.refs
    planes = synth_planes(nplanes, 1234 + rank)
    descs = np.concatenate([pkg.me_descs_for_frame(W, H, STRIDE, PAD, PAD, aw, ah, PLANE, n_refs=a.refs, src_plane=f, ref_plane0=f + 1)
                            for f in range(a.frames)])
    n = len(descs)
    d_planes = torch.from_numpy(planes.reshape(-1)).cuda()
    d_descs = torch.from_numpy(descs.view(np.uint8)).cuda()
    d_sad = torch.zeros(n * 85, dtype=torch.int32, device="cuda")
    d_mv = torch.zeros(n * 85, dtype=torch.int32, device="cuda")
    ws_bytes = lib.svt_hip_me_fullpel_search_workspace(n, aw, ah)
    d_ws = torch.zeros(max(ws_bytes, 8), dtype=torch.uint8, device="cuda")

    def step(sub=0, w=aw, h=ah, dd=d_descs, nn=n):
        lib.svt_hip_me_fullpel_search_batch(d_planes.data_ptr(), d_planes.data_ptr(), dd.data_ptr(), nn, w, h, sub, d_sad.data_ptr(),
                                            d_mv.data_ptr(), d_ws.data_ptr() if ws_bytes else None, stream)
    wall, dev = time_steps(torch, step, a.steps, a.warmup, dist)
    t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    positions = n * aw * ah
    value = world * positions * a.steps / wall / 1e6
    # roofline of the dominant kernel: algorithmic bytes per (SB, ref) = 64*64 + (64+W-1)(64+H-1) + 85*8
    bytes_item = 64 * 64 + (64 + aw - 1) * (64 + ah - 1) + 85 * 8
    kernel_s = dev / a.steps
    achieved = n * bytes_item / kernel_s / 1e9
    default_workload = (a.frames, a.refs, a.area) == (32, 4, "16x9")  # the workload the committed PMC passes were run on
    me_traffic = (pmc_traffic("me_fullpel_kernel<false>") or {}) if default_workload else {}
    out = {
        "metric": "Mblocks/s per kernel (SAD, FwdTxfm2d, CDEF) + encoder fps @1080p preset 8", "value": value,
        "unit": "Mblocks/s (block = one search position of one 64x64 SB vs one reference = 85 block SADs)",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[1]: batched open-loop ME integer full-pel search (SAD 8x8..64x64), 1080p 8-bit, all 64x64 SBs",
                   "frames_per_step_per_gpu": a.frames, "refs": a.refs, "search_area": a.area, "sb_refs_per_step_per_gpu": n,
                   "sub_sad": 0, "parallelism": "frame-sharded x%d (no collective)" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": me_traffic.get("hbm_bytes_per_launch"), "traffic_detail": me_traffic or None,
                     "algorithmic_bytes_per_launch": n * bytes_item, "kernel": "me_fullpel_kernel<false>", "kernel_ms": kernel_s * 1e3,
                     "algorithmic_bytes_per_sb_ref": bytes_item,
                     "note": "search is VALU(packed-SAD)-bound, see sad_ops; HBM figure = SURVEY 8(d) algorithmic bytes / time",
                     "sad_ops_per_s": n * aw * ah * 4096 / kernel_s,
                     # measured v_qsad_pk_u16_u8 issue cost: 22.4 cycles per wave64 instruction per SIMD (profiles/r01_call1_valu_issue_rates.txt)
                     "valu_peak_sad_ops_per_s": QSAD_PEAK, "valu_frac": n * aw * ah * 4096 / kernel_s / QSAD_PEAK},
    }
    kernels = {}
    kernels["sad64x64_pairs"] = bench_sad_pairs(torch, lib, pkg, stream, a)
    kernels["fwd_txfm2d_32x32"] = bench_fwd_txfm(torch, lib, pkg, stream, a, cpu=(rank == 0 and world == 1 and not a.no_cpu))
    kernels.update(bench_cdef(torch, lib, pkg, stream, a, cpu=(rank == 0 and world == 1 and not a.no_cpu)))
    if a.extra:
        for (w2, h2, sub) in [(16, 9, 1), (64, 32, 0), (256, 256, 0)]:
            nf = a.frames if w2 < 64 else (4 if w2 < 256 else 1)
            dd = np.concatenate([pkg.me_descs_for_frame(W, H, STRIDE, PAD, PAD, w2, h2, PLANE, n_refs=(a.refs if w2 < 256 else 1), src_plane=f, ref_plane0=f + 1)
                                 for f in range(nf)])
            # search windows must stay inside the padded planes: clamp like integer_search_b64 does (motion_estimation.c:1440-1506)
            keep = []
            for k, d in enumerate(dd):
                y0 = (int(d["ref_off"]) % PLANE) // STRIDE
                x0 = (int(d["ref_off"]) % PLANE) % STRIDE
                if int(d["ref_off"]) % PLANE >= 0 and x0 + 64 + w2 - 1 <= STRIDE and y0 + 64 + h2 - 1 <= ROWS and d["ref_off"] >= 0 and (int(d["ref_off"]) // PLANE) == (int(d["ref_off"]) + (64 + h2 - 2) * STRIDE + 64 + w2 - 2) // PLANE:
                    keep.append(k)
            dd = dd[keep]
            tdd = torch.from_numpy(dd.view(np.uint8)).cuda()
            wsb = lib.svt_hip_me_fullpel_search_workspace(len(dd), w2, h2)
            ws2 = torch.zeros(max(wsb, 8), dtype=torch.uint8, device="cuda")
            f2 = lambda: lib.svt_hip_me_fullpel_search_batch(d_planes.data_ptr(), d_planes.data_ptr(), tdd.data_ptr(), len(dd), w2, h2, sub,  # noqa: E731
                                                             d_sad.data_ptr(), d_mv.data_ptr(), ws2.data_ptr() if wsb else None, stream)
            st = a.steps if w2 < 64 else 2
            _, dv = time_steps(torch, f2, st, 1)
            kernels["me_search_%dx%d_sub%d" % (w2, h2, sub)] = {"value": len(dd) * w2 * h2 / (dv / st) / 1e6, "unit": "Mblocks/s",
                                                               "sb_refs": len(dd), "sad_ops_per_s": len(dd) * w2 * h2 * (2048 if sub else 4096) / (dv / st)}
        import bench_legs
        kernels.update(bench_legs.hme_sad_loop(torch, lib, pkg, stream, a.steps, a.warmup))
        kernels.update(bench_legs.lr_frames(torch, lib, pkg, stream, max(a.steps // 2, 2), 2))
        kernels.update(bench_legs.picprep(torch, lib, pkg, stream, max(a.steps // 2, 2), 1))
        kernels.update(bench_legs.deblock(torch, lib, pkg, stream, max(a.steps // 2, 2), 1))
        kernels.update(bench_legs.lr_stats(torch, lib, pkg, stream, max(a.steps // 2, 2), 1))
        kernels.update(bench_legs.cdef_chain(torch, lib, pkg, stream, max(a.steps // 4, 2), 1))
        kernels.update(bench_legs.me_session(torch, lib, pkg, stream, a.steps, 1))
        kernels.update(bench_legs.me_results(torch, lib, pkg, stream, a.steps, 1))
        kernels.update(bench_legs.hme_chain(torch, lib, pkg, stream, a.steps, 1))
        kernels.update(bench_legs.me_stage(torch, lib, pkg, stream, a.steps, 1))
        kernels.update(bench_legs.me_session_stage(torch, lib, pkg, stream, a.steps, 1))
        kernels.update(bench_legs.tf_frames(torch, lib, pkg, stream, a.steps, 1))
        kernels["txfm_quant_roundtrip"] = bench_legs.txfm_roundtrip(torch, lib, pkg, stream, max(a.steps // 4, 3), 1)
    out["kernels"] = kernels
    if rank == 0 and world == 1 and not a.no_cpu:
        host_descs = pkg.me_descs_for_frame(W, H, STRIDE, PAD, PAD, aw, ah, PLANE, n_refs=1, src_plane=0, ref_plane0=1)
        out["cpu_baseline"] = cpu_me_baseline(host_descs, planes, planes, (aw, ah))
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
