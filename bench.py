#!/usr/bin/env python3
"""bench.py -- throughput of the block-DSP hot path on MI355X (contract: see the task statement / DESIGN.md section 6).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode frames|strips]

`--gpus N` with no RANK / WORLD_SIZE in the environment launches the N ranks itself (one process per GPU, RCCL rendezvous on 127.0.0.1);
under torch.distributed.run it uses the ranks it is given.  Rank 0 prints ONE JSON line.

Workload at N = 1 (BASELINE.json configs[1]): batched open-loop ME integer search, 1080p 8-bit, every 64x64 SB of FRAMES source frames
against REFS reference frames each, search area 16x9 (the preset-8 1080p maximum, Source/Lib/Codec/enc_mode_config.c:325-326; at the default
CRF 35 the reference's QP modulation shrinks it to 8x4, see pkg.m8_me_settings), inputs resident in HBM.  One "step" = one launch of
svt_hip_me_fullpel_search_batch over the whole batch.  `value` = M(SB x search position)/s, i.e. one "block" is one candidate position of one
64x64 SB against one reference = the 85 block SADs of SURVEY 8(d).

Multi-GPU, both measured in the same run when N > 1:
  * --mode frames (default, `value`): every rank searches its own batch of frames -- the reference's picture-level parallelism, no data-path
    collective, "weak" scaling;
  * frame partition (`frame_partition` object; `value` with --mode strips): ONE picture per step cut into contiguous SB-row strips
    (1080p: 17 rows -> 3,2,2,2,2,2,2,2 at N = 8), reference planes resident on every GPU (broadcast once, outside the timed region), every rank
    searches its strip against all references, the per-strip 85 x (SAD, MV) tables are all-gathered over RCCL -- "strong" scaling.

Before a leg is timed its output is compared with the CPU checker on a slice (oracle/ = test infrastructure, used here only as the checker
and for the `cpu_baseline` legs); a mismatch aborts the run.  Extra objects on the JSON line: `roofline` (dominant kernel, algorithmic bytes of
SURVEY 8(d) / event-timed kernel duration, against the 8 TB/s HBM peak and against the 6.3 TB/s copy ceiling of MI355X_MICROARCH.md),
`cpu_baseline` (the reference's own AVX2 kernels from oracle/_ref on the host cores, bounded sample) and `kernels` (the other primitives of
the metric and the stages around them, each with its own roofline), and `encoder_fps_1080p_preset8` (the metric's second half: the reference encoder built
C-only under oracle/_ref/enc, encoding one clip with SVT_HIP unset and then with the ME / CDEF / LR stage seams on this GPU -- identical bitstream required).
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
import bench_regions as regions  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBS = 6300.0  # measured copy ceiling (MI355X_MICROARCH.md:35,293): what a kernel that streams every byte once can reach
QSAD_PEAK = 1024 * 2.4e9 / 22.4 * 64 * 16  # |a-b| per second if every SIMD issued nothing but v_qsad_pk_u16_u8 (16 per lane)
W, H, PAD = 1920, 1080, 68  # luma plane padded 68 px each side (enc_handle.c:4084) -> stride 2056
STRIDE, ROWS = W + 2 * PAD, H + 2 * PAD
PLANE = STRIDE * ROWS


LIVE_PMC = None  # {kernel: {...}} collected by live_pmc() in THIS run; the committed summary is only the fallback


def _short_kernel(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES"))


def live_pmc(a):
    """Counters of THIS run, by timed region: three child runs of this script's GPU legs (`--pmc-child`: no CPU legs, no parity checks; every timer runs its function
    bench_regions.CALLS times between two marker launches) under `rocprofv3 --kernel-trace --pmc <counters>` -- FETCH_SIZE and WRITE_SIZE in separate passes (the two do
    not fit one: MI355X_MICROARCH.md; KiB units, FETCH_SIZE doubled = the guide's gfx950 correction) and one pass of SQ counters for the VALU fractions.  Returns
    {"regions": {tag: {"calls", "kernels": {kernel: {"launches", counter: total}}}}, "probe": {...}} or None when rocprofv3 is missing or a pass fails (then the traffic
    fields stay null)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    t0 = time.perf_counter()
    merged, probe, mem_known = {}, {}, {}
    for ctrs in PMC_PASSES:
        td = tempfile.mkdtemp(prefix="svt_pmc_", dir="/tmp")
        rfile = os.path.join(td, "regions.json")
        cmd = [rp, "--kernel-trace", "--pmc"] + list(ctrs) + ["--output-format", "csv", "-d", td, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
               "--frames", str(a.frames), "--refs", str(a.refs), "--area", a.area] + (["--legs", a.legs] if a.legs else []) + (["--only-me"] if a.only_me else [])
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", SVT_PMC_REGIONS_FILE=rfile), timeout=420, capture_output=True)
            files = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files or not os.path.exists(rfile):
                sys.stderr.write("bench.py: counter pass %s failed (rc %d)\n%s\n" % (ctrs, r.returncode, r.stderr.decode(errors="replace")[-800:]))
                return None
            meta = json.load(open(rfile))
            got = regions.parse_counter_csv(files[0], meta["tags"], set(ctrs))
            for tag, g in got.items():
                m = merged.setdefault(tag, {"calls": meta["calls"], "kernels": {}})
                for kn, kv in g["kernels"].items():
                    m["kernels"].setdefault(kn, {}).update(kv)
            for tag, nbytes in (meta.get("mem_probe") or {}).items():  # the known byte counts of the calibration regions (identical in every pass)
                mem_known[tag] = nbytes
            if "SQ_ACTIVE_INST_VALU" in ctrs and "valu_probe#0" in got and meta.get("probe", {}).get("seconds"):
                tot = {}
                for kv in got["valu_probe#0"]["kernels"].values():
                    for c, v in kv.items():
                        tot[c] = tot.get(c, 0.0) + v
                probe = dict(meta["probe"], counters=tot, active_per_s=tot.get("SQ_ACTIVE_INST_VALU", 0.0) / meta["calls"] / meta["probe"]["seconds"],
                             wave_insts_measured=tot.get("SQ_INSTS_VALU", 0.0) / meta["calls"])
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("bench.py: counter pass %s: %r\n" % (ctrs, e))
            return None
        finally:
            shutil.rmtree(td, ignore_errors=True)
    return {"regions": merged, "probe": probe, "traffic_calibration": traffic_calibration(merged, mem_known), "_seconds": time.perf_counter() - t0}


def traffic_calibration(merged, mem_known):
    """bytes per KiB-unit of FETCH_SIZE / WRITE_SIZE for every access shape of bench_regions.MEM_SHAPES: known bytes of the calibration region / what the counter reported for
    it.  {"read": {shape: factor}, "write": {shape: factor}}; a factor of 2048 = "the counter reports half the bytes" (MI355X_MICROARCH.md, wide streaming reads)."""
    cal = {"read": {}, "write": {}}
    for tag, nbytes in mem_known.items():
        g = merged.get(tag)
        if not g:
            continue
        kind, shape = ("write", tag.split("#")[0][len("mem_w_"):]) if tag.startswith("mem_w_") else ("read", tag.split("#")[0][len("mem_r_"):])
        ctr = "WRITE_SIZE" if kind == "write" else "FETCH_SIZE"
        tot = sum(kv.get(ctr, 0.0) for kv in g["kernels"].values()) / g.get("calls", regions.CALLS)
        if tot > 0:
            cal[kind][shape] = nbytes / tot
    return cal


def pmc_traffic(kernel):
    """(kept for the legs' constructors: the traffic of a leg is attached by region in finish_rooflines once every leg has run)"""
    return None


def synth_planes(n, seed):
    """(x + y) & 255 gradient moving (2i, 3i) per frame + uniform +-8 noise (BASELINE.md section 2 generator)."""
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:ROWS, 0:STRIDE].astype(np.int32)
    out = np.empty((n, ROWS, STRIDE), np.uint8)
    for i in range(n):
        base = (xx + 2 * i + yy + 3 * i) & 255
        out[i] = np.clip(base + g.integers(-8, 9, base.shape), 0, 255).astype(np.uint8)
    return out


MIN_TIMED_S = 0.3  # every leg is timed over at least this much device time (--steps 20 used to time 9 ms, before the clocks had ramped)


def calibrate_launches(torch, fn, steps, min_s, dist=None):
    """How many back-to-back launches make one `step` so that `steps` steps last >= min_s: one untimed launch, three event-timed ones, the MAX over ranks."""
    import math
    fn()
    torch.cuda.synchronize()
    t = 1.0
    for reps in (3, 12):  # the second, longer probe runs at ramped clocks
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = max(e0.elapsed_time(e1) / (reps * 1e3), 1e-7)
        if t * reps > 0.02 or min_s <= 0:
            break
    L = max(1, int(math.ceil(1.1 * min_s / (steps * t)))) if min_s > 0 else 1
    if dist is not None:
        tl = torch.tensor([L], dtype=torch.int64, device="cuda")
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        L = int(tl.item())
    return L


def time_steps(torch, fn, steps, warmup, dist=None):
    idx = regions.open_region()
    if regions.PMC_CHILD:
        t = regions.pmc_run(fn, idx)
        return t * steps, t * steps
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


def time_leg(torch, fn, min_s=None, batches=5):
    """Seconds per launch of a per-kernel leg: launches back to back, `batches` event-timed batches of >= min_s / batches each after a warm-up of the same
    length, the median batch reported (HIP events on the launch stream = torch's current stream).  Returns (seconds per launch, launches per batch)."""
    import math
    idx = regions.open_region()
    if regions.PMC_CHILD:
        return regions.pmc_run(fn, idx), regions.CALLS
    min_s = MIN_TIMED_S if min_s is None else min_s
    reps = calibrate_launches(torch, fn, batches, min_s)
    reps = max(reps, 3)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 1e3 / reps)
    return sorted(ts)[len(ts) // 2], reps


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


def cpu_has(*flags):
    """x86 features of the host as /proc/cpuinfo lists them (the gate the reference's RTCD applies through cpuinfo: common_dsp_rtcd.c:97-140)"""
    try:
        fl = set(next(ln for ln in open("/proc/cpuinfo") if ln.startswith("flags")).split(":", 1)[1].split())
    except Exception:  # noqa: BLE001
        return False
    return all(f in fl for f in flags)


AVX512 = ("avx512f", "avx512bw", "avx512dq", "avx512vl", "avx512cd")  # HAS_AVX512F of the reference = this set (common_dsp_rtcd.c:123-129)


def CPU_HELPERS():
    """what bench_cpu.py needs of this module"""
    return {"cpu_pool": cpu_pool, "aligned_zeros": aligned_zeros, "ref_libs": ref_libs, "cpu_has": cpu_has, "AVX512": AVX512, "host_cores": host_cores}


def ref_libs():
    ref_path, ora_path = os.path.join(ROOT, "oracle", "_ref", "libsvtref.so"), os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(ref_path) and os.path.exists(ora_path)):
        return None, None
    return C.CDLL(ref_path), C.CDLL(ora_path)


def cpu_tf_subpel(k, budget_s):
    """Checker + CPU baseline of the TF sub-pel leg: the reference's own tf_subpel_search (oracle/_ref/libsvtref_me.so, C kernels, one core) -- or the C
    restatement where the reference build is absent -- on a bounded random sample of the leg's blocks; every sampled block must equal the device result."""
    me_path, ref_path, ora_path = (os.path.join(ROOT, "oracle", "_ref", "libsvtref_me.so"), os.path.join(ROOT, "oracle", "_ref", "libsvtref.so"),
                                   os.path.join(ROOT, "oracle", "liboracle.so"))
    P, src, refs, descs, res, W, H = k["P"], k["src"], k["refs"], k["descs"], k["results"], k["W"], k["H"]
    if os.path.exists(me_path) and os.path.exists(ref_path):
        ref = C.CDLL(ref_path, mode=C.RTLD_GLOBAL)
        ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
        ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
        me, kind = C.CDLL(me_path), "reference"

        def one(d):
            mx, my, dist = C.c_int16(int(d["mv_x"])), C.c_int16(int(d["mv_y"])), C.c_uint64(0x7fffffff)
            b = int(d["bsize"])
            sbx, sby = int(d["pu_x"]) & ~63, int(d["pu_y"]) & ~63
            me.ref_tf_subpel_search(C.byref(P), vp(src, sby * W + sbx), W, vp(refs, int(d["ref_off"])), W, H, sbx, sby, b, (int(d["pu_x"]) - sbx) // b,
                                    (int(d["pu_y"]) - sby) // b, int(d["bilinear"]), C.byref(mx), C.byref(my), C.byref(dist))
            return dist.value, mx.value, my.value
    elif os.path.exists(ora_path):
        ora, kind = C.CDLL(ora_path), "port"

        def one(d):
            mx, my, dist = C.c_int16(int(d["mv_x"])), C.c_int16(int(d["mv_y"])), C.c_uint64(0x7fffffff)
            ora.oracle_tf_subpel_search(C.byref(P), vp(src, int(d["src_off"])), int(d["src_stride"]), vp(refs, int(d["ref_off"])), int(d["pu_x"]), int(d["pu_y"]),
                                        int(d["bsize"]), int(d["bilinear"]), C.byref(mx), C.byref(my), C.byref(dist))
            return dist.value, mx.value, my.value
    else:
        return {}
    order = np.random.default_rng(5).permutation(len(descs))
    t0, done, tcpu = time.perf_counter(), 0, 0.0
    while time.perf_counter() - t0 < budget_s and done < len(order):
        i = int(order[done])
        t1 = time.perf_counter()
        got = one(descs[i])
        tcpu += time.perf_counter() - t1
        if got != (int(res[i]["dist"]), int(res[i]["mv_x"]), int(res[i]["mv_y"])):
            sys.exit("bench.py: parity check FAILED for tf_subpel block %d: device %s, %s %s -- no numbers recorded" % (i, res[i], kind, got))
        done += 1
    return {"parity_checked_blocks": done, "cpu_baseline": {"value": done / tcpu, "unit": "blocks/s", "cores": 1, "kind": kind,
                                                            "sample": "%d random blocks of the leg's 64x64 / 32x32 / 16x16 mix, C kernels" % done}}


def cpu_lr_search(k, budget_s):
    """Checker + CPU baseline of an LR search leg: the reference's own search_norestore_seg / search_wiener_seg / search_sgrproj_seg (oracle/_ref/libsvtref_me.so,
    C kernels, one core) -- or the C restatement -- on a bounded random sample of the plane's restoration units; every sampled unit must equal the device result."""
    me_path, ref_path, ora_path = (os.path.join(ROOT, "oracle", "_ref", "libsvtref_me.so"), os.path.join(ROOT, "oracle", "_ref", "libsvtref.so"),
                                   os.path.join(ROOT, "oracle", "liboracle.so"))
    if not os.path.exists(ora_path):
        return {}
    ora = C.CDLL(ora_path)
    PD, src, dgd, pad, res = k["P"], k["src"], k["dgd"], k["pad"], k["results"]
    P = type(PD).from_buffer_copy(bytes(PD))  # the same parameters over the host copies of the planes
    P.src, P.dgd = src.ctypes.data, dgd.ctypes.data + (pad * dgd.shape[1] + pad) * dgd.itemsize
    n = ora.oracle_lr_unit_rect(C.byref(P), -1, None)
    rects = np.zeros((n, 4), np.int32)
    for u in range(n):
        ora.oracle_lr_unit_rect(C.byref(P), u, vp(rects[u]))
    use_ref = os.path.exists(me_path) and os.path.exists(ref_path)
    if use_ref:
        ref = C.CDLL(ref_path, mode=C.RTLD_GLOBAL)
        ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
        ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
        me = C.CDLL(me_path)
    order = np.random.default_rng(6).permutation(n)
    t0, done, tcpu = time.perf_counter(), 0, 0.0
    one = np.zeros(1, res.dtype)
    while time.perf_counter() - t0 < budget_s and done < n:
        u = int(order[done])
        t1 = time.perf_counter()
        if use_ref:
            me.ref_lr_search_plane(C.byref(P), None, vp(one), vp(rects[u]), 1)
        else:  # the restatement works on whole planes: crop the parameters to this unit's rectangle (+ the border the filters read)
            return {}
        tcpu += time.perf_counter() - t1
        for f in ("sse", "vfilter", "hfilter", "ep", "xqd"):
            if not np.array_equal(one[f][0], res[f][u]):
                sys.exit("bench.py: parity check FAILED for the LR search, unit %d field %s: device %s, reference %s -- no numbers recorded" % (u, f, res[f][u], one[f][0]))
        done += 1
    out = {"parity_checked_units": done, "cpu_baseline_c_one_core": {"value": done / tcpu, "unit": "units/s", "cores": 1, "kind": "reference",
                                                                     "sample": "%d random 256x256 units of the plane, C kernels (the parity check's own run)" % done}}
    # the CPU path the contract names: the SAME reference function with the dispatch pointers it goes through pointed at the AVX2 kernels oracle/_ref holds, the units
    # spread over every host core (the wrapper builds its own search context per call: thread-safe)
    if use_ref and " avx2 " in open("/proc/cpuinfo").read():
        patched = []
        for ptr, fnn in (("svt_av1_compute_stats", "svt_av1_compute_stats_avx2"), ("svt_av1_compute_stats_highbd", "svt_av1_compute_stats_highbd_avx2"),
                         ("svt_av1_highbd_pixel_proj_error", "svt_av1_highbd_pixel_proj_error_avx2"), ("svt_av1_lowbd_pixel_proj_error", "svt_av1_lowbd_pixel_proj_error_avx2"),
                         ("svt_av1_selfguided_restoration", "svt_av1_selfguided_restoration_avx2"), ("svt_apply_selfguided_restoration", "svt_apply_selfguided_restoration_avx2"),
                         ("svt_av1_highbd_wiener_convolve_add_src", "svt_av1_highbd_wiener_convolve_add_src_avx2"), ("svt_av1_wiener_convolve_add_src", "svt_av1_wiener_convolve_add_src_avx2"),
                         ("svt_get_proj_subspace", "svt_get_proj_subspace_avx2"), ("svt_spatial_full_distortion_kernel", "svt_spatial_full_distortion_kernel_avx2"),
                         ("svt_full_distortion_kernel16_bits", "svt_full_distortion_kernel16_bits_avx2")):
            if hasattr(ref, fnn):
                try:
                    C.c_void_p.in_dll(ref, ptr).value = C.cast(getattr(ref, fnn), C.c_void_p).value
                    patched.append(fnn)
                except ValueError:
                    pass
        import concurrent.futures as cf
        cores = host_cores()
        sample = [int(u) for u in order[:min(n, max(cores * 2, 16))]]
        shares = [sample[i::cores] for i in range(cores)]
        outs = [np.zeros(max(len(sh), 1), res.dtype) for sh in shares]

        def work(i):
            if shares[i]:
                rr = np.ascontiguousarray(rects[shares[i]])
                me.ref_lr_search_plane(C.byref(P), None, vp(outs[i]), vp(rr), len(shares[i]))
        t1 = time.perf_counter()
        with cf.ThreadPoolExecutor(cores) as ex:
            list(ex.map(work, range(cores)))
        dt = time.perf_counter() - t1
        for i, sh in enumerate(shares):  # (the AVX2 kernels must give the device's units too: the reference's own cross-ISA invariant)
            for j, u in enumerate(sh):
                for f in ("sse", "vfilter", "hfilter", "ep", "xqd"):
                    if not np.array_equal(outs[i][f][j], res[f][u]):
                        sys.exit("bench.py: parity check FAILED for the LR search (AVX2 reference kernels), unit %d field %s -- no numbers recorded" % (u, f))
        out["cpu_baseline"] = {"value": len(sample) / dt / n, "unit": "planes/s (%d units per plane)" % n, "cores": cores, "kind": "reference", "units_per_s": len(sample) / dt,
                               "sample": "%d units through the reference's search_norestore / search_wiener / search_sgrproj_seg (ref_wrap/ref_lr_search.c), %d of its dispatch "
                                         "pointers at AVX2 kernels (%s), all cores at once" % (len(sample), len(patched), ", ".join(x.replace("svt_", "").replace("_avx2", "") for x in patched))}
        ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))  # back to the C table for whatever runs next
        ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    else:
        out["cpu_baseline"] = dict(out["cpu_baseline_c_one_core"])
    return out


def encoder_fps():
    """The second half of BASELINE.json's metric: encoder fps at 1080p preset 8.  The reference's own encoder (oracle/_ref/enc, C-only build with the binding of
    INTEGRATION.md section 1) encodes one synthetic 60-frame 1080p clip with SVT_HIP unset, then with the ME (open-loop and the temporal filter's, incl. its sub-pel refinement), deblocking, CDEF (search + apply) and LR
    (search + filter) stage seams on this GPU; the two bitstreams must be identical or no number is recorded.  ~10 s; None when the encoder build is absent."""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("enc_identity", os.path.join(ROOT, "tools", "enc_identity.py"))
    ei = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ei)
    lib = os.path.join(ROOT, "svt-av1-psy_amd", "libsvtav1_hip.so")
    if not (os.path.exists(ei.ENC) and os.path.exists(lib)):
        return None
    with tempfile.TemporaryDirectory() as td:
        have_x = os.path.exists(ei.ENC_AVX2)
        CASE = "fps_1080p_p8_all_tplrecon"  # every stage seam, the TPL dispenser's reconstruction half included (round 4: one launch per picture)
        r = ei.run_case(CASE, lib, td, timeout=600, host="avx2" if have_x else "c")
        # two more (AVX2 alone, AVX2 + stages) pairs: a 0.5 s encode spreads by +- 5 % from run to run -- the fps quoted are the medians of three
        want_bits = open(os.path.join(td, CASE + "_c.ivf"), "rb").read() if r.get("identical") else b""
        rep = ei.repeat_pairs(CASE, lib, td, want_bits, pairs=4, host="avx2") if have_x and r.get("identical") else {}  # five pairs in all
        rc_ = ei.run_case(CASE, lib, td, timeout=600, host="c") if have_x else r  # the round-1/2 figure: C-only host + stages, for continuity
        r300 = ei.run_case(CASE + "_300", lib, td, timeout=600, host="avx2") if have_x else {}  # steady state: the clip looped five times
        # the AVX-512 build of the reference (EN_AVX512_SUPPORT=1 + ASM_AVX512) where the host has AVX-512: alone and with the stages
        have_512 = os.path.exists(ei.ENC_AVX512) and cpu_has(*AVX512)
        r512 = ei.run_case(CASE, lib, td, timeout=600, host="avx512") if have_512 else {}
        rep512 = ei.repeat_pairs(CASE, lib, td, want_bits, pairs=4, host="avx512") if have_512 and r512.get("identical") else {}
        # ... and the subset of stages that pays at this preset (ME, temporal filter, TPL, CDEF: tools/seam_subset_probe.py), five pairs per host
        PAY = "fps_1080p_p8_paying"
        pay = ei.repeat_pairs(PAY, lib, td, want_bits, pairs=3, host="avx2") if have_x and r.get("identical") else {}  # (three pairs since round 6: the 4K10 leg took their time)
        pay512 = ei.repeat_pairs(PAY, lib, td, want_bits, pairs=3, host="avx512") if have_512 and r.get("identical") else {}
        # K concurrent encodes sharing this GPU on the box's host cores: aggregate fps and host CPU seconds per frame, AVX2 host alone vs with the stages
        inst = ei.run_instances(CASE + "_300", lib, td, 4, host="avx2", timeout=900) if have_x else {}  # (300 frames: a 60-frame encode is over in 0.5 s, less than a process's start-up)
        # thread CPU time per stage (integration/seam_cpu.h), a run of its own: the brackets cost two clock reads per SB in the ME stage
        rcpu = ei.run_case(CASE, lib, td, timeout=600, host="avx2", cpu_stats=True) if have_x else {}
        # tpl level 1 (presets <= M2) inside the AVX2 encoder, no seam: thread CPU time of the dispenser (integration/seam_cpu.h) over a short 1080p preset-2 encode --
        # the reference-kind CPU figure of the tpl_l1_* legs (not a one-core C run)
        p2 = {}
        if have_x:
            clip2, n2 = os.path.join(td, "p2.yuv"), 6
            ei.make_clip(clip2, 1920, 1080, n2, 8)
            st2 = os.path.join(td, "p2_cpu.txt")
            r2, dt2 = ei.encode(clip2, 1920, 1080, n2, 8, ["--preset", "2"], os.path.join(td, "p2"), {"SVT_HIP_SEAM_CPU_STATS": st2}, timeout=900, enc=ei.ENC_AVX2)
            if r2.returncode == 0 and os.path.exists(st2):
                kv = dict(ln.split() for ln in open(st2).read().splitlines() if ln.strip())
                p2 = {"frames": n2, "seconds": round(dt2, 2), "tpl_cpu_ms": int(kv.get("tpl_cpu_ms", 0)), "tpl_calls": int(kv.get("tpl_calls", 0)), "sbs_per_picture": 510}
    if (rep and not rep.get("identical")) or (rep512 and not rep512.get("identical")):
        return {"bitstream_identical": False, "error": "a repeated encode's bitstream differs from the C-only encoder's: no encoder numbers recorded"}
    med = lambda v: sorted(v)[len(v) // 2] if v else None  # noqa: E731
    alone_all = [r.get("fps_avx2")] + (rep.get("fps_alone") or []) if have_x else []
    with_all = [r.get("fps_hip")] + (rep.get("fps_with_stages") or []) if have_x else []
    alone_512 = [v for v in [r512.get("fps_avx512")] + (rep512.get("fps_alone") or []) if v] if r512 else []
    with_512 = [v for v in [r512.get("fps_hip")] + (rep512.get("fps_with_stages") or []) if v] if r512 else []
    if not r.get("identical") or not rc_.get("identical") or (r300 and not r300.get("identical")) or (r512 and not r512.get("identical")) or (inst and not inst.get("identical")):
        # (the kernel legs of the line stand on their own parity checks: the line is still printed, the encoder half says what failed and carries no fps)
        bad = [n for n, x in (("avx2", r), ("c", rc_), ("300 frames", r300), ("avx512", r512), ("instances", inst)) if x and not x.get("identical")]
        return {"bitstream_identical": False, "error": "identity check failed for: %s (bitstream_equal %s; seam %s) -- no encoder numbers recorded" %
                (", ".join(bad), [x.get("bitstream_equal") for x in (r, rc_) if x], (r.get("seam") or {}).get("plane_reuploads_by_checksum"))}
    return {"fps_c_only": r.get("fps_c"), "fps_avx2_intrinsics": med([v for v in alone_all if v]) if have_x else None,
            "fps_avx2_host_with_stage_seams": med([v for v in with_all if v]) if have_x else None,
            "fps_avx2_pairs": {"alone": alone_all, "with_stages": with_all, "quoted": "median"} if have_x else None,
            "fps_c_host_with_stage_seams": rc_.get("fps_hip"), "bitstream_identical": True,
            "steady_state_300_frames": {"fps_c_only": r300.get("fps_c"), "fps_avx2_intrinsics": r300.get("fps_avx2"), "fps_avx2_host_with_stage_seams": r300.get("fps_hip"),
                                        "note": "single run each; run-to-run spread on this box class is +- 5 % (profiles/r03_call13..15)"} if r300 else None, "avx2_bitstream_identical_to_c": r.get("avx2_identical_to_c"),
            "fps_avx512_intrinsics": med(alone_512) if alone_512 else r512.get("fps_avx512"), "fps_avx512_host_with_stage_seams": med(with_512) if with_512 else r512.get("fps_hip"),
            "fps_avx512_pairs": {"alone": alone_512, "with_stages": with_512, "quoted": "median"} if alone_512 else None,
            # the same clip with only the stages that pay at preset 8 on the device (no LR / deblocking seam); identical bitstreams or no number
            "paying_stages": {"stages": "ME, temporal filter, TPL (both halves), CDEF", "identical": bool(pay.get("identical")) and (not pay512 or bool(pay512.get("identical"))),
                              "fps_avx2": med(pay.get("fps_alone") or []), "fps_avx2_with": med(pay.get("fps_with_stages") or []),
                              "fps_avx512": med(pay512.get("fps_alone") or []), "fps_avx512_with": med(pay512.get("fps_with_stages") or []),
                              "pairs": {"avx2": pay, "avx512": pay512}} if pay and pay.get("identical") and (not pay512 or pay512.get("identical")) else None,
            # user + system CPU seconds of the whole encoder process per frame (RUSAGE_CHILDREN): what the offload takes off the host
            "host_cpu_s_per_frame": dict(r.get("host_cpu_s_per_frame") or {}, **{k: v for k, v in (r512.get("host_cpu_s_per_frame") or {}).items() if k != "c"}),
            "instances": {"k": inst.get("instances"), "frames_each": inst.get("frames"), "fps_avx2": inst.get("fps_avx2"), "fps_avx2_with_stages": inst.get("fps_avx2_with_stages"),
                          "fps_sum_of_encoder_reports_avx2": inst.get("fps_sum_of_encoder_reports_avx2"),
                          "fps_sum_of_encoder_reports_avx2_with_stages": inst.get("fps_sum_of_encoder_reports_avx2_with_stages"),
                          "cpu_s_per_frame_avx2": inst.get("host_cpu_s_per_frame_avx2"), "cpu_s_per_frame_avx2_with_stages": inst.get("host_cpu_s_per_frame_avx2_with_stages"),
                          "identical": inst.get("identical")} if inst else None,
            "stage_cpu_ms_per_frame": rcpu.get("stage_cpu_ms_per_frame"), "host_cpu_s_per_frame_in_that_run": rcpu.get("host_cpu_s_per_frame"),
            "preset2_tpl_cpu": p2 or None,
            "frames": r["frames"], "host_threads": len(os.sched_getaffinity(0)), "host_cores": host_cores(),
            "host_ms_per_me_stage_call": (lambda m: round(m.get("ms_in_stage_calls", 0) / max(m.get("pictures_offloaded", 0) + m.get("tf_pairs_offloaded", 0), 1), 3))(r.get("seam") or {}),
            "host_ms_first_stage_call": (r.get("seam") or {}).get("ms_first_stage_call"),
            # wall time the encoder's threads spent inside stage calls (uploads + kernels + downloads), summed over the run: where the host side of the offload goes
            "host_ms_in_stage_calls": {k: (r.get(v) or {}).get("ms_in_stage_calls") for k, v in (("me", "seam"), ("tf_picture", "tfdriver"), ("tpl", "tplseam"), ("dlf", "dlfseam"),
                                                                                                   ("cdef", "cdefseam"), ("lr", "lrseam"))},
            "config": "1080p 8-bit, preset 8, CRF 35, all host threads; the reference encoder built (a) C-only and (b) with its SSE2..AVX2 intrinsic kernels (177 NASM kernels "
                      "stay at their C versions: no nasm here); the stage seams (ME, the temporal filter as one stage per central picture, both halves of the TPL dispenser, deblocking, CDEF, LR) on the MI355X",
            "stages_on_gpu": {"me": r.get("seam"), "tf_subpel": r.get("tfsubpel"), "tf_picture": r.get("tfdriver"), "tpl": r.get("tplseam"), "dlf": r.get("dlfseam"), "cdef": r.get("cdefseam"),
                              "lr": r.get("lrseam")}}


def encoder_fps_4k10(n_devices=1):
    """BASELINE.json configs[4] -- "full preset-8 encode of 4K30 10-bit synthetic YUV" -- at this node's GPU count: the reference encoder (AVX2 intrinsics host where built,
    else C-only) alone, then with every stage seam of SURVEY 8 on the device(s); the two bitstreams and the C-only encoder's must be IDENTICAL, decided on the first
    attempt (every encode of the comparison runs under tools/enc_identity.py's deterministic_env: the reference's 10-bit path is timing-dependent without it).
    n_devices > 1: pictures go round-robin over SVT_HIP_DEVICES=0..n-1 (integration/enc_handle_binding.c) -- physical GPUs where the node has them.  ~1 min."""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("enc_identity", os.path.join(ROOT, "tools", "enc_identity.py"))
    ei = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ei)
    lib = os.path.join(ROOT, "svt-av1-psy_amd", "libsvtav1_hip.so")
    if not (os.path.exists(ei.ENC) and os.path.exists(lib)):
        return None
    have_x = os.path.exists(ei.ENC_AVX2)
    host = "avx2" if have_x else "c"
    CASE = "fps_4k10_p8_30"
    if n_devices > 1:
        w_, h_, n_, bd_, extra_ = ei.CASES[CASE]
        CASE = CASE + "_%ddev" % n_devices
        ei.CASES[CASE] = (w_, h_, n_, bd_, extra_ + ["+devices:" + ",".join(str(i) for i in range(n_devices))])
    try:
        with tempfile.TemporaryDirectory() as td:
            os.environ["SVT_HIP_WARM_ARENA_MB"] = "448"  # (the loop-restoration search of a 4K plane: the pooled arena is made that large at initialisation, not inside the first picture)
            r = ei.run_case(CASE, lib, td, timeout=1200, host=host)
            want = open(os.path.join(td, CASE + "_c.ivf"), "rb").read() if r.get("identical") else b""
            rep = ei.repeat_pairs(CASE, lib, td, want, pairs=1, host=host, timeout=1200) if (r.get("identical") and n_devices == 1) else {}
    except Exception as e:  # noqa: BLE001  (the kernel legs stand on their own: the line is still printed)
        return {"bitstream_identical": False, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    finally:
        os.environ.pop("SVT_HIP_WARM_ARENA_MB", None)
    if not r.get("identical") or (rep and not rep.get("identical")):
        return {"bitstream_identical": False, "hip_encode_attempts": r.get("hip_encode_attempts"),
                "error": "identity check failed (bitstream_equal %s, repeated pairs %s): no fps recorded" % (r.get("bitstream_equal"), rep.get("identical") if rep else None)}
    med = lambda v: sorted(v)[len(v) // 2] if v else None  # noqa: E731
    alone = [v for v in [r.get("fps_" + host)] + (rep.get("fps_alone") or []) if v]
    with_ = [v for v in [r.get("fps_hip")] + (rep.get("fps_with_stages") or []) if v]
    return {"workload": "configs[4]: 3840x2160 10-bit 4:2:0, preset 8 (--psy-rd 0: the reference's multi-threaded 10-bit encode is irreproducible with psy-rd on), %d frames" % r["frames"], "frames": r["frames"], "host": host, "n_devices": n_devices,
            "fps_c_only": r.get("fps_c"), "fps_host_alone": med(alone), "fps_host_with_stage_seams": med(with_), "pairs": {"alone": alone, "with_stages": with_, "quoted": "median"},
            "bitstream_identical": True, "first_attempt": r.get("hip_encode_attempts") == 1, "host_cpu_s_per_frame": r.get("host_cpu_s_per_frame"),
            "devices": r.get("devices"), "host_threads": len(os.sched_getaffinity(0)),
            "note": "every encode (reference alone and with the stages) runs under the test-only deterministic harness: zero-filled heap + the child control set's 16-bit source "
                    "zero-filled at acquisition (tools/enc_identity.py deterministic_env); the AVX2 host runs its 177 NASM kernels as C (no nasm in the image)"}


def attach_encoder_baselines(kernels, enc):
    """The stage legs' CPU baseline of kind "reference": the reference's OWN functions with their AVX2 kernels, timed inside the reference encoder (oracle/_ref/enc_avx2,
    no seam) by the thread CPU clock around the stage entries (integration/seam_cpu.h) over the encoder leg's 1080p preset-8 clip -- CPU milliseconds per picture the
    stage handled, quoted as pictures per second of ONE host core.  (The one-core C restatements of oracle/ stay in the detail object as `cpu_baseline_port`: they are
    the parity checkers, not the reference's CPU path.)"""
    p2 = (enc or {}).get("preset2_tpl_cpu")
    if p2 and p2.get("tpl_calls") and p2.get("tpl_cpu_ms"):
        # one dispenser call = one picture = 1 + 510 bracketed entries (the dispenser + its per-SB function): pictures = calls / 511
        pictures = p2["tpl_calls"] / (p2["sbs_per_picture"] + 1.0)
        ms = p2["tpl_cpu_ms"] / max(pictures, 1e-9)
        for leg, key in (("tpl_l1_src_1080p8", "cpu_baseline"), ("tpl_l1_recon_1080p8", "cpu_baseline_both_halves")):
            k = kernels.get(leg)
            if isinstance(k, dict):
                if key in k:
                    k[key + "_c_one_core"] = k.pop(key)
                k[key] = {"value": 1e3 / ms, "unit": "pictures/s per host core (both halves of the dispenser)", "cores": 1, "kind": "reference", "cpu_ms_per_picture": ms,
                          "sample": "tpl_mc_flow_dispenser + its per-SB calls at tpl level 1 in oracle/_ref/enc_avx2 over a %d-frame 1080p preset-2 encode (%.1f dispenser "
                                    "calls), thread CPU time" % (p2["frames"], pictures)}
    if not enc or not enc.get("stage_cpu_ms_per_frame") or not enc["stage_cpu_ms_per_frame"].get("avx2"):
        return
    ref, frames, st = enc["stage_cpu_ms_per_frame"]["avx2"], enc.get("frames") or 0, enc.get("stages_on_gpu") or {}
    pics = {"tpl": (st.get("tpl") or {}).get("recon_pictures"), "tf": (st.get("tf_picture") or {}).get("pictures_filtered"), "me": (st.get("me") or {}).get("pictures_offloaded")}
    for leg, stage, what in (("tpl_stage_host_resident_1080p8", "tpl", "tpl_mc_flow_dispenser + its per-SB calls: both halves of the dispenser, what the leg's one call replaces"),
                             ("tpl_stage_host_1080p8", "tpl", "tpl_mc_flow_dispenser + its per-SB calls: both halves of the dispenser, what the leg's one call replaces"),
                             ("tpl_recon_stage_1080p8", "tpl", "tpl_mc_flow_dispenser + its per-SB calls (both halves of the dispenser; the leg times the reconstruction half)"),
                             ("tpl_src_stage_1080p8", "tpl", "tpl_mc_flow_dispenser + its per-SB calls (both halves of the dispenser; the leg times the source-based half)"),
                             ("tf_picture_stage_1080p8_4refs_host", "tf", "produce_temporally_filtered_pic, every segment"),
                             ("tf_picture_stage_1080p8_4refs_resident", "tf", "produce_temporally_filtered_pic, every segment"),
                             ("me_session_stage_1080p_host_preset8", "me", "svt_aom_motion_estimation_b64, every SB")):
        k = kernels.get(leg)
        if not isinstance(k, dict) or not pics.get(stage) or not ref.get(stage) or not frames:
            continue
        ms_per_picture = ref[stage] * frames / pics[stage]
        if "cpu_baseline" in k:
            k["cpu_baseline_port"] = k.pop("cpu_baseline")
        k["cpu_baseline"] = {"value": 1e3 / ms_per_picture, "unit": "pictures/s per host core", "cores": 1, "kind": "reference", "cpu_ms_per_picture": ms_per_picture,
                             "sample": "%s in oracle/_ref/enc_avx2 over the encoder leg's %d-frame 1080p preset-8 clip (%d pictures), thread CPU time" % (what, frames, pics[stage])}


def cpu_tpl_recon_stage(k):
    """Checker + CPU baseline of the TPL reconstruction leg: oracle_tpl_recon_picture (one core) on the leg's whole picture; statistics and reconstruction plane must
    equal the device's."""
    ora_path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(ora_path):
        return {}
    ora = C.CDLL(ora_path)
    P, planes = k["P"], k["planes"]
    want = np.zeros(k["cells"], k["recon_out"].dtype)
    rec = np.zeros_like(k["recon"])
    refs = (type(P.refs[0]) * 8)(*[P.refs[i] for i in range(8)])
    t0 = time.perf_counter()
    ora.oracle_tpl_recon_picture(C.byref(P), refs, 1, vp(planes), vp(planes), vp(k["out"]), C.c_void_p(rec.ctypes.data + int(P.src_off)), k["recon_stride"], vp(want))
    dt = time.perf_counter() - t0
    for name in ("srcrf_dist", "recrf_dist", "srcrf_rate", "recrf_rate", "written", "coded"):
        must_equal("tpl_recon_stage " + name, k["recon_out"][name], want[name])
    must_equal("tpl_recon_stage reconstruction", k["recon"], rec)
    for form, (f_rec, f_out) in sorted(k["recon_forms"].items()):  # every form of the stage the leg timed
        for name in ("srcrf_dist", "recrf_dist", "written", "coded"):
            must_equal("tpl_recon_stage (form %d) %s" % (form, name), f_out[name], want[name])
        must_equal("tpl_recon_stage (form %d) reconstruction" % form, f_rec, rec)
    return {"parity_checked_values": int(k["cells"]) * 6 + int(rec.size), "cpu_baseline": {"value": 1 / dt, "unit": "pictures/s", "cores": 1, "kind": "port",
                                                                                            "sample": "the leg's whole 1080p picture, oracle/oracle_tpl.c"}}


def cpu_tpl_level1(k):
    """Checker + CPU baseline (kind "reference") of the tpl-level-1 legs: the reference's own tpl_mc_flow_dispenser_sb_generic (oracle/_ref/libsvtref_me.so, one core) on
    the leg's whole picture with the leg's quantizer row; source-based statistics, the TplStats grid and the reconstruction must equal the device's."""
    path = os.path.join(ROOT, "oracle", "_ref", "libsvtref_me.so")
    if not os.path.exists(path):
        return {}
    me = C.CDLL(path)
    P, planes, pad = k["P"], k["planes"], k["pad"]
    Q = type(P).from_buffer_copy(P)
    want = np.zeros(k["cells"], k["out"].dtype)
    grid = np.zeros((k["cells"], 8), np.int64)
    rec = np.zeros((P.height, P.width), np.uint8)
    t0 = time.perf_counter()
    me.ref_tpl_dispenser_picture(C.byref(Q), 120, vp(planes), pad, pad, vp(planes), vp(k["tot"]), vp(k["mvs"]), vp(k["cand"]), k["n_pus"], vp(want), vp(grid), vp(rec))
    dt = time.perf_counter() - t0
    if any(Q.quant_fp[i] != P.quant_fp[i] or Q.round_fp[i] != P.round_fp[i] or Q.dequant[i] != P.dequant[i] for i in range(2)):
        raise SystemExit("bench: the tpl level 1 leg's quantizer row is not row 120 of the reference's tables")
    for name in want.dtype.names:
        if name != "pad":
            must_equal("tpl_level1 source-based " + name, k["out"][name], want[name])
    w = want["written"] > 0
    ro = k["recon_out"]
    for j, name in enumerate(("srcrf_dist", "recrf_dist", "srcrf_rate", "recrf_rate")):  # (16x16 blocks, synth size 16: result_model_store only clamps to >= 1)
        must_equal("tpl_level1 " + name, np.maximum(1, ro[name][w]), grid[w, j])
    must_equal("tpl_level1 reconstruction", k["recon"][pad:pad + P.height, pad:pad + P.width], rec)
    return {"parity_checked_values": int(k["cells"]) * 13 + int(rec.size), "parity_checked_recon": int(w.sum()) * 4 + int(rec.size), "cpu_baseline": {"value": 1 / dt, "unit": "pictures/s", "cores": 1, "kind": "reference",
                                                                                             "sample": "the leg's whole 1080p picture: both halves of the reference's dispenser (C kernels) through oracle/ref_wrap/ref_tpl.c"}}


def cpu_tpl_stage(k):
    """Checker + CPU baseline of the TPL leg: the C restatement (oracle/oracle_tpl.c, one core) on the leg's whole picture; every statistics record must equal
    the device's."""
    ora_path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(ora_path):
        return {}
    ora = C.CDLL(ora_path)
    want = np.zeros(k["cells"], k["out"].dtype)
    t0 = time.perf_counter()
    ora.oracle_tpl_src_picture(C.byref(k["P"]), vp(k["planes"]), vp(k["planes"]), vp(k["tot"]), vp(k["mvs"]), vp(k["cand"]), vp(want))
    dt = time.perf_counter() - t0
    for name in want.dtype.names:
        if name != "pad":
            must_equal("tpl_src_stage " + name, k["out"][name], want[name])
    return {"parity_checked_values": int(k["cells"]) * 9, "cpu_baseline": {"value": 1 / dt, "unit": "pictures/s", "cores": 1, "kind": "port",
                                                                            "sample": "the leg's whole 1080p picture, oracle/oracle_tpl.c"}}


def cpu_tf_picture(k):
    """Checker + CPU baseline of the temporal-filter picture leg: oracle/oracle_tf_picture.c (one core, the reference's block order) on the leg's picture; the
    filtered planes must equal the device's."""
    ora_path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(ora_path):
        return {}
    ora = C.CDLL(ora_path)
    pics, tabs, n_refs = k["pics"], k["tabs"], k["n_refs"]
    out = [x.copy() for x in pics[0]]
    cen = (C.c_void_p * 3)(*[x.ctypes.data for x in pics[0]])
    refs = (C.c_void_p * (3 * n_refs))(*[x.ctypes.data for pic in pics[1:] for x in pic])
    arr = lambda j: (C.c_void_p * n_refs)(*[t[j].ctypes.data for t in tabs])  # noqa: E731
    o = (C.c_void_p * 3)(*[x.ctypes.data for x in out])
    stats = np.zeros(5, np.uint32)
    t0 = time.perf_counter()
    ora.oracle_tf_picture(C.byref(k["P"]), cen, refs, arr(0), arr(1), arr(2), arr(3), n_refs, o, vp(stats), None, None)
    dt = time.perf_counter() - t0
    n = 0
    for pl in range(3):
        n += must_equal("tf_picture_stage plane %d" % pl, k["out"][pl], out[pl])
    return {"parity_checked_values": n, "cpu_baseline": {"value": 1 / dt, "unit": "pictures/s", "cores": 1, "kind": "port",
                                                          "sample": "the leg's whole picture, oracle/oracle_tf_picture.c (C kernels, the reference's lazy block order)"}}


def roofline(bytes_alg, seconds, kernel, traffic_kernel=None, **extra):
    """HBM roofline object of one leg: ALGORITHMIC bytes per launch / event-timed launch duration."""
    gbs = bytes_alg / seconds / 1e9
    tr = pmc_traffic(traffic_kernel or kernel) or {}
    r = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
         "frac_of_copy_ceiling": gbs / HBM_COPY_CEILING_GBS, "traffic": tr.get("hbm_bytes_per_launch"), "traffic_detail": tr or None,
         "algorithmic_bytes_per_launch": bytes_alg, "kernel": kernel, "kernel_us": seconds * 1e6, "region": regions.LAST}
    r.update(extra)
    return r


L3_BYTES = 256 << 20  # Infinity Cache (MI355X_MICROARCH.md)
VALU_CYCLES_PER_WAVE_INST = 4.5  # measured 4.4-5.0 for the plain integer opcodes (profiles/r01_call1_valu_issue_rates.txt); v_qsad_pk_u16_u8 costs 22.4
SIMDS, CLOCK_HZ = 1024, 2.4e9


def finish_rooflines(kernels, rf, me_kernel_s):
    """Attaches this run's counter passes to every leg's roofline object, by timed region: `traffic` = HBM bytes moved per call of the leg (FETCH_SIZE x 2 + WRITE_SIZE,
    KiB units: MI355X_MICROARCH.md), `valu_frac` = SQ_INSTS_VALU per call x 4.5 cycles / (1024 SIMDs x 2.4 GHz x the leg's event-timed seconds) -- the share of the
    chip's VALU issue slots the leg's instructions need at the plain-opcode rate (a lower bound where multi-cycle opcodes dominate: the ME search states its own
    figure from the measured v_qsad_pk_u16_u8 rate; ~1.0 = the leg runs at the VALU issue roof and only fewer instructions make it faster).  `binds` names the roof that
    is closer: "valu", "hbm", or "latency" when neither reaches 0.35 (dependent launches / occupancy / LDS: DESIGN.md names which)."""
    reg = (LIVE_PMC or {}).get("regions") or {}
    probe = (LIVE_PMC or {}).get("probe") or {}
    cal = (LIVE_PMC or {}).get("traffic_calibration") or {}
    rf["traffic_calibration"] = cal or None
    items = [("__me__", {"roofline": rf})] + [(n, k) for n, k in kernels.items() if isinstance(k, dict)]
    c3 = kernels.get("config3_roundtrip")
    if isinstance(c3, dict) and c3.get("size_rooflines"):
        worst = c3["roofline"].get("size")
        items = items[:1] + [("config3:" + sz, {"roofline": r}) for sz, r in c3["size_rooflines"].items()] + [i for i in items[1:] if i[0] != "config3_roundtrip"]
    for name, k in items:
        r = k.get("roofline")
        if not isinstance(r, dict) or r.get("bound") == "pcie":
            continue
        g = reg.get(r.get("region"))
        t = (r.get("kernel_us") or 0) * 1e-6
        if g and t > 0:
            tot = {}
            for kn, kv in g["kernels"].items():
                for c, v in kv.items():
                    tot[c] = tot.get(c, 0.0) + v
            calls = g.get("calls", regions.CALLS)
            # bytes per counter unit, calibrated in THIS run on the leg's access shape (roofline["access"], default a contiguous 16 B/lane stream); where the calibration
            # regions did not run, the guide's figures: FETCH_SIZE x 2 KiB, WRITE_SIZE x 1 KiB
            shape = r.get("access") or "stream16"  # a tuple = the leg reads equal byte counts in each of the shapes: the mean of their factors
            crd = cal.get("read") or {}
            fs = [crd.get(sh) for sh in (shape if isinstance(shape, (tuple, list)) else (shape,))]
            f_rd = (sum(fs) / len(fs)) if fs and all(fs) else (crd.get("stream16") or 2048.0)
            f_wr = (cal.get("write") or {}).get(r.get("access_write") or "stream16") or 1024.0
            moved = (f_rd * tot.get("FETCH_SIZE", 0.0) + f_wr * tot.get("WRITE_SIZE", 0.0)) / calls
            if "FETCH_SIZE" in tot or "WRITE_SIZE" in tot:
                r["traffic"] = moved
                r["traffic_detail"] = {"read": f_rd * tot.get("FETCH_SIZE", 0.0) / calls, "write": f_wr * tot.get("WRITE_SIZE", 0.0) / calls,
                                       "bytes_per_fetch_unit": f_rd, "bytes_per_write_unit": f_wr, "access": shape, "calibrated": bool(cal.get("read")),
                                       "source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run, launches between the region's markers; "
                                                 "counter units calibrated on svt_hip_mem_probe regions of known size in the leg's access shape"}
                if t > 0:
                    r["traffic_GBps"] = moved / t / 1e9  # what the memory side moved per second of the leg (L2 misses: Infinity-Cache hits are counted too)
                if r.get("algorithmic_bytes_per_launch"):
                    r["moved_over_algorithmic"] = moved / r["algorithmic_bytes_per_launch"]
            if "SQ_INSTS_VALU" in tot:
                r["valu_wave_insts_per_call"] = tot["SQ_INSTS_VALU"] / calls
                r.setdefault("valu_frac", tot["SQ_INSTS_VALU"] / calls * VALU_CYCLES_PER_WAVE_INST / (SIMDS * CLOCK_HZ * t))
                if name == "__me__":
                    r["valu_frac_plain_opcode_rate"] = tot["SQ_INSTS_VALU"] / calls * VALU_CYCLES_PER_WAVE_INST / (SIMDS * CLOCK_HZ * t)
            if "SQ_ACTIVE_INST_VALU" in tot:  # raw, for the record only: the counter runs ahead of time for the multi-cycle opcodes (> 1 "busy" for the packed-SAD search)
                r["sq_active_inst_valu_per_call"] = tot["SQ_ACTIVE_INST_VALU"] / calls
            r["kernels_per_call"] = {kn: round(kv["launches"] / calls, 2) for kn, kv in g["kernels"].items()}
        v = r.get("valu_frac") or 0.0
        h = r.get("frac") or 0.0
        if r.get("bound") == "mfma":
            r["binds"] = "mfma"
        else:
            r["binds"] = "valu" if (v >= h and v >= 0.35) else ("hbm" if (h > v and h >= 0.35) else "latency")
            if r["binds"] == "hbm" and r.get("footprint_bytes") is not None and r["footprint_bytes"] <= L3_BYTES:
                r["binds"] = "l3"  # the leg's whole working set sits in the 256 MiB Infinity Cache and every launch re-reads it: a cache rate, not a DRAM rate
    if isinstance(c3, dict) and c3.get("size_rooflines"):
        for sz, r in c3["size_rooflines"].items():
            if sz in c3["sizes"]:
                c3["sizes"][sz] = c3["sizes"][sz][:2] + [round(r.get("valu_frac") or 0.0, 3)]
        c3["roofline"] = dict(c3["size_rooflines"][worst], size=worst, note=c3["roofline"].get("note"))
        vb = [r.get("valu_frac") or 0.0 for r in c3["size_rooflines"].values()]
        c3["valu_frac_min_max"] = [round(min(vb), 3), round(max(vb), 3)]


def must_equal(name, got, want):
    """In-run parity gate: no number is recorded for a leg whose output differs from the CPU checker."""
    if not np.array_equal(np.asarray(got), np.asarray(want)):
        bad = np.nonzero(np.asarray(got).reshape(-1) != np.asarray(want).reshape(-1))[0]
        sys.exit("bench.py: parity check FAILED for %s (%d mismatches, first at %s) -- no numbers recorded" % (name, bad.size, bad[:4]))
    return int(np.asarray(want).size)


NO_CHECK = False  # --no-parity-check: profiling passes only (the small-frame check launches would pollute per-kernel averages)


def oracle_lib():
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    return C.CDLL(path) if os.path.exists(path) and not NO_CHECK else None


def vp(a, off=0):
    return C.c_void_p(a.ctypes.data + off)


def bench_sad_pairs(torch, lib, pkg, stream, a, cpu):
    """config 1 on the GPU: 64x64 SAD of co-located SB pairs.  DISJOINT source and reference plane sets, every plane read by exactly one launch
    item set, so no byte is fetched twice within a launch and nothing a previous launch left in the 256 MiB Infinity Cache helps (footprint
    1.2 GB): the byte rate is a DRAM rate.  Algorithmic bytes = 2 * 64 * 64 per block."""
    n_src = 240
    planes = torch.randint(0, 256, (2 * n_src * PLANE,), dtype=torch.uint8, device="cuda")
    pairs = np.zeros(n_src * 510, dtype=pkg.SadPair)
    o = ((PAD + np.arange(17)[:, None] * 64) * STRIDE + PAD + np.arange(30)[None, :] * 64).reshape(-1).astype(np.uint64)
    for f in range(n_src):
        pairs["src_off"][f * 510:(f + 1) * 510] = np.uint64(f * PLANE) + o
        pairs["ref_off"][f * 510:(f + 1) * 510] = np.uint64((n_src + f) * PLANE + 3 + 2 * STRIDE) + o  # reference blocks sit at arbitrary byte offsets in ME
    pairs["src_stride"] = pairs["ref_stride"] = STRIDE
    d_pairs = torch.from_numpy(pairs.view(np.uint8)).cuda()
    d_out = torch.zeros(len(pairs), dtype=torch.int32, device="cuda")
    fn = lambda: lib.svt_hip_sad_nxm_batch(planes.data_ptr(), planes.data_ptr(), d_pairs.data_ptr(), len(pairs), 64, 64, d_out.data_ptr(), stream)  # noqa: E731
    fn()
    torch.cuda.synchronize()
    # parity: the first and the last plane pair against a plain |a - b| sum
    got = d_out.cpu().numpy().view(np.uint32)
    checked = 0
    for f in (0, n_src - 1):
        hs = planes[f * PLANE:(f + 1) * PLANE].cpu().numpy().reshape(ROWS, STRIDE).astype(np.int32)
        hr = planes[(n_src + f) * PLANE:(n_src + f + 1) * PLANE].cpu().numpy().reshape(ROWS, STRIDE).astype(np.int32)
        want = np.array([np.abs(hs[PAD + sy * 64:PAD + sy * 64 + 64, PAD + sx * 64:PAD + sx * 64 + 64] -
                                hr[PAD + sy * 64 + 2:PAD + sy * 64 + 66, PAD + sx * 64 + 3:PAD + sx * 64 + 67]).sum() for sy in range(17) for sx in range(30)], np.uint32)
        checked += must_equal("sad64x64_pairs", got[f * 510:(f + 1) * 510], want)
    per, reps = time_leg(torch, fn, a.min_leg_s)
    out = {"launches_per_timed_batch": reps, "value": len(pairs) / per / 1e6, "unit": "Mblocks/s (64x64 pairs)", "footprint_MB": 2 * n_src * PLANE / 1e6, "parity_checked_values": checked,
           "roofline": roofline(len(pairs) * 8192, per, "sad_nxm_strip_kernel" if os.environ.get("SVT_HIP_SAD_FORM") == "1" else "sad_nxm_pipe_kernel", algorithmic_bytes_per_block=8192, access=("blocks64_aligned", "blocks64_off3"), footprint_bytes=int(2 * n_src * PLANE),
                                note="disjoint src / ref plane sets, each byte read once per launch; footprint 1.2 GB")}
    if cpu:
        ref, oracle = ref_libs()
        if ref is not None and " avx2 " in open("/proc/cpuinfo").read():
            f = oracle.oracle_time_sad_pairs
            f.restype = C.c_uint64
            f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_void_p]
            nh = 120  # host sample: 120 + 120 planes (600 MB: beyond the 256 MB of L3 an EPYC socket has), the same pair geometry
            hp = np.concatenate([planes[:nh * PLANE].cpu().numpy(), planes[n_src * PLANE:(n_src + nh) * PLANE].cpu().numpy()])
            hd = pairs[:nh * 510].copy()
            hd["ref_off"] -= np.uint64((n_src - nh) * PLANE)
            sums = np.zeros(host_cores() + 1, np.uint64)
            legs = [(0, "svt_aom_sad64x64_avx2", "cpu_baseline"), (1, "svt_nxm_sad_kernel_helper_avx2", "cpu_baseline_nxm_helper")]
            if cpu_has(*AVX512) and hasattr(ref, "svt_aom_sad64x64_avx512"):
                legs.append((0, "svt_aom_sad64x64_avx512", "cpu_baseline_avx512"))
            for kind, sym, key in legs:
                fp = C.cast(getattr(ref, sym), C.c_void_p)
                run = lambda i0, st, sec: f(fp, kind, hp.ctypes.data, hp.ctypes.data, hd.ctypes.data, len(hd), i0, st, sec, sums[i0:].ctypes.data)  # noqa: E731
                rate, one, cores = cpu_pool(run, 3.0)
                out[key] = {"value": rate / 1e6, "unit": "Mblocks/s (64x64 pairs)", "cores": cores, "kind": "reference", "single_thread_value": one / 1e6,
                            "sample": "%s over 120 + 120 host planes (600 MB), same pair geometry, 3 s per leg" % sym}
    return out


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
    fn()
    torch.cuda.synchronize()
    checked, oracle = 0, oracle_lib()
    coeff = d_out.cpu().numpy()
    if oracle is not None:
        for b in list(range(24)) + [n - 1]:
            want = np.zeros(1024, np.int32)
            oracle.oracle_fwd_txfm2d(vp(res, b * 2048), vp(want), 32, 0, ts, 10, 0)
            checked += must_equal("fwd_txfm2d_32x32", coeff[b * 1024:(b + 1) * 1024], want)
    per, reps = time_leg(torch, fn, a.min_leg_s)
    out = {"fwd_txfm2d_32x32": {"launches_per_timed_batch": reps, "value": n / per / 1e6, "unit": "Mblocks/s (32x32)", "blocks_per_step": n, "parity_checked_values": checked,
                                "roofline": roofline(n * 6144, per, "fwd_txfm2d_kernel<32,32>", "fwd_txfm2d_kernel<32, 32>", algorithmic_bytes_per_block=6144,
                                                     note="butterfly network: VALU/int32-multiply bound, not a dense contraction (DESIGN.md 4.2)")}}
    # ---- quantize_b (high bit depth form, log_scale 1) on the coefficients just produced: 4 B in + 8 B out per coefficient
    import bench_legs
    qd = np.zeros(n, dtype=pkg.QuantDesc)
    qpar = bench_legs._qparams(pkg, 88, 112, 1)
    iscan = np.arange(1024, dtype=np.int16)
    d_qd, d_qp, d_is = (torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).reshape(-1)).cuda() for x in (qd, qpar, iscan))
    d_q, d_dq = torch.zeros(n * 1024, dtype=torch.int32, device="cuda"), torch.zeros(n * 1024, dtype=torch.int32, device="cuda")
    d_eob = torch.zeros(n, dtype=torch.int16, device="cuda")
    fq = lambda: lib.svt_hip_quantize_batch(1, d_out.data_ptr(), n, 1024, d_qp.data_ptr(), d_is.data_ptr(), None, None, d_qd.data_ptr(), d_q.data_ptr(),  # noqa: E731
                                            d_dq.data_ptr(), d_eob.data_ptr(), stream)
    fq()
    torch.cuda.synchronize()
    checked = 0
    hq, hdq, heob = d_q.cpu().numpy(), d_dq.cpu().numpy(), d_eob.cpu().numpy().view(np.uint16)
    if oracle is not None:
        P = {k: np.ascontiguousarray(qpar[k][0]) for k in ("zbin", "round", "quant", "quant_shift", "dequant")}
        for b in list(range(16)) + [n - 1]:
            q, dq, eob = np.zeros(1024, np.int32), np.zeros(1024, np.int32), C.c_uint16(0)
            cb = np.ascontiguousarray(coeff[b * 1024:(b + 1) * 1024])
            oracle.oracle_quantize(1, vp(cb), 1024, vp(P["zbin"]), vp(P["round"]), vp(P["quant"]), vp(P["quant_shift"]), vp(q), vp(dq), vp(P["dequant"]), C.byref(eob),
                                   vp(iscan), None, None, 1)
            checked += must_equal("quantize_b_32x32 qcoeff", hq[b * 1024:(b + 1) * 1024], q) + must_equal("quantize_b_32x32 dqcoeff", hdq[b * 1024:(b + 1) * 1024], dq)
            must_equal("quantize_b_32x32 eob", [int(heob[b])], [eob.value])
    per, _ = time_leg(torch, fq, a.min_leg_s)
    out["quantize_b_32x32"] = {"value": n / per / 1e6, "unit": "Mblocks/s (32x32, highbd quantize_b, log_scale 1)", "parity_checked_values": checked,
                               "roofline": roofline(n * (12 * 1024 + 2), per, "quant_kernel<1, false>", algorithmic_bytes_per_block=12 * 1024 + 2)}
    # ---- inverse 32x32 + reconstruction (10-bit) from the dequantised coefficients: 4 B/coeff in + 2 B/px prediction + 2 B/px reconstruction
    idesc = np.zeros(n, dtype=pkg.InvTxfmDesc)
    idesc["coeff_off"] = np.arange(n, dtype=np.uint64) * 1024
    idesc["pred_off"] = idesc["recon_off"] = np.arange(n, dtype=np.uint64) * 1024
    idesc["pred_stride"] = idesc["recon_stride"] = 32
    pred = g.integers(0, 1024, n * 1024).astype(np.uint16)
    d_id, d_pred = torch.from_numpy(idesc.view(np.uint8)).cuda(), torch.from_numpy(pred.view(np.int16)).cuda()
    d_rec = torch.zeros(n * 1024, dtype=torch.int16, device="cuda")
    fi = lambda: lib.svt_hip_inv_txfm2d_add_batch(d_dq.data_ptr(), d_pred.data_ptr(), d_rec.data_ptr(), d_id.data_ptr(), n, ts, 10, stream)  # noqa: E731
    fi()
    torch.cuda.synchronize()
    checked = 0
    hrec = d_rec.cpu().numpy().view(np.uint16)
    if oracle is not None:
        for b in list(range(24)) + [n - 1]:
            want = np.zeros(1024, np.uint16)
            cb = np.ascontiguousarray(hdq[b * 1024:(b + 1) * 1024])
            oracle.oracle_inv_txfm2d_add(vp(cb), vp(pred, b * 2048), 32, vp(want), 32, 0, ts, 10)
            checked += must_equal("inv_txfm2d_add_32x32", hrec[b * 1024:(b + 1) * 1024], want)
    per, _ = time_leg(torch, fi, a.min_leg_s)
    out["inv_txfm2d_add_32x32"] = {"value": n / per / 1e6, "unit": "Mblocks/s (32x32, 10-bit)", "parity_checked_values": checked,
                                   "roofline": roofline(n * 8192, per, "inv_txfm2d_kernel<unsigned short, 32, 32>", algorithmic_bytes_per_block=8192)}
    if cpu:
        ref, oracle2 = ref_libs()
        if ref is not None and " avx2 " in open("/proc/cpuinfo").read():
            f = oracle2.oracle_time_fwd_txfm
            f.restype = C.c_uint64
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_uint8, C.c_uint8, C.c_uint32, C.c_uint32, C.c_double]
            ns = 512  # private input / output per thread (shared buffers serialise the cores through the cache hierarchy)
            bufs = [(aligned_zeros(ns * 1024, np.int16), aligned_zeros(2048, np.int32)) for _ in range(host_cores())]
            for b_in, _ in bufs:
                b_in[:] = res[:ns * 1024]
            legs = [("svt_av1_fwd_txfm2d_32x32_avx2", "cpu_baseline")]
            if cpu_has(*AVX512) and hasattr(ref, "svt_av1_fwd_txfm2d_32x32_avx512"):
                legs.append(("svt_av1_fwd_txfm2d_32x32_avx512", "cpu_baseline_avx512"))
            for sym, key in legs:
                fnp = C.cast(getattr(ref, sym), C.c_void_p)
                run = lambda i0, st, sec: f(fnp, bufs[i0][0].ctypes.data, ns, 32, 32, bufs[i0][1].ctypes.data, 0, 10, 0, 1, sec)  # noqa: E731
                rate, one, cores = cpu_pool(run, 3.0)
                out["fwd_txfm2d_32x32"][key] = {"value": rate / 1e6, "unit": "Mblocks/s (32x32)", "cores": cores, "kind": "reference", "single_thread_value": one / 1e6,
                                                "sample": "%s, 512 private blocks per thread, 3 s per leg" % sym}
            import bench_cpu
            bench_cpu.quantize(out, CPU_HELPERS(), coeff, qpar, iscan)
            # the inverse 32x32 + reconstruction (10 bit): the reference's AVX2 variant is dav1d assembly (NASM: not buildable here), so the intrinsic variants it has --
            # SSE4.1 and AVX-512 -- are timed (common_dsp_rtcd.c:515)
            if hasattr(oracle2, "oracle_time_inv_txfm"):
                fi2 = oracle2.oracle_time_inv_txfm
                fi2.restype = C.c_uint64
                fi2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_uint8, C.c_uint8, C.c_uint32, C.c_uint32, C.c_double]
                nsi = 256
                ib = [(aligned_zeros(nsi * 1024, np.int32), aligned_zeros(1024, np.uint16)) for _ in range(host_cores())]
                for b_in, _ in ib:
                    b_in[:] = hdq[:nsi * 1024]
                legs = [("svt_av1_inv_txfm2d_add_32x32_sse4_1", "cpu_baseline_sse4_1")]
                if cpu_has(*AVX512) and hasattr(ref, "svt_av1_inv_txfm2d_add_32x32_avx512"):
                    legs.append(("svt_av1_inv_txfm2d_add_32x32_avx512", "cpu_baseline_avx512"))
                for sym, key in legs:
                    if not hasattr(ref, sym):
                        continue
                    fnp = C.cast(getattr(ref, sym), C.c_void_p)
                    run = lambda i0, st, sec: fi2(fnp, ib[i0][0].ctypes.data, nsi, 32, 32, ib[i0][1].ctypes.data, 0, 10, 0, 1, sec)  # noqa: E731
                    rate, one, cores = cpu_pool(run, 3.0)
                    out["inv_txfm2d_add_32x32"][key] = {"value": rate / 1e6, "unit": "Mblocks/s (32x32, 10-bit)", "cores": cores, "kind": "reference",
                                                        "single_thread_value": one / 1e6, "sample": "%s, 256 private blocks per thread, 3 s per leg" % sym}
    return out


def bench_config3(torch, lib, pkg, stream, a):
    """BASELINE configs[2] / SURVEY 8(d) config 3 on the reference's own data: residual -> FwdTxfm2d -> (svt_handle_transform) -> quantize_b -> InvTxfm2d_add ->
    reconstruction in ONE launch (svt_hip_txfm_quant_roundtrip_batch), all 19 TX sizes x {8, 10} bit; N = 65536 / 16384 / 4096 blocks per launch, residuals uniform in
    +-(2^bd - 1) (seed 13596), tx types cycling through the allowed set, quantizer tables from svt_av1_build_quantizer at q in {0, 60, 120, 180, 255} and scans from
    av1_scan_orders (tests/golden/quant_tables.npz, frozen from the reference).  Before a size is timed, blocks covering every (q, tx type) pair are compared
    with the CPU checker's composition fwd -> handle_transform -> quantize -> inverse (qcoeff, dqcoeff, eob, recon)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from quant_common import oracle_roundtrip, real_qparams, real_scans
    oracle = oracle_lib()
    g = np.random.default_rng(13596)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).reshape(-1)).cuda()  # noqa: E731
    rows, fr, checked_sizes, checked, rl = {}, [], 0, 0, {}
    for ts, (w, h) in enumerate(pkg.TX_SIZES):
        big = max(w, h)
        n = 65536 if big <= 16 else (16384 if big == 32 else 4096)
        ncoef, pels = min(w, 32) * min(h, 32), w * h
        ls = int(pels > 256) + int(pels > 1024)
        types = pkg.allowed_tx_types(ts)
        scans, iscans = real_scans(ts)
        for bd in (8, 10):
            amp = (1 << bd) - 1
            dt = np.uint16 if bd > 8 else np.uint8
            plist = real_qparams(bd, False)
            params = np.zeros(len(plist), dtype=pkg.QuantParams)
            for i, P in enumerate(plist):
                params[i] = (P["zbin"], P["round"], P["quant"], P["quant_shift"], P["dequant"], ls)
            res = g.integers(-amp, amp + 1, n * pels, dtype=np.int16)
            pred = g.integers(0, amp + 1, n * pels).astype(dt)
            rd = np.zeros(n, dtype=pkg.RoundtripDesc)
            rd["in_off"] = rd["pred_off"] = rd["recon_off"] = np.arange(n, dtype=np.uint64) * pels
            rd["in_stride"] = rd["pred_stride"] = rd["recon_stride"] = w
            rd["tx_type"] = np.array(types, np.uint8)[np.arange(n) % len(types)]
            rd["qparam_idx"] = (np.arange(n) // len(types)) % len(plist)
            rd["iscan_idx"] = rd["tx_type"]
            d_res, d_pred, d_rd, d_par, d_is = t(res), t(pred), t(rd), t(params), t(iscans)
            d_rec = torch.zeros(n * pels * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda")
            d_q, d_dq = torch.zeros(n * ncoef, dtype=torch.int32, device="cuda"), torch.zeros(n * ncoef, dtype=torch.int32, device="cuda")
            d_eob = torch.zeros(n, dtype=torch.int16, device="cuda")
            mode = 1 if bd > 8 else 0
            fn = lambda dq=True: lib.svt_hip_txfm_quant_roundtrip_batch(d_res.data_ptr(), d_pred.data_ptr(), d_rec.data_ptr(), d_rd.data_ptr(), n, ts, bd, mode,  # noqa: E731
                                                                        d_par.data_ptr(), d_is.data_ptr(), None, None, d_q.data_ptr(), d_dq.data_ptr() if dq else None,
                                                                        d_eob.data_ptr(), stream)
            fn()
            torch.cuda.synchronize()
            if oracle is not None:
                hq, hdq, he = d_q.cpu().numpy().reshape(n, ncoef), d_dq.cpu().numpy().reshape(n, ncoef), d_eob.cpu().numpy().view(np.uint16)
                hrec = d_rec.cpu().numpy().view(dt).reshape(n, h, w)
                for b in list(range(min(len(types) * len(plist), 20 if pels >= 1024 else 80))) + [n - 1]:
                    tt, qi = int(rd["tx_type"][b]), int(rd["qparam_idx"][b])
                    wq, wdq, weob, wrec = oracle_roundtrip(oracle, res[b * pels:(b + 1) * pels], w, pred[b * pels:(b + 1) * pels].astype(np.uint16), w, w, h, tt, ts, bd,
                                                           mode, plist[qi], np.ascontiguousarray(scans[tt]), None, None, ls)
                    tag = "config3 %dx%d bd%d block %d " % (w, h, bd, b)
                    checked += must_equal(tag + "qcoeff", hq[b], wq) + must_equal(tag + "dqcoeff", hdq[b], wdq) + must_equal(tag + "recon", hrec[b].astype(np.uint16), wrec)
                    must_equal(tag + "eob", [int(he[b])], [weob])
                checked_sizes += 1
            per, _ = time_leg(torch, lambda: fn(False), a.min_leg_s / 4)
            b_alg = (2 + 2 * np.dtype(dt).itemsize) * pels + 4 * ncoef + 2  # residual in, prediction in, reconstruction out, qcoeff + eob out (10 B/px at 16-bit pixels)
            frac = n * b_alg / per / 1e9 / HBM_PEAK_GBS
            fr.append(frac)
            rows["%dx%d_bd%d" % (w, h, bd)] = [round(n / per / 1e6, 1), round(frac, 3)]
            rl["%dx%d_bd%d" % (w, h, bd)] = roofline(n * b_alg, per, "txfm_roundtrip_kernel<%s, %d, %d>" % ("unsigned short" if bd > 8 else "unsigned char", w, h))
            del d_res, d_pred, d_rec, d_q, d_dq
    worst = min(rl, key=lambda k: rl[k]["frac"])
    return {"unit": "[Mblocks/s, fraction of the 8 TB/s HBM peak at SURVEY 8(d)'s 10 B/px, VALU fraction] per TX size and bit depth, fused single launch", "sizes": rows,
            "size_rooflines": rl, "roofline": dict(rl[worst], size=worst, note="the size furthest below the HBM roof; every size under size_rooflines"),
            "sizes_checked": checked_sizes, "parity_checked_values": checked, "hbm_frac_min_max": [round(min(fr), 3), round(max(fr), 3)],
            "tables": "svt_av1_build_quantizer q in {0,60,120,180,255} + av1_scan_orders (tests/golden/quant_tables.npz)", "bound": "VALU (butterfly network, DESIGN.md 4.2)"}


def bench_cdef(torch, lib, pkg, stream, a, cpu):
    """config 4: CDEF over a 4K 10-bit luma plane: strength search (all 64 luma strengths) and apply (pri 4, sec 2)."""
    Wc, Hc, bd = 3840, 2160, 10
    g = np.random.default_rng(4)

    def synth(w, h):
        yy, xx = np.mgrid[0:h, 0:w]
        pl = np.clip(((xx * 2 + yy * 3) % 1024) // 2 + (((xx // 8 + yy // 8) % 5) << 5) + g.integers(-16, 17, (h, w)), 0, 1023).astype(np.uint16)
        return pl, np.clip(pl.astype(np.int32) + g.integers(-6, 7, pl.shape), 0, 1023).astype(np.uint16)
    cands = [(pr, sc) for pr in range(16) for sc in (0, 1, 2, 4)]
    pri, sec = np.array([c[0] for c in cands], np.int32), np.array([c[1] for c in cands], np.int32)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).reshape(-1)).cuda()  # noqa: E731

    def setup(w, h):
        plane, src = synth(w, h)
        nhfb, nvfb = (w + 63) // 64, (h + 63) // 64
        nfb = nhfb * nvfb
        skip = np.zeros((nvfb * 8, nhfb * 8), np.uint8)
        D = dict(plane=plane, src=src, nfb=nfb, skip=skip, d_pl=t(plane), d_src=t(src), d_out=t(plane), d_skip=t(skip), d_pri=t(pri), d_sec=t(sec),
                 d_dir=torch.zeros(nfb * 64, dtype=torch.uint8, device="cuda"), d_var=torch.zeros(nfb * 64, dtype=torch.int32, device="cuda"),
                 d_mse=torch.zeros(nfb * 64, dtype=torch.int64, device="cuda"), apri=t(np.full(nfb, 4, np.int32)), asec=t(np.full(nfb, 2, np.int32)), w=w, h=h)
        return D

    def params(D, mode):
        return pkg.CdefParams(D["d_pl"].data_ptr(), D["d_src"].data_ptr(), D["d_out"].data_ptr(), D["w"], D["w"], D["w"], D["w"], D["h"], 0, 0, 0, 1, bd - 8, 4, 4, 1,
                              64 if mode else 0, D["d_skip"].data_ptr(), (D["d_pri"] if mode else D["apri"]).data_ptr(), (D["d_sec"] if mode else D["asec"]).data_ptr(),
                              D["d_dir"].data_ptr(), D["d_var"].data_ptr(), D["d_mse"].data_ptr())
    # ---- parity on a small frame (same kernels, 5 x 3 filter blocks incl. partial ones): all 64 strengths + the apply pass against the CPU checker
    checked, oracle = 0, oracle_lib()
    if oracle is not None:
        S = setup(304, 168)
        for mode in (1, 0):
            P = params(S, mode)
            lib.svt_hip_cdef_frame(mode, C.byref(P), stream)
            torch.cuda.synchronize()
            o_out, o_dir, o_var = S["plane"].copy(), np.zeros(S["nfb"] * 64, np.uint8), np.zeros(S["nfb"] * 64, np.int32)
            o_mse = np.zeros(S["nfb"] * 64, np.uint64)
            pr, sc = (pri, sec) if mode else (np.full(S["nfb"], 4, np.int32), np.full(S["nfb"], 2, np.int32))
            oracle.oracle_cdef_frame(mode, vp(S["plane"]), 304, vp(S["src"]), 304, vp(o_out), 304, 304, 168, 0, 0, 0, 1, bd - 8, 4, 4, 1, vp(S["skip"]), vp(pr), vp(sc),
                                     64 if mode else 0, vp(o_dir), vp(o_var), vp(o_mse))
            if mode:
                checked += must_equal("cdef search mse", S["d_mse"].cpu().numpy().view(np.uint64), o_mse) + must_equal("cdef dir", S["d_dir"].cpu().numpy(), o_dir)
            else:
                checked += must_equal("cdef apply", S["d_out"].cpu().numpy().view(np.uint16).reshape(168, 304), o_out)
    D = setup(Wc, Hc)
    out = {}
    n8 = (Wc // 8) * (Hc // 8)
    for mode, name in ((1, "cdef_search_4k10_64strengths"), (0, "cdef_apply_4k10")):
        P = params(D, mode)
        fn = lambda: lib.svt_hip_cdef_frame(mode, C.byref(P), stream)  # noqa: E731
        per, reps = time_leg(torch, fn, a.min_leg_s)
        units = n8 * (64 if mode else 1)
        bytes_alg = Wc * Hc * 2 * 2 + (D["nfb"] * 64 * 8 if mode else 0)  # apply: read + write; search: recon + source, 8 B per (fb, strength)
        out[name] = {"launches_per_timed_batch": reps, "value": units / per / 1e6, "unit": "M(8x8 block x strength)/s" if mode else "M(8x8 blocks)/s", "frames_per_s": 1 / per,
                     "parity_checked_values": checked,
                     "roofline": roofline(bytes_alg, per, "cdef_frame_kernel<unsigned short, %d, %s>" % (mode, (os.environ.get("SVT_HIP_CDEF_MINB") or "3") if mode else "4"),
                                          algorithmic_bytes_per_frame=bytes_alg)}
    # ---- the whole 4:2:0 PICTURE of config 4 ("3840x2160 4:2:0 ... skip map all-non-skip and a 25 %-skip variant"): Y + U + V per call, both skip maps.
    # The legs above time one luma plane (the unit the per-plane kernels are designed and priced in); these are what a picture costs.
    Dc = [setup(Wc // 2, Hc // 2) for _ in range(2)]
    for dc in Dc:  # a chroma plane has the LUMA picture's filter-block grid (blocks of 64 >> xdec samples): per-block tables are sized by it
        dc.update(d_mse=torch.zeros(D["nfb"] * 64, dtype=torch.int64, device="cuda"), apri=t(np.full(D["nfb"], 4, np.int32)), asec=t(np.full(D["nfb"], 2, np.int32)))
    gs = np.random.default_rng(41)
    skip25 = (gs.random(D["skip"].shape) < 0.25).astype(np.uint8)  # 8x8 luma units left out (svt_sb_compute_cdef_list); chroma planes share the luma map
    d_skip25 = t(skip25)

    def pparams(Dp, mode, pl, skip_t, lum):
        dec = 1 if pl else 0
        return pkg.CdefParams(Dp["d_pl"].data_ptr(), Dp["d_src"].data_ptr(), Dp["d_out"].data_ptr(), Dp["w"], Dp["w"], Dp["w"], Dp["w"], Dp["h"], dec, dec, pl, 1, bd - 8, 4, 4, 1,
                              64 if mode else 0, skip_t.data_ptr(), (Dp["d_pri"] if mode else Dp["apri"]).data_ptr(), (Dp["d_sec"] if mode else Dp["asec"]).data_ptr(),
                              lum["d_dir"].data_ptr(), lum["d_var"].data_ptr(), Dp["d_mse"].data_ptr())
    for tag, skip_t, frac_on in (("", D["d_skip"], 1.0), ("_skip25", d_skip25, float(1.0 - skip25.mean()))):
        for mode, nm in ((1, "cdef_search_4k10_420%s" % tag), (0, "cdef_apply_4k10_420%s" % tag)):
            PP = [pparams(D, mode, 0, skip_t, D), pparams(Dc[0], mode, 1, skip_t, D), pparams(Dc[1], mode, 2, skip_t, D)]

            def fnp(PP=PP, mode=mode):
                for q in PP:  # luma first: it writes the directions / variances the chroma planes read (cdef.c:367-386)
                    lib.svt_hip_cdef_frame(mode, C.byref(q), stream)
            perp, _ = time_leg(torch, fnp, a.min_leg_s)
            px = Wc * Hc * 3 // 2
            bytes_alg = int(px * 2 * 2 * (frac_on if not mode else 1.0)) + (3 * D["nfb"] * 64 * 8 if mode else 0)  # apply moves the filtered blocks; the search reads both pictures whole
            out[nm] = {"frames_per_s": 1 / perp, "frame_us": perp * 1e6, "planes": 3, "units_on": round(frac_on, 4), "parity_checked_values": checked,
                       "value": (n8 * 1.5 * frac_on) * (64 if mode else 1) / perp / 1e6, "unit": "M(8x8 block x strength)/s" if mode else "M(8x8 blocks)/s",
                       "roofline": roofline(bytes_alg, perp, "cdef_frame_kernel<unsigned short, %d, ...> x 3 planes" % mode, algorithmic_bytes_per_frame=bytes_alg,
                                            note="one 4:2:0 picture = three launches; tests/test_cdef.py compares the chroma planes and skip maps with the checker")}
    # apply with the directions the search pass just wrote (mode 2: what the CDEF stage runs after its strength search)
    P2 = params(D, 0)
    fn2 = lambda: lib.svt_hip_cdef_frame(2, C.byref(P2), stream)  # noqa: E731
    per2, _ = time_leg(torch, fn2, a.min_leg_s)
    out["cdef_apply_4k10"]["frames_per_s_given_directions"] = 1 / per2
    if cpu:
        ref, oracle2 = ref_libs()
        if ref is not None and " avx2 " in open("/proc/cpuinfo").read():
            ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))
            ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
            for ptr, fnn in (("svt_aom_cdef_find_dir", "svt_aom_cdef_find_dir_avx2"), ("svt_aom_cdef_find_dir_dual", "svt_aom_cdef_find_dir_dual_avx2"),
                             ("svt_cdef_filter_block", "svt_cdef_filter_block_avx2"),
                             ("svt_cdef_filter_block_8xn_16", "svt_cdef_filter_block_8xn_16_avx2")):  # SIMD-internal pointer, NULL in a C-only setup
                C.c_void_p.in_dll(ref, ptr).value = C.cast(getattr(ref, fnn), C.c_void_p).value  # what RTCD would select with AVX2 detected
            f = oracle2.oracle_time_cdef_apply
            f.restype = C.c_uint64
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_double]
            fb = C.cast(ref.svt_cdef_filter_fb, C.c_void_p)
            pl = aligned_zeros(Wc * Hc, np.uint16).reshape(Hc, Wc)  # the AVX2 kernels use aligned loads / stores
            pl[:] = D["plane"]
            cpu_out = aligned_zeros(Wc * Hc, np.uint16).reshape(Hc, Wc)
            run = lambda i0, stp, s: f(fb, pl.ctypes.data, Wc, Wc, Hc, cpu_out.ctypes.data, 4, 2, 4, 2, i0, stp, s)  # noqa: E731
            rate, one, cores = cpu_pool(run, 3.0)
            out["cdef_apply_4k10"]["cpu_baseline"] = {"value": rate * 64 / 1e6, "unit": "M(8x8 blocks)/s", "cores": cores, "kind": "reference",
                                                      "single_thread_value": one * 64 / 1e6,
                                                      "sample": "svt_cdef_filter_fb + svt_cdef_filter_block_avx2 / find_dir_dual_avx2 over the same 4K plane, 3 s per leg"}
            if cpu_has(*AVX512) and hasattr(ref, "svt_cdef_filter_block_8xn_16_avx512"):  # what RTCD adds on an AVX-512 host (common_dsp_rtcd.c:821)
                C.c_void_p.in_dll(ref, "svt_cdef_filter_block_8xn_16").value = C.cast(ref.svt_cdef_filter_block_8xn_16_avx512, C.c_void_p).value
                rate, one, cores = cpu_pool(run, 3.0)
                out["cdef_apply_4k10"]["cpu_baseline_avx512"] = {"value": rate * 64 / 1e6, "unit": "M(8x8 blocks)/s", "cores": cores, "kind": "reference",
                                                                 "single_thread_value": one * 64 / 1e6,
                                                                 "sample": "as the AVX2 leg with svt_cdef_filter_block_8xn_16 = its _avx512 variant, 3 s per leg"}
    return out


def check_lr_small(torch, lib, pkg, stream):
    """parity of the loop-restoration frame kernel on a small 10-bit plane (Wiener / self-guided / none units mixed) before its 4K legs are timed"""
    oracle = oracle_lib()
    if oracle is None:
        return 0
    g = np.random.default_rng(9)
    w, h, us, bd = 328, 200, 64, 10
    yy, xx = np.mgrid[0:h, 0:w]
    plane = np.clip(((xx * 3 + yy * 2) % 1024) // 2 + g.integers(0, 256, (h, w)), 0, 1023).astype(np.uint16)
    nstripes = (h + 8 + 63) // 64
    above, below = g.integers(0, 1024, (2 * nstripes, w)).astype(np.uint16), g.integers(0, 1024, (2 * nstripes, w)).astype(np.uint16)
    nvu, nhu = max((h + us // 2) // us, 1), max((w + us // 2) // us, 1)
    units = np.zeros(nvu * nhu, dtype=pkg.LrUnit)
    for i in range(len(units)):
        f = [int(g.integers(-5, 11)), int(g.integers(-23, 9)), int(g.integers(-17, 47))]
        taps = [f[0], f[1], f[2], -2 * sum(f), f[2], f[1], f[0], 0]
        units[i] = ((1, 2, 0)[i % 3], taps, taps, int(g.integers(0, 16)), (int(g.integers(-96, 32)), int(g.integers(-32, 96))))
    want = np.zeros((h, w), np.uint16)
    oracle.oracle_lr_filter_frame(vp(plane), w, vp(above), vp(below), w, vp(want), w, w, h, 0, us, vp(units), bd, 1)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).reshape(-1)).cuda()  # noqa: E731
    d_pl, d_ab, d_bl, d_un = t(plane), t(above), t(below), t(units)
    d_out = torch.zeros(h * w, dtype=torch.int16, device="cuda")
    P = pkg.LrParams(d_pl.data_ptr(), d_ab.data_ptr(), d_bl.data_ptr(), d_out.data_ptr(), w, w, w, w, h, us, 0, 0, 1, bd, d_un.data_ptr())
    lib.svt_hip_lr_filter_frame(C.byref(P), stream)
    torch.cuda.synchronize()
    return must_equal("lr_filter_frame", d_out.cpu().numpy().view(np.uint16).reshape(h, w), want)


def self_launch(a):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks here (one process per GPU), rank 0's stdout is ours."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SVT_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out)
    sys.exit(max(rcs))


def strip_rows(total_rows, rank, world):
    """contiguous SB-row strip of `rank`: 1080p (17 rows) over 8 ranks -> 3,2,2,2,2,2,2,2"""
    base, rem = divmod(total_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def bench_frame_partition(torch, lib, pkg, stream, a, dist, rank, world, oracle):
    """north_star's frame-partition case: ONE 1080p picture per step, SB-row strips across the ranks, references resident everywhere, all-gather of the tables."""
    aw, ah = (int(v) for v in a.area.split("x"))
    planes = synth_planes(1 + a.refs, 99)            # the same picture on every rank ...
    d_planes = torch.from_numpy(planes.reshape(-1)).cuda()
    if dist is not None:
        dist.broadcast(d_planes, src=0)                # ... made so by ONE broadcast of the source + reference planes, outside the timed region
    full = pkg.me_descs_for_frame(W, H, STRIDE, PAD, PAD, aw, ah, PLANE, n_refs=a.refs, src_plane=0, ref_plane0=1)  # [ref][sb_row][sb_col]
    sbs_x, sbs_y = (W + 63) // 64, (H + 63) // 64
    r0, r1 = strip_rows(sbs_y, rank, world)
    max_rows = strip_rows(sbs_y, 0, world)[1]
    mine = np.concatenate([full[r * sbs_x * sbs_y + r0 * sbs_x:r * sbs_x * sbs_y + r1 * sbs_x] for r in range(a.refs)]) if r1 > r0 else full[:0]
    n_mine, n_pad = len(mine), a.refs * max_rows * sbs_x
    d_descs = torch.from_numpy(np.ascontiguousarray(mine).view(np.uint8).copy()).cuda() if n_mine else torch.zeros(32, dtype=torch.uint8, device="cuda")
    local = torch.zeros(2 * n_pad * 85, dtype=torch.int32, device="cuda")  # [sad | mv], padded to the largest strip so that the all-gather is regular
    gathered = torch.zeros(world * 2 * n_pad * 85, dtype=torch.int32, device="cuda")

    def step():
        if n_mine:
            lib.svt_hip_me_fullpel_search_batch(d_planes.data_ptr(), d_planes.data_ptr(), d_descs.data_ptr(), n_mine, aw, ah, 0, local.data_ptr(),
                                                local.data_ptr() + n_pad * 85 * 4, None, stream)
        if dist is not None:
            dist.all_gather_into_tensor(gathered, local)
    step()
    torch.cuda.synchronize()
    checked = 0
    if rank == 0:  # the assembled picture-wide table against the CPU checker (a sample of SBs from every strip)
        g_all = (gathered if dist is not None else local).cpu().numpy().view(np.uint32).reshape(world if dist is not None else 1, 2, n_pad, 85)
        if oracle is not None:
            for rk in range(world):
                q0, q1 = strip_rows(sbs_y, rk, world)
                for (rf, row, col) in ((0, q0, 0), (a.refs - 1, q1 - 1, sbs_x - 1)):
                    if q1 <= q0:
                        continue
                    d = full[rf * sbs_x * sbs_y + row * sbs_x + col]
                    ws, wm = np.zeros(85, np.uint32), np.zeros(85, np.uint32)
                    oracle.oracle_me_fullpel_search(vp(planes, int(d["src_off"])), STRIDE, vp(planes, int(d["ref_off"])), STRIDE, int(d["x_origin"]), int(d["y_origin"]), aw, ah, 0,
                                                    vp(ws), vp(wm))
                    k = rf * (q1 - q0) * sbs_x + (row - q0) * sbs_x + col
                    checked += must_equal("frame partition sad", g_all[rk, 0, k], ws) + must_equal("frame partition mv", g_all[rk, 1, k], wm)
    L = calibrate_launches(torch, step, a.steps, a.min_leg_s, dist)  # pictures per step, so that the timed region lasts >= min_leg_s

    def step_l():
        for _ in range(L):
            step()
    wall, dev = time_steps(torch, step_l, a.steps, a.warmup, dist)
    t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    n_total = len(full)
    return {"value": n_total * aw * ah * a.steps * L / wall / 1e6, "unit": "Mblocks/s (whole job: one 1080p picture x %d references per launch)" % a.refs, "scaling": "strong",
            "ms_per_step": wall / a.steps * 1e3, "pictures_per_step": L, "timed_region_s": wall, "pictures_per_s": a.steps * L / wall, "strip_rows": [strip_rows(sbs_y, k, world)[1] - strip_rows(sbs_y, k, world)[0] for k in range(world)],
            "collective": ("all_gather_into_tensor over RCCL, %d B per rank per step" % (2 * n_pad * 85 * 4)) if dist is not None else "none (1 GPU)",
            "parity_checked_values": checked}


def bench_filter_partition(torch, lib, pkg, stream, a, dist, rank, world, oracle, size=(3840, 2160)):
    """The in-loop filter half of the frame-partition case: ONE 4K 10-bit luma plane per step, CDEF apply over this rank's strip of filter-block rows
    (svt_hip_cdef_frame_rows: 34 rows -> 5,5,4,4,4,4,4,4 at N = 8; the strip's tiles read their 3-row halos from the full deblocked plane every rank holds) and
    loop restoration over its range of 64-row stripes (svt_hip_lr_filter_frame_stripes), each followed by an all-gather of the filtered strips so that every rank
    holds the whole filtered plane (the next stage / the reference picture needs it).  Before timing, the assembled planes are compared with the CPU checker."""
    Wc, Hc = size
    bd = 10
    g = np.random.default_rng(11)
    yy, xx = np.mgrid[0:Hc, 0:Wc]
    plane = np.clip(((xx * 2 + yy * 3) % 1024) // 2 + (((xx // 8 + yy // 8) % 5) << 5) + g.integers(-16, 17, (Hc, Wc)), 0, 1023).astype(np.uint16)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).reshape(-1)).cuda()  # noqa: E731
    d_pl = t(plane)
    if dist is not None:
        dist.broadcast(d_pl, src=0)  # the deblocked plane on every rank, once, outside the timed region
    nhfb, nvfb = (Wc + 63) // 64, (Hc + 63) // 64
    nfb = nhfb * nvfb
    skip = np.zeros((nvfb * 8, nhfb * 8), np.uint8)
    apri, asec = np.full(nfb, 4, np.int32), np.full(nfb, 2, np.int32)
    d_skip, d_pri, d_sec = t(skip), t(apri), t(asec)
    d_dir, d_var = torch.zeros(nfb * 64, dtype=torch.uint8, device="cuda"), torch.zeros(nfb * 64, dtype=torch.int32, device="cuda")
    d_cdef = d_pl.clone()
    Pc = pkg.CdefParams(d_pl.data_ptr(), d_pl.data_ptr(), d_cdef.data_ptr(), Wc, Wc, Wc, Wc, Hc, 0, 0, 0, 1, bd - 8, 4, 4, 1, 0, d_skip.data_ptr(), d_pri.data_ptr(), d_sec.data_ptr(),
                        d_dir.data_ptr(), d_var.data_ptr(), None)
    r0, r1 = strip_rows(nvfb, rank, world)
    max_rows = (strip_rows(nvfb, 0, world)[1]) * 64
    # loop restoration: stripes of 64 rows offset by 8 (stripe i = rows i * 64 - 8 .. (i + 1) * 64 - 8)
    us = 256
    nstripes = (Hc + 8 + 63) // 64
    s0, s1 = strip_rows(nstripes, rank, world)
    nvu, nhu = max((Hc + us // 2) // us, 1), max((Wc + us // 2) // us, 1)
    units = np.zeros(nvu * nhu, dtype=pkg.LrUnit)
    for i in range(len(units)):
        f = [int(g.integers(-5, 11)), int(g.integers(-23, 9)), int(g.integers(-17, 47))]
        taps = [f[0], f[1], f[2], -2 * sum(f), f[2], f[1], f[0], 0]
        units[i] = ((1, 2, 0)[i % 3], taps, taps, int(g.integers(0, 16)), (int(g.integers(-96, 32)), int(g.integers(-32, 96))))
    above, below = g.integers(0, 1024, (2 * nstripes, Wc)).astype(np.uint16), g.integers(0, 1024, (2 * nstripes, Wc)).astype(np.uint16)
    d_ab, d_bl, d_un = t(above), t(below), t(units)
    d_lr = torch.zeros(Hc * Wc, dtype=torch.int16, device="cuda")
    row_bytes = Wc * 2
    lr_rows = lambda k: (max(strip_rows(nstripes, k, world)[0] * 64 - 8, 0), min(strip_rows(nstripes, k, world)[1] * 64 - 8, Hc))  # noqa: E731
    max_lr_rows = max(lr_rows(k)[1] - lr_rows(k)[0] for k in range(world))
    loc_c = torch.zeros(max_rows * row_bytes, dtype=torch.uint8, device="cuda")
    loc_l = torch.zeros(max_lr_rows * row_bytes, dtype=torch.uint8, device="cuda")
    gat_c = torch.zeros(world * max_rows * row_bytes, dtype=torch.uint8, device="cuda")
    gat_l = torch.zeros(world * max_lr_rows * row_bytes, dtype=torch.uint8, device="cuda")
    cdef_u8 = d_cdef.view(torch.uint8) if d_cdef.dtype != torch.uint8 else d_cdef
    lr_u8 = d_lr.view(torch.uint8)
    Pl = pkg.LrParams(d_cdef.data_ptr(), d_ab.data_ptr(), d_bl.data_ptr(), d_lr.data_ptr(), Wc, Wc, Wc, Wc, Hc, us, 0, 0, 1, bd, d_un.data_ptr())

    def step():
        if r1 > r0:
            lib.svt_hip_cdef_frame_rows(0, C.byref(Pc), r0, r1, stream)
        if dist is not None:  # assemble the CDEF output everywhere: the LR stage reads 3 rows beyond its own stripes
            y0, y1 = r0 * 64, min(r1 * 64, Hc)
            loc_c[:(y1 - y0) * row_bytes].copy_(cdef_u8[y0 * row_bytes:y1 * row_bytes])
            dist.all_gather_into_tensor(gat_c, loc_c)
            for k in range(world):
                q0, q1 = strip_rows(nvfb, k, world)
                a0, a1 = q0 * 64, min(q1 * 64, Hc)
                if k != rank and a1 > a0:
                    cdef_u8[a0 * row_bytes:a1 * row_bytes].copy_(gat_c[k * max_rows * row_bytes:k * max_rows * row_bytes + (a1 - a0) * row_bytes])
        if s1 > s0:
            lib.svt_hip_lr_filter_frame_stripes(C.byref(Pl), s0, s1, stream)
        if dist is not None:
            y0, y1 = lr_rows(rank)
            loc_l[:(y1 - y0) * row_bytes].copy_(lr_u8[y0 * row_bytes:y1 * row_bytes])
            dist.all_gather_into_tensor(gat_l, loc_l)
            for k in range(world):
                a0, a1 = lr_rows(k)
                if k != rank and a1 > a0:
                    lr_u8[a0 * row_bytes:a1 * row_bytes].copy_(gat_l[k * max_lr_rows * row_bytes:k * max_lr_rows * row_bytes + (a1 - a0) * row_bytes])
    step()
    torch.cuda.synchronize()
    checked = 0
    if oracle is not None and rank == 0:  # the assembled planes against the CPU checker on a band of rows around every strip boundary
        want_c = plane.copy()
        o_dir, o_var, o_mse = np.zeros(nfb * 64, np.uint8), np.zeros(nfb * 64, np.int32), np.zeros(1, np.uint64)
        oracle.oracle_cdef_frame(0, vp(plane), Wc, vp(plane), Wc, vp(want_c), Wc, Wc, Hc, 0, 0, 0, 1, bd - 8, 4, 4, 1, vp(skip), vp(apri), vp(asec), 0, vp(o_dir), vp(o_var), vp(o_mse))
        got_c = d_cdef.cpu().numpy().view(np.uint16).reshape(Hc, Wc)
        checked += must_equal("strip-partitioned CDEF apply", got_c, want_c)
        want_l = np.zeros((Hc, Wc), np.uint16)
        oracle.oracle_lr_filter_frame(vp(want_c), Wc, vp(above), vp(below), Wc, vp(want_l), Wc, Wc, Hc, 0, us, vp(units), bd, 1)
        checked += must_equal("strip-partitioned loop restoration", d_lr.cpu().numpy().view(np.uint16).reshape(Hc, Wc), want_l)
    L = calibrate_launches(torch, step, a.steps, a.min_leg_s, dist)

    def step_l():
        for _ in range(L):
            step()
    wall, dev = time_steps(torch, step_l, a.steps, a.warmup, dist)
    tm = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    wall = float(tm.item())
    return {"planes_per_s": a.steps * L / wall, "ms_per_plane": wall / (a.steps * L) * 1e3, "plane": "%dx%d 10-bit luma" % (Wc, Hc), "scaling": "strong",
            "cdef_fb_row_strips": [strip_rows(nvfb, k, world)[1] - strip_rows(nvfb, k, world)[0] for k in range(world)],
            "lr_stripe_ranges": [strip_rows(nstripes, k, world)[1] - strip_rows(nstripes, k, world)[0] for k in range(world)],
            "collective": ("2 x all_gather_into_tensor over RCCL per plane (%d + %d B per rank)" % (max_rows * row_bytes, max_lr_rows * row_bytes)) if dist is not None else "none (1 GPU)",
            "parity_checked_values": checked}


def bench_c_partition(torch, lib, pkg, stream, a, world, oracle):
    """The library's OWN frame partition (csrc/partition.hip: svt_hip_frame_partition_{me,cdef,lr}) driven from ONE process -- rank 0 -- over min(GPUs, N) devices: the
    home device is this rank's, every other device is a peer with its own stream, `done` event and arena, inputs broadcast and strips gathered by hipMemcpyPeerAsync
    (xGMI between GPUs).  With a single GPU (N = 1) the peers are LOGICAL devices of that GPU (svt_hip_set_virtual_devices): the same code, streams, events and peer
    copies, running concurrently on one device -- it measures the protocol's overhead, not a speed-up.  Every partitioned result is compared with the single-device
    call's (bit-equal) before timing; the ME tables also with the CPU checker."""
    phys = lib.svt_hip_physical_device_count()
    n_dev = min(phys, world)
    virtual = n_dev < 2
    if virtual:
        n_dev = 4
        assert lib.svt_hip_set_virtual_devices(n_dev) == 0
    devices = list(range(n_dev))
    part = lib.svt_hip_frame_partition_create((C.c_int * n_dev)(*devices), n_dev)
    if not part:
        return {"error": "svt_hip_frame_partition_create refused devices %s" % devices}
    t8 = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).reshape(-1)).cuda()  # noqa: E731
    res = {"devices": n_dev, "virtual_peers": virtual, "home": 0}
    checked = 0
    min_s = min(a.min_leg_s, 0.15)

    def stats():
        c, bi, bo = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        lib.svt_hip_frame_partition_stats(part, C.byref(c), C.byref(bi), C.byref(bo))
        return c.value, bi.value, bo.value

    def leg(name, single, parted, outs, extra_check=None):
        nonlocal checked
        for t in outs:
            t.zero_()
        single()
        torch.cuda.synchronize()
        want = [t.cpu().numpy().copy() for t in outs]
        for t in outs:
            t.zero_()
        c0 = stats()
        home = torch.cuda.current_device()
        parted_raw = parted

        def parted():  # (real peers: whatever the library leaves current, torch's own calls -- synchronize(), events -- must see the home device)
            rc = parted_raw()
            torch.cuda.set_device(home)
            return rc
        assert parted() == 0
        torch.cuda.synchronize()
        c1 = stats()
        for w, t in zip(want, outs):
            checked += must_equal("c_partition " + name, t.cpu().numpy(), w)
        if extra_check:
            checked += extra_check()
        t_single, _ = time_leg(torch, single, min_s, batches=3)
        t_part, _ = time_leg(torch, parted, min_s, batches=3)
        res[name] = {"ms_single_device": t_single * 1e3, "ms_partition": t_part * 1e3, "peer_bytes_in_per_call": c1[1] - c0[1], "peer_bytes_out_per_call": c1[2] - c0[2]}

    # ---- ME: one 1080p picture x its references (the items ordered SB row after SB row within a reference) ----
    aw, ah = (int(v) for v in a.area.split("x"))
    planes = synth_planes(1 + a.refs, 99)
    d_pl = torch.from_numpy(planes.reshape(-1)).cuda()
    full = pkg.me_descs_for_frame(W, H, STRIDE, PAD, PAD, aw, ah, PLANE, n_refs=a.refs, src_plane=0, ref_plane0=1)
    n = len(full)
    d_d = t8(full)
    d_s, d_m = torch.zeros(n * 85, dtype=torch.int32, device="cuda"), torch.zeros(n * 85, dtype=torch.int32, device="cuda")

    def me_check():
        k = 0
        if oracle is None:
            return 0
        hs, hm = d_s.cpu().numpy().view(np.uint32).reshape(n, 85), d_m.cpu().numpy().view(np.uint32).reshape(n, 85)
        for i in np.linspace(0, n - 1, 4 * n_dev).astype(int):  # items of every strip
            d = full[i]
            ws, wm = np.zeros(85, np.uint32), np.zeros(85, np.uint32)
            oracle.oracle_me_fullpel_search(vp(planes, int(d["src_off"])), STRIDE, vp(planes, int(d["ref_off"])), STRIDE, int(d["x_origin"]), int(d["y_origin"]), aw, ah, 0, vp(ws), vp(wm))
            k += must_equal("c_partition me sad", hs[i], ws) + must_equal("c_partition me mv", hm[i], wm)
        return k
    leg("me_1080p",
        lambda: lib.svt_hip_me_fullpel_search_batch(d_pl.data_ptr(), d_pl.data_ptr(), d_d.data_ptr(), n, aw, ah, 0, d_s.data_ptr(), d_m.data_ptr(), None, stream),
        lambda: lib.svt_hip_frame_partition_me(part, d_pl.data_ptr(), d_pl.numel(), d_pl.data_ptr(), d_pl.numel(), d_d.data_ptr(), n, aw, ah, 0, d_s.data_ptr(), d_m.data_ptr(), None, stream),
        [d_s, d_m], me_check)
    res["me_1080p"]["sb_refs"] = n

    # ---- CDEF apply + loop restoration of one 4K 10-bit luma plane ----
    Wc, Hc, bd = 3840, 2160, 10
    g = np.random.default_rng(11)
    yy, xx = np.mgrid[0:Hc, 0:Wc]
    plane = np.clip(((xx * 2 + yy * 3) % 1024) // 2 + (((xx // 8 + yy // 8) % 5) << 5) + g.integers(-16, 17, (Hc, Wc)), 0, 1023).astype(np.uint16)
    d_rec = t8(plane)
    nhfb, nvfb = (Wc + 63) // 64, (Hc + 63) // 64
    nfb = nhfb * nvfb
    d_skip, d_pri, d_sec = t8(np.zeros((nvfb * 8, nhfb * 8), np.uint8)), t8(np.full(nfb, 4, np.int32)), t8(np.full(nfb, 2, np.int32))
    d_dir, d_var = torch.zeros(nfb * 64, dtype=torch.uint8, device="cuda"), torch.zeros(nfb * 64, dtype=torch.int32, device="cuda")
    d_out = d_rec.clone()
    Pc = pkg.CdefParams(d_rec.data_ptr(), d_rec.data_ptr(), d_out.data_ptr(), Wc, Wc, Wc, Wc, Hc, 0, 0, 0, 1, bd - 8, 4, 4, 1, 0, d_skip.data_ptr(), d_pri.data_ptr(), d_sec.data_ptr(),
                        d_dir.data_ptr(), d_var.data_ptr(), None)

    def cdef_single():
        d_out.copy_(d_rec)  # (apply writes the filtered units onto a pre-copied plane)
        lib.svt_hip_cdef_frame(0, C.byref(Pc), stream)

    def cdef_part():
        d_out.copy_(d_rec)
        return lib.svt_hip_frame_partition_cdef(part, 0, C.byref(Pc), stream)
    leg("cdef_apply_4k10", cdef_single, cdef_part, [d_out, d_dir, d_var])
    us = 256
    nstripes = (Hc + 8 + 63) // 64
    nvu, nhu = max((Hc + us // 2) // us, 1), max((Wc + us // 2) // us, 1)
    units = np.zeros(nvu * nhu, dtype=pkg.LrUnit)
    for i in range(len(units)):
        f = [int(g.integers(-5, 11)), int(g.integers(-23, 9)), int(g.integers(-17, 47))]
        taps = [f[0], f[1], f[2], -2 * sum(f), f[2], f[1], f[0], 0]
        units[i] = ((1, 2, 0)[i % 3], taps, taps, int(g.integers(0, 16)), (int(g.integers(-96, 32)), int(g.integers(-32, 96))))
    d_ab, d_bl, d_un = t8(g.integers(0, 1024, (2 * nstripes, Wc)).astype(np.uint16)), t8(g.integers(0, 1024, (2 * nstripes, Wc)).astype(np.uint16)), t8(units)
    d_lr = torch.zeros(Hc * Wc * 2, dtype=torch.uint8, device="cuda")
    Pl = pkg.LrParams(d_rec.data_ptr(), d_ab.data_ptr(), d_bl.data_ptr(), d_lr.data_ptr(), Wc, Wc, Wc, Wc, Hc, us, 0, 0, 1, bd, d_un.data_ptr())
    leg("lr_4k10", lambda: lib.svt_hip_lr_filter_frame(C.byref(Pl), stream), lambda: lib.svt_hip_frame_partition_lr(part, C.byref(Pl), stream), [d_lr])
    torch.cuda.synchronize()
    lib.svt_hip_frame_partition_destroy(part)
    res["parity_checked_values"] = checked
    res["note"] = ("peers are logical devices of the one GPU (SVT_HIP_VIRTUAL_DEVICES): protocol overhead on one device, not a scaling figure" if virtual
                   else "peers are GPUs: inputs and strips cross xGMI by hipMemcpyPeerAsync, home device as root")
    return res


def emit(out, a):
    """stdout gets ONE line of at most bench_line.MAX_LINE bytes (what the driver records and parses); the full object -- every leg with its roofline, CPU baseline,
    parity counts, the encoder's per-stage statistics -- goes to bench_detail.json (gpurun_out/ when it exists, else the working directory) and to stderr."""
    import bench_line
    detail = json.dumps(out)
    path = None
    for d in (os.path.join(ROOT, "gpurun_out"), os.getcwd(), "/tmp"):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_detail%s.json" % ("" if (a.mode == "frames" and a.gpus == 1) else "_%s_n%d" % (a.mode, a.gpus)))
            with open(path, "w") as f:
                f.write(detail + "\n")
            break
        except OSError:
            path = None
    out["detail"] = os.path.relpath(path, ROOT) if path and path.startswith(ROOT) else path
    sys.stderr.write("BENCH_DETAIL " + detail + "\n")
    sys.stderr.flush()
    sys.stdout.write(bench_line.compact(out) + "\n")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=("frames", "strips"), default="frames", help="what `value` reports at N > 1 (both are measured)")
    ap.add_argument("--frames", type=int, default=32, help="source frames per step and per GPU")
    ap.add_argument("--refs", type=int, default=4)
    ap.add_argument("--area", type=str, default="16x9")
    ap.add_argument("--probe", action="store_true", help="print VALU issue rates and exit")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true", help="profiling passes only: skip the in-run comparisons with the CPU checker")
    ap.add_argument("--only-me", action="store_true", help="skip the per-kernel legs (profiling passes over the dominant kernel)")
    ap.add_argument("--min-leg-s", type=float, default=MIN_TIMED_S, help="minimum device time of every timed region (a step = as many launches as that takes)")
    ap.add_argument("--no-pmc", action="store_true", help="skip this run's own rocprofv3 --pmc child passes (roofline.traffic then comes from the committed summary)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--legs", type=str, default="", help="comma list restricting the per-kernel legs (me, sad, txfm, config3, cdef, lr, hme, session, tf, tpl, tpl1, tfpic, lrstats, cdefchain, lrsearch, cpart, satd): A/B measurements")
    ap.add_argument("--extra", action="store_true", help="also sweep the other search areas / sub_sad and the remaining stages (reported under kernels)")
    a = ap.parse_args()
    if a.gpus > 1 and "RANK" not in os.environ:
        self_launch(a)
    global NO_CHECK, LIVE_PMC
    if a.pmc_child:  # one of live_pmc()'s passes: every GPU leg of the default line, a few launches each, no CPU legs, no checks, no nested PMC
        a.no_cpu = a.no_parity_check = a.no_pmc = True
        a.steps, a.warmup, a.min_leg_s = 3, 1, 0.0
    NO_CHECK = a.no_parity_check
    if not a.no_pmc and a.gpus == 1 and "RANK" not in os.environ:
        LIVE_PMC = live_pmc(a)  # before this process touches the GPU; None -> fallback to the committed summary

    import torch
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch.distributed as dist  # RCCL: barrier, max-over-ranks timing, and the frame-partition all-gather
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        assert dist.get_world_size() == world == a.gpus or "SVT_BENCH_CHILD" not in os.environ
    torch.cuda.set_device(local)
    pkg = entry._pkg()
    lib = pkg.load(init_device=local)
    stream = torch.cuda.current_stream().cuda_stream
    aw, ah = (int(v) for v in a.area.split("x"))
    oracle = oracle_lib()
    regions.setup(torch, lib, stream, a.pmc_child)
    regions.probe()
    regions.mem_probe()  # (counter passes only: known byte counts per access shape, the calibration of roofline.traffic)

    if a.probe:
        sink = torch.zeros(4, dtype=torch.int32, device="cuda")
        names = ["v_sad_u8", "v_qsad_pk_u16_u8", "v_add+v_xor", "v_mul_lo_u32(+add)", "v_mad_i64_i32", "v_alignbyte_b32",
                 "mix: 6 qsad + 2 x (3 alignbyte + 4 sad_u8)", "mix: 4 qsad + 4 x (3 alignbyte + 4 sad_u8)", "mix: 8 x (3 alignbyte + 4 sad_u8)"]  # (6-8: the same 128 abs-diff lanes per round as 8 qsads)
        blocks, iters = 256 * 8, 4096
        for k, nm in enumerate(names):
            fn = lambda: lib.svt_hip_rate_probe(k, iters, blocks, sink.data_ptr(), stream)  # noqa: E731
            _, dev = time_steps(torch, fn, 5, 2)
            ops = 5 * blocks * 256 * iters * 8
            print("%-22s %8.1f Gop/s/lane-op  -> %.2f cycles per wave64 instruction per SIMD (2.4 GHz, 1024 SIMDs)" %
                  (nm, ops / dev / 1e9, 2.4e9 * 1024 * 64 / (ops / dev)))
        return

    # ---------------- config 2: batched ME full-pel search, frame-sharded -----------------------------------------
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
    step()
    torch.cuda.synchronize()
    me_checked = 0
    if oracle is not None:  # in-run parity: 64 (SB, reference) items spread over the batch against the CPU checker
        hs, hm = d_sad.cpu().numpy().view(np.uint32).reshape(n, 85), d_mv.cpu().numpy().view(np.uint32).reshape(n, 85)
        for i in np.linspace(0, n - 1, 64).astype(int):
            d = descs[i]
            ws_, wm_ = np.zeros(85, np.uint32), np.zeros(85, np.uint32)
            oracle.oracle_me_fullpel_search(vp(planes, int(d["src_off"])), STRIDE, vp(planes, int(d["ref_off"])), STRIDE, int(d["x_origin"]), int(d["y_origin"]), aw, ah, 0,
                                            vp(ws_), vp(wm_))
            me_checked += must_equal("me_fullpel sad", hs[i], ws_) + must_equal("me_fullpel mv", hm[i], wm_)
    # a "step" = L back-to-back launches over the resident batch (L batches of frames x refs x 510 SBs), L chosen so that the K timed steps last >= min_leg_s
    L = calibrate_launches(torch, step, a.steps, a.min_leg_s, dist)

    def step_l():
        for _ in range(L):
            step()
    wall, dev = time_steps(torch, step_l, a.steps, a.warmup, dist)
    t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    positions = n * aw * ah * L
    value = world * positions * a.steps / wall / 1e6
    # roofline of the dominant kernel: algorithmic bytes per (SB, ref) = 64*64 + (64+W-1)(64+H-1) + 85*8
    bytes_item = 64 * 64 + (64 + aw - 1) * (64 + ah - 1) + 85 * 8
    kernel_s = dev / (a.steps * L)  # per launch
    default_workload = (a.frames, a.refs, a.area) == (32, 4, "16x9")  # the workload the committed PMC passes were run on
    # areas up to 24x16 take the one-wave-per-item kernel (window pitch 18 or 26 dwords), larger ones the tiled workgroup kernel (csrc/sad.hip)
    me_kernel = "me_fullpel_wave_kernel<false, %d>" % (18 if (aw + 3) // 4 <= 2 else 26) if (aw <= 24 and ah <= 16) else "me_fullpel_kernel<false>"
    rf = roofline(n * bytes_item, kernel_s, me_kernel, None if (default_workload or LIVE_PMC is not None) else "-", kernel_ms=kernel_s * 1e3,
                  algorithmic_bytes_per_sb_ref=bytes_item,
                  footprint_bytes=int(d_planes.numel() + d_descs.numel() + 2 * 4 * n * 85),  # planes + descriptors + both result tables: what the launches of a step re-read
                  note="search is VALU(packed-SAD)-bound, see valu_frac; HBM figure = SURVEY 8(d) algorithmic bytes / time",
                  sad_ops_per_s=n * aw * ah * 4096 / kernel_s,
                  # measured v_qsad_pk_u16_u8 issue cost: 22.4 cycles per wave64 instruction per SIMD (profiles/r01_call1_valu_issue_rates.txt)
                  valu_peak_sad_ops_per_s=QSAD_PEAK, valu_frac=n * aw * ah * 4096 / kernel_s / QSAD_PEAK)
    if not a.pmc_child and rank == 0:
        # the denominators of valu_frac, measured in THIS run: lane-operations per second of a kernel that issues nothing but independent
        # v_qsad_pk_u16_u8 / v_add + v_xor instructions on every SIMD -- rates, so no clock has to be assumed; the "cycles" figures divide by the 2.4 GHz the device
        # property reports and are only a unit conversion of those rates
        sink = torch.zeros(4, dtype=torch.int32, device="cuda")
        pb, pi, probe = 256 * 8, 4096, {}
        for kind, nm in ((1, "v_qsad_pk_u16_u8"), (2, "v_add_xor_pair")):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            lib.svt_hip_rate_probe(kind, pi, pb, sink.data_ptr(), stream)
            e0.record()
            for _ in range(4):
                lib.svt_hip_rate_probe(kind, pi, pb, sink.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            lane_ops = 4 * pb * 256 * pi * 8 / (e0.elapsed_time(e1) / 1e3)
            probe[nm] = {"lane_ops_per_s": lane_ops, "ns_per_wave64_inst_per_simd": 1e9 * 64 * 1024 / lane_ops, "cycles_per_wave64_inst_at_2.4GHz": 2.4e9 * 64 * 1024 / lane_ops}
        rf["valu_probe_this_run"] = probe
        rf["valu_cycles_per_inst"] = probe["v_add_xor_pair"]["cycles_per_wave64_inst_at_2.4GHz"]
        rf["valu_frac_vs_this_runs_qsad_rate"] = (n * aw * ah * 4096 / kernel_s) / (probe["v_qsad_pk_u16_u8"]["lane_ops_per_s"] * 16)
    fp = bench_frame_partition(torch, lib, pkg, stream, a, dist, rank, world, oracle)
    if (world > 1 or a.mode == "strips") and not a.pmc_child:
        fp["in_loop_filters"] = bench_filter_partition(torch, lib, pkg, stream, a, dist, rank, world, oracle)
    if not a.pmc_child and (not a.legs or "cpart" in a.legs.split(",")):
        # the product's own multi-device path (one process, csrc/partition.hip), while the other ranks wait at the barrier
        if dist is not None:
            dist.barrier()
        if rank == 0:
            try:
                fp["c_partition"] = bench_c_partition(torch, lib, pkg, stream, a, world, oracle)
            except Exception as e:  # (a failure here must not cost the run its headline)
                fp["c_partition"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if dist is not None:
            dist.barrier()
    out = {
        "metric": "Mblocks/s per kernel (SAD, FwdTxfm2d, CDEF) + encoder fps @1080p preset 8", "value": value,
        "unit": "Mblocks/s (block = one search position of one 64x64 SB vs one reference = 85 block SADs)",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[1]: batched open-loop ME integer full-pel search (SAD 8x8..64x64), 1080p 8-bit, all 64x64 SBs",
                   "launches_per_step": L, "frames_per_launch_per_gpu": a.frames, "frames_per_step_per_gpu": a.frames * L, "refs": a.refs, "search_area": a.area,
                   "sb_refs_per_launch_per_gpu": n, "sb_refs_per_step_per_gpu": n * L, "timed_region_s": wall,
                   "sub_sad": 0, "parallelism": "frame-sharded x%d (no collective)" % world, "mode": a.mode},
        "parity_checked_values": me_checked, "roofline": rf, "frame_partition": fp,
    }
    if a.mode == "strips":  # report the frame-partition figure as the headline instead
        out.update({"value": fp["value"], "ms_per_step": fp["ms_per_step"], "scaling": "strong"})
        out["config"]["parallelism"] = "SB-row strips of one picture x%d + RCCL all-gather" % world
        out["frames_mode_value"] = value
    kernels = {}
    cpu = rank == 0 and world == 1 and not a.no_cpu
    import bench_legs
    bench_legs.MIN_S = a.min_leg_s
    want = (lambda leg: not a.legs or leg in a.legs.split(","))
    if not a.only_me:
        if want("me"):  # the same batched search at the areas preset 8 really derives at its default CRF 35 (8 x 4 and 8 x 3: pkg.m8_me_settings): HBM-bound there
            for (w2, h2) in ((8, 4), (8, 3)):
                dd = np.concatenate([pkg.me_descs_for_frame(W, H, STRIDE, PAD, PAD, w2, h2, PLANE, n_refs=a.refs, src_plane=f, ref_plane0=f + 1) for f in range(a.frames)])
                tdd = torch.from_numpy(dd.view(np.uint8)).cuda()
                f2 = lambda: lib.svt_hip_me_fullpel_search_batch(d_planes.data_ptr(), d_planes.data_ptr(), tdd.data_ptr(), len(dd), w2, h2, 0, d_sad.data_ptr(),  # noqa: E731
                                                                 d_mv.data_ptr(), None, stream)
                f2()
                torch.cuda.synchronize()
                chk = 0
                if oracle is not None:
                    hs, hm = d_sad.cpu().numpy().view(np.uint32).reshape(-1, 85), d_mv.cpu().numpy().view(np.uint32).reshape(-1, 85)
                    for i in np.linspace(0, len(dd) - 1, 24).astype(int):
                        d = dd[i]
                        ws_, wm_ = np.zeros(85, np.uint32), np.zeros(85, np.uint32)
                        oracle.oracle_me_fullpel_search(vp(planes, int(d["src_off"])), STRIDE, vp(planes, int(d["ref_off"])), STRIDE, int(d["x_origin"]), int(d["y_origin"]), w2, h2, 0,
                                                        vp(ws_), vp(wm_))
                        chk += must_equal("me_fullpel %dx%d sad" % (w2, h2), hs[i], ws_) + must_equal("me_fullpel %dx%d mv" % (w2, h2), hm[i], wm_)
                per, _ = time_leg(torch, f2, a.min_leg_s)
                b_item = 64 * 64 + (64 + w2 - 1) * (64 + h2 - 1) + 85 * 8
                kernels["me_search_%dx%d_preset8_area" % (w2, h2)] = {
                    "value": len(dd) * w2 * h2 / per / 1e6, "unit": "Mblocks/s", "sb_refs": len(dd), "parity_checked_values": chk,
                    "roofline": roofline(len(dd) * b_item, per, "me_fullpel_wave_kernel<false, 18>", algorithmic_bytes_per_sb_ref=b_item,
                                         footprint_bytes=int(d_planes.numel() + tdd.numel() + 2 * 4 * len(dd) * 85),
                                         valu_frac=len(dd) * w2 * h2 * 4096 / per / QSAD_PEAK,
                                         note="valu_frac from the measured v_qsad_pk_u16_u8 rate, as the headline leg's; the 134 MB working set stays in the 256 MiB "
                                              "Infinity Cache from launch to launch (binds: l3) -- the _dram leg below rotates plane sets past it")}
                # the same launch over ROT plane sets and result tables used in turn (footprint ROT x 134 MB > 256 MiB): no launch finds its planes in the Infinity Cache.
                # Inside one launch every plane is still read as a source once and as a reference `refs` times -- that reuse is the algorithm's, the counters price it.
                if True:
                    ROT = 3
                    rot_planes = [d_planes] + [d_planes.clone() for _ in range(ROT - 1)]
                    rot_sad = [d_sad] + [torch.zeros_like(d_sad) for _ in range(ROT - 1)]
                    rot_mv = [d_mv] + [torch.zeros_like(d_mv) for _ in range(ROT - 1)]
                    turn = [0]

                    def f3():
                        k = turn[0] = (turn[0] + 1) % ROT
                        lib.svt_hip_me_fullpel_search_batch(rot_planes[k].data_ptr(), rot_planes[k].data_ptr(), tdd.data_ptr(), len(dd), w2, h2, 0, rot_sad[k].data_ptr(),
                                                            rot_mv[k].data_ptr(), None, stream)
                    per3, _ = time_leg(torch, f3, a.min_leg_s)
                    kernels["me_search_%dx%d_preset8_area_dram" % (w2, h2)] = {
                        "value": len(dd) * w2 * h2 / per3 / 1e6, "unit": "Mblocks/s", "sb_refs": len(dd), "plane_sets_rotated": ROT,
                        "roofline": roofline(len(dd) * b_item, per3, "me_fullpel_wave_kernel<false, 18>", algorithmic_bytes_per_sb_ref=b_item,
                                             footprint_bytes=int(ROT * (d_planes.numel() + 2 * 4 * len(dd) * 85) + tdd.numel()),
                                             valu_frac=len(dd) * w2 * h2 * 4096 / per3 / QSAD_PEAK,
                                             note="footprint beyond the Infinity Cache; identical results to the resident leg (same planes, same descriptors)")}
                    del rot_planes, rot_sad, rot_mv
            step()  # (the shared result arrays hold the headline search again)
        if want("sad"):
            kernels["sad64x64_pairs"] = bench_sad_pairs(torch, lib, pkg, stream, a, cpu)
        if want("txfm"):
            kernels.update(bench_fwd_txfm(torch, lib, pkg, stream, a, cpu))
        if want("config3"):
            kernels["config3_roundtrip"] = bench_config3(torch, lib, pkg, stream, a)
        if want("cdef"):
            kernels.update(bench_cdef(torch, lib, pkg, stream, a, cpu))
        if want("lr"):
            lr_checked = check_lr_small(torch, lib, pkg, stream)
            lr = bench_legs.lr_frames(torch, lib, pkg, stream, max(a.steps // 10, 4), 2)
            for v in lr.values():
                v["parity_checked_values"] = lr_checked
            kernels.update(lr)
        if want("hme"):
            kernels.update(bench_legs.hme_chain(torch, lib, pkg, stream, max(a.steps // 4, 8), 2))
        if want("satd"):
            kernels.update(bench_legs.hadamard_satd(torch, lib, pkg, stream, max(a.steps // 4, 8), 2, oracle))
        if want("session"):
            kernels.update(bench_legs.me_session_stage(torch, lib, pkg, stream, 12, 1))
        if want("tf"):
            keep = {}
            kernels.update(bench_legs.tf_subpel(torch, lib, pkg, stream, 5, 1, keep))
            if cpu:
                kernels["tf_subpel_1080p8_6refs"].update(cpu_tf_subpel(keep, budget_s=4.0))
        if want("tpl"):
            keep = {}
            kernels.update(bench_legs.tpl_src_stage(torch, lib, pkg, stream, 10, 2, keep))
            if cpu:
                kernels["tpl_src_stage_1080p8"].update(cpu_tpl_stage(keep))
            kernels.update(bench_legs.tpl_recon_stage(torch, lib, pkg, stream, 5, 1, keep))
            if cpu:
                kernels["tpl_recon_stage_1080p8"].update(cpu_tpl_recon_stage(keep))
        if want("tpl1"):
            keep = {}
            kernels.update(bench_legs.tpl_level1_stage(torch, lib, pkg, stream, 3, 1, keep))
            if cpu:
                chk = cpu_tpl_level1(keep)
                kernels["tpl_l1_src_1080p8"].update(chk)
                if "cpu_baseline" in chk:
                    kernels["tpl_l1_recon_1080p8"]["cpu_baseline_both_halves"] = chk["cpu_baseline"]
                if "parity_checked_recon" in chk:  # the reconstruction half's share of the same comparison: its four statistics per cell + the reconstructed picture
                    kernels["tpl_l1_recon_1080p8"]["parity_checked_values"] = chk["parity_checked_recon"]
                    kernels["tpl_l1_recon_1080p8"].setdefault("cpu_baseline", dict(chk.get("cpu_baseline") or {}, note="BOTH halves of the reference's dispenser on one core (the "
                                                                                     "reference does not run the halves separately); the AVX2 all-thread figure: cpu_baseline_both_halves"))
        if want("tfpic"):
            keep = {}
            kernels.update(bench_legs.tf_picture_stage(torch, lib, pkg, stream, 5, 1, keep))
            if cpu:
                kernels["tf_picture_stage_1080p8_4refs_host"].update(cpu_tf_picture(keep))
        if want("lrstats"):  # a24 on the matrix cores: the one MFMA kernel of the path, priced against the dense int8 peak (north_star: "MFMA utilisation against peak")
            kernels.update(bench_legs.lr_stats(torch, lib, pkg, stream, 4, 1))
        if want("cdefchain"):  # config 4's whole CDEF stage of a 4:2:0 picture: search (Y, U, V x 64 strengths) -> joint strength pick -> per-block assignment -> apply
            kernels.update(bench_legs.cdef_chain(torch, lib, pkg, stream, 3, 1))
        if want("lrsearch"):
            keep = {}
            kernels.update(bench_legs.lr_search(torch, lib, pkg, stream, 2, 1, keep))
            if cpu:
                for name in keep:
                    kernels[name].update(cpu_lr_search(keep[name], budget_s=5.0))
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
            st = min(a.steps, 40) if w2 < 64 else 2
            _, dv = time_steps(torch, f2, st, 1)
            kernels["me_search_%dx%d_sub%d" % (w2, h2, sub)] = {"value": len(dd) * w2 * h2 / (dv / st) / 1e6, "unit": "Mblocks/s",
                                                               "sb_refs": len(dd), "sad_ops_per_s": len(dd) * w2 * h2 * (2048 if sub else 4096) / (dv / st)}
        import bench_legs
        es = min(a.steps, 20)
        kernels.update(bench_legs.hme_sad_loop(torch, lib, pkg, stream, es, 3))
        kernels.update(bench_legs.picprep(torch, lib, pkg, stream, max(es // 2, 2), 1))
        kernels.update(bench_legs.deblock(torch, lib, pkg, stream, max(es // 2, 2), 1))
        kernels.update(bench_legs.me_session(torch, lib, pkg, stream, es, 1))
        kernels.update(bench_legs.me_results(torch, lib, pkg, stream, es, 1))
        kernels.update(bench_legs.me_stage(torch, lib, pkg, stream, es, 1))
        kernels.update(bench_legs.tf_frames(torch, lib, pkg, stream, es, 1))
        kernels.update(bench_legs.tf_inter_pred(torch, lib, pkg, stream, max(es // 2, 2), 1))
        kernels["txfm_quant_roundtrip"] = bench_legs.txfm_roundtrip(torch, lib, pkg, stream, max(es // 4, 3), 1)
    out["kernels"] = kernels
    if cpu and not a.only_me:
        import bench_cpu
        bench_cpu.attach(kernels, CPU_HELPERS())  # the reference's AVX2 / AVX-512 kernels on all host cores for the legs that had no such figure
    if cpu:
        host_descs = pkg.me_descs_for_frame(W, H, STRIDE, PAD, PAD, aw, ah, PLANE, n_refs=1, src_plane=0, ref_plane0=1)
        out["cpu_baseline"] = cpu_me_baseline(host_descs, planes, planes, (aw, ah), budget_s=10.0)
        out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        for (w2, h2) in ((8, 4), (8, 3)):  # the same reference kernels at the areas preset 8 derives
            leg = kernels.get("me_search_%dx%d_preset8_area" % (w2, h2))
            if isinstance(leg, dict):
                hd = pkg.me_descs_for_frame(W, H, STRIDE, PAD, PAD, w2, h2, PLANE, n_refs=1, src_plane=0, ref_plane0=1)
                leg["cpu_baseline"] = cpu_me_baseline(hd, planes, planes, (w2, h2), budget_s=3.0)
        if not a.only_me and not a.legs:
            out["encoder_fps_1080p_preset8"] = encoder_fps()
            attach_encoder_baselines(kernels, out["encoder_fps_1080p_preset8"])
            out["encoder_fps_4k10_preset8"] = encoder_fps_4k10(1)  # BASELINE configs[4] at one GPU (the N-GPU form: below, rank 0 of a --gpus N run)
    elif rank == 0:
        out["cpu_baseline"] = None
        if world > 1 and not a.no_cpu and not a.pmc_child and not a.legs and not a.only_me:
            # BASELINE configs[4] on this node's GPUs: the reference encoder with its pictures sharded over SVT_HIP_DEVICES=0..N-1 (the other ranks wait at the barrier below;
            # their GPUs are idle by now).  Identity against the C-only encoder is decided inside; a failure is reported in the object, never raised.
            out["encoder_fps_4k10_preset8"] = encoder_fps_4k10(world)
    rf["traffic_source"] = (rf.get("traffic_detail") or {}).get("source")
    rf["pmc_seconds"] = (LIVE_PMC or {}).get("_seconds")
    if rf.get("traffic") and rf.get("algorithmic_bytes_per_launch"):
        rf["moved_over_algorithmic"] = rf["traffic"] / rf["algorithmic_bytes_per_launch"]
    finish_rooflines(kernels, rf, kernel_s)
    if "sad64x64_pairs" in kernels:  # the kernel north_star's ">= 50 % of HBM on the SAD path" applies to
        rf["sad_path_hbm_frac"] = kernels["sad64x64_pairs"]["roofline"]["frac"]
    if a.legs and a.mode != "strips" and "cpart" not in a.legs.split(","):
        out.pop("frame_partition", None)
    if a.pmc_child:
        if os.environ.get("SVT_PMC_REGIONS_FILE"):
            regions.dump(os.environ["SVT_PMC_REGIONS_FILE"])
    elif rank == 0:
        emit(out, a)
    if dist is not None:
        # rank 0 may just have spent a minute in the N-device encode: the other ranks wait for it on the rendezvous STORE (host side) -- an RCCL barrier would keep a
        # spinning kernel on every GPU the encode is using
        try:
            import datetime
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("svt_bench_rank0_done", "1")
            else:
                store.wait(["svt_bench_rank0_done"], datetime.timedelta(seconds=3600))
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("bench.py: store wait: %r\n" % (e,))
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
