"""Timed regions of bench.py / bench_legs.py and their attribution in the counter passes.

Every timer call (`time_leg`, `time_steps`, `_time`, or an explicit `hook(fn)`) opens a REGION tagged "<calling function>#<k-th timer of that function>".  In a normal run
the tag is only remembered (`LAST`, stored into the leg's roofline object).  In a counter pass (`bench.py --pmc-child` under `rocprofv3 --pmc ...`) the timer instead
runs the function CALLS times between two marker launches -- `svt_hip_rate_kernel<5>` with grid size 256 * (1000 + 2 * index [+ 1]) -- so that the parent can cut the
counter CSV (ordered by dispatch id) into regions and attribute every kernel launch between a pair of markers to its leg: HBM bytes, VALU instructions and VALU-active
cycles PER CALL of the leg, however many kernels the call launches (a stage = several kernels, the TPL reconstruction = 187 launches).  The parent matches regions by
tag, not by order."""
import inspect
import json
import os

PMC_CHILD = False
CALLS = 2          # calls of the leg's function between the markers of a counter pass
MARK_KERNEL = "svt_hip_rate_kernel<5>"
MARK_BASE = 1000
TAGS = []          # index -> tag
LAST = None        # tag of the most recent region
PROBE = {}         # the VALU calibration kernel's event-timed duration (child only)
_per_fn = {}
_lib = _sink = _stream = _torch = None


def setup(torch, lib, stream, pmc_child):
    global _torch, _lib, _stream, _sink, PMC_CHILD
    _torch, _lib, _stream, PMC_CHILD = torch, lib, stream, bool(pmc_child)
    _sink = torch.zeros(4, dtype=torch.int32, device="cuda")


def open_region(depth=2, name=None):
    """allocates the tag of a new region for the function `depth` frames up (or `name`) and returns its index"""
    global LAST
    fn = name or inspect.stack()[depth].function
    k = _per_fn.get(fn, 0)
    _per_fn[fn] = k + 1
    LAST = "%s#%d" % (fn, k)
    TAGS.append(LAST)
    return len(TAGS) - 1


def _mark(code):
    _lib.svt_hip_rate_probe(5, 1, MARK_BASE + code, _sink.data_ptr(), _stream)


def pmc_run(fn, idx, sync_each=False):
    """counter pass: one untimed call, then CALLS calls between the region's markers; returns seconds per call (event-timed, perturbed by the collection)"""
    fn()
    _torch.cuda.synchronize()
    _mark(2 * idx)
    e0, e1 = _torch.cuda.Event(enable_timing=True), _torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(CALLS):
        fn()
        if sync_each:
            _torch.cuda.synchronize()
    e1.record()
    _mark(2 * idx + 1)
    _torch.cuda.synchronize()
    return max(e0.elapsed_time(e1) / 1e3 / CALLS, 1e-9)


def hook(fn, name=None, sync_each=False):
    """for legs that keep their own timing loop (host forms timed by the wall clock): opens the region; in a counter pass runs `fn` between the markers and returns
    the seconds per call, otherwise returns None and the caller times as before"""
    idx = open_region(2, name)
    if PMC_CHILD:
        return pmc_run(fn, idx, sync_each)
    return None


def probe():
    """the VALU calibration region: a kernel that issues nothing but independent VALU instructions (svt_hip_rate_kernel<2>: v_add + v_xor chains, 4.4 cycles per wave64
    instruction and SIMD measured = the issue ceiling).  Its SQ_ACTIVE_INST_VALU per second of kernel time is what `valu_busy` = 1.0 means."""
    if not PMC_CHILD:
        return
    blocks, iters = 256 * 8, 4096
    fn = lambda: _lib.svt_hip_rate_probe(2, iters, blocks, _sink.data_ptr(), _stream)  # noqa: E731
    idx = open_region(1, "valu_probe")
    PROBE["seconds"] = pmc_run(fn, idx)
    PROBE["wave_insts_expected"] = blocks * 4 * iters * 16  # 256 threads = 4 waves per block, 8 chains x 2 instructions per iteration


# access shapes the traffic counters are calibrated on: name -> (bytes per lane, segment bytes, pitch bytes); "stream" shapes are contiguous
MEM_SHAPES = {"stream16": (16, 1 << 20, 1 << 20), "stream8": (8, 1 << 20, 1 << 20), "stream4": (4, 1 << 20, 1 << 20),
              "rows64_of_2056": (16, 64, 2056),   # ISOLATED 64-byte segments, one per 2056-byte row (nothing else of the row is read): what a short strided access costs
              "rows128_of_2056": (4, 128, 2056)}  # isolated 128-byte segments read as dwords
# the block kernels' shape (svt_hip_mem_probe_blocks): 64x64-byte blocks tiling padded 1080p luma planes (pitch 2056, 30 blocks per block row), aligned and 3 bytes off --
# the source and reference halves of the independent-pair SAD kernel's reads
MEM_BLOCK_SHAPES = {"blocks64_aligned": 0, "blocks64_off3": 3}
MEM_PROBE = {}     # tag -> bytes moved per call (child only)


def mem_probe():
    """the traffic calibration regions: svt_hip_mem_probe moves a KNOWN number of bytes in each access shape of MEM_SHAPES, reading and writing, over a buffer larger than the
    256 MiB Infinity Cache; the parent divides the known bytes by the FETCH_SIZE / WRITE_SIZE the pass reports for the region."""
    if not PMC_CHILD:
        return
    total = 768 << 20
    buf = _torch.empty(total + (4 << 20), dtype=_torch.uint8, device="cuda")
    for name, (w, seg, pitch) in MEM_SHAPES.items():
        lanes = (total // pitch) * (seg // w)
        for wr in (0, 1):
            fn = lambda wr=wr, w=w, seg=seg, pitch=pitch, lanes=lanes: _lib.svt_hip_mem_probe(wr, w, buf.data_ptr(), lanes, seg, pitch, _sink.data_ptr(), _stream)  # noqa: E731
            idx = open_region(1, "mem_%s_%s" % ("w" if wr else "r", name))
            pmc_run(fn, idx)
            MEM_PROBE[TAGS[idx]] = lanes * w
    pitch, bpr = 2056, 30
    nblk = (total // (64 * pitch)) * bpr
    for name, mis in MEM_BLOCK_SHAPES.items():
        fn = lambda mis=mis: _lib.svt_hip_mem_probe_blocks(buf.data_ptr(), nblk * 256, pitch, bpr, mis, _sink.data_ptr(), _stream)  # noqa: E731
        idx = open_region(1, "mem_r_%s" % name)
        pmc_run(fn, idx)
        MEM_PROBE[TAGS[idx]] = nblk * 256 * 16
    del buf


def dump(path):
    with open(path, "w") as f:
        json.dump({"tags": TAGS, "calls": CALLS, "probe": PROBE, "mem_probe": MEM_PROBE}, f)


def parse_counter_csv(path, tags, counters):
    """-> {tag: {"kernels": {kernel: {"launches": n, counter: total over the region}}}} from one rocprofv3 counter_collection.csv"""
    import collections
    import csv
    rows = collections.defaultdict(dict)  # dispatch id -> {"k": kernel, "g": grid, counter: value}
    for r in csv.DictReader(open(path)):
        d = int(r["Dispatch_Id"])
        e = rows[d]
        e["k"] = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        e["g"] = int(r["Grid_Size"])
        if r["Counter_Name"] in counters:
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out, cur = {}, None
    for d in sorted(rows):
        e = rows[d]
        if e["k"] == MARK_KERNEL and e["g"] % 256 == 0 and e["g"] // 256 >= MARK_BASE:
            code = e["g"] // 256 - MARK_BASE
            if code % 2 == 0 and code // 2 < len(tags):
                cur = tags[code // 2]
                out[cur] = {"kernels": {}}
            else:
                cur = None
            continue
        if cur is None or e["k"].startswith("at::") or "rocclr" in e["k"]:
            continue
        k = out[cur]["kernels"].setdefault(e["k"], {"launches": 0})
        k["launches"] += 1
        for c in counters:
            if c in e:
                k[c] = k.get(c, 0.0) + e[c]
    return out
