"""The ONE line bench.py prints on stdout, built from the full result object.

The driver keeps only a few KB of stdout: round 3's line had grown to 19 KB and its record came back unparsed.  So the full object goes to
`bench_detail.json` (and to stderr, one line); stdout carries a line of at most MAX_LINE bytes that holds exactly the claims the contract names: metric / value /
config, `roofline` of the dominant kernel, `cpu_baseline`, the encoder fps half of the metric, and a fixed-width row per leg (time, HBM fraction, VALU fraction, which
roof binds).  compact() never raises on a missing field -- a leg that did not run is simply absent -- and shrinks itself (drops the least important parts first) until
the line fits; tests/test_bench_line.py runs it on canned results.
"""
import json

MAX_LINE = 4090  # bytes (the driver asks for < 4096); asserted before printing


def _r(v, sig=5):
    """numbers rounded to `sig` significant digits (floats only), everything else unchanged"""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, v))
    return v


def _pick(d, keys, sig=5):
    return {k: _r(d[k], sig) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def leg_row(k):
    """[microseconds per launch or stage call, HBM fraction (algorithmic bytes / time / 8 TB/s), VALU fraction, binding roof] of one leg; None where unmeasured"""
    r = k.get("roofline") if isinstance(k, dict) else None
    if not isinstance(r, dict):
        return None
    us = r.get("kernel_us")
    frac = r.get("frac")
    vf = r.get("valu_frac")
    return [_r(us, 4), _r(frac, 3), _r(vf, 3), r.get("binds")]


def compact(out):
    rf, cb, enc = out.get("roofline") or {}, out.get("cpu_baseline") or {}, out.get("encoder_fps_1080p_preset8") or {}
    kernels = out.get("kernels") or {}
    line = {k: _r(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["unit"] = "Mblocks/s"  # (block = one search position of one 64x64 SB against one reference = 85 block SADs: config.workload / DESIGN.md 6)
    line["config"] = _pick(out.get("config") or {}, ("workload", "launches_per_step", "frames_per_step_per_gpu", "refs", "search_area", "sb_refs_per_step_per_gpu",
                                                     "timed_region_s", "parallelism", "mode"))
    line["parity_checked_values"] = out.get("parity_checked_values")
    line["roofline"] = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel", "kernel_us", "valu_frac",
                                  "binds", "sad_path_hbm_frac", "moved_over_algorithmic", "traffic_source", "valu_cycles_per_inst", "valu_frac_vs_this_runs_qsad_rate"))
    if "traffic" not in line["roofline"]:
        line["roofline"]["traffic"] = None
    a84 = (kernels.get("me_search_8x4_preset8_area") or {}).get("roofline")
    if isinstance(a84, dict) and a84.get("frac") is not None:  # the same kernel at the area preset 8 derives at its default CRF: where the HBM roof binds
        line["roofline"]["hbm_frac_at_preset8_area_8x4"] = _r(a84["frac"], 4)
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "single_thread_value", "gpu_over_cpu", "sample"))
        ck = {}
        for name in ("sad64x64_pairs", "fwd_txfm2d_32x32", "inv_txfm2d_add_32x32", "cdef_apply_4k10"):
            k = kernels.get(name) or {}
            e = {}
            if "value" in k:
                e["gpu"] = _r(k["value"], 4)
            for tag, key in (("avx2", "cpu_baseline"), ("avx512", "cpu_baseline_avx512"), ("sse4_1", "cpu_baseline_sse4_1")):
                if isinstance(k.get(key), dict) and "value" in k[key]:
                    e[tag] = _r(k[key]["value"], 4)
                    e["cores"] = k[key].get("cores")
            if e:
                ck[name] = e
        if ck:
            c["kernels"] = ck
        have = [n for n, k in kernels.items() if isinstance(k, dict) and any(isinstance(k.get(x), dict) and k[x].get("kind") == "reference" for x in k if x.startswith("cpu_baseline"))]
        c["legs_with_reference_cpu"] = [len(have), len([n for n, k in kernels.items() if isinstance(k, dict) and not n.startswith("_")])]  # (every figure: the detail file)
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    if enc:
        e = _pick(enc, ("fps_c_only", "fps_avx2_intrinsics", "fps_avx512_intrinsics", "fps_avx2_host_with_stage_seams", "fps_avx512_host_with_stage_seams",
                        "fps_c_host_with_stage_seams", "frames", "bitstream_identical", "host_cpu_s_per_frame", "instances", "error"), 4)
        ps = enc.get("paying_stages")
        if isinstance(ps, dict):  # only ME / TF / TPL / CDEF on the device: [alone, with] medians of five pairs per host
            e["paying_stages"] = {"avx2": [_r(ps.get("fps_avx2"), 4), _r(ps.get("fps_avx2_with"), 4)], "avx512": [_r(ps.get("fps_avx512"), 4), _r(ps.get("fps_avx512_with"), 4)]}
        sp = {}
        for tag, key in (("avx2", "fps_avx2_pairs"), ("avx512", "fps_avx512_pairs")):
            pr = enc.get(key)
            if isinstance(pr, dict) and pr.get("alone") and pr.get("with_stages"):  # [pairs, min / max alone, min / max with the stages]: the medians are quoted above
                al, wi = [v for v in pr["alone"] if v], [v for v in pr["with_stages"] if v]
                if al and wi:
                    sp[tag] = [len(al), _r(min(al), 4), _r(max(al), 4), _r(min(wi), 4), _r(max(wi), 4)]
        if sp:
            e["fps_pairs"] = dict(sp, _cols="n,min,max alone,min,max with stages")
        sc = enc.get("stage_cpu_ms_per_frame")
        if isinstance(sc, dict):  # host CPU ms per frame inside the stages of SURVEY 8: the reference's AVX2 code vs the device stage calls
            e["stage_cpu_ms_per_frame"] = {k: v for k, v in sc.items() if k != "c" and v}
        ss = enc.get("steady_state_300_frames")
        if isinstance(ss, dict):
            e["steady_state_300_frames"] = _pick(ss, ("fps_avx2_intrinsics", "fps_avx2_host_with_stage_seams", "fps_avx512_intrinsics", "fps_avx512_host_with_stage_seams"), 4)
        line["encoder_fps_1080p_preset8"] = e
    e4 = out.get("encoder_fps_4k10_preset8")
    if isinstance(e4, dict):  # BASELINE configs[4]: [host alone, host + every stage seam] fps, identity decided on the first attempt
        line["encoder_fps_4k10_preset8"] = _pick(e4, ("frames", "host", "n_devices", "fps_c_only", "fps_host_alone", "fps_host_with_stage_seams", "bitstream_identical",
                                                      "first_attempt", "error"), 4)
    fp = out.get("frame_partition")
    if isinstance(fp, dict):
        line["frame_partition"] = _pick(fp, ("value", "ms_per_step", "scaling", "collective"), 4)  # (Mblocks/s of one picture x its references per launch: detail file)
        cp = fp.get("c_partition")
        if isinstance(cp, dict):  # the library's own partition from one process: [ms on one device, ms partitioned] per primitive
            c = _pick(cp, ("devices", "virtual_peers", "parity_checked_values", "error"))
            for k in ("me_1080p", "cdef_apply_4k10", "lr_4k10"):
                if isinstance(cp.get(k), dict):
                    c[k] = [_r(cp[k].get("ms_single_device"), 3), _r(cp[k].get("ms_partition"), 3)]
            line["frame_partition"]["c_partition"] = c
    legs = {}
    for name, k in kernels.items():
        row = leg_row(k)
        if row is not None:
            legs[name] = row
    if legs:
        line["legs"] = {"_columns": ["us", "hbm_frac", "valu_frac", "binds"], **legs}
    if out.get("detail"):
        line["detail"] = out["detail"]
    # shrink until it fits, the least important parts first (everything is in the detail file); the leg table goes last
    def size():
        return len(json.dumps(line, separators=(",", ":")))

    def steps():
        if "legs" in line:  # fewer digits
            line["legs"] = {k: (v if k == "_columns" else [_r(x, 3) if not isinstance(x, str) else x for x in v]) for k, v in line["legs"].items()}
        yield
        if isinstance(line.get("cpu_baseline"), dict) and isinstance(line["cpu_baseline"].get("sample"), str):
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:72]
        yield
        e = line.get("encoder_fps_1080p_preset8")
        if isinstance(e, dict) and isinstance(e.get("instances"), dict):
            for k in ("fps_sum_of_encoder_reports_avx2", "fps_sum_of_encoder_reports_avx2_with_stages", "frames_each"):
                e["instances"].pop(k, None)
        yield
        if isinstance(e, dict):
            e.pop("steady_state_300_frames", None)
        yield
        if isinstance(line.get("config"), dict):
            for k in ("launches_per_step", "frames_per_step_per_gpu", "sb_refs_per_step_per_gpu", "mode"):
                line["config"].pop(k, None)
        yield
        if isinstance(e, dict):
            e.pop("instances", None)
        yield
        if isinstance(line.get("frame_partition"), dict):
            line["frame_partition"].pop("collective", None)
        yield
        if isinstance(e, dict):
            e.pop("stage_cpu_ms_per_frame", None)
        yield
        if "legs" in line:  # coarser rows: whole microseconds, two digits, the roof's first letter (v / h / l / m / p)
            line["legs"] = {k: (["us", "hbm", "valu", "roof"] if k == "_columns" else [None if v[0] is None else (int(round(v[0])) if v[0] >= 10 else v[0]),
                                                                                     None if v[1] is None else round(v[1], 2), None if v[2] is None else round(v[2], 2),
                                                                                     (v[3] or "?")[0]]) for k, v in line["legs"].items()}
        yield
        line.pop("detail", None)
        yield
        # then leg by leg, the least important first (host forms and sub-variants before the kernels the metric names); the MFMA leg and the metric's kernels go last
        keep_last = ("sad64x64_pairs", "fwd_txfm2d_32x32", "cdef_apply_4k10", "cdef_search_4k10_64strengths", "lr_compute_stats_4k10_win7", "inv_txfm2d_add_32x32", "quantize_b_32x32",
                     "me_search_8x4_preset8_area_dram", "me_search_8x4_preset8_area", "lr_search_4k10_full", "cdef_stage_4k10_420", "cdef_apply_4k10_420", "lr_mixed_4k10_420",
                     "hme_3level_1080p_4refs", "tf_picture_stage_1080p8_4refs_resident", "tpl_recon_stage_1080p8", "lr_wiener_4k10", "lr_sgrproj_4k10", "config3_roundtrip")
        if "legs" in line:
            order = [k for k in line["legs"] if k != "_columns" and k not in keep_last] + [k for k in reversed(keep_last) if k in line["legs"]]
            for k in order:
                if size() <= MAX_LINE:
                    break
                line["legs"].pop(k, None)
            yield
        for drop in ("frame_partition", "legs"):
            line.pop(drop, None)
            yield
    for _ in steps():
        if size() <= MAX_LINE:
            break
    s = json.dumps(line, separators=(",", ":"))
    assert len(s) <= MAX_LINE, "bench line is %d bytes" % len(s)
    return s
