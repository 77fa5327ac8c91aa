"""bench.py's stdout contract: ONE line, at most bench_line.MAX_LINE bytes, that json.loads and carries metric / value / config / roofline /
cpu_baseline / the encoder fps half of the metric -- built here from canned leg results: the complete object of a real run (round 3's 19 KB line, kept under
profiles/) plus every field this round adds, and degenerate objects (no CPU legs, no counters, a strips-mode run)."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_line  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
            "cpu_baseline")


def canned():
    path = os.path.join(ROOT, "profiles", "r03_final_bench_default.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    # what round 4 adds to every leg
    for name, k in d["kernels"].items():
        if isinstance(k, dict) and isinstance(k.get("roofline"), dict):
            k["roofline"].update(valu_frac=0.4321, valu_busy=0.51234, binds="valu", region="x#0", kernels_per_call={"k": 1.0}, traffic=123456789.0)
    d["kernels"]["sad64x64_pairs"]["cpu_baseline_avx512"] = dict(d["kernels"]["sad64x64_pairs"]["cpu_baseline"], value=1234.5)
    d["kernels"]["hme_3level_1080p_4refs"]["roofline"] = {"bound": "hbm", "frac": 0.03, "kernel_us": 62.0, "valu_frac": 0.7, "binds": "valu"}
    d["encoder_fps_1080p_preset8"]["host_cpu_s_per_frame"] = {"c": 0.61, "avx2": 0.082, "avx2_with_stages": 0.074}
    d["encoder_fps_1080p_preset8"]["instances"] = {"k": 4, "fps_avx2": 201.2, "fps_avx2_with_stages": 214.9, "identical": True}
    return d


def check(line, want_cpu=True):
    assert "\n" not in line and len(line) <= bench_line.MAX_LINE
    o = json.loads(line)
    for k in REQUIRED:
        assert k in o, k
    assert o["metric"].startswith("Mblocks/s per kernel") and isinstance(o["value"], float) and o["value"] > 0
    assert o["config"]["workload"].startswith("configs[1]")
    r = o["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3 and "traffic" in r and r["unit"] == "GB/s"
    if want_cpu:
        c = o["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    return o


def test_full_object_fits_and_parses():
    d = canned()
    assert len(json.dumps(d)) > 15000  # the object that used to be printed whole
    o = check(bench_line.compact(d))
    assert o["roofline"]["kernel"].startswith("me_fullpel") and o["roofline"]["sad_path_hbm_frac"] > 0.5
    assert o["cpu_baseline"]["kernels"]["sad64x64_pairs"]["avx512"] == pytest.approx(1234.5, rel=1e-3) and "gpu" in o["cpu_baseline"]["kernels"]["fwd_txfm2d_32x32"]
    e = o["encoder_fps_1080p_preset8"]
    assert e["bitstream_identical"] is True and e["fps_avx2_intrinsics"] > 0 and e["host_cpu_s_per_frame"]["avx2"] == 0.082 and e["instances"]["k"] == 4
    legs = o["legs"]
    assert legs["_columns"] == ["us", "hbm_frac", "valu_frac", "binds"]
    assert legs["cdef_search_4k10_64strengths"][3] == "valu" and legs["hme_3level_1080p_4refs"][0] == 62.0
    assert o["value"] == pytest.approx(d["value"], rel=1e-4) and o["ms_per_step"] == pytest.approx(d["ms_per_step"], rel=1e-4)


def test_degenerate_objects():
    d = canned()
    for drop in (("cpu_baseline", "encoder_fps_1080p_preset8"), ("kernels",), ("frame_partition",)):
        e = copy.deepcopy(d)
        for k in drop:
            e[k] = None
        check(bench_line.compact(e), want_cpu="cpu_baseline" not in drop)
    e = copy.deepcopy(d)
    e["roofline"]["traffic"] = None  # counters unavailable: the field stays, null
    assert json.loads(bench_line.compact(e))["roofline"]["traffic"] is None
    e = copy.deepcopy(d)  # a run with hundreds of legs still fits (the leg table gives way first)
    for i in range(400):
        e["kernels"]["extra_leg_%03d" % i] = {"roofline": {"kernel_us": 12.345, "frac": 0.123, "valu_frac": 0.456, "binds": "valu"}}
    o = check(bench_line.compact(e))
    # the leg table gives way leg by leg, the least important first: the kernels the metric names (and the MFMA leg) are still there when 400 others are not
    assert "legs" in o and "sad64x64_pairs" in o["legs"] and sum(k.startswith("extra_leg_") for k in o["legs"]) < 400
    e = copy.deepcopy(d)  # NaN / inf never reach the line
    e["roofline"]["valu_frac"] = float("nan")
    assert "NaN" not in bench_line.compact(e)


def test_strips_mode_object():
    d = canned()
    d.update(scaling="strong", n_gpus=8)
    d["frame_partition"] = {"value": 1234.5, "unit": "pictures/s", "ms_per_step": 0.81, "scaling": "strong", "collective": "all_gather_into_tensor", "in_loop_filters": {"x": 1}}
    o = check(bench_line.compact(d))
    assert o["frame_partition"]["collective"] == "all_gather_into_tensor" and o["n_gpus"] == 8
