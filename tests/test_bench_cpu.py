"""bench.py's CPU-baseline column (bench_cpu.py + oracle/ref_drivers2.c): every leg's time-bounded loop over the reference's own AVX2 kernels runs (0.1 s each), returns a
positive reference-kind figure and leaves no error -- on the CPU, without a GPU.  Skipped where oracle/_ref is not built or the host has no AVX2."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from conftest import REF_LIB, ROOT


@pytest.mark.skipif(not os.path.exists(REF_LIB) or " avx2 " not in open("/proc/cpuinfo").read(), reason="needs oracle/_ref/libsvtref.so and an AVX2 host")
def test_cpu_baseline_legs_run(oracle):
    sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    import bench_cpu
    import bench_legs
    bench_cpu.SECONDS = 0.1
    legs = ("quantize_b_32x32", "lr_wiener_4k10", "lr_sgrproj_4k10", "lr_mixed_4k10", "lr_search_4k10_full", "hme_3level_1080p_4refs", "cdef_search_4k10_64strengths",
            "hadamard_satd_32x32")
    k = {n: {"value": 1.0} for n in legs}
    H = b.CPU_HELPERS()
    pkg = b.entry._pkg()
    coeff = np.random.default_rng(1).integers(-3000, 3000, 256 * 1024).astype(np.int32)
    bench_cpu.quantize(k, H, coeff, bench_legs._qparams(pkg, 88, 112, 1), np.arange(1024, dtype=np.int16))
    bench_cpu.attach(k, H)
    assert "_cpu_leg_errors" not in k, k.get("_cpu_leg_errors")
    for n in legs:
        cb = k[n].get("cpu_baseline") or k[n].get("cpu_baseline_compute_stats")
        assert cb and cb["kind"] == "reference" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"], (n, k[n])
