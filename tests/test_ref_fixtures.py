"""The reference's OWN gtest fixtures with the `_hip` functions as the function under test ("reuse the fixtures with the HIP symbol as the test function").

tests/ref_fixtures/ builds one binary from the reference's test files (included where they lie under /root/reference/test), gtest from the reference's third_party tree, and
INSTANTIATE_TEST_SUITE_P(HIP, ...) blocks that pass `svt_*_hip` where the reference passes `*_avx2`.  The binary is built in the container that has the reference
(__graft_entry__.build()) and travels to the GPU box with the snapshot (oracle/_ref/ is git-ignored, not gpurun-ignored); the `_c` side of every comparison is
oracle/_ref/libsvtref.so = the reference compiled in place.

  -m gpu      : every HIP-instantiated suite on the MI355X through svt-av1-psy_amd/libsvtav1_hip.so, sharded over the host's cores with gtest's own sharding
  -m "not gpu": the same objects linked with the CPU interpreter build of the kernel sources -- the suite list is checked and a slice small enough for it runs
"""
import json
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

from conftest import ROOT

FIX_DIR = os.path.join(ROOT, "oracle", "_ref", "fixtures")
BIN_GPU = os.path.join(FIX_DIR, "SvtAv1HipFixtures")
BIN_EMU = os.path.join(FIX_DIR, "SvtAv1HipFixturesEmu")
HAVE_REF = os.path.isdir("/root/reference/test")

# every suite the wrappers of tests/ref_fixtures instantiate: name -> number of parameterised cases (DISABLED speed tests not counted)
SUITES = {
    "HIP/SADTest": 4, "HIP/sad_LoopTest": 256, "HIP/Allsad8x8_CalculationTest": 16, "HIP/Allsad32x32_CalculationTest": 16, "HIP/Extsad8x8_CalculationTest": 16,
    "HIP/Extsad32x32_CalculationTest": 16, "HIP/InitializeBuffer32": 9, "HIP/SADTestSubSample16bit": 4, "HIP/PmeSadLoopTest": 128, "HIP_MotionEstimation": 2,
    "HIP/SatdTest": 12, "HIP/HadamardLowbdTest": 8,
    "HIP/FwdTxfm2dAsmTest": 38, "HIP_N2/FwdTxfm2dAsmTest": 38, "HIP_N4/FwdTxfm2dAsmTest": 38, "HIP/InvTxfm2dAsmSqrTest": 10, "HIP/InvTxfm2dAsmType1Test": 20,
    "HIP/InvTxfm2dAsmType2Test": 8, "HIP/InvTxfm2dAddTest": 2, "HIP/HandleTransformTest": 10,
    "HIP_LBD/QuantizeBTest": 12, "HIP_HBD/QuantizeBTest": 12, "HIP_LBD/QuantizeBQmTest": 12, "HIP_HBD/QuantizeBQmTest": 12, "HIP/QuantizeLbdTest": 54,
    "HIP/QuantizeHbdTest": 108, "HIP/QuantizeQmTest": 30, "HIP/QuantizeQmHbdTest": 30,
    "HIP/QuantizeLbdFewerBlocksTest": 9, "HIP/QuantizeHbdFewerBlocksTest": 18, "HIP/QuantizeQmFewerBlocksTest": 5, "HIP_HBD/QuantizeQmFewerBlocksTest": 5,
    "HIP/CDEFBlockTest": 180, "HIP/CDEFBlockInteriorTest": 12, "HIP/CDEFFindDirFewerRepeatsTest": 1, "HIP/CDEFFindDirDualFewerRepeatsTest": 1, "HIP/CDEFCopyRectTest": 1,
    "HIP/CDEFComputeCdefDist16Bit": 1, "HIP/CDEFComputeCdefDist8BitTest": 1, "HIP/CDEFSearchOneDualTest": 1,
    "HIP/LbdLoopFilterTest": 8, "HIP/HbdLoopFilterTest": 24,
    "HIP/AV1WienerConvolveLbdTest": 22, "HIP/AV1WienerConvolveHbdTest": 66, "HIP/AV1SelfguidedFilterTest": 1, "HIP/AV1HighbdSelfguidedFilterTest": 3,
    "HIP/PixelProjErrorLbdTest": 3, "HIP/PixelProjErrorHbdTest": 3, "HIP/GetProjSubspaceTestLbd": 1, "HIP/GetProjSubspaceTestHbd": 1,
    "HIP/av1_compute_stats_test": 432, "HIP/av1_compute_stats_test_hbd": 1728,
    "HIP/ResidualKernel8BitTest": 66, "HIP/ResidualKernel16BitTest": 66, "HIP/Downsample2DTest": 12, "HIP/EstimateNoiseTestFP": 49, "HIP/EstimateNoiseTestFPHbd": 49,
}
# The reference sized three of its tests for a function call that costs nanoseconds: MultipleQ (256 000 calls per parameter set, 47 sets) and the two CDEF direction
# tests (3.9 million calls each) are 45 + 21 CPU-minutes of PCIe round trips through the per-call `_hip` symbols.  They run -- and pass: profiles/r06_reference_fixtures.txt --
# with SVT_HIP_FIXTURES=full; the default run replaces each with a derived fixture that keeps the generator, the reference call and the checks and cuts the repetition
# count (tests/ref_fixtures/hip_quantize_func_test.cc, hip_CdefTest.cc).
FULL = os.environ.get("SVT_HIP_FIXTURES") == "full"
FILTER = "HIP*" if FULL else "HIP*:-HIPFULL*:*.MultipleQ/*"
FULL_ONLY = {"HIPFULL/CDEFFindDirTest": 1, "HIPFULL/CDEFFindDirDualTest": 1}


def _list(binary):
    out = subprocess.run([binary, "--gtest_list_tests", "--gtest_filter=" + FILTER], capture_output=True, text=True, check=True).stdout
    suites, cur = {}, None
    for line in out.splitlines():
        if line and not line.startswith(" "):
            cur = line.split("#")[0].strip().rstrip(".")
            suites[cur] = 0
        elif line.strip() and cur is not None and not line.strip().startswith("DISABLED_"):
            suites[cur] += 1
    return suites


def _check_suite_list(binary):
    got = _list(binary)
    want_suites = dict(SUITES, **(FULL_ONLY if FULL else {}))
    assert set(got) == set(want_suites), (sorted(set(want_suites) - set(got)), sorted(set(got) - set(want_suites)))
    if not FULL:  # (the MultipleQ cases of the four quantiser suites are left to the full run: their suites list 5 instead of 6 tests per parameter set)
        want_suites = dict(want_suites, **{k: want_suites[k] // 6 * 5 for k in ("HIP/QuantizeLbdTest", "HIP/QuantizeHbdTest", "HIP/QuantizeQmTest", "HIP/QuantizeQmHbdTest")})
    for name, want in want_suites.items():
        assert got[name] > 0 and (want is None or got[name] == want), (name, got[name], want)
    return sum(got.values())


def _run_sharded(binary, gtest_filter, shards, timeout):
    """gtest's own sharding (GTEST_TOTAL_SHARDS / GTEST_SHARD_INDEX): `shards` processes side by side, each running every shards-th test."""
    tmp = tempfile.mkdtemp(prefix="ref_fixtures_")

    def one(i):
        env = dict(os.environ, GTEST_TOTAL_SHARDS=str(shards), GTEST_SHARD_INDEX=str(i))
        js = os.path.join(tmp, "shard%d.json" % i)
        r = subprocess.run([binary, "--gtest_filter=" + gtest_filter, "--gtest_output=json:" + js], capture_output=True, text=True, env=env, timeout=timeout)
        return r, js

    with ThreadPoolExecutor(shards) as ex:
        results = list(ex.map(one, range(shards)))
    keep = os.path.join(ROOT, "gpurun_out", "ref_fixtures")  # (the per-test times of the last run, for profiles/: gpurun merges gpurun_out/ back)
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")) and "Emu" not in binary:
        import shutil
        shutil.rmtree(keep, ignore_errors=True)
        shutil.copytree(tmp, keep)
    tests = failures = 0
    failed = []
    for r, js in results:
        assert os.path.isfile(js), "a shard died before writing its report:\n" + (r.stdout[-3000:] + r.stderr[-3000:])
        rep = json.load(open(js))
        tests += rep["tests"] - rep.get("disabled", 0)
        failures += rep["failures"] + rep.get("errors", 0)
        for s in rep["testsuites"]:
            for t in s["testsuite"]:
                if t.get("failures"):
                    failed.append("%s.%s" % (s["name"], t["name"]))
        assert r.returncode == 0 or rep["failures"], (r.returncode, r.stdout[-3000:], r.stderr[-3000:])  # (rc 3 / 4: no device, or the device path switched itself off)
    return tests, failures, failed


@pytest.mark.gpu
def test_reference_fixtures_on_the_hip_symbols():
    assert os.path.isfile(BIN_GPU), "oracle/_ref/fixtures/SvtAv1HipFixtures is missing: __graft_entry__.build() makes it where /root/reference exists, and it ships with the snapshot"
    total = _check_suite_list(BIN_GPU)
    shards = max(2, min(24, (os.cpu_count() or 4)))
    tests, failures, failed = _run_sharded(BIN_GPU, FILTER, shards, timeout=2400 if FULL else 900)
    assert failures == 0, failed[:40]
    assert tests >= total, (tests, total)


# what the CPU interpreter gets through in well under a minute: whole suites where they are small, single cases of the heavy ones
EMU_SLICE = ("HIP/SADTest.*:HIP/Allsad*:HIP/Extsad*:HIP/InitializeBuffer32.*:HIP/SADTestSubSample16bit.*:HIP_MotionEstimation.*:HIP/SatdTest.*:HIP/HadamardLowbdTest.*:"
             "HIP/HandleTransformTest.*:HIP/sad_LoopTest.sad_LoopTest/0:HIP/PmeSadLoopTest.PmeSadLoopTest/0:HIP/FwdTxfm2dAsmTest.match_test/0:"
             "HIP_N2/FwdTxfm2dAsmTest.match_test/3:HIP/InvTxfm2dAsmSqrTest.sqr_txfm_match_test/2:HIP/InvTxfm2dAddTest.*:HIP_LBD/QuantizeBTest.input_zero_all/0:"
             "HIP/QuantizeLbdTest.DcOnlyInput/0:HIP/CDEFBlockTest.MatchTest/0:HIP/LbdLoopFilterTest.MatchTestRandomData/0:HIP/ResidualKernel8BitTest.MatchTest/0")


@pytest.mark.skipif(not (os.path.isfile(BIN_EMU) or HAVE_REF), reason="the fixture binary is built where the reference is")
def test_reference_fixtures_slice_on_the_emulator():
    if not os.path.isfile(BIN_EMU):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "svt-av1-psy_amd", "csrc"), "-j8", "emu"], check=True)
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "ref_fixtures"), "-j8"], check=True)
    _check_suite_list(BIN_EMU)
    tests, failures, failed = _run_sharded(BIN_EMU, EMU_SLICE, 4, timeout=600)
    assert failures == 0, failed
    assert tests >= 100, tests


def test_wrappers_hold_no_reference_text():
    """Each hip_<File>.cc includes the reference's file where it lies; nothing of it is copied in (the longest run of wrapper lines found verbatim in the included file stays
    at the size of an INSTANTIATE parameter list)."""
    wdir = os.path.join(ROOT, "tests", "ref_fixtures")
    wraps = sorted(f for f in os.listdir(wdir) if re.fullmatch(r"hip_\w+\.cc", f))
    assert len(wraps) >= 17
    for w in wraps:
        text = open(os.path.join(wdir, w)).read()
        inc = re.search(r'#include "(\w+\.cc)"', text)
        assert inc and inc.group(1) == w[len("hip_"):], w
        assert "INSTANTIATE_TEST_SUITE_P(" in text or "TEST(HIP_" in text, w
        if HAVE_REF:
            ref_lines = {l.strip() for l in open(os.path.join("/root/reference/test", inc.group(1))).read().splitlines() if len(l.strip()) >= 30}
            own = [l.strip() for l in text.splitlines() if len(l.strip()) >= 30 and not l.strip().startswith("//")]
            same = [l for l in own if l in ref_lines]
            assert len(same) <= max(3, len(own) // 5), (w, same[:5])
