"""SURVEY 8b "errors": a `_hip` variant must never propagate an error, and the library must not kill the encoder (VERDICT r4 missing #3).
  * the sticky switch-off through the C ABI (tests/error_policy_worker.py: emulator here, the MI355X with -m gpu);
  * inside the REAL reference encoder on the emulator with a HIP failure injected at the N-th device operation (SVT_HIPEMU_FAIL_AFTER, tests/emu/hipemu.h): the encode
    completes, the bitstream equals the C-only encoder's, the library reported the switch-off -- whether the failure hits the initialisation, a stage in the middle
    of the encode (every seam on), or a per-call `_hip` variant in flight (which finishes through the saved dispatch pointer)."""
import importlib.util
import os
import subprocess
import sys

import pytest

from conftest import EMU_LIB, ROOT

spec = importlib.util.spec_from_file_location("enc_identity", os.path.join(ROOT, "tools", "enc_identity.py"))
enc_identity = importlib.util.module_from_spec(spec)
spec.loader.exec_module(enc_identity)


def _worker(backend):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "error_policy_worker.py"), ROOT, backend], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ERROR_POLICY_OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "the device path is off from here on" in r.stderr


def test_error_policy_c_abi_emulator():
    from conftest import EmuBackend
    EmuBackend()
    _worker("emu")


@pytest.mark.gpu
def test_error_policy_c_abi_gpu():
    _worker("gpu")


@pytest.mark.skipif(not os.path.exists(enc_identity.ENC), reason="oracle/_ref/enc/SvtAv1EncApp not built (reference sources absent)")
@pytest.mark.parametrize("case", ["tiny_fail_at_init", "tiny_fail_hooks_p8", "tiny_fail_everyseam_p8_a", "tiny_fail_everyseam_p8_b", "tiny_fail_everyseam_p4"])
def test_encoder_survives_injected_failure(case, tmp_path):
    from conftest import EmuBackend
    EmuBackend()
    res = enc_identity.run_case(case, EMU_LIB, str(tmp_path), timeout=900)
    assert res["rc_c"] == 0 and res["rc_hip"] == 0, res.get("stderr_tail")
    assert res["bitstream_equal"], "the bitstream differs after the injected failure: %s" % case
    assert res["device_path_switched_off"], "the failure was not injected (too few device operations in this case?)"


@pytest.mark.skipif(not os.path.exists(enc_identity.ENC), reason="oracle/_ref/enc/SvtAv1EncApp not built (reference sources absent)")
@pytest.mark.parametrize("host", ["c", "avx2"])
@pytest.mark.parametrize("case", ["tiny_fail_dlfseam_p4", "tiny_fail_dlfseam_sb_p8"])
def test_deblocking_replay_after_injected_failure(case, host, tmp_path):
    """The deblocking seam's fallback: its device call fails, the recorded segments go through the reference's OWN edge filters.  On the AVX2 host those are the SSE2 / AVX2
    kernels, which load 16 bytes from each threshold pointer (dlf_intrin_sse2.c:273,605,634): the replay hands them replicated 16-byte arrays."""
    if host == "avx2" and not os.path.exists(enc_identity.ENC_AVX2):
        pytest.skip("oracle/_ref/enc_avx2 not built")
    from conftest import EmuBackend
    EmuBackend()
    res = enc_identity.run_case(case, EMU_LIB, str(tmp_path), timeout=900, host=host)
    assert res["rc_c"] == 0 and res["rc_hip"] == 0, res.get("stderr_tail")
    assert res["device_path_switched_off"], "the failure was not injected"
    assert res["bitstream_equal"], "the bitstream differs after the deblocking replay (%s host)" % host
    st = dict(ln.split() for ln in open(os.path.join(str(tmp_path), case + "_dlfseam.txt")).read().splitlines() if ln.strip())
    assert int(st["planes_declined"]) > 0, "no plane went through the replay"
