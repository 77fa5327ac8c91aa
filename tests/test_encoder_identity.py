"""Integration-level parity (VERDICT r1 row g): the HIP variant installed in the REAL reference encoder must leave the
bitstream byte-identical to the C-only encoder (and with it the reconstruction, a function of the bitstream) -- the reference's own CI invariant across ISA levels
(.gitlab/workflows/linux/.gitlab-ci.yml:351-367).  oracle/_ref/enc/SvtAv1EncApp = the reference built C-only by oracle/Makefile
with the binding of INTEGRATION.md §1 (integration/enc_handle_binding.c).

CPU (`-m "not gpu"`): tiny clips through the lock-step emulator build of the same kernel sources.
GPU (`-m gpu`): 256x144 clips, presets 4 / 6 / 8, 8- and 10-bit, --lp 1 / 2 / 4, quantisation matrices, lossless, through libsvtav1_hip.so;
the per-pointer call counts land in gpurun_out/identity/identity.json (copied to profiles/ per round)."""
import importlib.util
import os
import sys

import pytest

from conftest import EMU_LIB, PKG_DIR, ROOT

spec = importlib.util.spec_from_file_location("enc_identity", os.path.join(ROOT, "tools", "enc_identity.py"))
enc_identity = importlib.util.module_from_spec(spec)
spec.loader.exec_module(enc_identity)

needs_encoder = pytest.mark.skipif(not os.path.exists(enc_identity.ENC), reason="oracle/_ref/enc/SvtAv1EncApp not built (reference sources absent)")


def _check(res):
    if not res.get("reference_deterministic", True):  # (no such case in the committed lists; see tools/enc_identity.py on 10-bit preset 8 with --lp >= 2)
        pytest.skip("the C-only reference encoder does not reproduce its own bitstream for this configuration")
    assert res["rc_c"] == 0 and res["rc_hip"] == 0, res.get("stderr_tail")
    assert res["hook_line"], "the encoder did not install the HIP variant"
    assert res["identical"], "the bitstream differs from the C-only encoder: %s" % res["case"]
    if "dlfseam" in res and res["case"].startswith(("dlfseam_", "tiny_dlfseam")):  # deblocking segments were filtered on the device
        assert res["dlfseam"].get("segments", 0) > 0, res["dlfseam"]
    if "dlfseam" in res and "_sb_" in res["case"]:  # presets >= 7: the segments came from the per-SB records of the coding loop
        assert res["dlfseam"].get("pictures_filtered_from_sb_records", 0) > 0, res["dlfseam"]
    if "cdefseam" in res:  # pictures were CDEF-filtered on the device, none declined
        assert res["cdefseam"]["filter_blocks"] > 0 and res["cdefseam"]["pictures_declined"] == 0, res["cdefseam"]
    if "lrseam" in res:  # restoration units were searched on the device
        assert res["lrseam"]["units_searched"] > 0 and res["lrseam"]["pictures_offloaded"] > 0, res["lrseam"]
    if "seam" in res:  # the ME stage ran as one device call per picture for EVERY inter picture (a declined picture would run the reference's C code)
        assert res["seam"]["pictures_offloaded"] > 0 and res["seam"]["pictures_declined"] == 0, res["seam"]
    if "strips" in res:  # the frame-partition case: frame launches of the in-loop filter host forms went through a partition
        assert res["strips"]["frame_launches_through_a_partition"] > 0, res["strips"]
    if "devices" in res:  # one encode over several (emulated) GPUs: every device received stage calls
        assert len(res["devices"]) >= 2 and all(v > 0 for v in res["devices"].values()), res["devices"]
    if "tplseam" in res and res["case"].startswith(("tplseam_", "tiny_tplseam")):  # the TPL source-based statistics of every picture came from the device stage
        assert res["tplseam"]["pictures_offloaded"] > 0 and res["tplseam"]["pictures_declined"] == 0 and res["tplseam"]["blocks"] > 0, res["tplseam"]
    if "tplseam" in res and "tplrecon" in res["case"]:  # the reconstruction half of every dispenser call ran on the device; the reference's per-SB function was skipped
        t = res["tplseam"]
        assert t["recon_pictures"] > 0 and t["recon_blocks_coded"] > 0 and t["sb_calls_skipped"] > 0 and t["pictures_declined"] == 0, t
    if "tfdriver" in res:  # central pictures were temporally filtered by the device stage, none left to the reference
        assert res["tfdriver"]["pictures_filtered"] > 0 and res["tfdriver"]["pictures_declined"] == 0 and res["tfdriver"]["reference_frames"] > 0, res["tfdriver"]
    if "seam" in res:
        pass
    elif "lrseam" not in res and "cdefseam" not in res and "dlfseam" not in res and "tplseam" not in res and "tfdriver" not in res:
        assert res["pointers_hit"] >= 20 and res["calls"] > 1000, res


@needs_encoder
@pytest.mark.parametrize("case", ["tiny_p8_8bit", "tiny_p8_10bit", "tiny_p8_lossless", "tiny_seam_p8", "tiny_seam_p5_lp2", "tiny_lrseam_p4", "tiny_cdefseam_p8", "tiny_dlfseam_p4", "tiny_tfseam_p8", "tiny_tfsubpel_p8", "tiny_tfdriver_p8", "tiny_tfdriver_p8_10bit", "tiny_tplseam_p8", "tiny_tplseam_p10", "tiny_dlfseam_sb_p8", "tiny_dlfseam_sb_p8_lp2", "tiny_2dev_everyseam_p8",
                                  "tiny_lowdelay_p8", "tiny_lowdelay_p10_10bit", "tiny_lowdelay_720p_tf",
                                  "tiny_screen_p8", "tiny_screen_lowdelay_p9",
                                  "tiny_tplrecon_p8", "tiny_tplrecon_p10", "tiny_tiles_p8",
                                  "tiny_tplrecon_p2",  # tpl level 1 (presets <= M2): every intra mode, SATD costs, quarter-pel vectors, rate (csrc/tpl_full.hip)
                                  "tiny_strips_cdef_lr_p4"])  # SVT_HIP_STRIPS: one picture's CDEF / LR launches over two emulated devices  # both halves of the TPL dispenser as device stages  # screen content: enable_me_sr_adjustment == 2  # low delay: level-0 HME areas from list-0 motion; the zero-motion temporal filter (on from 720p)
def test_encoder_identity_emulator(case, tmp_path):
    from conftest import EmuBackend  # builds the emulator library if needed
    EmuBackend()
    _check(enc_identity.run_case(case, EMU_LIB, str(tmp_path), timeout=900))


@needs_encoder
@pytest.mark.gpu
@pytest.mark.parametrize("case", enc_identity.GPU_CASES)
def test_encoder_identity_gpu(case):
    lib = os.path.join(PKG_DIR, "libsvtav1_hip.so")
    assert os.path.exists(lib), "libsvtav1_hip.so missing (no CPU fallback)"
    out = os.path.join(ROOT, "gpurun_out", "identity")
    res = enc_identity.run_case(case, lib, out, timeout=1500)
    import json
    with open(os.path.join(out, case + ".json"), "w") as f:
        json.dump(res, f, indent=1)
    _check(res)


@needs_encoder
@pytest.mark.gpu
@pytest.mark.parametrize("case", ["everyseam_p4_8bit_lp2", "tplseam_1080p_p8", "allseams_1080p_p6", "p8_8bit_lp1"])
def test_encoder_identity_gpu_avx2_host(case):
    """The same seams inside the reference built WITH its x86 intrinsic kernels (oracle/_ref/enc_avx2): that encoder alone reproduces the C-only bitstream, and so
    does it with the stages on the MI355X -- the configuration a deployment would run."""
    if not os.path.exists(enc_identity.ENC_AVX2):
        pytest.skip("oracle/_ref/enc_avx2 not built")
    lib = os.path.join(PKG_DIR, "libsvtav1_hip.so")
    assert os.path.exists(lib), "libsvtav1_hip.so missing (no CPU fallback)"
    res = enc_identity.run_case(case, lib, os.path.join(ROOT, "gpurun_out", "identity_avx2"), timeout=1500, host="avx2")
    assert res.get("avx2_identical_to_c"), "the intrinsics encoder does not reproduce the C-only bitstream"
    _check(res)


@needs_encoder
@pytest.mark.parametrize("case", ["tiny_tplseam_p8", "tiny_seam_p8"])
def test_encoder_identity_emulator_avx2_host(case, tmp_path):
    if not os.path.exists(enc_identity.ENC_AVX2):
        pytest.skip("oracle/_ref/enc_avx2 not built")
    from conftest import EmuBackend
    EmuBackend()
    res = enc_identity.run_case(case, EMU_LIB, str(tmp_path), timeout=900, host="avx2")
    assert res.get("avx2_identical_to_c")
    _check(res)
