"""Compile-time proof of the drop-in boundary: every `_hip` variant csrc/rtcd_hook.hip installs has exactly the prototype of the reference's dispatch
pointer it is installed into, and the header's self-contained PODs have the layout of the reference structs they stand for.

tests/abi/abi_typecheck.c turns every line of svt-av1-psy_amd/csrc/rtcd_hooks.def -- the one list rtcd_hook.hip declares the weak pointers from and installs from --
into `pointer = variant;`, compiled against the reference's own aom_dsp_rtcd.h / common_dsp_rtcd.h with -Werror=incompatible-pointer-types.  Needs the reference's
headers, so it runs in the build container only (the GPU box has no /root/reference)."""
import os
import re
import subprocess

import pytest

from conftest import PKG_DIR, ROOT

REF = os.environ.get("SVT_REF", "/root/reference")
SRC = os.path.join(REF, "Source")
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(SRC, "Lib", "Codec", "aom_dsp_rtcd.h")), reason="the reference's headers are not on this machine")

INC = ["-I" + os.path.join(SRC, "API"), "-I" + os.path.join(SRC, "Lib", "Codec"), "-I" + os.path.join(SRC, "Lib", "C_DEFAULT"), "-I" + os.path.join(SRC, "Lib", "Globals"),
       "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG_DIR, "csrc")]
STRICT = ["-std=gnu11", "-fsyntax-only", "-Wall", "-Werror=incompatible-pointer-types", "-Werror=discarded-qualifiers", "-Werror=int-conversion",
          "-Werror=implicit-function-declaration"]
N_HOOKS = 193


def _cc(args, src_text=None, path=None):
    cmd = ["gcc", *STRICT, *INC, *args]
    if src_text is not None:
        return subprocess.run(cmd + ["-x", "c", "-"], input=src_text, capture_output=True, text=True)
    return subprocess.run(cmd + [path], capture_output=True, text=True)


def _hooks():
    txt = open(os.path.join(PKG_DIR, "csrc", "rtcd_hooks.def")).read()
    return re.findall(r"^HOOK\((\w+), (\w+)\)", txt, re.M)


def test_hook_list_is_the_193_pointers_and_the_only_list():
    hooks = _hooks()
    assert len(hooks) == N_HOOKS
    assert len({p for p, _ in hooks}) == N_HOOKS, "a dispatch pointer is listed twice"
    hip = open(os.path.join(PKG_DIR, "csrc", "rtcd_hook.hip")).read()
    # rtcd_hook.hip holds no list of its own: both the weak declarations and the installation come from the .def
    assert hip.count('#include "rtcd_hooks.def"') == 2 and not re.search(r"^\s*HOOK\(svt_", hip, re.M)
    # every pointer is declared RTCD_EXTERN by one of the reference's two dispatch headers
    decl = open(os.path.join(SRC, "Lib", "Codec", "aom_dsp_rtcd.h")).read() + open(os.path.join(SRC, "Lib", "Codec", "common_dsp_rtcd.h")).read()
    missing = [p for p, _ in hooks if not re.search(r"RTCD_EXTERN[^;]*\(\s*\*\s*%s\s*\)" % re.escape(p), decl)]
    assert not missing, missing


def test_every_hip_variant_has_its_pointers_prototype():
    r = _cc([], path=os.path.join(ROOT, "tests", "abi", "abi_typecheck.c"))
    assert r.returncode == 0 and "warning" not in r.stderr, r.stderr[-6000:]
    # the proof has teeth: the same translation unit with one variant installed into a pointer of another prototype is rejected ...
    base = open(os.path.join(ROOT, "tests", "abi", "abi_typecheck.c")).read()
    for ptr, wrong in (("svt_aom_satd", "svt_nxm_sad_kernel_hip"),                      # different arity
                       ("svt_aom_highbd_quantize_b", "svt_av1_highbd_quantize_fp_hip"), # last parameter int32_t vs int16_t, one more / fewer pointer
                       ("svt_av1_inv_txfm2d_add_16x16", "svt_av1_inv_txfm2d_add_16x8_hip"),  # rectangular sizes carry tx_size
                       ("svt_aom_sad64x64x4d", "svt_aom_sad64x64_hip")):
        bad = base.replace("    return n;", "    %s = %s;\n    return n;" % (ptr, wrong))
        r = _cc([], src_text=bad)
        assert r.returncode != 0 and "incompatible-pointer-types" in r.stderr, (ptr, wrong)
    # ... and so is a by-value struct or enum parameter of the wrong type
    bad = base.replace('#include "svtav1_hip.h"', '#include "svtav1_hip.h"\nuint32_t hadamard_path_wrong(SvtHipBuf2D a, SvtHipBuf2D b, SvtHipBuf2D c, const SvtHipBuf2D *d, SvtHipBlockSize e);')
    bad = bad.replace("    return n;", "    hadamard_path = hadamard_path_wrong;\n    return n;")
    r = _cc([], src_text=bad)
    assert r.returncode != 0 and "incompatible-pointer-types" in r.stderr


def test_mirrored_pods_have_the_reference_layout():
    r = _cc([], path=os.path.join(ROOT, "tests", "abi", "abi_layout.c"))
    assert r.returncode == 0 and "warning" not in r.stderr, r.stderr[-6000:]
    # teeth: the mirror this header shipped until round 6 (tx_set_type as an int32_t; the reference's TxSetType is a packed one-byte enum) is caught
    bad = open(os.path.join(ROOT, "tests", "abi", "abi_layout.c")).read() + (
        "typedef struct { uint8_t tx_type, tx_size; int32_t lossless, bd, is_hbd; int32_t tx_set_type; int32_t eob; } OldTxfmParam;\n"
        "SAME_FIELD(OldTxfmParam, TxfmParam, tx_set_type);\n")
    r = _cc([], src_text=bad)
    assert r.returncode != 0 and "static assertion failed" in r.stderr
