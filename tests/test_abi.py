"""The drop-in boundary itself: every function include/svtav1_hip.h declares is exported by the product library (svt-av1-psy_amd/libsvtav1_hip.so, loaded here WITHOUT
touching a device) and by the CPU emulator build of the same sources, and the ctypes table of the package lists exactly the declared names -- no compute calls."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT, load_pkg

HEADER = os.path.join(ROOT, "include", "svtav1_hip.h")
PRODUCT = os.path.join(ROOT, "svt-av1-psy_amd", "libsvtav1_hip.so")
EMU = os.path.join(ROOT, "tests", "emu", "_build", "libsvtav1_hipemu.so")


def declared():
    # the header through the C preprocessor (families of prototypes are declared by macros), system headers dropped
    src = subprocess.run(["gcc", "-E", "-P", "-x", "c", HEADER], capture_output=True, text=True, check=True).stdout
    src = re.sub(r"typedef\s+struct[^{;]*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)  # struct bodies (function-pointer fields are not exports)
    names = re.findall(r"\b((?:svt_|hadamard_)[A-Za-z0-9_]*)\s*\(", src)
    return sorted(set(names))


def exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if " T " in ln or " W " in ln}


def test_header_declares_functions():
    names = declared()
    assert len(names) > 150 and "svt_hip_init" in names and "svt_hip_tf_picture_host" in names and "svt_hip_tpl_src_stage" in names


@pytest.mark.parametrize("which", ["product", "emulator"])
def test_library_exports_every_declared_symbol(which):
    path = PRODUCT if which == "product" else EMU
    if not os.path.exists(path):
        if which == "product":
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "svt-av1-psy_amd", "csrc"), "-j8"], check=True)
        else:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "svt-av1-psy_amd", "csrc"), "-j8", "emu"], check=True)
    have = exported(path)
    missing = [n for n in declared() if n not in have]
    assert not missing, "declared in include/svtav1_hip.h but not exported by %s: %s" % (os.path.basename(path), missing)


def test_product_library_loads_without_a_device():
    lib = C.CDLL(PRODUCT)  # resolves its ROCm dependencies; no HIP call is made
    for n in declared():
        assert hasattr(lib, n), n


def test_package_table_matches_the_header():
    pkg = load_pkg()
    table, names = set(pkg.PROTOTYPES), set(declared())
    assert not (names - table), "declared but missing from the package's ctypes table: %s" % sorted(names - table)
    assert not (table - names), "in the package's ctypes table but not declared by the header: %s" % sorted(table - names)
