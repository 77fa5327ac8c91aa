"""Shared test plumbing.

Backends
--------
``be`` fixture = how a test reaches the kernels:
  * ``emu``  (CPU, unmarked)   -- the SAME kernel sources compiled by g++ against tests/emu/hipemu.h (lock-step SIMT
                                  interpreter, test infrastructure).  Catches indexing / reduction-order bugs here.
  * ``gpu``  (``-m gpu``)      -- the product: svt-av1-psy_amd/libsvtav1_hip.so through its C ABI on cuda:0, device
                                  buffers owned by torch.  If the .so is missing these tests FAIL (no fallback).
Checkers: ``oracle`` = oracle/liboracle.so (our C restatement), ``ref`` = oracle/_ref/libsvtref.so (the real
reference, only where it has been built: this container or the prebuilt copy shipped to the GPU box).
"""
import ctypes as C
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "svt-av1-psy_amd")
# (SVT_HIP_EMU_LIB: another build of the emulator -- e.g. g++ -fsanitize=address over the same sources, run as `LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest ...`:
#  device buffers are plain heap blocks there, so a kernel's out-of-bounds access is a report with a stack)
EMU_LIB = os.environ.get("SVT_HIP_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "_build", "libsvtav1_hipemu.so")
ORACLE_LIB = os.environ.get("SVT_HIP_ORACLE_LIB") or os.path.join(ROOT, "oracle", "liboracle.so")  # (the override: a sanitizer build of the checker, see tools/sanitizer_emulator.sh)
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libsvtref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "slow: long CPU-side sweep")


def load_pkg():
    if "svt_av1_psy_amd" in sys.modules:
        return sys.modules["svt_av1_psy_amd"]
    spec = importlib.util.spec_from_file_location("svt_av1_psy_amd", os.path.join(PKG_DIR, "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["svt_av1_psy_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def _make(target_dir, *args):
    r = subprocess.run(["make", "-s", "-C", target_dir, "-j8", *args], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("make %s failed:\n%s\n%s" % (" ".join(args), r.stdout[-4000:], r.stderr[-4000:]))


class EmuBackend:
    name = "emu"
    is_gpu = False

    def __init__(self):
        _make(os.path.join(PKG_DIR, "csrc"), "emu")
        self.pkg = load_pkg()
        self.lib = self.pkg.bind(C.CDLL(EMU_LIB))
        assert self.lib.svt_hip_init(0) == 0
        self.stream = None

    def dev(self, a):
        return np.ascontiguousarray(a).copy()

    def empty(self, shape, dtype):
        return np.zeros(shape, dtype=dtype)

    def ptr(self, a):
        return a.ctypes.data

    def host(self, a):
        return np.array(a, copy=True)

    def sync(self):
        pass


class GpuBackend:
    name = "gpu"
    is_gpu = True

    def __init__(self):
        import torch
        self.torch = torch
        assert torch.cuda.is_available(), "gpu backend requested without a GPU"
        self.pkg = load_pkg()
        self.lib = self.pkg.load(init_device=0)  # raises if libsvtav1_hip.so is missing: no silent fallback
        self.stream = torch.cuda.current_stream().cuda_stream

    def dev(self, a):
        a = np.ascontiguousarray(a)
        t = self.torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda()
        t._np_dtype, t._np_shape = a.dtype, a.shape
        return t

    def empty(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        t = self.torch.zeros(max(n, 1), dtype=self.torch.uint8, device="cuda")
        t._np_dtype, t._np_shape = np.dtype(dtype), tuple(np.atleast_1d(shape))
        return t

    def ptr(self, t):
        return t.data_ptr()

    def host(self, t):
        self.torch.cuda.synchronize()
        n = int(np.prod(t._np_shape)) * np.dtype(t._np_dtype).itemsize
        return t.cpu().numpy()[:n].view(t._np_dtype).reshape(t._np_shape).copy()

    def sync(self):
        self.torch.cuda.synchronize()


_backends = {}


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def be(request):
    k = request.param
    if k not in _backends:
        _backends[k] = EmuBackend() if k == "emu" else GpuBackend()
    return _backends[k]


@pytest.fixture(scope="session")
def oracle():
    _make(os.path.join(ROOT, "oracle"), "oracle")
    return C.CDLL(ORACLE_LIB)


@pytest.fixture(scope="session")
def ref():
    """The real reference (its `*_c` functions).  Built here from /root/reference; on the GPU box the prebuilt copy."""
    if os.path.isdir("/root/reference/Source"):
        _make(os.path.join(ROOT, "oracle"), "ref")
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libsvtref.so not available (reference sources absent and no prebuilt copy)")
    return C.CDLL(REF_LIB)


def p(a):
    """ctypes void* of a numpy array"""
    return C.c_void_p(a.ctypes.data)


def rng(seed=13596):  # the reference tests seed mt19937 with 13596 (test/random.h:141)
    return np.random.default_rng(seed)
