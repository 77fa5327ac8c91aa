"""Drop-in boundary: svt_hip_setup_rtcd() must overwrite the REFERENCE's own dispatch pointers (weak symbols resolved against
libsvtref.so = the reference's Codec objects) and calls made THROUGH those pointers must land in our variants."""
import ctypes as C
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import EMU_LIB, REF_LIB, ROOT

CHILD = textwrap.dedent('''
    import ctypes as C, os, sys
    import numpy as np
    ref = C.CDLL(sys.argv[1], mode=os.RTLD_GLOBAL | os.RTLD_NOW)          # defines the RTCD pointer globals
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0)); ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    before = C.c_void_p.in_dll(ref, "svt_nxm_sad_kernel").value
    hip = C.CDLL(sys.argv[2], mode=os.RTLD_GLOBAL | os.RTLD_NOW)          # weak references bind to the globals above
    hip.svt_hip_init(0)
    n = hip.svt_hip_setup_rtcd(C.c_uint64(0))
    after = C.c_void_p.in_dll(ref, "svt_nxm_sad_kernel").value
    ours = C.cast(hip.svt_nxm_sad_kernel_hip, C.c_void_p).value
    assert n == 193, n   # one number everywhere: DESIGN.md section 1, INTEGRATION.md section 1, README.md
    # what is installed is the per-pointer GUARD of the `_hip` variant (csrc/rtcd_hook.hip: same signature; it finishes the call through the saved pointer on a HIP error)
    assert before != after and after not in (None, 0) and ours not in (None, 0), (before, after, ours)
    for name in ("svt_av1_fwd_txfm2d_32x32", "svt_av1_inv_txfm2d_add_16x64", "svt_aom_quantize_b", "svt_cdef_filter_block", "svt_av1_wiener_convolve_add_src",
                 "svt_aom_cdef_find_dir", "svt_av1_compute_stats", "svt_handle_transform64x64", "svt_ext_all_sad_calculation_8x8_16x16", "hadamard_path"):
        assert C.c_void_p.in_dll(ref, name).value not in (None, 0)
    # call through the reference's pointer: 64x64 SAD and a 16x16 forward transform
    g = np.random.default_rng(0)
    a, b = g.integers(0, 256, 64 * 64, dtype=np.uint8), g.integers(0, 256, 64 * 64, dtype=np.uint8)
    fp = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32)(after)
    assert fp(a.ctypes.data, 64, b.ctypes.data, 64, 64, 64) == int(np.abs(a.astype(np.int32) - b).sum())
    res = g.integers(-255, 256, 256).astype(np.int16)
    o1, o2 = np.zeros(256, np.int32), np.zeros(256, np.int32)
    ft = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint8, C.c_uint8)(C.c_void_p.in_dll(ref, "svt_av1_fwd_txfm2d_16x16").value)
    ft(res.ctypes.data, o1.ctypes.data, 16, 3, 8)
    ref.svt_av1_transform_two_d_16x16_c(C.c_void_p(res.ctypes.data), C.c_void_p(o2.ctypes.data), 16, 3, 8)
    assert np.array_equal(o1, o2)
    # ---- more pointers, each called THROUGH the reference's table and compared with the reference's own _c function on the same arguments
    def through(name, restype, *argtypes):
        return C.CFUNCTYPE(restype, *argtypes)(C.c_void_p.in_dll(ref, name).value)
    vp, i32, u32, u8 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint8
    # svt_av1_inv_txfm_add with the whole TxfmParam: regular 8x8 / 16x16, and lossless 4x4 (Walsh-Hadamard) in both eob forms (inv_transforms.c:2826-2848)
    class TxfmParam(C.Structure):
        _fields_ = [("tx_type", u8), ("tx_size", u8), ("lossless", i32), ("bd", i32), ("is_hbd", i32), ("tx_set_type", u8), ("eob", i32)]
    f = through("svt_av1_inv_txfm_add", None, vp, vp, i32, vp, i32, vp)
    for (ts, w, h, lossless, eob, tt) in ((1, 8, 8, 0, 64, 3), (2, 16, 16, 0, 256, 0), (0, 4, 4, 1, 16, 0), (0, 4, 4, 1, 1, 0), (0, 4, 4, 0, 16, 5)):
        co = g.integers(-2000, 2001, w * h).astype(np.int32)
        dst = g.integers(0, 256, h * (w + 3)).astype(np.uint8)
        a, b = dst.copy(), dst.copy()
        tp = TxfmParam(tt, ts, lossless, 8, 0, 0, eob)
        f(co.ctypes.data, a.ctypes.data, w + 3, a.ctypes.data, w + 3, C.addressof(tp))
        ref.svt_av1_inv_txfm_add_c(C.c_void_p(co.ctypes.data), C.c_void_p(b.ctypes.data), w + 3, C.c_void_p(b.ctypes.data), w + 3, C.byref(tp))
        assert np.array_equal(a, b), ("svt_av1_inv_txfm_add", ts, lossless, eob)
    # svt_av1_fwht4x4
    r4 = g.integers(-255, 256, 4 * 9).astype(np.int16)
    w1, w2 = np.zeros(16, np.int32), np.zeros(16, np.int32)
    through("svt_av1_fwht4x4", None, vp, vp, u32)(r4.ctypes.data, w1.ctypes.data, 9)
    ref.svt_av1_fwht4x4_c(C.c_void_p(r4.ctypes.data), C.c_void_p(w2.ctypes.data), 9)
    assert np.array_equal(w1, w2)
    # hadamard_path: four Buf2D structs BY VALUE (definitions.h:243-249) + BlockSize, every square block size from 8x8 to 64x64
    class Buf2D(C.Structure):
        _fields_ = [("buf", vp), ("buf0", vp), ("width", C.c_int), ("height", C.c_int), ("stride", C.c_int)]
    fh = through("hadamard_path", u32, Buf2D, Buf2D, Buf2D, Buf2D, u8)
    ref.hadamard_path_c.restype = u32
    ref.hadamard_path_c.argtypes = [Buf2D, Buf2D, Buf2D, Buf2D, u8]
    inp, prd = g.integers(0, 256, 64 * 80).astype(np.uint8), g.integers(0, 256, 64 * 72).astype(np.uint8)
    for bsize, bw in ((3, 8), (6, 16), (9, 32), (12, 64)):  # BLOCK_8X8, BLOCK_16X16, BLOCK_32X32, BLOCK_64X64 (definitions.h BlockSize)
        res, cof = np.zeros(64 * 64, np.int16), np.zeros(64 * 64, np.int32)
        mk = lambda arr, st: Buf2D(arr.ctypes.data, None, 0, 0, st)
        got = fh(mk(res, bw), mk(cof, bw), mk(inp, 80), mk(prd, 72), bsize)
        want = ref.hadamard_path_c(mk(res, bw), mk(cof, bw), mk(inp, 80), mk(prd, 72), bsize)
        assert got == want, ("hadamard_path", bsize, got, want)
    # svt_aom_satd, svt_aom_hadamard_16x16
    cf = g.integers(-30000, 30001, 1024).astype(np.int32)
    ref.svt_aom_satd_c.restype = C.c_int
    assert through("svt_aom_satd", C.c_int, vp, C.c_int)(cf.ctypes.data, 1024) == ref.svt_aom_satd_c(C.c_void_p(cf.ctypes.data), 1024)
    rs = g.integers(-255, 256, 16 * 20).astype(np.int16)
    h1, h2 = np.zeros(256, np.int32), np.zeros(256, np.int32)
    through("svt_aom_hadamard_16x16", None, vp, C.c_ssize_t, vp)(rs.ctypes.data, 20, h1.ctypes.data)
    ref.svt_aom_hadamard_16x16_c(C.c_void_p(rs.ctypes.data), C.c_ssize_t(20), C.c_void_p(h2.ctypes.data))
    assert np.array_equal(h1, h2)
    # svt_sad_loop_kernel (in / out best SAD and centre)
    src, rf = g.integers(0, 256, 16 * 40).astype(np.uint8), g.integers(0, 256, 40 * 64).astype(np.uint8)
    outs = []
    for fn in (through("svt_sad_loop_kernel", None, vp, u32, vp, u32, u32, u32, vp, vp, vp, u32, u8, C.c_int16, C.c_int16), None):
        bs, xs, ys = C.c_uint64(0), C.c_int16(0), C.c_int16(0)
        args = (src.ctypes.data, 40, rf.ctypes.data, 64, 16, 16, C.addressof(bs), C.addressof(xs), C.addressof(ys), 64, 0, 24, 12)
        if fn:
            fn(*args)
        else:
            ref.svt_sad_loop_kernel_c(*[C.c_void_p(a) if i in (0, 2, 6, 7, 8) else a for i, a in enumerate(args)])
        outs.append((bs.value, xs.value, ys.value))
    assert outs[0] == outs[1], outs
    # the error policy (SURVEY 8b "errors"): the first HIP error puts every overwritten pointer back; a guard still held by a caller finishes through the saved pointer
    hip.svt_hip_debug_inject_failure()
    assert hip.svt_hip_failed() == 1
    assert C.c_void_p.in_dll(ref, "svt_nxm_sad_kernel").value == before, "the dispatch pointer was not restored"
    sa, sb = g.integers(0, 256, 64 * 64, dtype=np.uint8), g.integers(0, 256, 64 * 64, dtype=np.uint8)
    assert fp(sa.ctypes.data, 64, sb.ctypes.data, 64, 64, 64) == int(np.abs(sa.astype(np.int32) - sb).sum())  # (fp = the guard: now the reference's own kernel answers)
    print("HOOK_OK", n)
''')


def test_rtcd_hook_overwrites_reference_pointers(be):
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libsvtref.so not available")
    lib = EMU_LIB if not be.is_gpu else os.path.join(ROOT, "svt-av1-psy_amd", "libsvtav1_hip.so")
    r = subprocess.run([sys.executable, "-c", CHILD, REF_LIB, lib], capture_output=True, text=True, timeout=300)
    assert "HOOK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
