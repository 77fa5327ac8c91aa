"""Picture preparation for ME: 2x2 decimation (downsample_2d) and border replication (svt_aom_generate_padding).
The oracle restatement is pinned against the reference's objects; the HIP paths (RTCD single call, device-resident padded pyramid level,
in-place padding) are compared with the oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

from conftest import p, rng

SHAPES = [(64, 48), (70, 30), (1920, 1080), (426, 240), (33, 17), (8, 8)]


def test_picprep_oracle_vs_reference(oracle, ref):
    g = rng(91)
    ref.svt_aom_setup_common_rtcd_internal(C.c_uint64(0))  # svt_aom_generate_padding copies rows through the svt_memcpy pointer
    ref.svt_aom_setup_rtcd_internal(C.c_uint64(0))
    for (w, h) in SHAPES:
        for step in (2, 4):
            stride = w + 9
            src = g.integers(0, 256, (h + 1, stride), dtype=np.uint8)
            ow, oh = max((w - step // 2 + step - 1) // step, 0), max((h - step // 2 + step - 1) // step, 0)
            a, b = np.zeros((oh + 1, ow + 5), np.uint8), np.zeros((oh + 1, ow + 5), np.uint8)
            oracle.oracle_downsample_2d(p(src), stride, w, h, p(a), ow + 5, step)
            ref.svt_aom_downsample_2d_c(p(src), C.c_uint32(stride), C.c_uint32(w), C.c_uint32(h), p(b), C.c_uint32(ow + 5), C.c_uint32(step))
            assert np.array_equal(a, b), (w, h, step)
        for (px, py) in ((4, 4), (17, 5), (68, 68)):
            stride = w + 2 * px + 3
            a = g.integers(0, 256, (h + 2 * py, stride), dtype=np.uint8)
            b = a.copy()
            oracle.oracle_generate_padding(p(a), stride, w, h, px, py)
            ref.svt_aom_generate_padding(p(b), C.c_uint32(stride), C.c_uint32(w), C.c_uint32(h), C.c_uint32(px), C.c_uint32(py))
            assert np.array_equal(a[:, :w + 2 * px], b[:, :w + 2 * px]), (w, h, px, py)


def test_downsample_single_call(be, oracle):
    g = rng(92)
    for (w, h) in (SHAPES if be.is_gpu else SHAPES[:2] + SHAPES[4:]):
        for step in (2, 4):
            stride = w + 9
            src = g.integers(0, 256, (h + 1, stride), dtype=np.uint8)
            ow, oh = max((w - step // 2 + step - 1) // step, 0), max((h - step // 2 + step - 1) // step, 0)
            a, b = np.zeros((oh + 1, ow + 5), np.uint8), np.zeros((oh + 1, ow + 5), np.uint8)
            oracle.oracle_downsample_2d(p(src), stride, w, h, p(a), ow + 5, step)
            be.lib.svt_aom_downsample_2d_hip(p(src), stride, w, h, p(b), ow + 5, step)
            assert np.array_equal(a, b), (w, h, step)


def test_pyramid_and_padding_device(be, oracle):
    """full-res padded plane -> 1/4 plane (pad 32) -> 1/16 plane (pad 16), as svt_aom_downsample_filtering_input_picture, plus the
    1/16-from-full (step 4) variant and in-place padding of the full-resolution plane"""
    g = rng(93)
    for (w, h, pad) in ([(1920, 1080, 68), (426, 240, 68)] if be.is_gpu else [(96, 64, 12), (70, 34, 9)]):
        stride = w + 2 * pad + (0 if w % 8 == 0 else 5)
        full = g.integers(0, 256, (h + 2 * pad, stride), dtype=np.uint8)
        want_full = full.copy()
        oracle.oracle_generate_padding(p(want_full), stride, w, h, pad, pad)
        d_full = be.dev(full)
        be.lib.svt_hip_generate_padding(be.ptr(d_full), stride, w, h, pad, pad, be.stream)
        got_full = be.host(d_full).reshape(full.shape)
        assert np.array_equal(got_full[:, :w + 2 * pad], want_full[:, :w + 2 * pad])
        org = pad * stride + pad
        levels = [(2, 32), (4, 16)]
        for step, opad in levels:
            ow, oh = (w - step // 2 + step - 1) // step, (h - step // 2 + step - 1) // step
            ostride = ow + 2 * opad + 3
            want = np.zeros((oh + 2 * opad, ostride), np.uint8)
            oracle.oracle_downsample_2d(C.c_void_p(want_full.ctypes.data + org), stride, w, h, C.c_void_p(want.ctypes.data + opad * ostride + opad), ostride, step)
            oracle.oracle_generate_padding(p(want), ostride, ow, oh, opad, opad)
            d_out = be.dev(np.zeros_like(want))
            be.lib.svt_hip_downsample_2d_padded(be.ptr(d_full) + org, stride, w, h, be.ptr(d_out), ostride, opad, opad, step, be.stream)
            got = be.host(d_out).reshape(want.shape)
            assert np.array_equal(got[:, :ow + 2 * opad], want[:, :ow + 2 * opad]), (w, h, step)
            if step == 2:  # second level from the first (quarter -> sixteenth)
                ow2, oh2 = (ow - 1 + 1) // 2, (oh - 1 + 1) // 2
                os2 = ow2 + 2 * 16 + 1
                want2 = np.zeros((oh2 + 32, os2), np.uint8)
                oracle.oracle_downsample_2d(C.c_void_p(want.ctypes.data + opad * ostride + opad), ostride, ow, oh, C.c_void_p(want2.ctypes.data + 16 * os2 + 16), os2, 2)
                oracle.oracle_generate_padding(p(want2), os2, ow2, oh2, 16, 16)
                d2 = be.dev(np.zeros_like(want2))
                be.lib.svt_hip_downsample_2d_padded(be.ptr(d_out) + opad * ostride + opad, ostride, ow, oh, be.ptr(d2), os2, 16, 16, 2, be.stream)
                got2 = be.host(d2).reshape(want2.shape)
                assert np.array_equal(got2[:, :ow2 + 32], want2[:, :ow2 + 32]), (w, h, "level2")
