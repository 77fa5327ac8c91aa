"""svt-av1-psy_amd -- host-side binding of libsvtav1_hip.so (MI355X / gfx950 variant of SVT-AV1-PSY's block-DSP hot path).

The product is the C-ABI shared library declared in ``include/svtav1_hip.h``; the reference's host code is C and binds
it through its RTCD function-pointer table.  This module is the thin ctypes mirror used by
``bench.py``, ``__graft_entry__.py`` and the tests: it loads the library, declares every prototype, and offers
numpy/torch-friendly descriptors.  It has NO CPU implementation: if the HIP library is missing or no GPU is present,
``load()`` raises / the library aborts -- nothing silently falls back.

The directory name contains a hyphen (it mirrors the reference's name), so import it with
``importlib`` -- see ``tests/conftest.py::load_pkg``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsvtav1_hip.so")
ME_NUM_BLOCKS = 85
MAX_SAD_VALUE = 128 * 128 * 255

u8p, u16p, u32p, u64p, i16p, i32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_int16, C.c_int32))
vp = C.c_void_p

# numpy views of the descriptor structs of include/svtav1_hip.h
SadPair = np.dtype([("src_off", "<u8"), ("ref_off", "<u8"), ("src_stride", "<u4"), ("ref_stride", "<u4")])
SadLoopDesc = np.dtype([("src_off", "<u8"), ("ref_off", "<u8"), ("src_stride", "<u4"), ("ref_stride", "<u4"),
                        ("src_stride_raw", "<u4"), ("block_width", "<u2"), ("block_height", "<u2"),
                        ("search_area_width", "<i2"), ("search_area_height", "<i2"), ("skip_search_line", "u1"),
                        ("pad", "u1", (3,))], align=True)
SadLoopResult = np.dtype([("best_sad", "<u8"), ("x", "<i2"), ("y", "<i2"), ("valid", "<u4")])
MeSearchDesc = np.dtype([("src_off", "<u8"), ("ref_off", "<u8"), ("src_stride", "<u4"), ("ref_stride", "<u4"),
                         ("x_origin", "<i2"), ("y_origin", "<i2"), ("width", "<u2"), ("height", "<u2")])
assert SadPair.itemsize == 24 and SadLoopDesc.itemsize == 40 and SadLoopResult.itemsize == 16 and MeSearchDesc.itemsize == 32

# symbol -> (restype, argtypes); exactly the functions include/svtav1_hip.h declares (tests/test_abi.py checks both directions and the libraries' exports)
PROTOTYPES = {
    "svt_hip_init": (C.c_int, [C.c_int]),
    "svt_hip_shutdown": (None, []),
    "svt_hip_device_name": (C.c_char_p, []),
    "svt_hip_tuning_reload": (None, []),
    "svt_hip_warmup": (None, []),
    "svt_hip_host_register": (C.c_int, [vp, C.c_size_t]),
    "svt_hip_host_unregister": (C.c_int, [vp]),
    "svt_hip_device_count": (C.c_int, []),
    "svt_hip_last_error": (C.c_char_p, []),
    "svt_hip_failed": (C.c_int, []),
    "svt_hip_rtcd_unhook": (None, []),
    "svt_hip_debug_commit_violations": (C.c_uint64, []),
    "svt_hip_warmup_sized": (None, [C.c_int, C.c_uint32]),
    "svt_hip_mem_probe": (None, [C.c_int, C.c_int, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp]),
    "svt_hip_mem_probe_blocks": (None, [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]),
    "svt_hip_debug_inject_failure": (C.c_int, []),
    "svt_hip_physical_device_count": (C.c_int, []),
    "svt_hip_physical_device": (C.c_int, [C.c_int]),
    "svt_hip_set_virtual_devices": (C.c_int, [C.c_int]),
    "svt_hip_set_thread_device": (C.c_int, [C.c_int]),
    "svt_hip_get_thread_device": (C.c_int, []),
    "svt_hip_tpl_src_stage": (None, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "svt_hip_tpl_src_stage_host": (C.c_int, [vp, vp, vp, vp, vp, vp]),
    "svt_hip_tpl_recon_stage": (None, [vp, vp, vp, vp, vp, vp, vp]),
    "svt_hip_tpl_recon_stage_host": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp]),
    "svt_hip_tpl_stage_host": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint32, vp]),
    "svt_hip_tpl_stage_host_resident": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint32, vp]),
    "svt_hip_tpl_plane_drop": (None, [vp]),
    "svt_hip_tpl_plane_counts": (None, [vp, vp]),
    "svt_hip_frame_partition_create": (vp, [vp, C.c_int]),
    "svt_hip_frame_partition_destroy": (None, [vp]),
    "svt_hip_frame_partition_size": (C.c_int, [vp]),
    "svt_hip_debug_spin": (None, [vp, C.c_uint32]),
    "svt_hip_frame_partition_set_jitter": (None, [vp, C.c_uint32, C.c_uint32]),
    "svt_hip_frame_partition_stats": (None, [vp, vp, vp, vp]),
    "svt_hip_frame_partition_me": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp, vp]),
    "svt_hip_frame_partition_cdef": (C.c_int, [vp, C.c_int, vp, vp]),
    "svt_hip_frame_partition_lr": (C.c_int, [vp, vp, vp]),
    "svt_hip_set_frame_partition": (C.c_int, [vp, C.c_int]),
    "svt_hip_frame_partition_host_calls": (C.c_ulonglong, []),
    "svt_hip_setup_rtcd": (C.c_int, [C.c_uint64]),
    "svt_hip_selftest": (C.c_int, [vp, vp]),
    "svt_hip_rate_probe": (None, [C.c_int, C.c_uint32, C.c_uint32, vp, vp]),
    # SAD family
    "svt_nxm_sad_kernel_hip": (C.c_uint32, [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "svt_aom_sad_16b_kernel_hip": (C.c_uint32, [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "svt_aom_sad_wxh_hip": (C.c_uint32, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]),
    "svt_aom_sad_wxh_x4d_hip": (None, [vp, C.c_int, C.POINTER(vp), C.c_int, vp, C.c_int, C.c_int]),
    "svt_sad_loop_kernel_hip": (None, [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, u64p, i16p, i16p, C.c_uint32,
                                       C.c_uint8, C.c_int16, C.c_int16]),
    "svt_ext_all_sad_calculation_8x8_16x16_hip": (None, [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp, C.c_bool]),
    "svt_ext_eight_sad_calculation_32x32_64x64_hip": (None, [vp, vp, vp, vp, vp, C.c_uint32, vp]),
    "svt_ext_sad_calculation_8x8_16x16_hip": (None, [vp, C.c_uint32, vp, C.c_uint32, vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_bool]),
    "svt_ext_sad_calculation_32x32_64x64_hip": (None, [vp, vp, vp, vp, vp, C.c_uint32, vp]),
    "svt_initialize_buffer_32bits_hip": (None, [vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "svt_pme_sad_loop_kernel_hip": (None, [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, i16p, i16p, C.c_int16, C.c_int16,
                                           C.c_int16, C.c_int16, C.c_int16, C.c_int16, C.c_int16]),
    "svt_aom_downsample_2d_hip": (None, [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_uint32]),
    "svt_hip_downsample_2d_padded": (None, [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "svt_hip_generate_padding": (None, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "svt_hip_cdef_search_one_dual": (None, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "svt_search_one_dual_hip": (C.c_uint64, [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]),
    "svt_hip_lpf_edges_batch": (None, [vp, C.c_uint32, C.c_int, C.c_int, vp, C.c_uint32, vp]),
    "svt_hip_stream_create": (vp, []),
    "svt_hip_stream_destroy": (None, [vp]),
    "svt_hip_stream_synchronize": (None, [vp]),
    "svt_hip_graph_capture_begin": (None, [vp]),
    "svt_hip_graph_capture_end": (vp, [vp]),
    "svt_hip_graph_launch": (None, [vp, vp]),
    "svt_hip_graph_destroy": (None, [vp]),
    "svt_hip_cdef_joint_strength_search": (None, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "svt_hip_cdef_assign_fb_strengths": (None, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "svt_hip_host_alloc": (vp, [C.c_size_t]),
    "svt_hip_host_free": (None, [vp]),
    "svt_hip_me_session_create": (vp, [C.c_uint32] * 11),
    "svt_hip_me_session_create_on": (vp, [C.c_int] + [C.c_uint32] * 11),
    "svt_hip_me_session_destroy": (None, [vp]),
    "svt_hip_me_session_submit": (C.c_int, [vp, C.c_int64, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp]),
    "svt_hip_me_session_wait": (C.c_int, [vp, C.c_int]),
    "svt_hip_me_session_invalidate": (None, [vp, C.c_int64]),
    "svt_hip_me_session_resident": (C.c_int, [vp, C.c_int64]),
    "svt_hip_me_session_enable_stage": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "svt_hip_me_session_submit_stage": (C.c_int, [vp, C.c_int64, vp, vp, C.c_uint32, vp, vp]),
    "svt_hip_me_session_submit_results": (C.c_int, [vp, C.c_int64, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp]),
    "svt_hip_me_results_batch": (None, [vp] * 10),
    "svt_hip_me_integer_search_workspace": (C.c_size_t, [vp]),
    "svt_hip_me_integer_search_batch": (None, [vp] * 14),
    "svt_hip_hme_chain_batch": (None, [vp] * 7),
    "svt_hip_me_zz_sad_batch": (None, [vp] * 5),
    "svt_hip_me_ref_gate_batch": (None, [vp, vp, C.c_uint32, C.c_uint32, C.c_int, vp, vp]),
    "svt_hip_me_ref_safe_limit_batch": (None, [vp, vp, C.c_uint32, vp, vp]),
    "svt_hip_prehme_batch": (None, [vp] * 7),
    "svt_hip_hme_level_workspace": (C.c_size_t, [vp]),
    "svt_hip_hme_level_batch": (None, [vp] * 9),
    "svt_av1_apply_temporal_filter_planewise_medium_hip": (None, [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "svt_av1_apply_temporal_filter_planewise_medium_hbd_hip": (None, [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp] + [C.c_uint32]),
    "svt_av1_apply_zz_based_temporal_filter_planewise_medium_hip": (None, [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "svt_av1_apply_zz_based_temporal_filter_planewise_medium_hbd_hip": (None, [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp] + [C.c_uint32]),
    "svt_estimate_noise_fp16_hip": (C.c_int32, [vp, C.c_uint16, C.c_uint16, C.c_uint16]),
    "svt_estimate_noise_highbd_fp16_hip": (C.c_int32, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "svt_hip_estimate_noise_workspace": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "svt_hip_estimate_noise_batch": (None, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp]),
    "svt_hip_tf_filter_frame": (None, [vp, vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, vp]),
    "svt_hip_tf_subpel_search_batch": (None, [vp, vp, vp, vp, C.c_uint32, vp, vp]),
    "svt_hip_tf_inter_pred_batch": (None, [vp, vp, vp, C.c_uint32, C.c_int, vp]),
    "svt_hip_tf_inter_pred_list": (None, [vp, vp, vp, C.c_uint32, vp, C.c_int, vp]),
    "svt_hip_tf_subpel_search_host": (C.c_int, [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_uint32, vp]),
    "svt_hip_tf_picture_host": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, vp, vp, vp]),
    "svt_hip_tf_picture_workspace": (C.c_size_t, [vp, C.c_uint32]),
    "svt_hip_tf_picture": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, vp]),
    "svt_hip_tf_filter_frame_workspace": (C.c_size_t, [vp, C.c_uint32, C.c_uint32]),
    "svt_hip_tf_filter_frame_chunked": (None, [vp, vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, vp, vp]),
    "svt_hip_lr_filter_frame_host": (C.c_int, [vp]),
    "svt_hip_cdef_apply_host": (C.c_int, [vp]),
    "svt_hip_lpf_plane_host": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp, C.c_uint32, vp, C.c_uint32]),
    "svt_hip_cdef_search_host": (C.c_int, [vp]),
    "svt_hip_lr_search_workspace": (C.c_size_t, [vp]),
    "svt_hip_lr_search_plane": (C.c_int, [vp, vp, vp, vp, vp]),
    "svt_hip_lr_search_plane_host": (C.c_int, [vp, vp, vp]),
    "svt_hip_sad_nxm_batch": (None, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]),
    "svt_hip_sad_loop_batch": (None, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp]),
    "svt_hip_me_fullpel_search_workspace": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "svt_hip_me_fullpel_search_batch": (None, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp, vp]),
}
TX_SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (32, 64), (64, 32), (4, 16),
            (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]  # TxSize order of definitions.h


def allowed_tx_types(tx_size):
    """TxType values legal for a TxSize index: 64-point and 2:1/4:1 shapes touching 64 carry DCT_DCT only, 32-point shapes DCT_DCT and
    IDTX (32x32 also V_DCT / H_DCT), everything else all 16 (is_txfm_allowed, test/TxfmCommon.h:160-209)."""
    w, h = TX_SIZES[tx_size]
    if (w, h) == (32, 32):
        return [0, 9, 10, 11]
    if (w, h) in ((32, 64), (64, 32), (16, 64), (64, 16)):
        return [0]
    if max(w, h) >= 32:
        return [0, 9]
    return list(range(16))


class Buf2D(C.Structure):  # Buf2D, definitions.h:243-249 (passed BY VALUE to hadamard_path)
    _fields_ = [("buf", C.c_void_p), ("buf0", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("stride", C.c_int)]


FwdTxfmDesc = np.dtype([("in_off", "<u8"), ("in_stride", "<u4"), ("tx_type", "u1"), ("pad", "u1", (3,))])
InvTxfmDesc = np.dtype([("coeff_off", "<u8"), ("pred_off", "<u8"), ("recon_off", "<u8"), ("pred_stride", "<u4"), ("recon_stride", "<u4"),
                        ("tx_type", "u1"), ("wht_full", "u1"), ("pad", "u1", (6,))])
TxfmParam = np.dtype([("tx_type", "u1"), ("tx_size", "u1"), ("lossless", "<i4"), ("bd", "<i4"), ("is_hbd", "<i4"), ("tx_set_type", "u1"),
                      ("eob", "<i4")], align=True)  # TxfmParam, definitions.h:1043-1055
assert TxfmParam.itemsize == 24
RoundtripDesc = np.dtype([("in_off", "<u8"), ("pred_off", "<u8"), ("recon_off", "<u8"), ("in_stride", "<u4"), ("pred_stride", "<u4"), ("recon_stride", "<u4"),
                          ("qparam_idx", "<u4"), ("iscan_idx", "<u4"), ("qm_idx", "<u4"), ("tx_type", "u1"), ("pad", "u1", (7,))])
assert FwdTxfmDesc.itemsize == 16 and InvTxfmDesc.itemsize == 40 and RoundtripDesc.itemsize == 56
PROTOTYPES.update({
    "svt_hip_fwd_txfm2d_batch": (None, [vp, vp, C.c_uint32, C.c_int, C.c_int, C.c_int, vp, vp]),
    "svt_hip_inv_txfm2d_add_batch": (None, [vp, vp, vp, vp, C.c_uint32, C.c_int, C.c_int, vp]),
    "svt_hip_inv_txfm2d_add_batch_u8": (None, [vp, vp, vp, vp, C.c_uint32, C.c_int, vp]),
    "svt_hip_inv_txfm2d_add_batch_any_type": (None, [vp, vp, vp, vp, C.c_uint32, C.c_int, C.c_int, vp]),
    "svt_hip_inv_txfm2d_add_batch_any_type_u8": (None, [vp, vp, vp, vp, C.c_uint32, C.c_int, vp]),
    "svt_av1_fwd_txfm2d_hip": (None, [vp, vp, C.c_uint32, C.c_int, C.c_int, C.c_uint8, C.c_int]),
    "svt_av1_inv_txfm2d_add_hip": (None, [vp, vp, C.c_int32, vp, C.c_int32, C.c_int, C.c_int, C.c_int32]),
    "svt_av1_inv_txfm_add_u8_hip": (None, [vp, vp, C.c_int32, vp, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int]),
    "svt_av1_inv_txfm_add_hip": (None, [vp, vp, C.c_int32, vp, C.c_int32, vp]),
    "svt_hip_txfm_quant_roundtrip_batch": (None, [vp, vp, vp, vp, C.c_uint32, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "svt_av1_fwht4x4_hip": (None, [vp, vp, C.c_uint32]),
    "svt_hip_fwht4x4_batch": (None, [vp, vp, C.c_uint32, vp, vp]),
    "svt_hip_iwht4x4_add_batch": (None, [vp, vp, vp, vp, C.c_uint32, C.c_int, vp]),
    "svt_hip_iwht4x4_add_batch_u8": (None, [vp, vp, vp, vp, C.c_uint32, vp]),
    "svt_hip_rtcd_call_counts": (C.c_int, [vp, vp, C.c_int]),
})
for _i, (_w, _h) in enumerate(TX_SIZES):
    for _sfx in ("", "_N2", "_N4"):
        PROTOTYPES["svt_av1_fwd_txfm2d_%dx%d%s_hip" % (_w, _h, _sfx)] = (None, [vp, vp, C.c_uint32, C.c_int, C.c_uint8])
    if _w == _h:
        _a = [vp, vp, C.c_int32, vp, C.c_int32, C.c_int, C.c_int32]
    elif (_w, _h) in ((4, 8), (8, 4), (4, 16), (16, 4)):
        _a = [vp, vp, C.c_int32, vp, C.c_int32, C.c_int, C.c_int, C.c_int32]
    else:
        _a = [vp, vp, C.c_int32, vp, C.c_int32, C.c_int, C.c_int, C.c_int32, C.c_int32]
    PROTOTYPES["svt_av1_inv_txfm2d_add_%dx%d_hip" % (_w, _h)] = (None, _a)
QuantParams = np.dtype([("zbin", "<i2", (2,)), ("round", "<i2", (2,)), ("quant", "<i2", (2,)), ("quant_shift", "<i2", (2,)),
                        ("dequant", "<i2", (2,)), ("log_scale", "<i4")])
QuantDesc = np.dtype([("qparam_idx", "<u4"), ("iscan_idx", "<u4"), ("qm_idx", "<u4"), ("reserved", "<u4")])
assert QuantParams.itemsize == 24 and QuantDesc.itemsize == 16
_Q = [vp, C.c_ssize_t, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
PROTOTYPES.update({
    "svt_hip_quantize_batch": (None, [C.c_int, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "svt_hip_handle_transform_batch": (None, [vp, C.c_uint32, C.c_int, C.c_int, vp, vp]),
    "svt_quantize_hip": (None, [C.c_int] + _Q + [vp, vp, C.c_int]),
    "svt_handle_transform_hip": (C.c_uint64, [vp, C.c_int, C.c_int]),
    "svt_aom_quantize_b_hip": (None, _Q + [vp, vp, C.c_int32]),
    "svt_aom_highbd_quantize_b_hip": (None, _Q + [vp, vp, C.c_int32]),
    "svt_av1_quantize_fp_hip": (None, _Q), "svt_av1_quantize_fp_32x32_hip": (None, _Q), "svt_av1_quantize_fp_64x64_hip": (None, _Q),
    "svt_av1_quantize_fp_qm_hip": (None, _Q + [vp, vp, C.c_int16]),
    "svt_av1_highbd_quantize_fp_hip": (None, _Q + [C.c_int16]),
    "svt_av1_highbd_quantize_fp_qm_hip": (None, _Q + [vp, vp, C.c_int16]),
})
for _n in ("64x64", "32x64", "64x32", "16x64", "64x16"):
    PROTOTYPES["svt_handle_transform%s_hip" % _n] = (C.c_uint64, [vp])
    PROTOTYPES["svt_handle_transform%s_N2_N4_hip" % _n] = (C.c_uint64, [vp])


LpfEdge = np.dtype([("x", "<u4"), ("y", "<u4"), ("vertical", "u1"), ("length", "u1"), ("blimit", "u1"), ("limit", "u1"), ("thresh", "u1"), ("pad", "u1", (3,))])
assert LpfEdge.itemsize == 16
for _len in (4, 6, 8, 14):
    for _d in ("horizontal", "vertical"):
        PROTOTYPES["svt_aom_lpf_%s_%d_hip" % (_d, _len)] = (None, [vp, C.c_int32, vp, vp, vp])
        PROTOTYPES["svt_aom_highbd_lpf_%s_%d_hip" % (_d, _len)] = (None, [vp, C.c_int32, vp, vp, vp, C.c_int32])


class Mv(C.Structure):
    _fields_ = [("row", C.c_int16), ("col", C.c_int16)]


class MvCostParams(C.Structure):
    """MV_COST_PARAMS (mcomp.h:37-48)."""
    _fields_ = [("ref_mv", C.POINTER(Mv)), ("full_ref_mv", Mv), ("mv_cost_type", C.c_uint8), ("mvjcost", C.POINTER(C.c_int)),
                ("mvcost", C.POINTER(C.c_int) * 2), ("error_per_bit", C.c_int), ("early_exit_th", C.c_int), ("sad_per_bit", C.c_int)]


class MeResultsParams(C.Structure):
    """SvtHipMeResultsParams (include/svtav1_hip.h)."""
    _fields_ = [("n_sb", C.c_uint32), ("num_of_list_to_search", C.c_uint8), ("num_of_ref_pic_to_search", C.c_uint8 * 2), ("max_cand", C.c_uint8),
                ("max_refs", C.c_uint8), ("max_l0", C.c_uint8), ("enable_me_16x16", C.c_uint8), ("enable_me_8x8", C.c_uint8), ("only_l_bwd", C.c_uint8),
                ("use_best_unipred_cand_only", C.c_uint8), ("prune_ref", C.c_uint8), ("low_resolution", C.c_uint8), ("gm_enabled", C.c_uint8),
                ("gm_use_distance_based_active_th", C.c_uint8), ("prune_ref_if_me_sad_dev_bigger_than_th", C.c_uint16),
                ("prune_me_candidates_th", C.c_int32), ("picture_number", C.c_uint64), ("ref_picture_number", (C.c_uint64 * 4) * 2)]


assert C.sizeof(MeResultsParams) == 96


class HmeLevelParams(C.Structure):
    """SvtHipHmeLevelParams (include/svtav1_hip.h)."""
    _fields_ = [("level", C.c_uint8), ("sub_sampled", C.c_uint8), ("num_hme_sa_w", C.c_uint8), ("num_hme_sa_h", C.c_uint8), ("sa_width", C.c_int16),
                ("sa_height", C.c_int16), ("sbs_x", C.c_uint32), ("sbs_y", C.c_uint32), ("n_refs", C.c_uint32), ("prev_shift", C.c_uint32), ("aligned_width", C.c_uint32),
                ("aligned_height", C.c_uint32), ("src_off", C.c_uint64), ("src_stride", C.c_uint32), ("ref_stride", C.c_uint32), ("ref_org_x", C.c_uint32),
                ("ref_org_y", C.c_uint32), ("ref_width", C.c_uint32), ("ref_height", C.c_uint32), ("ref_off", C.c_uint64 * 8),
                ("per_ref_area", C.c_uint8), ("pad1", C.c_uint8 * 3), ("sa_width_ref", C.c_int16 * 8), ("sa_height_ref", C.c_int16 * 8),
                ("n_refs_list0", C.c_uint8), ("ref_pic_index", C.c_uint8 * 8), ("prehme_enabled", C.c_uint8), ("pad2", C.c_uint8 * 2),
                ("zz_skip_th", C.c_uint32), ("l0_mv_th_min", C.c_uint16), ("l0_mv_th_max", C.c_uint16), ("sa_width_ref2", C.c_int16 * 8), ("sa_height_ref2", C.c_int16 * 8),
                ("l0_still_rule", C.c_uint8), ("pad3", C.c_uint8), ("sa_width_ref4", C.c_int16 * 8), ("sa_height_ref4", C.c_int16 * 8)]


class HmeChainInputs(C.Structure):
    _fields_ = [("zz_sad", vp), ("do_ref", vp), ("prehme", vp), ("prev_me_stage_based_exit_th", C.c_uint32), ("n_levels", C.c_uint8), ("list1_no_hme", C.c_uint8),
                ("pad", C.c_uint8 * 2)]


class PrehmeParams(C.Structure):
    _fields_ = [("plane", HmeLevelParams), ("sa_min_width", C.c_uint16 * 2), ("sa_min_height", C.c_uint16 * 2), ("sa_max_width", C.c_uint16 * 2),
                ("sa_max_height", C.c_uint16 * 2), ("hme_sr_factor", C.c_uint16 * 8), ("skip_search_line", C.c_uint8), ("l1_early_exit", C.c_uint8),
                ("temporal_layer_gt0", C.c_uint8), ("pad", C.c_uint8), ("me_early_exit_th", C.c_uint32), ("phme_sad_th", C.c_uint32),
                ("phme_sad_pct", C.c_uint16), ("pad2", C.c_uint16)]


PrehmeResult = np.dtype([("sad", "<u8"), ("mv_x", "<i2"), ("mv_y", "<i2"), ("valid", "u1"), ("performed", "u1"), ("pad", "u1", (2,))])
assert PrehmeResult.itemsize == 16


class MeIntegerSearchParams(C.Structure):
    """SvtHipMeIntegerSearchParams (include/svtav1_hip.h)."""
    _fields_ = [("sbs_x", C.c_uint32), ("sbs_y", C.c_uint32), ("n_refs", C.c_uint32), ("regions", C.c_uint32), ("aligned_width", C.c_uint32),
                ("aligned_height", C.c_uint32), ("sa_min_width", C.c_int16), ("sa_min_height", C.c_int16), ("sa_max_width", C.c_int16),
                ("sa_max_height", C.c_int16), ("sub_sad", C.c_uint8), ("mv_adj_enabled", C.c_uint8), ("mv_adj_nearest_ref_only", C.c_uint8),
                ("list1_no_hme", C.c_uint8), ("mv_adj_mv_size_th", C.c_uint16), ("mv_adj_sa_multiplier", C.c_uint16), ("dist", C.c_uint16 * 8),
                ("ref_pic_index", C.c_uint8 * 8), ("src_off", C.c_uint64), ("src_stride", C.c_uint32), ("ref_stride", C.c_uint32),
                ("ref_org_x", C.c_uint32), ("ref_org_y", C.c_uint32), ("ref_off", C.c_uint64 * 8), ("n_refs_list0", C.c_uint8),
                ("hme_prune_enabled", C.c_uint8), ("prune_ref_if_hme_sad_dev_bigger_than_th", C.c_uint16), ("sr_adjustment", C.c_uint8), ("pad2", C.c_uint8),
                ("reduce_me_sr_based_on_mv_length_th", C.c_uint16), ("stationary_hme_sad_abs_th", C.c_uint16), ("stationary_me_sr_divisor", C.c_uint16),
                ("reduce_me_sr_based_on_hme_sad_abs_th", C.c_uint16), ("me_sr_divisor_for_low_hme_sad", C.c_uint16), ("me_early_exit_th", C.c_uint32),
                ("is_ref", C.c_uint8), ("me_8x8_var_enabled", C.c_uint8), ("pad3", C.c_uint8 * 2), ("me_sr_div4_th", C.c_uint32), ("me_sr_div2_th", C.c_uint32),
                ("me_sr_mult2_th", C.c_uint32), ("ref_width", C.c_uint32), ("ref_height", C.c_uint32), ("tf_me_exit_th", C.c_uint32), ("pad4", C.c_uint32)]


class TfParams(C.Structure):
    """SvtHipTfParams: the per-picture MeContext fields read by the temporal filter's pixel kernels."""
    _fields_ = [("tf_decay_factor_fp16", C.c_uint32 * 3), ("tf_mv_dist_th", C.c_uint16), ("tf_chroma", C.c_uint8), ("use_zz_based_filter", C.c_uint8),
                ("encoder_bit_depth", C.c_uint8), ("ss_x", C.c_uint8), ("ss_y", C.c_uint8), ("pad", C.c_uint8)]


assert C.sizeof(TfParams) == 20
TfBlock = np.dtype([("block_error", "<u8", (4,)), ("mv_x", "<i2", (4,)), ("mv_y", "<i2", (4,)), ("split", "u1"), ("pad", "u1", (7,))])
assert TfBlock.itemsize == 56


class TfSubpelParams(C.Structure):
    """SvtHipTfSubpelParams: the picture-level inputs of tf_subpel_search (temporal_filtering.c:1670)."""
    _fields_ = [("half_pel_mode", C.c_uint8), ("quarter_pel_mode", C.c_uint8), ("eight_pel_mode", C.c_uint8), ("subsampling_shift", C.c_uint8),
                ("bit_depth", C.c_uint8), ("pad", C.c_uint8 * 3), ("early_exit_th", C.c_uint32), ("mi_rows", C.c_uint32), ("mi_cols", C.c_uint32),
                ("ref_org_x", C.c_uint32), ("ref_org_y", C.c_uint32), ("ref_stride", C.c_uint32)]


assert C.sizeof(TfSubpelParams) == 32
TfSubpelDesc = np.dtype([("src_off", "<u8"), ("ref_off", "<u8"), ("src_stride", "<u4"), ("pu_x", "<u2"), ("pu_y", "<u2"), ("bsize", "u1"), ("bilinear", "u1"),
                         ("mv_x", "<i2"), ("mv_y", "<i2"), ("pad", "<u2")])
TfMcDesc = np.dtype([("ref_off", "<u8", (3,)), ("pred_off", "<u8", (3,)), ("pu_x", "<u2"), ("pu_y", "<u2"), ("bsize", "u1"), ("pad", "u1"), ("mv_x", "<i2"), ("mv_y", "<i2"),
                     ("pad2", "<u2", (3,))])
assert TfMcDesc.itemsize == 64


class TfMcPlanes(C.Structure):
    """SvtHipTfMcPlanes: reference and prediction plane bases + strides (samples)."""
    _fields_ = [("ref", vp * 3), ("pred", vp * 3), ("ref_stride", C.c_uint32 * 3), ("pred_stride", C.c_uint32 * 3)]


TfSubpelResult = np.dtype([("dist", "<u8"), ("mv_x", "<i2"), ("mv_y", "<i2"), ("pad", "<u4")])
assert TfSubpelDesc.itemsize == 32 and TfSubpelResult.itemsize == 16


class LrSearchParams(C.Structure):
    """SvtHipLrSearchParams: one plane of restoration_seg_search (restoration_pick.c:1448)."""
    _fields_ = [("dgd", C.c_void_p), ("src", C.c_void_p), ("dgd_stride", C.c_uint32), ("src_stride", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("unit_size", C.c_uint32), ("ss_y", C.c_uint8), ("highbd", C.c_uint8), ("bit_depth", C.c_uint8), ("wn_enabled", C.c_uint8), ("wiener_win", C.c_uint8),
                ("wn_use_refinement", C.c_uint8), ("wn_max_one_refinement_step", C.c_uint8), ("sg_enabled", C.c_uint8), ("sg_start_ep", C.c_uint8),
                ("sg_end_ep", C.c_uint8), ("sg_ep_inc", C.c_uint8), ("sg_refine", C.c_uint8), ("pad", C.c_uint8 * 3)]


assert C.sizeof(LrSearchParams) == 56
LrSearchUnit = np.dtype([("sse", "<i8", (3,)), ("vfilter", "<i2", (8,)), ("hfilter", "<i2", (8,)), ("ep", "<i4"), ("xqd", "<i4", (2,)), ("pad", "<i4")])
LrPrevUnit = np.dtype([("use", "<i4"), ("vfilter", "<i2", (8,)), ("hfilter", "<i2", (8,))])
assert LrSearchUnit.itemsize == 72 and LrPrevUnit.itemsize == 36


class TplRef(C.Structure):
    """SvtHipTplRef: one reference picture of the TPL stage."""
    _fields_ = [("plane_off", C.c_uint64), ("picture_number", C.c_uint64), ("stride", C.c_uint32), ("org_x", C.c_uint32), ("org_y", C.c_uint32),
                ("max_width", C.c_uint16), ("max_height", C.c_uint16), ("valid", C.c_uint8), ("pad", C.c_uint8 * 3)]


class TplSrcParams(C.Structure):
    """SvtHipTplSrcParams: the TPL dispenser's source-based half of one picture (svt_hip_tpl_src_stage)."""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("aligned_width", C.c_uint32), ("sbs_x", C.c_uint32), ("n_sb", C.c_uint32), ("src_stride", C.c_uint32),
                ("src_off", C.c_uint64), ("dispenser_search_level", C.c_uint8), ("subsample_tx", C.c_uint8), ("pf_shape", C.c_uint8), ("disable_intra_pred", C.c_uint8),
                ("i_slice", C.c_uint8), ("enable_me_16x16", C.c_uint8), ("enable_me_8x8", C.c_uint8), ("max_cand", C.c_uint8), ("max_refs", C.c_uint8),
                ("max_l0", C.c_uint8), ("intra_mode_end", C.c_uint8), ("search_flags", C.c_uint8), ("quant_fp", C.c_int16 * 2), ("round_fp", C.c_int16 * 2), ("dequant", C.c_int16 * 2),
                ("refs", TplRef * 8)]


assert C.sizeof(TplRef) == 40 and C.sizeof(TplSrcParams) == 376


class TplReconParams(C.Structure):
    """SvtHipTplReconParams: the TPL dispenser's reconstruction half of one picture (svt_hip_tpl_recon_stage)."""
    _fields_ = [("src", TplSrcParams), ("rec_refs", TplRef * 8), ("recon_off", C.c_uint64), ("recon_stride", C.c_uint32), ("is_ref", C.c_uint8), ("pad", C.c_uint8 * 3)]


class TplReconStats(C.Structure):
    _fields_ = [("srcrf_dist", C.c_int64), ("recrf_dist", C.c_int64), ("srcrf_rate", C.c_int64), ("recrf_rate", C.c_int64), ("written", C.c_uint8), ("coded", C.c_uint8),
                ("pad", C.c_uint8 * 2), ("reserved", C.c_uint32)]


assert C.sizeof(TplReconParams) == 376 + 320 + 16 and C.sizeof(TplReconStats) == 40
TplSrcStats = np.dtype([("srcrf_dist", "<i8"), ("srcrf_rate", "<i8"), ("ref_frame_poc", "<u8"), ("mv_row", "<i2"), ("mv_col", "<i2"), ("best_rf_idx", "<i4"),
                        ("best_mode", "u1"), ("best_intra_mode", "u1"), ("written", "u1"), ("pad", "u1", (5,))])
assert TplSrcStats.itemsize == 40


class TfPictureParams(C.Structure):
    """SvtHipTfPictureParams: one central picture of the temporal filter as a device stage (svt_hip_tf_picture_host)."""
    _fields_ = [("sp", TfSubpelParams), ("tf", TfParams), ("pic_w_sb", C.c_uint32), ("pic_h_sb", C.c_uint32), ("uv_stride", C.c_uint32), ("me_exit_th", C.c_uint32),
                ("pred_error_32x32_th", C.c_uint64), ("use_2tap", C.c_uint8), ("enable_8x8_pred", C.c_uint8), ("use_pred_64x64_only_th", C.c_uint8), ("subpel_8bit", C.c_uint8), ("zero_motion", C.c_uint8), ("pad", C.c_uint8 * 3)]


class TfHostPicture(C.Structure):
    """SvtHipTfHostPicture: whole padded host buffers of one picture."""
    _fields_ = [("y", vp), ("u", vp), ("v", vp), ("y_samples", C.c_size_t), ("uv_samples", C.c_size_t), ("y8", vp)]


class TfDevicePictures(C.Structure):
    """SvtHipTfDevicePictures: device-resident pictures of svt_hip_tf_picture."""
    _fields_ = [("central", vp * 3), ("refs", vp * 3), ("ref_pitch", C.c_uint64), ("ref_uv_pitch", C.c_uint64), ("central_y8", vp), ("refs_y8", vp), ("ref_y8_pitch", C.c_uint64)]


class TfMeTables(C.Structure):
    """SvtHipTfMeTables: the ME results of one (central, reference) pair."""
    _fields_ = [("best_sad", vp), ("best_mv", vp), ("hme_sc", vp), ("hme_sad", vp)]


class TfPictureStats(C.Structure):
    _fields_ = [("blocks_64x64", C.c_uint32), ("blocks_32x32", C.c_uint32), ("blocks_16x16", C.c_uint32), ("blocks_8x8", C.c_uint32), ("early_exit_blocks", C.c_uint32),
                ("pad", C.c_uint32 * 3)]


assert C.sizeof(TfPictureParams) == 88 and C.sizeof(TfHostPicture) == 48 and C.sizeof(TfMeTables) == 32 and C.sizeof(TfPictureStats) == 32


class TfPlanes(C.Structure):
    """SvtHipTfPlanes: device planes (strides in samples)."""
    _fields_ = [("y", vp), ("u", vp), ("v", vp), ("y_stride", C.c_uint32), ("uv_stride", C.c_uint32)]


class MeStageParams(C.Structure):
    """SvtHipMeStageParams (include/svtav1_hip.h)."""
    _fields_ = [("num_hme_sa_w", C.c_uint8), ("num_hme_sa_h", C.c_uint8), ("hme_sub_sampled", C.c_uint8), ("me_sub_sad", C.c_uint8),
                ("hme_sa_width", C.c_int16 * 3), ("hme_sa_height", C.c_int16 * 3), ("me_sa_min_width", C.c_int16), ("me_sa_min_height", C.c_int16),
                ("me_sa_max_width", C.c_int16), ("me_sa_max_height", C.c_int16), ("mv_adj_enabled", C.c_uint8), ("mv_adj_nearest_ref_only", C.c_uint8),
                ("mv_adj_mv_size_th", C.c_uint16), ("mv_adj_sa_multiplier", C.c_uint16), ("dist", C.c_uint16 * 8), ("ref_pic_index", C.c_uint8 * 8),
                ("hme_l0_per_ref", C.c_uint8), ("hme_prune_enabled", C.c_uint8), ("sr_adjustment", C.c_uint8), ("me_type_mctf", C.c_uint8),
                ("hme_l0_sa_width_ref", C.c_int16 * 8), ("hme_l0_sa_height_ref", C.c_int16 * 8), ("prune_ref_if_hme_sad_dev_bigger_than_th", C.c_uint16),
                ("reduce_me_sr_based_on_mv_length_th", C.c_uint16), ("stationary_hme_sad_abs_th", C.c_uint16), ("stationary_me_sr_divisor", C.c_uint16),
                ("reduce_me_sr_based_on_hme_sad_abs_th", C.c_uint16), ("me_sr_divisor_for_low_hme_sad", C.c_uint16), ("me_early_exit_th", C.c_uint32),
                ("is_ref", C.c_uint8), ("me_8x8_var_enabled", C.c_uint8), ("hme_levels", C.c_uint8), ("pad1", C.c_uint8), ("me_sr_div4_th", C.c_uint32), ("me_sr_div2_th", C.c_uint32),
                ("me_sr_mult2_th", C.c_uint32), ("temporal_layer_gt0", C.c_uint8), ("prehme_enabled", C.c_uint8), ("prehme_skip_search_line", C.c_uint8),
                ("prehme_l1_early_exit", C.c_uint8), ("prehme_sa_min_width", C.c_uint16 * 2), ("prehme_sa_min_height", C.c_uint16 * 2),
                ("prehme_sa_max_width", C.c_uint16 * 2), ("prehme_sa_max_height", C.c_uint16 * 2), ("zz_sad_th", C.c_uint32), ("phme_sad_th", C.c_uint32),
                ("zz_sad_pct", C.c_uint16), ("phme_sad_pct", C.c_uint16), ("prev_me_stage_based_exit_th", C.c_uint32), ("me_safe_limit_zz_th", C.c_uint32), ("tf_me_exit_th", C.c_uint32), ("results", MeResultsParams),
                ("reduce_hme_l0_sr_th_min", C.c_uint16), ("reduce_hme_l0_sr_th_max", C.c_uint16), ("hme_l0_sa_width_ref2", C.c_int16 * 8), ("hme_l0_sa_height_ref2", C.c_int16 * 8),
                ("hme_l0_sa_width_ref4", C.c_int16 * 8), ("hme_l0_sa_height_ref4", C.c_int16 * 8)]


class MeResultsHost(C.Structure):
    """SvtHipMeResultsHost: host destinations of svt_hip_me_session_submit_results."""
    _fields_ = [("do_ref", vp), ("total_me_candidate_index", vp), ("me_mv_array", vp), ("me_candidate_array", vp), ("sb_stats", vp), ("best_sad", vp),
                ("best_mv", vp), ("hme_sc", vp), ("hme_sad", vp)]


MeSbStats = np.dtype([("me_64x64_distortion", "<u4"), ("me_32x32_distortion", "<u4"), ("me_16x16_distortion", "<u4"), ("me_8x8_distortion", "<u4"),
                      ("me_8x8_cost_variance", "<u4"), ("rc_me_distortion", "<u4"), ("stationary_block_present_sb", "u1"), ("rc_me_allow_gm", "u1"),
                      ("pad", "u1", (2,))])
assert MeSbStats.itemsize == 28


def scaled_picture_distance(d):
    """svt_aom_get_scaled_picture_distance (motion_estimation.c:1239-1243)."""
    return d * 5 // 8 + (1 if d % 8 else 0)


def m8_me_settings(qp=35, temporal_layer=1, resolution_1080p_or_above=True):
    """What svt_aom_sig_deriv_me (enc_mode_config.c:681-815) derives for preset 8 (ENC_M8), random access, non-screen content, >= 5 hierarchical levels
    -- restated as a dict so that bench.py and the tests use ONE table; tests/test_hme.py::test_m8_me_settings_vs_reference pins every entry against the
    reference function itself (oracle/ref_wrap/ref_static_me.c: ref_sig_deriv_me).  Areas are modulated by the sequence QP (:203-209, :326-333)."""
    clip = lambda lo, hi, v: max(lo, min(hi, v))  # noqa: E731
    qw_hme = clip(500, 1000, 3 * (8 * qp - 125))            # q_mult = 3 (:166-201)
    qw_me = clip(500, 1000, (7 * (31 * qp - 700)) >> 3)     # q_mult = 7 (:306-321)
    me = (16, 6, 16, 9) if resolution_1080p_or_above else (16, 16, 32, 16)
    base = temporal_layer == 0
    return dict(
        num_hme_sa=(2, 2), hme_levels=2,                    # enable_hme_level2_flag = 0 above M6 (enc_mode_config.c:1636-1640)
        hme_l0=(max(8, 16 * qw_hme // 1000), max(8, 16 * qw_hme // 1000), max(96, 192 * qw_hme // 1000), max(96, 192 * qw_hme // 1000)),
        hme_l1=(8, 3), hme_l2=(8, 3),
        me=(max(8, me[0] * qw_me // 1000), max(3, me[1] * qw_me // 1000), max(8, me[2] * qw_me // 1000), max(3, me[3] * qw_me // 1000)),
        sub_sampled=1,                                       # hme_search_method = me_search_method = SUB_SAD_SEARCH (:698-699)
        prehme=dict(skip=1, l1=1, sa=((8, 100, 8, 350), (32, 7, 128, 7))),  # level 4 (:731-733, :594-603)
        hme_prune=80 if base else 5, me_prune=0xffff if base else 60,        # ref-prune level 1 (base) / 6 (:762-763)
        zz=(0, 0) if base else (20 * 64 * 64, 5), phme=(0, 0) if base else (10 * 64 * 64, 5),
        sr=dict(level=1, mv_length_th=4, stationary_th=12000 // 4, stationary_div=8, low_sad_th=12000 // 4, low_sad_div=8, distance_based=1),  # level 3, / 4: no level 2 (:479-486, :510-515)
        mv_adj=0, var=(80000, 150000, 0xffffffff),           # me_8x8_var level 2 (:533-538)
        prune_me_candidates_th=65, me_early_exit_th=64 * 64 * 8, prev_me_stage_based_exit_th=0)


def fill_m8_stage_params(S, dists, ref_pic_index, qp=35, temporal_layer=1, is_ref=1):
    """SvtHipMeStageParams (search part) for preset 8 from m8_me_settings; dists = raw picture distances per reference slot (list 0 first)."""
    m = m8_me_settings(qp, temporal_layer)
    nw, nh = m["num_hme_sa"]
    S.num_hme_sa_w, S.num_hme_sa_h, S.hme_sub_sampled, S.me_sub_sad, S.hme_levels = nw, nh, m["sub_sampled"], m["sub_sampled"], m["hme_levels"]
    S.hme_l0_per_ref = 1
    for r, (d, ri) in enumerate(zip(dists, ref_pic_index)):
        f = scaled_picture_distance(d)
        S.dist[r], S.ref_pic_index[r] = f, ri
        b = [v // (1 + ri) for v in m["hme_l0"]] if m["sr"]["distance_based"] else m["hme_l0"]  # get_hme_l0_search_area (:1806-1866), non-RTC form
        S.hme_l0_sa_width_ref[r] = min((((b[0] // nw) * f) + 15) & ~15, ((b[2] // nw) + 15) & ~15)
        S.hme_l0_sa_height_ref[r] = min((b[1] // nh) * f, b[3] // nh)
    S.hme_sa_width[1], S.hme_sa_height[1] = m["hme_l1"]
    S.hme_sa_width[2], S.hme_sa_height[2] = m["hme_l2"]
    S.me_sa_min_width, S.me_sa_min_height, S.me_sa_max_width, S.me_sa_max_height = m["me"]
    S.me_early_exit_th, S.is_ref, S.temporal_layer_gt0 = m["me_early_exit_th"], is_ref, int(temporal_layer > 0)
    S.me_8x8_var_enabled, (S.me_sr_div4_th, S.me_sr_div2_th, S.me_sr_mult2_th) = 1, m["var"]
    S.hme_prune_enabled, S.prune_ref_if_hme_sad_dev_bigger_than_th = 1, m["hme_prune"]
    sr = m["sr"]
    (S.sr_adjustment, S.reduce_me_sr_based_on_mv_length_th, S.stationary_hme_sad_abs_th, S.stationary_me_sr_divisor, S.reduce_me_sr_based_on_hme_sad_abs_th,
     S.me_sr_divisor_for_low_hme_sad) = sr["level"], sr["mv_length_th"], sr["stationary_th"], sr["stationary_div"], sr["low_sad_th"], sr["low_sad_div"]
    S.zz_sad_th, S.zz_sad_pct = m["zz"]
    S.phme_sad_th, S.phme_sad_pct = m["phme"]
    ph = m["prehme"]
    S.prehme_enabled, S.prehme_skip_search_line, S.prehme_l1_early_exit = 1, ph["skip"], ph["l1"]
    for k, v in enumerate(ph["sa"]):
        S.prehme_sa_min_width[k], S.prehme_sa_min_height[k], S.prehme_sa_max_width[k], S.prehme_sa_max_height[k] = v
    S.results.prune_ref = int(m["me_prune"] != 0xffff)
    S.results.prune_ref_if_me_sad_dev_bigger_than_th = m["me_prune"]
    S.results.prune_me_candidates_th = m["prune_me_candidates_th"]
    return m


def me_max_allocated_refs(l0, l1):
    """svt_aom_get_max_allocated_me_refs (pcs.c:91-96) -> (max_refs, max_cand)."""
    return l0 + l1, l0 + l1 + l0 * l1 + (l0 - 1) + (1 if l1 == 3 else 0)


class CdefParams(C.Structure):
    _fields_ = [("recon", vp), ("source", vp), ("out", vp), ("recon_stride", C.c_uint32), ("source_stride", C.c_uint32), ("out_stride", C.c_uint32),
                ("width", C.c_uint32), ("height", C.c_uint32), ("xdec", C.c_uint8), ("ydec", C.c_uint8), ("pli", C.c_uint8), ("is_16bit", C.c_uint8),
                ("coeff_shift", C.c_uint8), ("pri_damping", C.c_uint8), ("sec_damping", C.c_uint8), ("subsampling", C.c_uint8), ("ncand", C.c_uint32),
                ("skip", vp), ("pri", vp), ("sec", vp), ("dir", vp), ("var", vp), ("mse", vp)]


PROTOTYPES.update({
    "svt_hip_cdef_frame": (None, [C.c_int, C.POINTER(CdefParams), vp]),
    "svt_hip_cdef_frame_rows": (None, [C.c_int, C.POINTER(CdefParams), C.c_int, C.c_int, vp]),
    "svt_aom_cdef_find_dir_hip": (C.c_uint8, [vp, C.c_int32, vp, C.c_int32]),
    "svt_aom_cdef_find_dir_dual_hip": (None, [vp, vp, C.c_int, vp, vp, C.c_int32, vp, vp]),
    "svt_cdef_filter_block_hip": (None, [vp, vp, C.c_int32, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint8]),
    "svt_compute_cdef_dist_16bit_hip": (C.c_uint64, [vp, C.c_int32, vp, vp, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_uint8]),
    "svt_compute_cdef_dist_8bit_hip": (C.c_uint64, [vp, C.c_int32, vp, vp, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_uint8]),
    "svt_aom_copy_rect8_8bit_to_16bit_hip": (None, [vp, C.c_int32, vp, C.c_int32, C.c_int32, C.c_int32]),
})
class CdefApplyHost(C.Structure):
    """SvtHipCdefApplyHost: svt_av1_cdef_frame from host planes (in place)."""
    _fields_ = [("plane", vp * 3), ("stride", C.c_uint32 * 3), ("width", C.c_uint32), ("height", C.c_uint32), ("num_planes", C.c_uint8), ("is_16bit", C.c_uint8),
                ("coeff_shift", C.c_uint8), ("damping", C.c_uint8), ("skip", vp), ("pri_y", vp), ("sec_y", vp), ("pri_uv", vp), ("sec_uv", vp)]


class CdefSearchHost(C.Structure):
    """SvtHipCdefSearchHost: cdef_seg_search for all filter blocks of a picture from host planes."""
    _fields_ = [("recon", vp * 3), ("source", vp * 3), ("recon_stride", C.c_uint32 * 3), ("source_stride", C.c_uint32 * 3), ("width", C.c_uint32),
                ("height", C.c_uint32), ("is_16bit", C.c_uint8), ("coeff_shift", C.c_uint8), ("damping", C.c_uint8), ("subsampling", C.c_uint8 * 2),
                ("pad", C.c_uint8 * 3), ("skip", vp), ("ncand_y", C.c_uint32), ("ncand_uv", C.c_uint32), ("pri_y", vp), ("sec_y", vp), ("pri_uv", vp),
                ("sec_uv", vp), ("mse_y", vp), ("mse_u", vp), ("mse_v", vp), ("dir", vp), ("var", vp)]


LrUnit = np.dtype([("rtype", "<i4"), ("vfilter", "<i2", (8,)), ("hfilter", "<i2", (8,)), ("ep", "<i4"), ("xqd", "<i4", (2,))])
assert LrUnit.itemsize == 48


class LrParams(C.Structure):
    _fields_ = [("data", vp), ("boundary_above", vp), ("boundary_below", vp), ("dst", vp), ("stride", C.c_uint32), ("boundary_stride", C.c_uint32),
                ("dst_stride", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("unit_size", C.c_uint32), ("ss_x", C.c_uint8),
                ("ss_y", C.c_uint8), ("highbd", C.c_uint8), ("bit_depth", C.c_uint8), ("units", vp)]


PROTOTYPES.update({
    "svt_hip_lr_filter_frame": (None, [C.POINTER(LrParams), vp]),
    "svt_hip_lr_filter_frame_stripes": (None, [C.POINTER(LrParams), C.c_int, C.c_int, vp]),
    "svt_av1_wiener_convolve_add_src_hip": (None, [vp, C.c_ssize_t, vp, C.c_ssize_t, vp, vp, C.c_int32, C.c_int32, vp]),
    "svt_av1_highbd_wiener_convolve_add_src_hip": (None, [vp, C.c_ssize_t, vp, C.c_ssize_t, vp, vp, C.c_int32, C.c_int32, vp, C.c_int32]),
    "svt_av1_selfguided_restoration_hip": (None, [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "svt_apply_selfguided_restoration_hip": (None, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_int32, vp, C.c_int32, C.c_int32]),
})
SatdDesc = np.dtype([("in_off", "<u8"), ("pred_off", "<u8"), ("in_stride", "<u4"), ("pred_stride", "<u4")])
Rect = np.dtype([("h_start", "<i4"), ("h_end", "<i4"), ("v_start", "<i4"), ("v_end", "<i4")])
_PE = [vp, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, vp, C.c_int32, vp, C.c_int32, vp, vp]
PROTOTYPES.update({
    "svt_hip_hadamard_satd_batch": (None, [vp, vp, vp, C.c_uint32, C.c_int, vp, vp, vp]),
    "svt_hip_lr_compute_stats_batch": (None, [vp, vp, vp, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "svt_hip_lr_compute_stats_batch_samples": (None, [vp, vp, vp, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "svt_aom_satd_hip": (C.c_int, [vp, C.c_int]),
    "svt_aom_hadamard_nxn_hip": (None, [vp, C.c_ssize_t, vp, C.c_int]),
    "svt_hadamard_path_hip": (C.c_uint32, [vp, C.c_uint32, vp, C.c_uint32, C.c_int]),
    "hadamard_path_hip": (C.c_uint32, [Buf2D, Buf2D, Buf2D, Buf2D, C.c_uint8]),
    "svt_residual_kernel8bit_hip": (None, [vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "svt_residual_kernel16bit_hip": (None, [vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "svt_av1_compute_stats_hip": (None, [C.c_int32, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp]),
    "svt_av1_compute_stats_highbd_hip": (None, [C.c_int32, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_int]),
    "svt_av1_lowbd_pixel_proj_error_hip": (C.c_int64, _PE),
    "svt_av1_highbd_pixel_proj_error_hip": (C.c_int64, _PE),
    "svt_get_proj_subspace_hip": (None, [vp, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, C.c_int32, vp, C.c_int32, vp, C.c_int32, vp, vp]),
})
for _n in (4, 8, 16, 32):
    PROTOTYPES["svt_aom_hadamard_%dx%d_hip" % (_n, _n)] = (None, [vp, C.c_ssize_t, vp])
for _m, _n in [(128, 128), (128, 64), (64, 128), (64, 64), (64, 32), (32, 64), (32, 32), (32, 16), (16, 32), (16, 16), (16, 8),
               (8, 16), (8, 8), (8, 4), (4, 8), (4, 4), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]:
    PROTOTYPES["svt_aom_sad%dx%d_hip" % (_m, _n)] = (C.c_uint32, [vp, C.c_int, vp, C.c_int])
    PROTOTYPES["svt_aom_sad%dx%dx4d_hip" % (_m, _n)] = (None, [vp, C.c_int, C.POINTER(vp), C.c_int, vp])


def bind(lib):
    """Declare argtypes/restype on a loaded library (the product .so, or the test-only emulated build)."""
    missing = []
    for name, (res, args) in PROTOTYPES.items():
        try:
            f = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        f.restype = res
        f.argtypes = args
    if missing:
        raise RuntimeError("libsvtav1_hip: missing symbols: " + ", ".join(missing))
    return lib


_lib = None


def load(init_device=None):
    """Load libsvtav1_hip.so (built in-tree by ``__graft_entry__.build()``).  Fails loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        _lib = bind(C.CDLL(LIB_PATH))
    if init_device is not None:
        if _lib.svt_hip_init(int(init_device)) != 0:
            raise RuntimeError("svt_hip_init(%d) failed: no usable HIP device; libsvtav1_hip has no CPU path" % init_device)
    return _lib


def me_descs_for_frame(width, height, stride, org_x, org_y, area_w, area_h, plane_bytes, n_refs=1, src_plane=0, ref_plane0=0,
                       sb=64):
    """Descriptor table for config 2 of BASELINE.json: every 64x64 SB of one padded luma plane against ``n_refs``
    reference planes, fixed search area ``area_w x area_h`` centred on the co-located block, exactly the operands
    open_loop_me_fullpel_search_sblock (motion_estimation.c:781) receives for a (0,0) search centre."""
    sbs_x, sbs_y = (width + sb - 1) // sb, (height + sb - 1) // sb
    d = np.zeros(sbs_x * sbs_y * n_refs, dtype=MeSearchDesc)
    xo, yo = -(area_w >> 1), -(area_h >> 1)
    i = 0
    for r in range(n_refs):
        for sy in range(sbs_y):
            for sx in range(sbs_x):
                px, py = org_x + sx * sb, org_y + sy * sb
                d[i] = (src_plane * plane_bytes + py * stride + px,
                        (ref_plane0 + r) * plane_bytes + (py + yo) * stride + (px + xo), stride, stride, xo, yo, area_w, area_h)
                i += 1
    return d


def shard_range(n_units, rank, world):
    """Contiguous share of `n_units` independent units (frames / SB rows / filter blocks) for `rank` of `world`:
    the path partitions without any exchange, so sharding is just a range split."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
