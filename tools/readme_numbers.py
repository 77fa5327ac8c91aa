#!/usr/bin/env python3
"""The "Numbers of the round's final regression" block of README.md from a bench_detail.json (and the previous regression's, for the fps spread):
    python tools/readme_numbers.py profiles/r06_final_bench_detail.json profiles/r06_reg3_bench_detail.json"""
import json
import sys

d, d2 = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
rf, cb, e, e4, K, e2 = d['roofline'], d['cpu_baseline'], d['encoder_fps_1080p_preset8'], d['encoder_fps_4k10_preset8'], d['kernels'], d2['encoder_fps_1080p_preset8']


def us(n):
    r = K[n].get('roofline') or {}
    return r.get('kernel_us') or K[n]['ms'] * 1e3


print('''* headline (configs[1], batched ME full-pel search 16x9, 1080p 8-bit, 65 280 SB-refs per launch): **%.0f M(SB x position)/s**, `me_fullpel_wave_kernel` %.1f us per launch (rocprofv3 average of
  the same grid: see `r06_final_kernel_stats.txt`), VALU %.2f of the packed-SAD issue roof measured in the same run, %.0f GB/s algorithmic = %.3f of HBM peak (`binds: valu`); the
  reference's AVX2 kernels on the box's 16 cores: %.0f M/s (`kind: reference`) -> %.0fx.
* SAD path at DRAM footprint: `sad64x64_pairs` %.0f us, %.2f of 8 TB/s; preset-8 ME areas 8x4 / 8x3 with plane sets larger than the Infinity Cache: %.2f / %.2f of HBM peak.
* transforms / quantiser: `fwd_txfm2d_32x32` %.0f us (VALU %.2f), `inv_txfm2d_add_32x32` %.0f us (VALU %.2f), `quantize_b_32x32` %.0f us (HBM %.2f).
* config 4 (4K 10-bit): `cdef_search` 64 strengths %.0f us per luma plane / %.0f us per 4:2:0 picture (VALU %.2f), `cdef_apply` %.0f / %.0f us, whole CDEF stage (search -> pick -> apply, 3 planes)
  %.2f ms; LR apply %.0f-%.0f us per plane; **`lr_compute_stats` (MFMA) %.0f us, %.2f of the dense int8 peak as issued** (205 us, 0.20 at the start of the round); **LR search %.2f ms (full) / %.2f ms (fast)**
  (3.16 / 0.95 at the start of the round; the two legs alone on one box: 2.71-2.80 / 0.69-0.70).
* callers: `hme_3level_1080p_4refs` %.0f us (moved / algorithmic %.1f), `tf_picture_stage` resident %.0f us, `tpl_src_stage` %.1f us, **`tpl_recon_stage` %.0f us, `tpl_l1_recon` %.0f us** (340 / 483 before the
  write-through hand-off).
* encoder, 1080p preset 8, 60 frames, medians of five pairs, every bitstream identical: AVX2 host %.1f -> **%.1f fps** with every stage on the MI355X (+%.0f %%; the previous regression's box: %.1f -> %.1f, +%.0f %%),
  AVX-512 host %.1f -> %.1f (previous box %.1f -> %.1f); 300 frames %.1f -> %.1f; configs[4] (4K 10-bit preset 8, 30 frames, identity on the first attempt): %.2f -> %.2f fps (bound by the host's
  mode decision).''' % (
    d['value'], rf['kernel_us'], rf['valu_frac'], rf['achieved'], rf['frac'], cb['value'], cb['gpu_over_cpu'],
    us('sad64x64_pairs'), K['sad64x64_pairs']['roofline']['frac'], K['me_search_8x4_preset8_area_dram']['roofline']['frac'], K['me_search_8x3_preset8_area_dram']['roofline']['frac'],
    us('fwd_txfm2d_32x32'), K['fwd_txfm2d_32x32']['roofline']['valu_frac'], us('inv_txfm2d_add_32x32'), K['inv_txfm2d_add_32x32']['roofline']['valu_frac'], us('quantize_b_32x32'),
    K['quantize_b_32x32']['roofline']['frac'],
    us('cdef_search_4k10_64strengths'), us('cdef_search_4k10_420'), K['cdef_search_4k10_64strengths']['roofline']['valu_frac'], us('cdef_apply_4k10'), us('cdef_apply_4k10_420'),
    K['cdef_stage_4k10_420']['ms'],
    us('lr_wiener_4k10'), us('lr_sgrproj_4k10'), us('lr_compute_stats_4k10_win7'), K['lr_compute_stats_4k10_win7']['roofline']['frac'], K['lr_search_4k10_full']['ms'], K['lr_search_4k10_fast']['ms'],
    us('hme_3level_1080p_4refs'), K['hme_3level_1080p_4refs']['roofline']['moved_over_algorithmic'], us('tf_picture_stage_1080p8_4refs_resident'), us('tpl_src_stage_1080p8'),
    us('tpl_recon_stage_1080p8'), us('tpl_l1_recon_1080p8'),
    e['fps_avx2_intrinsics'], e['fps_avx2_host_with_stage_seams'], 100 * (e['fps_avx2_host_with_stage_seams'] / e['fps_avx2_intrinsics'] - 1),
    e2['fps_avx2_intrinsics'], e2['fps_avx2_host_with_stage_seams'], 100 * (e2['fps_avx2_host_with_stage_seams'] / e2['fps_avx2_intrinsics'] - 1),
    e['fps_avx512_intrinsics'], e['fps_avx512_host_with_stage_seams'], e2['fps_avx512_intrinsics'], e2['fps_avx512_host_with_stage_seams'],
    e['steady_state_300_frames']['fps_avx2_intrinsics'], e['steady_state_300_frames']['fps_avx2_host_with_stage_seams'], e4['fps_host_alone'], e4['fps_host_with_stage_seams']))
