#!/usr/bin/env python3
"""Bitstream-identity check of the HIP variant inside the REAL reference encoder (SURVEY 8c "integration-level parity").

The reference's own CI invariant is that every ISA level produces the same bitstream
(/root/reference/.gitlab/workflows/linux/.gitlab-ci.yml:351-367).  oracle/_ref/enc/SvtAv1EncApp is the reference encoder built
C-only by oracle/Makefile with the binding of INTEGRATION.md §1 (integration/enc_handle_binding.c): with SVT_HIP unset it is
the `--asm c` encoder; with SVT_HIP=<device> svt_hip_setup_rtcd() overwrites the dispatch pointers right after
enc_handle.c:1444-1445.  This script encodes a synthetic clip both ways and compares the .ivf byte by byte;
SVT_HIP_COUNT gives the number of calls that went through every installed pointer.

    python tools/enc_identity.py --lib svt-av1-psy_amd/libsvtav1_hip.so --case all --out gpurun_out/identity
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENC = os.path.join(ROOT, "oracle", "_ref", "enc", "SvtAv1EncApp")
# the same reference encoder with its x86 intrinsic kernels compiled in (oracle/Makefile `enc_avx2`: ASM_SSE2 .. ASM_AVX2/*.c, NASM kernels at their C versions):
# the CPU baseline a user of the reference actually runs, and -- with the seams on -- the host half of a deployment.  Its bitstream equals the C-only encoder's.
ENC_AVX2 = os.path.join(ROOT, "oracle", "_ref", "enc_avx2", "SvtAv1EncApp")
# ... and with EN_AVX512_SUPPORT=1 + ASM_AVX512/*.c (oracle/Makefile `enc_avx512`): what the reference selects on an AVX-512 host
ENC_AVX512 = os.path.join(ROOT, "oracle", "_ref", "enc_avx512", "SvtAv1EncApp")
HOST_ENC = {"c": None, "avx2": ENC_AVX2, "avx512": ENC_AVX512}
LAST_CPU_S = [0.0]  # user + system CPU seconds of the most recent encode() child (RUSAGE_CHILDREN delta)

# name: (width, height, frames, bit depth, extra encoder arguments)
CASES = {
    "p8_8bit_lp1": (256, 144, 8, 8, ["--preset", "8", "--lp", "1"]),
    "p8_10bit_lp1": (256, 144, 8, 10, ["--preset", "8", "--lp", "1"]),
    "p4_8bit_lp1": (256, 144, 6, 8, ["--preset", "4", "--lp", "1"]),
    "p4_10bit_lp4": (256, 144, 6, 10, ["--preset", "4", "--lp", "4"]),
    "p8_8bit_lp4": (256, 144, 8, 8, ["--preset", "8", "--lp", "4"]),
    "p8_8bit_lossless": (128, 128, 4, 8, ["--preset", "8", "--lp", "1", "--lossless", "1", "--tune", "1"]),
    "p8_10bit_lossless": (128, 128, 4, 10, ["--preset", "8", "--lp", "1", "--lossless", "1", "--tune", "1"]),
    "p6_8bit_qm_lp2": (256, 144, 6, 8, ["--preset", "6", "--lp", "2", "--enable-qm", "1", "--qm-min", "2", "--qm-max", "10"]),
    # the open-loop ME stage as ONE device call per picture (integration/me_process_seam.c): SVT_HIP_ME_SEAM=1, parameters from the reference's own
    # svt_aom_sig_deriv_me for every picture; "+hook" = the per-call RTCD variants are installed as well
    "seam_p8_8bit": (448, 264, 10, 8, ["--preset", "8", "--lp", "1", "+seam"]),
    # (10-bit preset 8 with --lp >= 2 is not used: the C-only reference encoder itself produces a different bitstream from run to run there -- 5 different
    # .ivf files in 6 runs at 448x264 --, so there is nothing to be identical to; run_case() detects that situation for any case, see "reference_deterministic")
    "seam_p8_10bit": (448, 264, 10, 10, ["--preset", "8", "--lp", "1", "+seam"]),
    "seam_p8_8bit_lp4": (448, 264, 10, 8, ["--preset", "8", "--lp", "4", "+seam"]),
    "seam_p4_8bit_lp2": (256, 144, 8, 8, ["--preset", "4", "--lp", "2", "+seam"]),
    "seam_p6_8bit_hook": (256, 144, 8, 8, ["--preset", "6", "--lp", "1", "+seam", "+hook"]),
    "seam_p10_8bit": (448, 264, 10, 8, ["--preset", "10", "--lp", "1", "+seam"]),
    "seam_p2_8bit": (256, 144, 6, 8, ["--preset", "2", "--lp", "1", "+seam"]),
    # the per-unit half of the loop-restoration search as one device stage per plane (integration/rest_process_seam.c): SVT_HIP_LR_SEAM=1
    "lrseam_p4_8bit": (256, 144, 6, 8, ["--preset", "4", "--lp", "1", "+lrseam"]),
    "lrseam_p2_10bit": (256, 144, 5, 10, ["--preset", "2", "--lp", "1", "+lrseam"]),
    "lrseam_p6_8bit_lp4": (448, 264, 8, 8, ["--preset", "6", "--lp", "4", "+lrseam"]),
    "lrseam_me_p5_8bit_lp2": (448, 264, 8, 8, ["--preset", "5", "--lp", "2", "+lrseam", "+seam"]),  # both seams at once
    "lrseam_p4_8bit_crf55": (448, 264, 6, 8, ["--preset", "4", "--lp", "1", "--crf", "55", "+lrseam"]),  # coarse quantisation: restoration wins more often
    "lrseam_p3_10bit_crf50": (256, 144, 5, 10, ["--preset", "3", "--lp", "1", "--crf", "50", "+lrseam"]),
    # CDEF applied to the whole picture by one device call (integration/cdef_process_seam.c): SVT_HIP_CDEF_SEAM=1
    "cdefseam_p8_8bit": (448, 264, 10, 8, ["--preset", "8", "--lp", "1", "+cdefseam"]),
    "cdefseam_p4_10bit": (256, 144, 6, 10, ["--preset", "4", "--lp", "1", "+cdefseam"]),
    "cdefseam_p6_8bit_lp4": (448, 264, 8, 8, ["--preset", "6", "--lp", "4", "--crf", "45", "+cdefseam"]),
    "allseams_p5_8bit_lp2": (448, 264, 8, 8, ["--preset", "5", "--lp", "2", "+seam", "+tfseam", "+tfsubpel", "+lrseam", "+cdefseam", "+dlfseam"]),
    "allseams_1080p_p6": (1920, 1080, 6, 8, ["--preset", "6", "+seam", "+tfseam", "+tfsubpel", "+lrseam", "+cdefseam", "+dlfseam"]),
    # the deblocking filter of a picture as one device call per plane, segments recorded from the reference's own driver (integration/dlf_process_seam.c)
    "dlfseam_p5_8bit": (448, 264, 8, 8, ["--preset", "5", "--lp", "1", "+dlfseam"]),
    "dlfseam_p2_10bit": (256, 144, 5, 10, ["--preset", "2", "--lp", "1", "+dlfseam"]),
    "dlfseam_p6_8bit_lp4": (448, 264, 8, 8, ["--preset", "6", "--lp", "4", "+dlfseam"]),
    # presets >= 7 deblock SB by SB inside the coding loop (coding_loop.c:2278): recorded per SB, filtered per picture on the device
    "dlfseam_sb_p8_8bit": (448, 264, 12, 8, ["--preset", "8", "--lp", "1", "+dlfseam"]),
    "dlfseam_sb_p8_8bit_lp4": (448, 264, 12, 8, ["--preset", "8", "--lp", "4", "+dlfseam"]),
    "dlfseam_sb_p10_10bit": (256, 144, 10, 10, ["--preset", "10", "--lp", "1", "+dlfseam"]),
    "dlfseam_sb_1080p_p8": (1920, 1080, 10, 8, ["--preset", "8", "+dlfseam"]),
    "everyseam_p4_8bit_lp2": (448, 264, 8, 8, ["--preset", "4", "--lp", "2", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+dlfseam", "+cdefseam", "+lrseam", "+tplseam"]),
    "everyseam_4k10_p8_lp1": (3840, 2160, 6, 10, ["--preset", "8", "--lp", "1", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),  # config-5 format, single-threaded (reproducible)
    # the temporal filter's ME (ME_MCTF form of the stage, one call per (central picture, reference picture) pair): SVT_HIP_TF_ME_SEAM=1 on top of the ME seam
    "tfseam_p8_8bit": (448, 264, 10, 8, ["--preset", "8", "--lp", "1", "+seam", "+tfseam"]),
    "tfseam_p4_10bit": (256, 144, 8, 10, ["--preset", "4", "--lp", "1", "+seam", "+tfseam"]),
    "tfseam_p6_8bit_lp4": (448, 264, 10, 8, ["--preset", "6", "--lp", "4", "+seam", "+tfseam"]),
    # the temporal filter's sub-pel refinement as one device call per (picture, reference) pair (tf_subpel_search served from the batch): SVT_HIP_TF_SUBPEL_SEAM=1
    "tfsubpel_p8_8bit": (448, 264, 10, 8, ["--preset", "8", "--lp", "1", "+seam", "+tfseam", "+tfsubpel"]),
    "tfsubpel_p4_8bit": (256, 144, 8, 8, ["--preset", "4", "--lp", "1", "+seam", "+tfseam", "+tfsubpel"]),
    "tfsubpel_p6_8bit_lp4": (448, 264, 10, 8, ["--preset", "6", "--lp", "4", "+seam", "+tfseam", "+tfsubpel"]),
    # the temporal filter of a central picture as ONE device stage (sub-pel refinement, block-size decisions, final motion compensation, filter): SVT_HIP_TF_SEAM=1
    "tfdriver_p8_8bit": (448, 264, 20, 8, ["--preset", "8", "--lp", "1", "+seam", "+tfseam", "+tfdriver"]),
    "tfdriver_p4_8bit": (256, 144, 18, 8, ["--preset", "4", "--lp", "1", "+seam", "+tfseam", "+tfdriver"]),
    "tfdriver_p6_8bit_lp4": (448, 264, 20, 8, ["--preset", "6", "--lp", "4", "+seam", "+tfseam", "+tfdriver"]),
    "tfdriver_p2_8bit": (256, 144, 10, 8, ["--preset", "2", "--lp", "1", "+seam", "+tfseam", "+tfdriver"]),
    "tfdriver_p10_8bit": (448, 264, 20, 8, ["--preset", "10", "--lp", "1", "+seam", "+tfseam", "+tfdriver"]),  # the temporal filter's HME runs level 0 only there (tf_ctrls.hme_me_level 3 / 4)
    "tfdriver_1080p_p8": (1920, 1080, 20, 8, ["--preset", "8", "+seam", "+tfseam", "+tfdriver"]),
    "tfdriver_p8_10bit": (256, 144, 18, 10, ["--preset", "8", "--lp", "1", "+seam", "+tfseam", "+tfdriver"]),  # high bit depth: the packed 16-bit planes, searches on the 8-bit luma
    "tfdriver_p4_10bit": (256, 144, 12, 10, ["--preset", "4", "--lp", "1", "+seam", "+tfseam", "+tfdriver"]),
    # low delay (--pred-struct 1) from 720p up, a still scene: HME level-0 areas resized from list 0 / reference 0's motion, the zero-motion temporal filter
    "lowdelay_720p_p8_8bit": (1280, 720, 24, 8, ["--preset", "8", "--lp", "1", "--pred-struct", "1", "--tune", "1", "+static", "+seam", "+tfseam", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam"]),
    "lowdelay_720p_p10_10bit": (1280, 720, 24, 10, ["--preset", "10", "--lp", "1", "--pred-struct", "1", "--tune", "1", "+static", "+seam", "+tfseam", "+tfdriver", "+cdefseam", "+dlfseam"]),
    "lowdelay_1080p_p9_lp4": (1920, 1080, 24, 8, ["--preset", "9", "--lp", "4", "--pred-struct", "1", "--tune", "1", "+seam", "+tfseam", "+tfdriver", "+cdefseam", "+dlfseam"]),  # (moving scene)
    # screen content (--scm 1): enable_me_sr_adjustment == 2 -- search areas halved from check_00_center's SAD / the first reference's final SAD; with low delay also
    # the (4 + index) level-0 areas
    "screen_p8_8bit": (448, 264, 16, 8, ["--preset", "8", "--lp", "1", "--scm", "1", "+seam", "+tfseam", "+tfdriver", "+cdefseam", "+dlfseam"]),
    "screen_p5_8bit_lp2": (448, 264, 12, 8, ["--preset", "5", "--lp", "2", "--scm", "1", "+seam", "+tfseam", "+tfdriver"]),
    "screen_lowdelay_720p_p9": (1280, 720, 16, 8, ["--preset", "9", "--lp", "1", "--scm", "1", "--pred-struct", "1", "--tune", "1", "+static", "+seam", "+cdefseam", "+dlfseam"]),  # (no temporal filter: off for screen content in low delay, enc_handle.c:3309)
    "tiles_p8_8bit": (448, 264, 16, 8, ["--preset", "8", "--lp", "1", "--tile-columns", "1", "--tile-rows", "1", "+seam", "+tfseam", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    "tfsubpel_p2_10bit": (256, 144, 6, 10, ["--preset", "2", "--lp", "1", "+seam", "+tfseam", "+tfsubpel"]),  # high bit depth: the seam hands those searches to the reference
    "lrseam_1080p_p6": (1920, 1080, 5, 8, ["--preset", "6", "+lrseam"]),  # 1080p: tens of restoration units per plane, all host cores
    "seam_1080p_p8": (1920, 1080, 10, 8, ["--preset", "8", "+seam"]),  # every picture's MeContext from svt_aom_sig_deriv_me at the real 1080p derivation, all 510 SBs
    # BASELINE.json metric, second half: encoder fps @1080p preset 8 (C-only reference vs the same encoder with the ME stage on the MI355X), all host cores
    "fps_1080p_p8": (1920, 1080, 24, 8, ["--preset", "8", "+seam"]),
    "fps_1080p_p8_all": (1920, 1080, 60, 8, ["--preset", "8", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),
    # 3840x2160 10-bit, preset 8, 60 frames (10-bit preset 8 is where the multi-threaded C-only reference was seen not to reproduce its own
    # bitstream; run_case reports `reference_deterministic` and the identity verdict next to the two speeds)
    "fps_1080p_p8_all_tplrecon": (1920, 1080, 60, 8, ["--preset", "8", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),  # A/B against fps_1080p_p8_all: the reconstruction half of the TPL dispenser on the device too
    # the stages that PAY at preset 8 against a vectorised host (tools/seam_subset_probe.py, profiles/r05_seam_subsets_fps.txt): without the loop-restoration and deblocking seams,
    # whose reference code is cheap at this preset (0.35 / 0.8 ms of CPU per frame) while their stage calls sit at the end of every picture's pipeline
    "fps_1080p_p8_paying": (1920, 1080, 60, 8, ["--preset", "8", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+cdefseam", "+tplseam", "+tplrecon"]),
    "fps_4k10_p8_all": (3840, 2160, 60, 10, ["--preset", "8", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),
    # larger pictures: more work per stage call against the same fixed latency
    "fps_4k8_p8_all": (3840, 2160, 30, 8, ["--preset", "8", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),
    "fps_4k10_p8_all_tplrecon": (3840, 2160, 60, 10, ["--preset", "8", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    # (--psy-rd 0: with this fork's psy-rd on, a MULTI-THREADED 10-bit encode of the plain reference gives a different bitstream in every run -- five md5s in five 720p
    #  preset-8 runs, one md5 with --psy-rd 0, one md5 at 8 bit, one md5 at --lp 1: its distortion reads per-thread scratch past what the block in hand wrote, and which
    #  thread gets which superblock is timing -- so there is nothing to be identical to; the single-threaded 10-bit cases above keep psy-rd on)
    "fps_4k10_p8_30": (3840, 2160, 30, 10, ["--preset", "8", "--psy-rd", "0", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),  # BASELINE configs[4] at one GPU: bench.py's encoder_fps_4k10_preset8 (30 frames)
    "fps_4k8_p8_all_tplrecon": (3840, 2160, 30, 8, ["--preset", "8", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    "fps_1080p_p6_all_tplrecon": (1920, 1080, 32, 8, ["--preset", "6", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    "fps_1080p_p4_all_tplrecon": (1920, 1080, 12, 8, ["--preset", "4", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    "fps_1080p_p10_all_tplrecon": (1920, 1080, 60, 8, ["--preset", "10", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    "fps_1080p_p10_all_tplrecon_300": (1920, 1080, 300, 8, ["--preset", "10", "+clip60", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    "fps_1080p_p8_me": (1920, 1080, 60, 8, ["--preset", "8", "+seam"]),
    # steady state: 300 frames (the 60-frame clip looped by the application), so that one-time costs (HIP context, session, kernel code loading) amortise
    "fps_1080p_p8_all_300": (1920, 1080, 300, 8, ["--preset", "8", "+clip60", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),
    "fps_1080p_p8_all_tplrecon_300": (1920, 1080, 300, 8, ["--preset", "8", "+clip60", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    # the same clip with the host side limited to a few threads (--lp): where the host is the bottleneck, what do the device stages buy
    "fps_1080p_p8_all_lp4": (1920, 1080, 60, 8, ["--preset", "8", "--lp", "4", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),
    "fps_1080p_p8_all_lp8": (1920, 1080, 60, 8, ["--preset", "8", "--lp", "8", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),
    "fps_1080p_p8_all_lp16": (1920, 1080, 60, 8, ["--preset", "8", "--lp", "16", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),
    "fps_1080p_p8_metf_300": (1920, 1080, 300, 8, ["--preset", "8", "+clip60", "+seam", "+tfseam", "+tfsubpel", "+tfdriver"]),
    "fps_1080p_p6_all": (1920, 1080, 32, 8, ["--preset", "6", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),
    "fps_1080p_p4_all": (1920, 1080, 12, 8, ["--preset", "4", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),
    # the TPL dispenser's source-based half as one device stage per picture (integration/src_ops_process_seam.c): SVT_HIP_TPL_SEAM=1
    "tplseam_p8_8bit": (448, 264, 20, 8, ["--preset", "8", "--lp", "1", "+tplseam"]),
    "tplseam_p6_8bit_lp4": (448, 264, 20, 8, ["--preset", "6", "--lp", "4", "+tplseam"]),
    "tplseam_p4_8bit": (256, 144, 18, 8, ["--preset", "4", "--lp", "1", "+tplseam"]),
    "tplseam_p10_8bit": (448, 264, 20, 8, ["--preset", "10", "--lp", "1", "+tplseam"]),  # tpl level 5: 32x32 blocks, TX_32X8, partial SBs at 16x16
    "tplseam_p8_10bit": (256, 144, 18, 10, ["--preset", "8", "--lp", "1", "+tplseam"]),  # TPL works on the 8-bit MSB picture of a 10-bit encode
    "tplseam_1080p_p8": (1920, 1080, 20, 8, ["--preset", "8", "+tplseam"]),
    # both halves of the TPL dispenser on the device (the reconstruction half walks the block grid by anti-diagonals)
    "tplrecon_p8_8bit": (448, 264, 20, 8, ["--preset", "8", "--lp", "1", "+tplseam", "+tplrecon"]),
    "tplrecon_p4_8bit_lp2": (448, 264, 12, 8, ["--preset", "4", "--lp", "2", "+tplseam", "+tplrecon"]),
    "tplrecon_everyseam_1080p_p8": (1920, 1080, 20, 8, ["--preset", "8", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    "tplrecon_p2_8bit": (448, 264, 18, 8, ["--preset", "2", "--lp", "2", "+tplseam", "+tplrecon"]),  # tpl level 1 (csrc/tpl_full.hip)
    "tplrecon_everyseam_p1_8bit": (256, 144, 12, 8, ["--preset", "1", "--lp", "2", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    "tplseam_me_p8_8bit": (448, 264, 20, 8, ["--preset", "8", "--lp", "1", "+seam", "+tfseam", "+tfsubpel", "+tplseam"]),  # ME results produced by the device stage feed the TPL stage
    # the multi-device paths on the MI355X itself: logical devices of the one GPU (SVT_HIP_VIRTUAL_DEVICES), per-device sessions / arenas / resident planes; peer streams,
    # events and peer copies of the frame partition -- an asynchronous device, unlike the emulator
    "vdev3_everyseam_p8": (448, 264, 16, 8, ["--preset", "8", "--lp", "4", "+devices:0,1,2", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    "vdev2_everyseam_p4_lp2": (256, 144, 8, 8, ["--preset", "4", "--lp", "2", "+devices:0,1", "+seam", "+tfseam", "+tfsubpel", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),
    "vstrips3_cdef_lr_p4": (448, 264, 6, 8, ["--preset", "4", "--lp", "2", "+strips:0,1,2", "+lrseam", "+cdefseam"]),
    "vstrips2_cdef_lr_p8_10bit": (448, 264, 8, 10, ["--preset", "8", "--lp", "1", "+strips:0,1", "+lrseam", "+cdefseam"]),
    "vstrips4_cdef_lr_1080p_p6": (1920, 1080, 5, 8, ["--preset", "6", "+strips:0,1,2,3", "+lrseam", "+cdefseam"]),
    # the error policy (SURVEY 8b "errors"): a HIP failure in the middle of the encode -- at the N-th device allocation / copy / synchronisation -- switches the device
    # path off; the encoder finishes on the reference's own kernels and functions with the same bitstream (emulator: SVT_HIPEMU_FAIL_AFTER)
    "tiny_fail_everyseam_p8_a": (128, 128, 10, 8, ["--preset", "8", "--lp", "2", "+failafter:40", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    "tiny_fail_everyseam_p8_b": (128, 128, 10, 8, ["--preset", "8", "--lp", "2", "+failafter:400", "+seam", "+tfseam", "+tfsubpel", "+tfdriver", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    "tiny_fail_everyseam_p4": (128, 128, 6, 8, ["--preset", "4", "--lp", "1", "+failafter:150", "+seam", "+tfseam", "+tfsubpel", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),
    "tiny_fail_dlfseam_p4": (128, 128, 6, 8, ["--preset", "4", "--lp", "1", "+failafter:6", "+dlfseam"]),  # the deblocking stage alone: its device call fails and the recorded segments are replayed through the reference's own edge filters (run with host="avx2" too: those load 16 threshold bytes)
    "tiny_fail_dlfseam_sb_p8": (128, 128, 8, 8, ["--preset", "8", "--lp", "1", "+failafter:6", "+dlfseam"]),  # the same through the SB-based recorder (presets >= 7)
    "tiny_fail_hooks_p8": (64, 64, 3, 8, ["--preset", "8", "--lp", "1", "+failafter:3000"]),  # the per-call dispatch pointers: a `_hip` call fails in flight and finishes through the saved pointer
    "tiny_fail_at_init": (64, 64, 3, 8, ["--preset", "8", "--lp", "1", "+failafter:0", "+seam", "+cdefseam"]),  # the very first device operation fails
    # 60 frames with the sampled checksum of every resident ME plane compared on top of the explicit invalidation: longer than the checksum table, so
    # that its eviction is exercised too -- plane_reuploads_by_checksum must stay 0
    "seam_hash_60f_p8": (448, 264, 60, 8, ["--preset", "8", "--lp", "4", "+seam", "+tfseam", "+tfsubpel", "+tfdriver"]),
    # small cases for the CPU lock-step emulator (tests/test_encoder_identity.py, -m "not gpu")
    # one encode over TWO (emulated) devices: SVT_HIP_DEVICES=0,1 shards the pictures by picture number; every seam at once
    "tiny_2dev_everyseam_p8": (128, 64, 12, 8, ["--preset", "8", "--lp", "2", "+devices:0,1", "+seam", "+tfseam", "+tfsubpel", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),
    # the frame-partition case from the C host: ONE picture's CDEF / loop-restoration launches cut into strips over two (emulated) devices (SVT_HIP_STRIPS)
    "tiny_strips_cdef_lr_p4": (128, 136, 3, 8, ["--preset", "4", "--lp", "2", "+strips:0,1", "+lrseam", "+cdefseam"]),
    "tiny_strips_cdef_lr_p8_10bit": (128, 136, 4, 10, ["--preset", "8", "--lp", "1", "+strips:0,1", "+lrseam", "+cdefseam"]),
    "tiny_2dev_everyseam_p4": (128, 128, 6, 8, ["--preset", "4", "--lp", "2", "+devices:0,1", "+seam", "+tfseam", "+tfsubpel", "+lrseam", "+cdefseam", "+dlfseam", "+tplseam"]),
    "tiny_tplseam_p8": (192, 128, 18, 8, ["--preset", "8", "--lp", "1", "+tplseam"]),
    "tiny_tplrecon_p8": (192, 128, 18, 8, ["--preset", "8", "--lp", "1", "+tplseam", "+tplrecon"]),
    "tiny_tplrecon_p10": (192, 136, 18, 8, ["--preset", "10", "--lp", "1", "+tplseam", "+tplrecon"]),
    "tiny_tplrecon_p4_lp2": (192, 136, 12, 8, ["--preset", "4", "--lp", "2", "+tplseam", "+tplrecon"]),
    "tiny_tplseam_p10": (192, 136, 18, 8, ["--preset", "10", "--lp", "1", "+tplseam"]),
    # tpl level 1 (presets <= M2: every intra mode, SATD costs, quarter-pel vectors, rate, per-layer quantizer; csrc/tpl_full.hip), both halves on the device
    "tiny_tplrecon_p2": (128, 72, 10, 8, ["--preset", "2", "--lp", "1", "+tplseam", "+tplrecon"]),
    "tiny_tplseam_p2": (128, 72, 10, 8, ["--preset", "2", "--lp", "1", "+tplseam"]),
    "tiny_2dev_tplrecon_p2": (128, 72, 10, 8, ["--preset", "2", "--lp", "2", "+devices:0,1", "+tplseam", "+tplrecon"]),  # the same over two (emulated) devices: resident planes by content id
    "tiny_tfdriver_p8": (128, 128, 12, 8, ["--preset", "8", "--lp", "1", "+seam", "+tfseam", "+tfdriver"]),
    "tiny_tfdriver_p8_10bit": (128, 128, 12, 10, ["--preset", "8", "--lp", "1", "+seam", "+tfseam", "+tfdriver"]),
    "tiny_tfdriver_p4_lp2": (128, 128, 10, 8, ["--preset", "4", "--lp", "2", "+seam", "+tfseam", "+tfdriver"]),
    # low delay (--pred-struct 1): HME level-0 areas resized from list 0 / reference 0's motion, the zero-motion temporal filter, no TPL
    "tiny_lowdelay_p8": (128, 128, 12, 8, ["--preset", "8", "--lp", "1", "--pred-struct", "1", "--tune", "1", "+seam", "+cdefseam", "+dlfseam"]),
    "tiny_lowdelay_p10_10bit": (128, 128, 10, 10, ["--preset", "10", "--lp", "1", "--pred-struct", "1", "--tune", "1", "+seam", "+cdefseam", "+dlfseam"]),
    "tiny_lowdelay_720p_tf_10bit": (1280, 720, 10, 10, ["--preset", "8", "--lp", "1", "--pred-struct", "1", "--tune", "1", "+static", "+tfseam", "+tfdriver"]),
    "tiny_lowdelay_720p_tf": (1280, 720, 10, 8, ["--preset", "8", "--lp", "1", "--pred-struct", "1", "--tune", "1", "+static", "+tfseam", "+tfdriver"]),  # the low-delay temporal filter is on from 720p up (enc_handle.c:3303-3310)
    "tiny_screen_p8": (128, 128, 6, 8, ["--preset", "8", "--lp", "1", "--scm", "1", "+seam"]),
    "tiny_screen_p5_tf": (192, 128, 10, 8, ["--preset", "5", "--lp", "1", "--scm", "1", "+seam", "+tfseam", "+tfdriver"]),
    "tiny_screen_lowdelay_p9": (192, 128, 12, 8, ["--preset", "9", "--lp", "1", "--scm", "1", "--pred-struct", "1", "--tune", "1", "+seam"]),
    "tiny_tiles_p8": (256, 128, 8, 8, ["--preset", "8", "--lp", "1", "--tile-columns", "1", "--tile-rows", "1", "+seam", "+tfseam", "+tfdriver", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]),  # 2 x 2 tiles: the TPL dispenser's tile path calls the per-SB function without segments (src_ops_process.c:2068-2080)
    "tiny_tfsubpel_p8": (192, 128, 8, 8, ["--preset", "8", "--lp", "1", "+seam", "+tfseam", "+tfsubpel"]),
    "tiny_tfseam_p8": (192, 128, 8, 8, ["--preset", "8", "--lp", "1", "+seam", "+tfseam"]),
    "tiny_dlfseam_p4": (128, 64, 3, 8, ["--preset", "4", "--lp", "1", "+dlfseam"]),
    "tiny_dlfseam_sb_p8": (192, 128, 6, 8, ["--preset", "8", "--lp", "1", "+dlfseam"]),
    "tiny_dlfseam_sb_p8_lp2": (192, 128, 6, 8, ["--preset", "8", "--lp", "2", "+dlfseam"]),
    "tiny_cdefseam_p8": (128, 64, 3, 8, ["--preset", "8", "--lp", "1", "+cdefseam"]),
    "tiny_lrseam_p4": (96, 64, 3, 8, ["--preset", "4", "--lp", "1", "+lrseam"]),
    "tiny_seam_p8": (192, 128, 8, 8, ["--preset", "8", "--lp", "1", "+seam"]),
    "tiny_seam_p5_lp2": (128, 128, 6, 8, ["--preset", "5", "--lp", "2", "+seam"]),
    "tiny_p8_8bit": (64, 64, 3, 8, ["--preset", "8", "--lp", "1"]),
    "tiny_p8_10bit": (64, 64, 2, 10, ["--preset", "8", "--lp", "1"]),
    "tiny_p8_lossless": (64, 64, 2, 8, ["--preset", "8", "--lp", "1", "--lossless", "1", "--tune", "1"]),
}
# Option sweep (--case sweep): small encodes with the ME / TF / CDEF / deblocking / TPL (both halves) seams on, one encoder option set each; the claim is bitstream
# equality and no declined picture (since the end of round 4 also at tpl level 1 = presets <= 2: csrc/tpl_full.hip).  Sized for the CPU emulator (`--lib tests/emu/_build/libsvtav1_hipemu.so`).
_SW = ["+seam", "+tfseam", "+tfdriver", "+cdefseam", "+dlfseam", "+tplseam", "+tplrecon"]
SWEEP = {
    "sweep_overlays": (192, 128, 14, 8, ["--preset", "8", "--lp", "1", "--enable-overlays", "1"]),
    "sweep_hl2": (192, 128, 14, 8, ["--preset", "8", "--lp", "1", "--hierarchical-levels", "2"]),
    "sweep_keyint6": (192, 128, 14, 8, ["--preset", "8", "--lp", "1", "--keyint", "6"]),
    "sweep_scd": (192, 128, 18, 8, ["--preset", "8", "--lp", "1", "--scd", "1", "--irefresh-type", "1", "--keyint", "8"]),
    "sweep_no_cdef_dlf": (192, 128, 14, 8, ["--preset", "8", "--lp", "1", "--enable-cdef", "0", "--enable-dlf", "0"]),
    "sweep_no_tf": (192, 128, 14, 8, ["--preset", "8", "--lp", "1", "--enable-tf", "0"]),
    "sweep_vbr": (192, 128, 14, 8, ["--preset", "8", "--lp", "1", "--tune", "1", "--rc", "1", "--tbr", "500"]),
    "sweep_vbr_p10": (192, 128, 14, 8, ["--preset", "10", "--lp", "1", "--tune", "1", "--rc", "1", "--tbr", "300"]),
    "sweep_two_pass": (192, 128, 14, 8, ["--preset", "8", "--lp", "1", "--passes", "2", "--rc", "1", "--tbr", "400", "--tune", "1"]),
    "sweep_cbr_lowdelay": (192, 128, 14, 8, ["--preset", "8", "--lp", "1", "--tune", "1", "--rc", "2", "--tbr", "500", "--pred-struct", "1"]),
    "sweep_lookahead0": (192, 128, 14, 8, ["--preset", "8", "--lp", "1", "--lookahead", "0"]),
    "sweep_film_grain": (192, 128, 14, 8, ["--preset", "8", "--lp", "1", "--film-grain", "8"]),
    "sweep_fast_decode": (192, 128, 14, 8, ["--preset", "8", "--lp", "1", "--fast-decode", "1"]),
    "sweep_tune0": (192, 128, 14, 8, ["--preset", "8", "--lp", "1", "--tune", "0"]),
    "sweep_tiles_10bit": (192, 128, 14, 10, ["--preset", "8", "--lp", "1", "--tile-columns", "1"]),
    "sweep_lp3": (192, 128, 14, 8, ["--preset", "8", "--lp", "3"]),
    "sweep_odd_size": (190, 130, 12, 8, ["--preset", "8", "--lp", "1"]),
    "sweep_odd_size_10bit": (202, 138, 10, 10, ["--preset", "8", "--lp", "1"]),
    "sweep_p12": (192, 128, 14, 8, ["--preset", "12", "--lp", "1"]),
    "sweep_p6": (192, 128, 14, 8, ["--preset", "6", "--lp", "1"]),
    "sweep_p3": (128, 64, 8, 8, ["--preset", "3", "--lp", "1"]),
    "sweep_p2": (96, 64, 6, 8, ["--preset", "2", "--lp", "1"]),
    "sweep_p0": (64, 64, 4, 8, ["--preset", "0", "--lp", "1"]),
}
for _k, (_w, _h, _n, _bd, _a) in SWEEP.items():
    CASES[_k] = (_w, _h, _n, _bd, _a + _SW)
GPU_CASES = [k for k in CASES if not k.startswith(("tiny_", "fps_", "sweep_"))]
SEAM_CASES = [k for k in GPU_CASES if k.startswith(("seam_", "lrseam_", "cdefseam_", "allseams_", "dlfseam_", "everyseam_", "tfseam_", "tfsubpel_", "tfdriver_", "tplseam_", "tplrecon_", "lowdelay_", "screen_", "tiles_"))]


def make_clip(path, w, h, n, bd, seed=7, static=False):
    """Textured luma sliding by (2, 1) pixels per frame plus noise, smooth chroma; 4:2:0 planar, 16-bit little endian above 8 bit.  static: no slide (a fixed
    camera; the low-delay temporal filter predicts every block from the co-located one, so only a still scene gives it weights above zero)."""
    g = np.random.default_rng(seed)
    W, H = w + max(64, 2 * n + 2), h + max(64, n + 2)  # (margin for the slide; unchanged for the clips of up to 31 frames)
    base = np.kron(g.integers(0, 256, (H // 8 + 2, W // 8 + 2)).astype(np.float32), np.ones((8, 8), np.float32))[:H, :W]
    yy, xx = np.mgrid[0:H, 0:W]
    tex = base * 0.5 + 64 + 40 * np.sin(xx / 9.0) + 30 * np.cos(yy / 7.0)
    with open(path, "wb") as f:
        for i in range(n):
            ox, oy = (0, 0) if static else (2 * i, i)
            y = np.clip(tex[oy:oy + h, ox:ox + w] + g.normal(0, 2, (h, w)), 0, 255)
            u = np.clip(128 + 20 * np.sin((xx[:h // 2, :w // 2] + ox) / 5.0), 0, 255)
            v = np.clip(128 + 20 * np.cos((yy[:h // 2, :w // 2] + oy) / 6.0), 0, 255)
            for p in (y, u, v):
                f.write(p.astype(np.uint8).tobytes() if bd == 8 else (p.astype(np.uint16) << (bd - 8)).astype("<u2").tobytes())


DETHEAP_DIR = os.path.join(ROOT, "tests", "detheap")


def deterministic_env(bd):
    """What EVERY encode of a high-bit-depth comparison runs under (the reference alone and the reference with the device stages alike), so that equality can be demanded
    on the first attempt.  The reference's 10-bit path reads memory nobody wrote for the picture in hand (MemorySanitizer on the plain C encoder,
    profiles/r05_reference_msan_10bit.txt: svt_psy_distortion / svt_sa8d_8x8 / svt_satd_4x4, psy_rd.c:94-165, on the 16-bit source picture of a re-used picture control
    set; heap arrays of pcs.c:539), so its bitstream depends on which pool object a picture happens to get -- on thread timing -- and it does not always reproduce itself
    (profiles/r05_race_probe_10bit.txt).  Two test-only instruments take the timing out: SVT_HIP_TEST_SCRUB_PCS=1 zero-fills that 16-bit source picture whenever a
    control set is taken from its pool (integration/pic_manager_process_seam.c), and tests/detheap zero-fills every heap block at allocation.  Until round 5 a flipped
    10-bit encode was simply repeated up to three times."""
    if bd <= 8:
        return {}
    so = os.path.join(DETHEAP_DIR, "libdetheap.so")
    src = os.path.join(DETHEAP_DIR, "detheap.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", so, src], check=True)
    return {"SVT_HIP_TEST_SCRUB_PCS": "1", "LD_PRELOAD": so}


def encode(clip, w, h, n, bd, extra, out_prefix, env_extra=None, timeout=1800, enc=None):
    env = dict(os.environ)
    for k in ("SVT_HIP", "SVT_HIP_LIB", "SVT_HIP_COUNT", "SVT_HIP_ONLY", "SVT_HIP_SKIP"):
        env.pop(k, None)
    env.update(deterministic_env(bd))
    env.update(env_extra or {})
    cmd = [enc or ENC, "-i", clip, "-w", str(w), "-h", str(h), "--fps", "30", "-n", str(n), "--input-depth", str(bd)] + extra + \
          ["-b", out_prefix + ".ivf"]
    # (no `-o` reconstruction file: the reconstruction is a function of the bitstream, and with -o the reference APP busy-polls svt_av1_get_recon on its
    # main thread, which starves the encoder's own threads on a CPU-limited box -- a 1 s encode was seen to take minutes there)
    import resource
    u0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)
    dt = time.time() - t0
    u1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    LAST_CPU_S[0] = (u1.ru_utime - u0.ru_utime) + (u1.ru_stime - u0.ru_stime)  # (encodes run one at a time here: the delta is this child's)
    return r, dt


def _stage_cpu(path, n):
    """integration/seam_cpu.h's per-stage thread-CPU sums of one encode -> {stage: ms per frame}"""
    if not os.path.exists(path):
        return None
    st = dict(ln.split() for ln in open(path).read().splitlines() if ln.strip())
    return {k[:-7]: round(int(v) / n, 3) for k, v in st.items() if k.endswith("_cpu_ms")}


def run_case(name, lib, outdir, device=0, only=None, skip=None, timeout=1800, host="c", cpu_stats=False):
    """host = "c": the C-only encoder with and without the HIP library (the identity anchor).  host = "avx2": additionally the intrinsics encoder without anything
    (`fps_avx2`, bitstream must equal the C-only one) and the HIP run uses THAT encoder (`fps_hip` = AVX2 host kernels + device stages)."""
    w, h, n, bd, extra = CASES[name]
    os.makedirs(outdir, exist_ok=True)
    clip = os.path.join(outdir, name + ".yuv")
    clip_frames = next((int(a[5:]) for a in extra if a.startswith("+clip")), n)
    make_clip(clip, w, h, min(clip_frames, n), bd, static="+static" in extra)
    seam, with_hook, lrseam, cdefseam, dlfseam = "+seam" in extra, "+hook" in extra, "+lrseam" in extra, "+cdefseam" in extra, "+dlfseam" in extra
    tplseam = "+tplseam" in extra
    extra = [a for a in extra if not a.startswith("+")]
    cpu_env = (lambda tag: {"SVT_HIP_SEAM_CPU_STATS": os.path.join(outdir, "%s_%s_cpu.txt" % (name, tag))}) if cpu_stats else (lambda tag: {})
    rc, tc = encode(clip, w, h, n, bd, extra, os.path.join(outdir, name + "_c"), cpu_env("c"), timeout=timeout)
    cpu_s = {"c": LAST_CPU_S[0]}
    deterministic = True
    if "--lp" not in extra or extra[extra.index("--lp") + 1] != "1":  # multi-threaded: is the C-only reference reproducible at all for this configuration?
        rc2, _ = encode(clip, w, h, n, bd, extra, os.path.join(outdir, name + "_c2"), timeout=timeout)
        deterministic = rc2.returncode == 0 and open(os.path.join(outdir, name + "_c.ivf"), "rb").read() == open(os.path.join(outdir, name + "_c2.ivf"), "rb").read()
        os.remove(os.path.join(outdir, name + "_c2.ivf"))
    counts_file = os.path.join(outdir, name + "_counts.txt")
    seam_file = os.path.join(outdir, name + "_seam.txt")
    env = {"SVT_HIP": str(device), "SVT_HIP_LIB": lib, "SVT_HIP_COUNT": counts_file}
    lrseam_file = os.path.join(outdir, name + "_lrseam.txt")
    tf_alone = not seam and "+tfdriver" in CASES[name][4] and "--pred-struct" in extra  # the low-delay temporal filter has no ME: its stage can run without the ME seams' session
    if seam:
        # SVT_HIP_ME_SEAM_HASH=1: the residency of the ME planes rests on explicit invalidation; the identity suites ALSO compare a sampled checksum of every
        # resident plane and count what the explicit rule missed -- any non-zero count voids the case (below)
        env.update({"SVT_HIP_ME_SEAM": "1", "SVT_HIP_ME_SEAM_STATS": seam_file})
        if not name.startswith("fps_"):  # (the fps cases time the product's default path: no hashing; their bitstreams are still compared)
            env["SVT_HIP_ME_SEAM_HASH"] = "1"
    if seam or tf_alone:
        if "+tfseam" in CASES[name][4]:
            env["SVT_HIP_TF_ME_SEAM"] = "1"
        if "+tfsubpel" in CASES[name][4]:
            env.update({"SVT_HIP_TF_SUBPEL_SEAM": "1", "SVT_HIP_TF_SUBPEL_SEAM_STATS": os.path.join(outdir, name + "_tfsubpel.txt")})
        if "+tfdriver" in CASES[name][4]:
            env.update({"SVT_HIP_TF_SEAM": "1", "SVT_HIP_TF_SEAM_STATS": os.path.join(outdir, name + "_tfdriver.txt")})
    if lrseam:
        env.update({"SVT_HIP_LR_SEAM": "1", "SVT_HIP_LR_SEAM_STATS": lrseam_file})
    cdefseam_file = os.path.join(outdir, name + "_cdefseam.txt")
    if cdefseam:
        env.update({"SVT_HIP_CDEF_SEAM": "1", "SVT_HIP_CDEF_SEAM_STATS": cdefseam_file})
    dlfseam_file = os.path.join(outdir, name + "_dlfseam.txt")
    if dlfseam:
        env.update({"SVT_HIP_DLF_SEAM": "1", "SVT_HIP_DLF_SEAM_STATS": dlfseam_file})
    tplseam_file = os.path.join(outdir, name + "_tplseam.txt")
    if tplseam:
        env.update({"SVT_HIP_TPL_SEAM": "1", "SVT_HIP_TPL_SEAM_STATS": tplseam_file})
        if "+tplrecon" in CASES[name][4]:
            env["SVT_HIP_TPL_RECON_SEAM"] = "1"
    devices = next((a[9:] for a in CASES[name][4] if a.startswith("+devices:")), None)
    shard_file = os.path.join(outdir, name + "_devices.txt")
    if devices:
        env.update({"SVT_HIP_DEVICES": devices, "SVT_HIP_DEVICES_STATS": shard_file})
        # the emulator emulates the devices; on a GPU box with fewer GPUs than the list names, logical devices of the one GPU (svt_hip_device_count = GPUs x V)
        env.update({"SVT_HIPEMU_DEVICES": str(len(devices.split(",")))} if "hipemu" in lib else {"SVT_HIP_VIRTUAL_DEVICES": str(max(int(d) for d in devices.split(",")) + 1)})
    strips = next((a[8:] for a in CASES[name][4] if a.startswith("+strips:")), None)
    strips_file = os.path.join(outdir, name + "_strips.txt")
    if strips:
        env.update({"SVT_HIP_STRIPS": strips, "SVT_HIP_STRIPS_STATS": strips_file})
        env.update({"SVT_HIPEMU_DEVICES": str(len(strips.split(",")))} if "hipemu" in lib else {"SVT_HIP_VIRTUAL_DEVICES": str(max(int(d) for d in strips.split(",")) + 1)})
    # "+failafter:N" (emulator): the N-th device allocation / copy / synchronisation of the encode and every later one fail (tests/emu/hipemu.h).  The claim (SURVEY 8b
    # "errors"): the encoder finishes, the bitstream is identical, the library reported the switch-off -- stages before the failure ran on the device, later ones declined
    fail_after = next((a[11:] for a in CASES[name][4] if a.startswith("+failafter:")), None)
    if fail_after is not None:
        env["SVT_HIPEMU_FAIL_AFTER"] = fail_after
    if (seam or lrseam or cdefseam or dlfseam or tplseam or tf_alone) and not with_hook and not only:
        only = "-"  # no RTCD pointer matches: the seam(s) alone
    if only:
        env["SVT_HIP_ONLY"] = only
    if skip:
        env["SVT_HIP_SKIP"] = skip
    rx = None
    if host != "c":
        rx, tx = encode(clip, w, h, n, bd, extra, os.path.join(outdir, name + "_x"), cpu_env(host), timeout=timeout, enc=HOST_ENC[host])
        cpu_s[host] = LAST_CPU_S[0]
    env.update(cpu_env(host + "_with_stages"))
    rh, th = encode(clip, w, h, n, bd, extra, os.path.join(outdir, name + "_hip"), env, timeout=timeout, enc=HOST_ENC[host])
    cpu_s[host + "_with_stages"] = LAST_CPU_S[0]
    attempts = 1  # (kept in the record: every case is decided by its first and only attempt since round 6 -- see deterministic_env)
    res = {"case": name, "host": host, "width": w, "height": h, "frames": n, "bit_depth": bd, "args": extra, "rc_c": rc.returncode, "rc_hip": rh.returncode,
           "seconds_c": round(tc, 2), "seconds_hip": round(th, 2), "reference_deterministic": deterministic, "hip_encode_attempts": attempts,
           # host CPU seconds (user + system, every thread) per frame: what the offload takes off the host
           "host_cpu_s_per_frame": {k: round(v / n, 5) for k, v in cpu_s.items()}}
    if cpu_stats:  # thread CPU time inside each stage of SURVEY 8 (the reference's own functions without the seams, the device stage calls with them)
        res["stage_cpu_ms_per_frame"] = {tag: _stage_cpu(os.path.join(outdir, "%s_%s_cpu.txt" % (name, tag)), n) for tag in cpu_s}
    for tag, r in (("c", rc), ("hip", rh)) + (((host, rx),) if rx is not None else ()):  # the encoder's own speed line
        for ln in (r.stdout + r.stderr).splitlines():
            if "Average Speed" in ln:
                res["fps_" + tag] = float(ln.split(":")[1].split()[0])
    hooked = [ln for ln in rh.stderr.splitlines() if ln.startswith("SVT_HIP:")]
    diag = [ln for ln in rh.stderr.splitlines() if ln.startswith("SVT_HIP_ME_SEAM") and "runs as one device stage" not in ln]  # (not the seam's start-up line)
    if diag:  # SVT_HIP_ME_SEAM_VERIFY=1: per-SB differences against the reference's own function
        res["seam_diagnostics"] = diag[:40]
        print("\n".join(diag[:40]), file=sys.stderr)
    res["hook_line"] = hooked[0] if hooked else None
    if rc.returncode or rh.returncode or not hooked:
        res["identical"] = False
        res["stderr_tail"] = (rc.stderr[-1500:] if rc.returncode else rh.stderr[-1500:])
        return res
    same = True
    for ext in (".ivf",):
        a = open(os.path.join(outdir, name + "_c" + ext), "rb").read()
        b = open(os.path.join(outdir, name + "_hip" + ext), "rb").read()
        res["bytes" + ext] = len(a)
        same = same and len(a) > 0 and a == b
    if rx is not None:  # the intrinsics encoder alone must reproduce the C-only bitstream too
        x = open(os.path.join(outdir, name + "_x.ivf"), "rb").read() if rx.returncode == 0 else b""
        res[host + "_identical_to_c"] = len(x) > 0 and x == open(os.path.join(outdir, name + "_c.ivf"), "rb").read()
        same = same and res[host + "_identical_to_c"]
    res["identical"] = same
    res["bitstream_equal"] = bool(same)  # the files alone; "identical" additionally demands that the stages the case names really ran
    if fail_after is not None:
        res["device_path_switched_off"] = "the device path is off from here on" in rh.stderr
        res["last_error_line"] = next((ln for ln in rh.stderr.splitlines() if ln.startswith("libsvtav1_hip:")), None)
        res["identical"] = bool(same) and res["device_path_switched_off"]
        return res
    if seam:
        st = dict(ln.split(None, 1) for ln in open(seam_file).read().splitlines()) if os.path.exists(seam_file) else {}
        res["seam"] = {k: (int(v) if v.strip().isdigit() else v.strip()) for k, v in st.items()}
        # the claim is void unless every picture really went through the device stage
        res["identical"] = same and res["seam"].get("pictures_offloaded", 0) > 0 and res["seam"].get("pictures_declined", 1) == 0
        if not name.startswith("fps_"):
            res["identical"] = res["identical"] and res["seam"].get("plane_reuploads_by_checksum", 1) == 0  # a resident plane was rewritten behind the explicit invalidation
        if "+tfseam" in CASES[name][4]:  # temporal-filter pairs really went through the stage, none declined
            ld = "--pred-struct" in CASES[name][4]  # the low-delay temporal filter performs no ME (produce_temporally_filtered_pic_ld): no pair exists
            res["identical"] = res["identical"] and (ld or res["seam"].get("tf_pairs_offloaded", 0) > 0) and res["seam"].get("tf_pairs_declined", 1) == 0
    if "+tfsubpel" in CASES[name][4]:
        f = os.path.join(outdir, name + "_tfsubpel.txt")
        st = dict(ln.split(None, 1) for ln in open(f).read().splitlines()) if os.path.exists(f) else {}
        res["tfsubpel"] = {k: int(v) for k, v in st.items()}
        if bd == 8 and "+tfdriver" not in CASES[name][4]:  # the dedicated 8-bit cases must really be served from the device batch (the driver seam replaces the callers)
            res["identical"] = res["identical"] and res["tfsubpel"].get("searches_served", 0) > 0
    if "+tfdriver" in CASES[name][4]:
        f = os.path.join(outdir, name + "_tfdriver.txt")
        st = dict(ln.split(None, 1) for ln in open(f).read().splitlines()) if os.path.exists(f) else {}
        res["tfdriver"] = {k: (int(v) if v.strip().isdigit() else v.strip()) for k, v in st.items()}
        if True:  # void unless central pictures really went through the device stage, none left to the reference (8 and 10 bit)
            res["identical"] = res["identical"] and res["tfdriver"].get("pictures_filtered", 0) > 0 and res["tfdriver"].get("pictures_declined", 1) == 0
    if lrseam:
        st = dict(ln.split(None, 1) for ln in open(lrseam_file).read().splitlines()) if os.path.exists(lrseam_file) else {}
        res["lrseam"] = {k: int(v) for k, v in st.items()}
        res["identical"] = res["identical"] and res["lrseam"].get("units_searched", 0) > 0  # void unless restoration units really went through the device stage
    if dlfseam:
        st = dict(ln.split(None, 1) for ln in open(dlfseam_file).read().splitlines()) if os.path.exists(dlfseam_file) else {}
        res["dlfseam"] = {k: int(v) for k, v in st.items()}
        if name.startswith(("dlfseam_", "tiny_dlfseam", "fps_1080p_p8_all", "everyseam_4k10_p8")):  # the dedicated cases and the metric's preset-8 case must really
            res["identical"] = res["identical"] and res["dlfseam"].get("segments", 0) > 0               # filter segments on the device (other configurations may pick level 0)
        if "_sb_" in name or name.startswith("fps_1080p_p8_all"):  # ... and, at presets >= 7, from the per-SB records of the coding loop
            res["identical"] = res["identical"] and res["dlfseam"].get("pictures_filtered_from_sb_records", 0) > 0
    if devices:  # every listed device must really have received stage calls
        st = dict(ln.split(None, 1) for ln in open(shard_file).read().splitlines()) if os.path.exists(shard_file) else {}
        res["devices"] = {k: int(v) for k, v in st.items()}
        res["identical"] = res["identical"] and len(res["devices"]) == len(devices.split(",")) and all(v > 0 for v in res["devices"].values())
    if strips:  # frame launches really went through a partition of the listed devices
        st = dict(ln.split(None, 1) for ln in open(strips_file).read().splitlines()) if os.path.exists(strips_file) else {}
        res["strips"] = {k: int(v) for k, v in st.items()}
        res["identical"] = res["identical"] and res["strips"].get("frame_launches_through_a_partition", 0) > 0
    if tplseam:
        st = dict(ln.split(None, 1) for ln in open(tplseam_file).read().splitlines()) if os.path.exists(tplseam_file) else {}
        res["tplseam"] = {k: int(float(v)) for k, v in st.items()}
        if name.startswith(("tplseam_", "tiny_tplseam")):  # void unless pictures really went through the device stage and none was declined
            res["identical"] = res["identical"] and res["tplseam"].get("pictures_offloaded", 0) > 0 and res["tplseam"].get("pictures_declined", 1) == 0
        if "+tplrecon" in CASES[name][4]:  # ... and the reconstruction half of every dispenser call ran on the device, the per-SB function skipped
            t = res["tplseam"]
            res["identical"] = res["identical"] and t.get("recon_pictures", 0) > 0 and t.get("recon_blocks_coded", 0) > 0 and t.get("sb_calls_skipped", 0) > 0 and \
                t.get("recon_pictures", 0) == t.get("pictures_offloaded", 0) + t.get("pictures_with_stored_statistics", 0) and t.get("pictures_declined", 1) == 0
    if cdefseam:
        st = dict(ln.split(None, 1) for ln in open(cdefseam_file).read().splitlines()) if os.path.exists(cdefseam_file) else {}
        res["cdefseam"] = {k: int(v) for k, v in st.items()}
        res["identical"] = res["identical"] and res["cdefseam"].get("filter_blocks", 0) > 0 and res["cdefseam"].get("pictures_declined", 1) == 0
    if name.startswith("sweep_"):  # the sweep's claim: equal bitstreams and no picture declined by the ME stage (a stage an option switches off has nothing to run)
        res["identical"] = bool(res["bitstream_equal"]) and res.get("seam", {}).get("pictures_declined", 1) == 0
    counts = {}
    if os.path.exists(counts_file):
        for ln in open(counts_file):
            k, v = ln.split()
            counts[k] = int(v)
    res["pointers_installed"] = len(counts)
    res["pointers_hit"] = sum(1 for v in counts.values() if v)
    res["calls"] = sum(counts.values())
    res["counts"] = {k: v for k, v in sorted(counts.items(), key=lambda kv: -kv[1]) if v}
    for f in (clip,):
        os.remove(f)
    return res


def seam_env(name, lib, outdir, tag, device=0):
    """the environment run_case() gives the HIP run of `name` (every seam the case names), statistics files suffixed with `tag`"""
    flags = CASES[name][4]
    env = {"SVT_HIP": str(device), "SVT_HIP_LIB": lib}
    f = lambda k: os.path.join(outdir, "%s_%s_%s.txt" % (name, tag, k))  # noqa: E731
    if "+seam" in flags:
        env.update({"SVT_HIP_ME_SEAM": "1", "SVT_HIP_ME_SEAM_STATS": f("seam")})
        if "+tfseam" in flags:
            env["SVT_HIP_TF_ME_SEAM"] = "1"
        if "+tfsubpel" in flags:
            env.update({"SVT_HIP_TF_SUBPEL_SEAM": "1", "SVT_HIP_TF_SUBPEL_SEAM_STATS": f("tfsubpel")})
        if "+tfdriver" in flags:
            env.update({"SVT_HIP_TF_SEAM": "1", "SVT_HIP_TF_SEAM_STATS": f("tfdriver")})
    for flag, var in (("+lrseam", "SVT_HIP_LR_SEAM"), ("+cdefseam", "SVT_HIP_CDEF_SEAM"), ("+dlfseam", "SVT_HIP_DLF_SEAM"), ("+tplseam", "SVT_HIP_TPL_SEAM")):
        if flag in flags:
            env.update({var: "1", var + "_STATS": f(var.lower())})
    if "+tplrecon" in flags:
        env["SVT_HIP_TPL_RECON_SEAM"] = "1"
    if "+hook" not in flags:
        env["SVT_HIP_ONLY"] = "-"
    return env


def repeat_pairs(name, lib, outdir, want_ivf, pairs=2, host="avx2", timeout=600):
    """`pairs` more (host alone, host + the case's stage seams) encodes of case `name`, each bitstream compared with `want_ivf` (bytes): the fps of a 0.5 s encode spreads
    by +- 5 % from run to run, so bench.py quotes the median of several pairs.  -> {"fps_alone": [...], "fps_with_stages": [...], "identical": bool}"""
    w, h, n, bd, extra = CASES[name]
    clip = os.path.join(outdir, name + "_rep.yuv")
    clip_frames = next((int(a[5:]) for a in extra if a.startswith("+clip")), n)
    make_clip(clip, w, h, min(clip_frames, n), bd, static="+static" in extra)
    args = [a for a in extra if not a.startswith("+")]
    res = {"fps_alone": [], "fps_with_stages": [], "identical": True}
    for i in range(pairs):
        for key, env in (("fps_alone", None), ("fps_with_stages", seam_env(name, lib, outdir, "rep%d" % i))):
            out = os.path.join(outdir, "%s_rep_%s_%d" % (name, key, i))
            r, _ = encode(clip, w, h, n, bd, args, out, env, timeout=timeout, enc=HOST_ENC[host])
            ok = r.returncode == 0 and open(out + ".ivf", "rb").read() == want_ivf
            res["identical"] = res["identical"] and ok
            for ln in (r.stdout + r.stderr).splitlines():
                if "Average Speed" in ln:
                    res[key].append(float(ln.split(":")[1].split()[0]))
            if os.path.exists(out + ".ivf"):
                os.remove(out + ".ivf")
    os.remove(clip)
    return res


def run_instances(name, lib, outdir, k, host="avx2", timeout=1800):
    """K concurrent encodes of case `name` (each its own process, all host threads, one shared MI355X): aggregate fps and host CPU seconds per frame of the intrinsics
    host alone vs the same host with the case's stage seams on the GPU.  Every one of the 2K bitstreams must equal the C-only encoder's (the one
    thing offload can buy a host that is already fast is CPU time -- visible as aggregate throughput once the host cores are saturated)."""
    import resource
    w, h, n, bd, extra = CASES[name]
    os.makedirs(outdir, exist_ok=True)
    clip = os.path.join(outdir, name + "_inst.yuv")
    clip_frames = next((int(a[5:]) for a in extra if a.startswith("+clip")), n)
    make_clip(clip, w, h, min(clip_frames, n), bd, static="+static" in extra)
    args = [a for a in extra if not a.startswith("+")]
    rc, _ = encode(clip, w, h, n, bd, args, os.path.join(outdir, name + "_inst_c"), timeout=timeout)
    want = open(os.path.join(outdir, name + "_inst_c.ivf"), "rb").read() if rc.returncode == 0 else b""
    res = {"case": name, "instances": k, "host": host, "frames": n, "identical": len(want) > 0}
    for tag, with_stages in ((host, False), (host + "_with_stages", True)):
        procs = []
        u0 = resource.getrusage(resource.RUSAGE_CHILDREN)
        t0 = time.time()
        for i in range(k):
            env = dict(os.environ)
            for v in ("SVT_HIP", "SVT_HIP_LIB", "SVT_HIP_COUNT", "SVT_HIP_ONLY", "SVT_HIP_SKIP"):
                env.pop(v, None)
            if with_stages:
                env.update(seam_env(name, lib, outdir, "inst%d" % i))
            out = os.path.join(outdir, "%s_inst_%s_%d" % (name, "s" if with_stages else "x", i))
            cmd = [HOST_ENC[host], "-i", clip, "-w", str(w), "-h", str(h), "--fps", "30", "-n", str(n), "--input-depth", str(bd)] + args + ["-b", out + ".ivf"]
            procs.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env), out))
        outs = [(p.communicate(timeout=timeout), p.returncode, o) for p, o in procs]
        wall = time.time() - t0
        u1 = resource.getrusage(resource.RUSAGE_CHILDREN)
        same = all(rcode == 0 and open(o + ".ivf", "rb").read() == want for _, rcode, o in outs)
        own = [float(ln.split(":")[1].split()[0]) for (so, se), _, _ in outs for ln in (so + se).splitlines() if "Average Speed" in ln]
        res["fps_sum_of_encoder_reports_" + tag] = round(sum(own), 2)  # (the encoders' own speed lines: without process start-up and device initialisation)
        res["identical"] = res["identical"] and same
        res["fps_" + tag] = round(k * n / wall, 2)
        res["host_cpu_s_per_frame_" + tag] = round(((u1.ru_utime - u0.ru_utime) + (u1.ru_stime - u0.ru_stime)) / (k * n), 5)
        if not same:
            res["stderr_tail"] = next((e[-1200:] for (_, e), rcode, _o in outs if rcode), "")
        for _, _, o in outs:
            if os.path.exists(o + ".ivf"):
                os.remove(o + ".ivf")
    os.remove(clip)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "svt-av1-psy_amd", "libsvtav1_hip.so"))
    ap.add_argument("--case", default="all", help="case name, comma list, or 'all' (the GPU set)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "identity"))
    ap.add_argument("--only", default=None)
    ap.add_argument("--skip", default=None)
    ap.add_argument("--timeout", type=int, default=1800)
    ap.add_argument("--host", choices=("c", "avx2", "avx512"), default="c", help="host build the HIP run uses (avx2 / avx512: oracle/_ref/enc_avx2 / enc_avx512, also timed alone)")
    ap.add_argument("--cpu-stats", action="store_true", help="per-stage thread CPU time of every encode (integration/seam_cpu.h)")
    ap.add_argument("--instances", type=int, default=0, help="K > 0: run_instances() instead -- K concurrent encodes of the case on one GPU, host alone vs host + stages")
    a = ap.parse_args()
    if not os.path.exists(ENC):
        sys.exit("oracle/_ref/enc/SvtAv1EncApp is missing: run `make -C oracle enc` where /root/reference exists")
    names = GPU_CASES if a.case == "all" else (list(SWEEP) if a.case == "sweep" else a.case.split(","))
    if a.instances:
        ok = True
        for nme in names:
            r = run_instances(nme, os.path.abspath(a.lib), a.out, a.instances, host=a.host if a.host != "c" else "avx2", timeout=a.timeout)
            print(json.dumps(r), flush=True)
            ok = ok and r["identical"]
        sys.exit(0 if ok else 1)
    results, union = [], {}
    for nme in names:
        r = run_case(nme, os.path.abspath(a.lib), a.out, only=a.only, skip=a.skip, timeout=a.timeout, host=a.host, cpu_stats=a.cpu_stats)
        results.append(r)
        for k, v in r.get("counts", {}).items():
            union[k] = union.get(k, 0) + v
        print("%-20s identical=%s (bitstream %s)  calls=%s  pointers hit=%s/%s  C %.1fs  HIP %.1fs  %s" % (nme, r["identical"], r.get("bitstream_equal"), r.get("calls"), r.get("pointers_hit"),
                                                                                        r.get("pointers_installed"), r["seconds_c"], r["seconds_hip"],
                                                                                        str(r.get("seam", "")) + " " + str(r.get("lrseam", "")) + " " + str(r.get("cdefseam", "")) + " " + str(r.get("dlfseam", "")) + " " + str(r.get("tfsubpel", "")) + " " + str(r.get("tfdriver", "")) + " " + str(r.get("tplseam", "")) + " " + str(r.get("devices", ""))), flush=True)
        if "fps_c" in r:
            print("    encoder fps: C-only %.2f, %swith HIP (host = %s) %.2f; host CPU s / frame %s" % (
                r["fps_c"], ("%s intrinsics %.2f, " % (r["host"], r["fps_" + r["host"]])) if ("fps_" + r["host"]) in r and r["host"] != "c" else "", r["host"],
                r.get("fps_hip", 0.0), r.get("host_cpu_s_per_frame")), flush=True)
        if r.get("stage_cpu_ms_per_frame"):
            print("    stage CPU ms / frame: %s" % json.dumps(r["stage_cpu_ms_per_frame"]), flush=True)
        if not r["identical"]:
            print(r.get("stderr_tail", ""))
    summary = {"all_identical": all(r["identical"] for r in results), "pointers_hit_union": len(union), "calls_total": sum(union.values()),
               "hit_table": dict(sorted(union.items(), key=lambda kv: -kv[1])), "cases": results}
    with open(os.path.join(a.out, "identity.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("ALL IDENTICAL" if summary["all_identical"] else "MISMATCH", "- %d distinct pointers hit, %d calls" % (len(union), sum(union.values())))
    sys.exit(0 if summary["all_identical"] else 1)


if __name__ == "__main__":
    main()
