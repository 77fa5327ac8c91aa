/*
 * svtav1_hip.h -- C ABI of libsvtav1_hip.so: the MI355X (gfx950) variant of SVT-AV1-PSY's block-DSP hot path.
 *
 * Two families of entry points:
 *
 *  (1) `*_hip` functions whose prototypes are IDENTICAL to the reference's RTCD function pointers
 *      (Source/Lib/Codec/aom_dsp_rtcd.h, common_dsp_rtcd.h).  They take HOST pointers, are synchronous and
 *      re-entrant, exactly like the `_c/_avx2` variants, so `svt_hip_setup_rtcd()` can install them as
 *      "just another SIMD variant".  Each one stages its operands to the GPU, runs the same kernel the batched
 *      entry point uses with n = 1, and copies the results back.  They exist for drop-in parity, not speed.
 *
 *  (2) `svt_hip_*_batch` functions: DEVICE pointers, asynchronous on the caller's HIP stream, one launch per
 *      primitive over all superblocks / blocks of a frame (or several frames).  These are the throughput path.
 *
 * Error policy: the reference's DSP kernels return void and cannot fail.  This library never falls back to a
 * CPU implementation: a missing device or a HIP error prints the failing call to stderr and abort()s.
 *
 * All kernels are integer and bit-exact with the reference `*_c` functions cited per declaration.
 */
#ifndef SVTAV1_HIP_H
#define SVTAV1_HIP_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* PODs of this header that mirror a struct of the reference (SvtHipMv, SvtHipMvCostParams, SvtHipTxfmParam, SvtHipBuf2D, SvtHipCdefList, SvtHipSgrParams,
 * SvtHipConvolveParams) are self-contained definitions with the reference's layout, so the header needs none of the reference's.  A translation unit that HAS the
 * reference's headers (the in-encoder binding; tests/abi/abi_typecheck.c) defines SVT_HIP_REFERENCE_TYPES after including them: each of those names is then a
 * typedef of the reference's own struct and every `_hip` prototype is spelled in the reference's types -- which is how tests/abi/abi_typecheck.c proves, at
 * compile time, that each of the 193 variants of csrc/rtcd_hooks.def has its dispatch pointer's exact prototype.  tests/abi/abi_layout.c static-asserts that the
 * self-contained definitions and the reference's structs agree in size and in every field offset. */

#ifdef SVT_HIP_REFERENCE_TYPES
typedef BlockSize SvtHipBlockSize;
#else
typedef uint8_t SvtHipBlockSize; /* BlockSize (definitions.h:764-791) is a packed one-byte enum */
#endif

/* ---------------------------------------------------------------- runtime ------------------------------------ */
/* Bind the calling process to HIP device `device` (one process per GPU). 0 on success, -1 if no usable device. */
int         svt_hip_init(int device);
void        svt_hip_shutdown(void);
const char *svt_hip_device_name(void);
/* one-time costs of the calling thread's device (context, code-object loading, first allocations) paid now instead of inside the first real call */
void svt_hip_warmup(void);
/* the same with the pooled stage arenas sized by the caller: stage_arenas = how many stage-sized host forms (`svt_hip_*_host`, `*_stage*`) will be in flight at once
 * (0-3; 0 = dispatch pointers only: nothing is pre-reserved), first_arena_mb = device MB of the first one (the loop-restoration search of a 1080p plane takes ~100 MB, of a
 * 4K plane ~400 MB; the other arenas get 96 MB).  svt_hip_warmup() = svt_hip_warmup_sized(3, 192). */
void svt_hip_warmup_sized(int stage_arenas, uint32_t first_arena_mb);
/* Page-locks a host buffer the caller will hand to the host-pointer forms again and again (an encoder's picture buffers): copies from / to it then go by DMA without
 * the runtime's staging pass.  Portable across devices.  0 on success; a buffer that is already registered, or that the runtime refuses, is left as it is (non-zero) --
 * registration is an optimisation, never a requirement.  Unregister before the buffer is freed. */
int svt_hip_host_register(void *buffer, size_t bytes);
int svt_hip_host_unregister(void *buffer);
/* Several GPUs from one process (frame-level sharding): the device is selected PER HOST THREAD, like HIP's own current device.  svt_hip_set_thread_device(d)
 * binds the calling thread to device d for every later call of this library on that thread (host-call arenas and streams are per thread AND device); -1 returns the
 * thread to the default device of svt_hip_init.  Returns 0, or -1 when d is not a device.  Objects that live on a device (ME sessions) remember it and make it current
 * inside their own entry points, whatever the calling thread's selection is.
 * Device numbers are LOGICAL: device d runs on GPU d % svt_hip_physical_device_count(), and everything the library keeps per device (arenas, streams, sessions,
 * resident planes, partition peers) is keyed by the logical number.  SVT_HIP_VIRTUAL_DEVICES=V in the environment, or svt_hip_set_virtual_devices(V) before the
 * extra devices are used, gives every GPU V logical devices (default 1: logical = physical): the multi-device paths -- SVT_HIP_DEVICES picture sharding, frame
 * partitions with their peer streams, events and peer copies -- then run concurrently on a node with a single GPU. */
int svt_hip_device_count(void);          /* logical devices */
int svt_hip_physical_device_count(void); /* GPUs */
int svt_hip_physical_device(int device); /* the GPU behind a logical device, -1 if it is not one */
int svt_hip_set_virtual_devices(int per_gpu);
int svt_hip_set_thread_device(int device);
int svt_hip_get_thread_device(void);
/* The measurement knobs SVT_HIP_LR_UR / SVT_HIP_CDEF_GPW are read from the environment once, at first use; this re-reads them (tests that sweep a knob). */
void svt_hip_tuning_reload(void);
/* Overwrite the reference's RTCD pointers (weak symbols; present only when linked into libSvtAv1Enc) with the
 * `_hip` variants.  Call right after svt_aom_setup_rtcd_internal() (Source/Lib/Globals/enc_handle.c:1444-1445).
 * Returns the number of pointers installed. */
int         svt_hip_setup_rtcd(uint64_t flags);
/* Error policy (SURVEY 8b "errors": a `_hip` variant must never propagate an error; Source/Lib/Codec/aom_dsp_rtcd.c:188: the reference's kernels cannot fail).
 * The FIRST HIP error inside the library (a failed allocation, copy, launch, synchronisation, or an arena limit) switches the device path OFF for the rest of the
 * process: it is recorded (svt_hip_last_error), every dispatch pointer svt_hip_setup_rtcd overwrote is put back to the variant the reference had selected, the
 * `_hip` call in flight finishes through that saved pointer, and every stage / host-form entry point (`svt_hip_*_host`, `svt_hip_*_stage*`, the ME session) returns
 * SVT_HIP_E_DEVICE (or NULL) from then on -- the encoder's seams decline and run the reference's own function.  Nothing aborts; there is no CPU path in here.
 * The device-pointer batch entry points (callers that own streams and device buffers: tests, bench) throw through to their caller's guard or terminate. */
#define SVT_HIP_E_DEVICE (-100)
const char *svt_hip_last_error(void); /* NULL while the device path is on */
int         svt_hip_failed(void);
void        svt_hip_rtcd_unhook(void); /* puts the saved dispatch pointers back (what the first error does) */
int         svt_hip_debug_inject_failure(void); /* test instrument: behaves as if a HIP call had just failed */
/* test instrument: how often a host form issued a HIP operation AFTER it had already written caller memory (must stay 0: a failed call is finished through the
 * reference's own function, csrc/rtcd_hook.hip, so nothing may be half done when an operation fails) */
uint64_t    svt_hip_debug_commit_violations(void);
/* SVT_HIP_COUNT mode (csrc/rtcd_hook.hip): calls made so far through each installed pointer; returns the number of counted pointers */
int         svt_hip_rtcd_call_counts(const char **names, uint64_t *counts, int max);
/* Cross-lane / packed-byte instruction self-test used by the GPU test-suite (returns 0 when the silicon agrees
 * with the C model that the CPU-side interpreter in tests/emu uses). */
int         svt_hip_selftest(uint32_t *results /* device, 64*8 u32 */, void *stream);

/* VALU issue-rate probe (kind: 0 v_sad_u8, 1 v_qsad_pk_u16_u8, 2 v_add/xor pair, 3 v_mul_lo_u32+add, 4 v_mad_i64_i32,
 * 5 v_alignbyte): each of blocks*256 lanes issues iters*8 independent ops.  Used by bench.py --probe. */
/* Streams and HIP graphs.  Every svt_hip_*_batch / *_frame entry point only enqueues work on the stream it is given (no host synchronisation),
 * so a per-picture launch sequence can be captured once (begin ... calls ... end) and replayed with one launch per picture. */
void       *svt_hip_stream_create(void);
void        svt_hip_stream_destroy(void *stream);
void        svt_hip_stream_synchronize(void *stream);
void        svt_hip_graph_capture_begin(void *stream);
void       *svt_hip_graph_capture_end(void *stream);   /* returns an executable graph */
void        svt_hip_graph_launch(void *graph_exec, void *stream);
void        svt_hip_graph_destroy(void *graph_exec);
void        svt_hip_rate_probe(int kind, uint32_t iters, uint32_t blocks, uint32_t *sink, void *stream);
/* Memory-traffic probe (bench.py calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on it, per access shape): grid lane t moves `width` (4 / 8 / 16) bytes of segment
 * t / (seg / width) -- segments of `seg` bytes every `pitch` bytes from `base` -- reading (write = 0) or storing (write = 1) every byte exactly once: lanes * width bytes. */
void        svt_hip_mem_probe(int write, int width, void *base, uint64_t lanes, uint32_t seg, uint32_t pitch, uint32_t *sink, void *stream);
/* the block kernels' shape: 64x64-byte blocks tiling a picture of `pitch`-byte rows (blocks_per_row per block row, each starting `misalign` bytes into its 64-byte column),
 * 256 lanes per block reading 64 rows x 4 x 16 bytes: lanes * 16 bytes, every byte once */
void        svt_hip_mem_probe_blocks(const void *base, uint64_t lanes, uint32_t pitch, uint32_t blocks_per_row, uint32_t misalign, uint32_t *sink, void *stream);

/* ---------------------------------------------------------------- SAD family (a1-a6) -------------- */
/* a1. svt_nxm_sad_kernel -> svt_nxm_sad_kernel_helper_c (Source/Lib/C_DEFAULT/compute_sad_c.c:209, body :20-37) */
uint32_t svt_nxm_sad_kernel_hip(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                                uint32_t height, uint32_t width);
/*     svt_aom_sad_16b_kernel_c (compute_sad_c.c:39-56) */
uint32_t svt_aom_sad_16b_kernel_hip(uint16_t *src, uint32_t src_stride, uint16_t *ref, uint32_t ref_stride,
                                    uint32_t height, uint32_t width);
/*     svt_aom_sad{W}x{H}_c / x4d (compute_sad_c.c:117-131): generic-size forms; the 22 fixed-size symbols
 *     svt_aom_sadWxH_hip / svt_aom_sadWxHx4d_hip are generated from these in sad.hip. */
uint32_t svt_aom_sad_wxh_hip(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, int w, int h);
void     svt_aom_sad_wxh_x4d_hip(const uint8_t *src, int src_stride, const uint8_t *const ref_array[4], int ref_stride,
                                 uint32_t *sad_array, int w, int h);

/* a2. svt_sad_loop_kernel -> svt_sad_loop_kernel_c (compute_sad_c.c:58-100) */
void svt_sad_loop_kernel_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height,
                             uint32_t block_width, uint64_t *best_sad, int16_t *x_search_center, int16_t *y_search_center,
                             uint32_t src_stride_raw, uint8_t skip_search_line, int16_t search_area_width,
                             int16_t search_area_height);

/* a3. svt_ext_all_sad_calculation_8x8_16x16 -> _c (Source/Lib/Codec/motion_estimation.c:335-362) */
void svt_ext_all_sad_calculation_8x8_16x16_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride,
                                               uint32_t mv, uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16,
                                               uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16,
                                               uint32_t p_eight_sad16x16[16][8], uint32_t p_eight_sad8x8[64][8],
                                               bool sub_sad);
/* a4. svt_ext_eight_sad_calculation_32x32_64x64 -> _c (motion_estimation.c:369-425) */
void svt_ext_eight_sad_calculation_32x32_64x64_hip(uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32,
                                                   uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32,
                                                   uint32_t *p_best_mv64x64, uint32_t mv, uint32_t p_sad32x32[4][8]);
/* a5. svt_ext_sad_calculation_8x8_16x16 -> _c (motion_estimation.c:98-164), ..._32x32_64x64 -> _c (:171-205) */
void svt_ext_sad_calculation_8x8_16x16_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride,
                                           uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8,
                                           uint32_t *p_best_mv16x16, uint32_t mv, uint32_t *p_sad16x16,
                                           uint32_t *p_sad8x8, bool sub_sad);
void svt_ext_sad_calculation_32x32_64x64_hip(uint32_t *p_sad16x16, uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64,
                                             uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv,
                                             uint32_t *p_sad32x32);
/* a6. svt_initialize_buffer_32bits -> _c (Source/Lib/Codec/me_sad_calculation.c:14-17) */
void svt_initialize_buffer_32bits_hip(uint32_t *pointer, uint32_t count128, uint32_t count32, uint32_t value);

/* a7. svt_pme_sad_loop_kernel -> svt_pme_sad_loop_kernel_c (aom_dsp_rtcd.h:868, product_coding_loop.c:1900-1951): SAD + MV-rate search
 * of MD's predictive ME.  SvtHipMvCostParams is layout-identical to the reference's MV_COST_PARAMS (`struct svt_mv_cost_param`,
 * mcomp.h:37-48; MV = {int16 row, col}, block_structures.h:26; MV_COST_TYPE is a 1-byte enum, mcomp.h:29-36). */
#ifdef SVT_HIP_REFERENCE_TYPES
typedef MV SvtHipMv;
typedef struct svt_mv_cost_param SvtHipMvCostParams;
#else
typedef struct SvtHipMv { int16_t row, col; } SvtHipMv;
typedef struct SvtHipMvCostParams {
    const SvtHipMv *ref_mv;
    SvtHipMv        full_ref_mv;
    uint8_t         mv_cost_type; /* 0 ENTROPY, 1 L1_LOWRES, 2 L1_MIDRES, 3 L1_HDRES, 4 OPT, 5 NONE */
    const int      *mvjcost;
    const int      *mvcost[2];    /* centred tables, index range [MV_LOW, MV_UPP] */
    int             error_per_bit, early_exit_th, sad_per_bit;
} SvtHipMvCostParams;
#endif
void svt_pme_sad_loop_kernel_hip(const SvtHipMvCostParams *mv_cost_params, uint8_t *src, uint32_t src_stride, uint8_t *ref,
                                 uint32_t ref_stride, uint32_t block_height, uint32_t block_width, uint32_t *best_cost, int16_t *best_mvx,
                                 int16_t *best_mvy, int16_t search_position_start_x, int16_t search_position_start_y,
                                 int16_t search_area_width, int16_t search_area_height, int16_t search_step, int16_t mvx, int16_t mvy);

/* ---- the ME stage as a per-picture service over HOST pictures (the caller owns host buffers) ----
 * A session keeps the last `ring_planes` padded luma planes (stride * rows bytes each, picture origin at (org_x, org_y)) resident in HBM: a picture is
 * uploaded once, when it is submitted as a source, and then serves as a reference by id.  Each submission runs on its own stream (upload -> descriptor
 * build -> svt_hip_me_fullpel_search_batch over every 64x64 SB x n_refs, search centre (0, 0) -> result download), so the copies of one picture overlap
 * the search of another.  Host buffers should come from svt_hip_host_alloc (pinned) for the copies to be asynchronous.
 * submit returns a slot (>= 0) to wait on; -1: a reference (or the unsent source) is not resident, -2: n_refs > max_refs, -3: ring too small.
 * Results: best_sad_host / best_mv_host [n_refs][SBs][85], valid after svt_hip_me_session_wait(slot).
 * THREADING: a session is NOT internally synchronised -- submit / invalidate / resident / wait of one session must be serialised by the caller (the ME
 * seam holds its picture lock around them); different sessions are independent.  A re-upload after invalidate reuses the freed ring entry. */
void *svt_hip_host_alloc(size_t bytes);
void  svt_hip_host_free(void *p);
void *svt_hip_me_session_create(uint32_t width, uint32_t height, uint32_t stride, uint32_t org_x, uint32_t org_y, uint32_t rows, uint32_t ring_planes,
                                uint32_t max_refs, uint32_t max_area_width, uint32_t max_area_height, uint32_t n_slots);
/* the same on a named GPU: the session's ring, slots and launches live on `device`, and its entry points make that device current for their duration whatever
 * the calling thread has selected -- one session per GPU is how an encoder shards pictures over several devices (frame-level sharding) */
void *svt_hip_me_session_create_on(int device, uint32_t width, uint32_t height, uint32_t stride, uint32_t org_x, uint32_t org_y, uint32_t rows, uint32_t ring_planes,
                                   uint32_t max_refs, uint32_t max_area_width, uint32_t max_area_height, uint32_t n_slots);
void  svt_hip_me_session_destroy(void *session);
int   svt_hip_me_session_submit(void *session, int64_t pic_id, const uint8_t *plane_host, const int64_t *ref_ids, uint32_t n_refs, uint32_t area_w,
                                uint32_t area_h, int sub_sad, uint32_t *best_sad_host, uint32_t *best_mv_host);
int  svt_hip_me_session_wait(void *session, int slot);
/* forget a resident picture whose host content changed (the next submission naming it as the source uploads it again); is a picture resident? */
void  svt_hip_me_session_invalidate(void *session, int64_t pic_id);
int   svt_hip_me_session_resident(void *session, int64_t pic_id);

/* ---- batched forms (device pointers) ---- */
typedef struct SvtHipSadPair {
    uint64_t src_off;    /* byte offset of the block's top-left sample from src_base */
    uint64_t ref_off;    /* same, from ref_base */
    uint32_t src_stride; /* bytes */
    uint32_t ref_stride;
} SvtHipSadPair;
/* n independent WxH SADs (a1 batched; BASELINE config 1 = 64x64 over all co-located SBs). sad_out[n]. */
void svt_hip_sad_nxm_batch(const uint8_t *src_base, const uint8_t *ref_base, const SvtHipSadPair *pairs, uint32_t n,
                           uint32_t width, uint32_t height, uint32_t *sad_out, void *stream);

typedef struct SvtHipSadLoopDesc {
    uint64_t src_off, ref_off;
    uint32_t src_stride, ref_stride, src_stride_raw;
    uint16_t block_width, block_height;
    int16_t  search_area_width, search_area_height;
    uint8_t  skip_search_line, pad[3];
} SvtHipSadLoopDesc;
typedef struct SvtHipSadLoopResult {
    uint64_t best_sad;
    int16_t  x_search_center, y_search_center;
    uint32_t valid; /* 0 when the search area was empty (reference leaves x/y untouched, best_sad = 0xffffff) */
} SvtHipSadLoopResult;
/* n exhaustive searches (a2 batched: pre-HME / HME level 0-2 of every SB). `keys` = device scratch, n*8 bytes.  The maxima over the batch
 * size the position tile and the LDS window (descriptors live on the device and are not read back): widest / tallest search area, widest /
 * tallest block, and the largest ref_stride / src_stride_raw (1 = full SAD, 2 = sub-sampled HME form; ref_stride must be a multiple of
 * src_stride_raw, as it is at every reference call site). */
void svt_hip_sad_loop_batch(const uint8_t *src_base, const uint8_t *ref_base, const SvtHipSadLoopDesc *descs, uint32_t n,
                            uint32_t max_area_width, uint32_t max_area_height, uint32_t max_block_width, uint32_t max_block_height,
                            int max_ref_step, SvtHipSadLoopResult *results, uint64_t *keys, void *stream);

/* One hierarchical-ME level for a whole picture = hme_level_0 / hme_level_1 / hme_level_2 (motion_estimation.c:820-921, 923-1018, 1020-1116) for
 * every (reference, 64x64 SB, search region): search-area placement (level 0: the region's cell of the num_hme_sa_w x num_hme_sa_h grid around
 * the co-located block; levels 1-2: centred on the previous level's result), clipping to the reference picture, svt_sad_loop_kernel, result
 * scaled to the next level's resolution (x4, x2, x1).  Descriptors are built and results post-processed on the device; the search itself is
 * svt_hip_sad_loop_batch.  Items are ordered ((ref * n_sb + sb) * num_hme_sa_h + sr_h) * num_hme_sa_w + sr_w.
 * prev_sc: [items][2] int16 (x, y) = the previous level's centres (levels 1, 2; ignored at level 0).  sad_out [items]; sc_out [items][2] is in/out:
 * an empty clipped area leaves the reference's centre variable untouched, so pre-fill with what init_me_hme_data leaves there (0). */
typedef struct SvtHipHmeLevelParams {
    uint8_t  level;                  /* 0: 1/16-area planes (x, y / 4), 1: quarter planes (/ 2), 2: full resolution */
    uint8_t  sub_sampled;            /* me_ctx->hme_search_method != FULL_SAD_SEARCH: every other block row, SAD doubled */
    uint8_t  num_hme_sa_w, num_hme_sa_h;
    int16_t  sa_width, sa_height;    /* per-region area: get_hme_l0_search_area at level 0, hme_l1_sa / hme_l2_sa above */
    uint32_t sbs_x, sbs_y, n_refs;   /* n_refs <= 8 */
    uint32_t prev_shift;             /* prev_sc >> prev_shift before use: 1 when level 1 is fed with level-0 output (hme_level1_b64, :2105-2110), else 0 */
    uint32_t aligned_width, aligned_height; /* full-resolution picture size rounded up to 8 (b64_width / b64_height, :3093-3100) */
    uint64_t src_off;                /* picture sample (0, 0) of the source plane at this level's resolution, from src_base */
    uint32_t src_stride;
    uint32_t ref_stride, ref_org_x, ref_org_y, ref_width, ref_height; /* EbPictureBufferDesc of the references at this resolution */
    uint64_t ref_off[8];             /* buffer_y[0] of each reference, from ref_base */
    uint8_t  per_ref_area;           /* 1: level-0 areas differ per reference (get_hme_l0_search_area, :1800-1866): use sa_*_ref[ref] instead of sa_width / sa_height */
    uint8_t  pad1[3];
    int16_t  sa_width_ref[8], sa_height_ref[8];
    uint8_t  n_refs_list0;           /* slots [0, n_refs_list0) are list 0 (needed with do_ref / pre-HME inputs) */
    uint8_t  ref_pic_index[8];       /* index of each slot inside its list */
    uint8_t  prehme_enabled;         /* chain form: after level 0 the worst of the 2 x 2 regions is replaced by the better pre-HME result if that beats it
                                        (hme_level0_b64 :2001-2031) */
    uint8_t  pad2[2];
    uint32_t zz_skip_th;             /* me_early_exit_th >> 2 (0 = off): items whose zero-motion SAD zz_sad[ref][sb] is below it skip the level with centre (0, 0) and SAD 0 (hme_level0_b64 :1922-1935, hme_level1_b64 :2057-2070; never applied at level 2) */
    /* level-0 areas that depend on the motion list 0 / reference 0 found for the SAME superblock (get_hme_l0_search_area, :1809-1836; the low-delay settings,
     * enc_mode_config.c:702-714): for every slot but 0 the divisor of the width is (1 + ref index) when that motion is horizontal (|x| > th_max, |y| < th_min), else
     * (2 + ref index); likewise the height with "vertical".  sa_*_ref[] hold the (1 + index) areas, sa_*_ref2[] the (2 + index) ones; 0 / 0 = off.  Chain form only. */
    uint16_t l0_mv_th_min, l0_mv_th_max;
    int16_t  sa_width_ref2[8], sa_height_ref2[8];
    /* enable_me_sr_adjustment == 2 (the screen-content levels 4 / 5, enc_mode_config.c:485-505) adds: when that motion is small on both axes (|x|, |y| < 3 th_min) both
     * divisors are (4 + ref index) (:1836-1841) -- sa_*_ref4[] */
    uint8_t  l0_still_rule, pad3;
    int16_t  sa_width_ref4[8], sa_height_ref4[8];
} SvtHipHmeLevelParams;
size_t svt_hip_hme_level_workspace(const SvtHipHmeLevelParams *params);
void   svt_hip_hme_level_batch(const SvtHipHmeLevelParams *params, const uint8_t *src_base, const uint8_t *ref_base, const int16_t *prev_sc,
                               const uint32_t *zz_sad, uint64_t *sad_out, int16_t *sc_out, void *workspace, void *stream);
/* Levels 0, 1 and 2 of every item in ONE launch (one wave walks an item through the three levels: level N+1 only needs level N's result of the same
 * item).  params[3] = the three levels (same n_refs / SB grid / regions; params[1].prev_shift = 1); src_base / ref_base / sad_out / sc_out: one
 * pointer per level; results identical to three svt_hip_hme_level_batch calls.  sc_out[lv] is in/out like there. */
typedef struct SvtHipPrehmeResult { /* me_ctx->prehme_data[list][ref][sr_i] (SearchInfo) of one (reference slot, SB, search region) */
    uint64_t sad;
    int16_t  mv_x, mv_y; /* best_mv.as_mv.col / row, full-resolution units */
    uint8_t  valid, performed, pad[2];
} SvtHipPrehmeResult;
typedef struct SvtHipHmeChainInputs { /* optional device inputs of the chain, any may be NULL */
    const uint32_t           *zz_sad;  /* [ref][sb]: see zz_skip_th */
    const uint8_t            *do_ref;  /* [sb][2][4] = search_results[list][ref].do_ref: a 0 entry skips levels 0 and 1 with centre (0, 0), SAD MAX_U32 (:1963-1973, :2071-2082) */
    const SvtHipPrehmeResult *prehme;  /* [ref][sb][2]: svt_hip_prehme_batch's output, used when params[0].prehme_enabled */
    uint32_t prev_me_stage_based_exit_th; /* me_ctx->prev_me_stage_based_exit_th (0 = off): a level is skipped, keeping the previous stage's centre and SAD, when that
                                           * SAD is already small -- level 0 from the better performed pre-HME region below th >> 4 (:1937-1957), level 1 from
                                           * level 0 below th >> 5 (:2086-2096), level 2 from level 1 below th >> 2 (:2144-2154) */
    uint8_t  n_levels;     /* 3 (or 0): levels 0-2; 1: level 0 only (final centre = the best level-0 region, :2215-2262); 2: enable_hme_level2_flag = 0 (presets M7 and above, enc_mode_config.c:1636-1640): levels 0 and 1 only,
                            * params[2] / sad_out[2] / sc_out[2] are not touched and the final centre is taken from level 1 (:2267-2309) */
    uint8_t  list1_no_hme; /* temporal_layer_index == 0: list 1's references take no part in HME (:1983, :2055, :2127): their items are skipped */
    uint8_t  pad[2];
} SvtHipHmeChainInputs;
void   svt_hip_hme_chain_batch(const SvtHipHmeLevelParams *params, const uint8_t *const *src_base, const uint8_t *const *ref_base,
                               const SvtHipHmeChainInputs *inputs, uint64_t *const *sad_out, int16_t *const *sc_out, void *stream);
/* What precedes the HME levels in hme_b64 (motion_estimation.c:2441-2450), for every SB of a picture:
 * svt_hip_me_ref_gate_batch = the second half of init_zz_sad (:2402-2417): when the best zero-motion SAD of the SB is below zz_sad_th, references other than the
 * first of each list whose zz_sad is zz_sad_pct % above it are dropped (temporal layers > 0 only).  do_ref [sb][2][4] in/out.
 * svt_hip_prehme_batch = prehme_b64 (:1722-1798): two long thin search regions (vertical, horizontal) on the 1/16-area planes per reference through
 * prehme_core (:1568-1666), with check_prehme_early_exit (zero-motion SAD below me_early_exit_th; list-1 shortcut from the list-0 result of the same
 * index), then reference pruning on the pre-HME SADs (:1781-1797).  out [ref][sb][2]; do_ref in/out. */
typedef struct SvtHipPrehmeParams {
    SvtHipHmeLevelParams plane;       /* level = 0 plane geometry, SB grid, n_refs, n_refs_list0, ref_pic_index, sub_sampled (areas / regions unused) */
    uint16_t sa_min_width[2], sa_min_height[2], sa_max_width[2], sa_max_height[2]; /* prehme_ctrl.prehme_sa_cfg[sr_i] */
    uint16_t hme_sr_factor[8];        /* per slot: svt_aom_get_scaled_picture_distance(picture distance) */
    uint8_t  skip_search_line, l1_early_exit, temporal_layer_gt0, pad;
    uint32_t me_early_exit_th;        /* 0 = off */
    uint32_t phme_sad_th; uint16_t phme_sad_pct; uint16_t pad2; /* me_hme_prune_ctrls */
} SvtHipPrehmeParams;
void   svt_hip_me_ref_gate_batch(const SvtHipHmeLevelParams *plane, const uint32_t *zz_sad, uint32_t zz_sad_th, uint32_t zz_sad_pct,
                                 int temporal_layer_gt0, uint8_t *do_ref, void *stream);
/* me_safe_limit_zz_th (init_zz_sad :2416-2436): where zz_sad of both lists' nearest references is below the threshold, every reference with ref_pic_index > 0
 * is dropped.  The picture-level conditions (hierarchical_levels > 0, top temporal layer, similar_brightness_refs) are the caller's: pass 0 when they fail. */
void   svt_hip_me_ref_safe_limit_batch(const SvtHipHmeLevelParams *plane, const uint32_t *zz_sad, uint32_t safe_limit_zz_th, uint8_t *do_ref, void *stream);
void   svt_hip_prehme_batch(const SvtHipPrehmeParams *params, const uint8_t *src_base, const uint8_t *ref_base, const uint32_t *zz_sad, uint8_t *do_ref,
                            SvtHipPrehmeResult *out, void *stream);

/* Integer ME of a whole picture from its HME results = set_final_seach_centre_sb (motion_estimation.c:2182-2368: per (reference, SB) the first
 * strictly smallest SAD over the search regions) + integer_search_b64's search-area geometry (:1294-1325, :1458-1508: min(sa_min * dist, sa_max),
 * enlargement for long search-centre components, division by reduce_me_sr_divisor, width rounded up to 8, centred on the HME result, clipped to the
 * picture + 63-sample border) + svt_hip_me_fullpel_search_batch, including hme_prune_ref_and_adjust_sr, the zero-motion early exit, check_00_center
 * and the 8x8-variance probe.  With me_sr_adjustment = 2 the second rule makes the references of an SB depend on the first one's final SADs: two passes.
 * hme_sad / hme_sc: the last HME level's outputs in svt_hip_hme_level_batch's item order ((ref * n_sb + sb) * regions + region).
 * do_ref ([n_sb][2][4] = search_results[list][ref].do_ref as svt_hip_me_results_batch takes it, or NULL = all; in/out: HME-based pruning clears entries): a 0 entry gets a 1 x 1 placeholder search whose results must
 * be ignored (the reference skips the reference picture, :1292-1293).  divisor ([n_sb][n_refs] uint32, or NULL = 1) = me_ctx->reduce_me_sr_divisor
 * as an input when sr_adjustment = 0; with sr_adjustment = 1 it is derived from the HME results (and written back when non-NULL).
 * zz_sad ([ref][sb], needed when me_early_exit_th != 0): svt_hip_me_zz_sad_batch.
 * Outputs: best_sad / best_mv [ref][sb][85] (as svt_hip_me_fullpel_search_batch), sc_out [ref][sb][2] + sad_out [ref][sb] = search_results[].hme_sc_x/y,
 * hme_sad. */
typedef struct SvtHipMeIntegerSearchParams {
    uint32_t sbs_x, sbs_y, n_refs;            /* reference slots (list 0 first), <= 8 */
    uint32_t regions;                         /* num_hme_sa_w * num_hme_sa_h entries per (ref, SB) in hme_sad / hme_sc */
    uint32_t aligned_width, aligned_height;   /* pcs->aligned_width / aligned_height */
    int16_t  sa_min_width, sa_min_height, sa_max_width, sa_max_height; /* me_ctx->me_sa */
    uint8_t  sub_sad;                         /* me_search_method == SUB_SAD_SEARCH */
    uint8_t  mv_adj_enabled, mv_adj_nearest_ref_only; /* me_ctx->mv_based_sa_adj */
    uint8_t  list1_no_hme;                    /* temporal_layer_index == 0 with two lists: set_final_seach_centre_sb (:2211, :2362-2373) gives list 1's slots the centre (0, 0)
                                               * and -- the variable is not reset -- the HME SAD of the last list-0 slot; their zz_sad stays ~0 (init_zz_sad :2390) */
    uint16_t mv_adj_mv_size_th, mv_adj_sa_multiplier;
    uint16_t dist[8];                         /* per slot: picture distance, through svt_aom_get_scaled_picture_distance unless ME_MCTF (:1300-1302) */
    uint8_t  ref_pic_index[8];                /* per slot: index inside its list (nearest_ref_only) */
    uint64_t src_off;                         /* picture sample (0, 0) of the source plane, from src_base */
    uint32_t src_stride;
    uint32_t ref_stride, ref_org_x, ref_org_y;
    uint64_t ref_off[8];                      /* buffer_y[0] of each reference, from ref_base */
    /* hme_prune_ref_and_adjust_sr (:2477-2518), between the final centres and the search: */
    uint8_t  n_refs_list0;                    /* slots [0, n_refs_list0) are list 0 */
    uint8_t  hme_prune_enabled;               /* me_hme_prune_ctrls.enable_me_hme_ref_pruning with a threshold != (uint16_t)~0 */
    uint16_t prune_ref_if_hme_sad_dev_bigger_than_th;
    uint8_t  sr_adjustment;                   /* me_sr_adjustment_ctrls.enable_me_sr_adjustment: 0, 1 or 2.  2 (screen-content levels 4 / 5) adds two rules to integer_search_b64
                                               * (:1349-1364, only without me_early_exit_th): the height halves when the HME result is already good (check_00_center's SAD of an
                                               * accurate centre, or -- is_ref -- the HME SAD, below 24 * 24); else both sides halve for every slot but the first when the first
                                               * slot's final 64x64 SAD of the same SB is below 5000 -- the first slot is therefore searched before the geometry of the others */
    uint8_t  pad2;
    uint16_t reduce_me_sr_based_on_mv_length_th, stationary_hme_sad_abs_th, stationary_me_sr_divisor, reduce_me_sr_based_on_hme_sad_abs_th,
             me_sr_divisor_for_low_hme_sad;
    uint32_t me_early_exit_th;                /* 0 = off; else zz_sad < th / 6 searches a single point (:1322-1327) */
    /* content-dependent steps of integer_search_b64 (extra per-item SADs / a single-point pre-search on the device): */
    uint8_t  is_ref;                          /* me_ctx->is_ref: with me_early_exit_th = 0, a non-zero search centre is checked against (0, 0) on the
                                                 sub-sampled 64x64 SAD (check_00_center, :1139-1206), after being clipped to the picture + 63 */
    uint8_t  me_8x8_var_enabled;              /* me_8x8_var_ctrls: areas > 24 positions first search the centre alone; the variance of its 64 8x8 SADs
                                                 scales the area (:1388-1420); the centre's results take part in the final minimum */
    uint8_t  pad3[2];
    uint32_t me_sr_div4_th, me_sr_div2_th, me_sr_mult2_th;
    uint32_t ref_width, ref_height;           /* EbPictureBufferDesc width / height of the references (check_00_center clips against them) */
    uint32_t tf_me_exit_th;                   /* ME_MCTF (svt_aom_motion_estimation_b64 :3109-3113): 0 = off; an SB whose first reference's HME SAD is below it is not
                                               * searched at all (its tables keep their initial content; the temporal filter then uses the 64x64 prediction only) */
    uint32_t pad4;
} SvtHipMeIntegerSearchParams;
size_t svt_hip_me_integer_search_workspace(const SvtHipMeIntegerSearchParams *params);
void   svt_hip_me_integer_search_batch(const SvtHipMeIntegerSearchParams *params, const uint8_t *src_base, const uint8_t *ref_base,
                                       const uint64_t *hme_sad, const int16_t *hme_sc, uint8_t *do_ref, uint32_t *divisor, const uint32_t *zz_sad,
                                       uint32_t *best_sad, uint32_t *best_mv, int16_t *sc_out, uint64_t *sad_out, void *workspace, void *stream);
/* init_zz_sad's per-(reference, SB) value (motion_estimation.c:2382-2400, get_zz_sad :1667-1689): SAD of the 64x64 block against the co-located
 * reference block on every other row, doubled, normalised by 4096 / (b64 width x height).  Geometry fields of params only.  zz_out [ref][sb]. */
void   svt_hip_me_zz_sad_batch(const SvtHipMeIntegerSearchParams *params, const uint8_t *src_base, const uint8_t *ref_base, uint32_t *zz_out,
                               void *stream);

/* Frame-batched integer full-pel search = open_loop_me_fullpel_search_sblock (motion_estimation.c:781-816), i.e.
 * a3+a4+a5+a6 fused: for every (64x64 SB, reference) item, all 85 block SADs (8x8..64x64) at every position of the
 * search area, keeping the first minimum in raster order (strict `<`), bests initialised to MAX_SAD_VALUE
 * (motion_estimation.h:85).  Output layout per item = p_sb_best_sad / p_sb_best_mv[85]: 64x64 @0, 32x32 @1-4,
 * 16x16 @5-20, 8x8 @21-84 (me_context.h:52-138); mv = (y << 16) | (uint16_t)x (motion_estimation.c:448-450). */
typedef struct SvtHipMeSearchDesc {
    uint64_t src_off;              /* me_ctx->b64_src_ptr relative to src_base */
    uint64_t ref_off;              /* top-left sample of the search area relative to ref_base */
    uint32_t src_stride;           /* me_ctx->b64_src_stride */
    uint32_t ref_stride;           /* interpolated_full_stride[list][ref] */
    int16_t  x_search_area_origin; /* added to the x index to form the MV */
    int16_t  y_search_area_origin;
    uint16_t search_area_width;
    uint16_t search_area_height;
} SvtHipMeSearchDesc;
#define SVT_HIP_ME_NUM_BLOCKS 85
/* bytes of device scratch needed for n items whose largest search area is max_w x max_h (0 for areas <= 64x32) */
size_t svt_hip_me_fullpel_search_workspace(uint32_t n, uint32_t max_w, uint32_t max_h);
void   svt_hip_me_fullpel_search_batch(const uint8_t *src_base, const uint8_t *ref_base, const SvtHipMeSearchDesc *descs,
                                       uint32_t n, uint32_t max_w, uint32_t max_h, int sub_sad, uint32_t *best_sad,
                                       uint32_t *best_mv, void *workspace, void *stream);

/* ---- ME result formatting: what svt_aom_motion_estimation_b64 does with the integer-search tables after the search,
 * motion_estimation.c:3121-3152 -- me_prune_ref (:1522-1566), construct_me_candidate_array{,_mrp_off,_single_ref} (:2532-2828),
 * compute_distortion (:2964-3008), perform_gm_detection (:2833-2961) -- for every 64x64 SB of a picture in one launch.
 * Inputs (device): best_sad / best_mv [ref slot][n_sb][85] as written by svt_hip_me_fullpel_search_batch with items ordered
 * slot * n_sb + sb, slot = (list ? num_of_ref_pic_to_search[0] : 0) + ref (a slot the search never wrote must hold 0 MVs, as
 * init_me_hme_data leaves them, :3049-3052); do_ref [n_sb][2][4] = search_results[list][ref].do_ref after HME pruning, updated in place;
 * sb_size [n_sb][2] = B64Geom width, height.
 * Outputs (device), addressed exactly as MeSbResults (me_sb_results.h:28-53) with n_pus = 85 / 21 / 5 (pcs.c:107-111):
 * total_me_candidate_index [n_sb][n_pus], me_mv_array [n_sb][n_pus * max_refs] (MvCandidate.as_int), me_candidate_array
 * [n_sb][n_pus * max_cand] (MeCandidate bit-field bytes).  Entries the reference does not write are left as the caller filled them. */
typedef struct SvtHipMeResultsParams {
    uint32_t n_sb;
    uint8_t  num_of_list_to_search;         /* MeContext */
    uint8_t  num_of_ref_pic_to_search[2];
    uint8_t  max_cand, max_refs, max_l0;    /* MotionEstimationData (pcs.h:500-502; svt_aom_get_max_allocated_me_refs, pcs.c:91-96) */
    uint8_t  enable_me_16x16, enable_me_8x8; /* PictureParentControlSet */
    uint8_t  only_l_bwd;                    /* scs->mrp_ctrls.only_l_bwd */
    uint8_t  use_best_unipred_cand_only;    /* MeContext */
    uint8_t  prune_ref;                     /* prune_ref && me_hme_prune_ctrls.enable_me_hme_ref_pruning (:3122) */
    uint8_t  low_resolution;                /* scs->input_resolution <= INPUT_SIZE_480p_RANGE */
    uint8_t  gm_enabled, gm_use_distance_based_active_th; /* pcs->gm_ctrls */
    uint16_t prune_ref_if_me_sad_dev_bigger_than_th;
    int32_t  prune_me_candidates_th;
    uint64_t picture_number;
    uint64_t ref_picture_number[2][4];      /* me_ds_ref_array[list][ref].picture_number */
} SvtHipMeResultsParams;
typedef struct SvtHipMeSbStats { /* the per-SB entries of the pcs arrays written by compute_distortion / perform_gm_detection */
    uint32_t me_64x64_distortion, me_32x32_distortion, me_16x16_distortion, me_8x8_distortion, me_8x8_cost_variance, rc_me_distortion;
    uint8_t  stationary_block_present_sb, rc_me_allow_gm, pad[2];
} SvtHipMeSbStats;
void svt_hip_me_results_batch(const SvtHipMeResultsParams *params, const uint32_t *best_sad, const uint32_t *best_mv, uint8_t *do_ref,
                              const uint8_t *sb_size, uint8_t *total_me_candidate_index, uint32_t *me_mv_array, uint8_t *me_candidate_array,
                              SvtHipMeSbStats *sb_stats, void *stream);
/* The session form: search as svt_hip_me_session_submit, then format on the device and download the final product instead of (or besides) the raw
 * tables.  params->n_sb is filled in by the session; num_of_ref_pic_to_search[0] + [1] must equal n_refs (list 0 references first in ref_ids),
 * else -4.  Host buffers (pinned for asynchronous copies): sizes as the device arrays above with n_pus from enable_me_16x16 / enable_me_8x8;
 * do_ref may be NULL (every reference allowed, nothing returned); best_sad / best_mv may be NULL.  Entries the reference leaves unwritten are 0. */
typedef struct SvtHipMeResultsHost {
    uint8_t         *do_ref;                   /* [n_sb][2][4] in/out, or NULL */
    uint8_t         *total_me_candidate_index; /* [n_sb][n_pus] */
    uint32_t        *me_mv_array;              /* [n_sb][n_pus * max_refs] */
    uint8_t         *me_candidate_array;       /* [n_sb][n_pus * max_cand] */
    SvtHipMeSbStats *sb_stats;                 /* [n_sb] */
    uint32_t        *best_sad, *best_mv;       /* [n_refs][n_sb][85], or NULL */
    int16_t         *hme_sc;                   /* stage form: [n_refs][n_sb][2] = search_results[list][ref].hme_sc_x / hme_sc_y, or NULL */
    uint64_t        *hme_sad;                  /* stage form: [n_refs][n_sb] = search_results[list][ref].hme_sad, or NULL */
} SvtHipMeResultsHost;
int svt_hip_me_session_submit_results(void *session, int64_t pic_id, const uint8_t *plane_host, const int64_t *ref_ids, uint32_t n_refs,
                                      uint32_t area_w, uint32_t area_h, int sub_sad, const SvtHipMeResultsParams *params,
                                      const SvtHipMeResultsHost *out);

/* ---- temporal filter pixel kernels (the DSP part of Codec/temporal_filtering.c; fixed point) ----
 * The reference's RTCD pointers take a `struct MeContext *` (aom_dsp_rtcd.h:797-826); its layout is private to the encoder, so the `_hip`
 * functions take the handful of fields the kernels read as two plain structs instead (INTEGRATION.md shows the 12-line adapter that fills
 * them from a MeContext).  Everything else of each prototype is the reference's. */
typedef struct SvtHipTfParams { /* per picture */
    uint32_t tf_decay_factor_fp16[3]; /* me_ctx->tf_decay_factor_fp16[C_Y, C_U, C_V] (me_context.h:372) */
    uint16_t tf_mv_dist_th;           /* me_context.h:490 */
    uint8_t  tf_chroma;               /* :469 */
    uint8_t  use_zz_based_filter;     /* tf_ctrls.use_zz_based_filter (frame form only: selects the kernel) */
    uint8_t  encoder_bit_depth;       /* frame form only; the symbols below take it as an argument like the reference */
    uint8_t  ss_x, ss_y;              /* frame form only */
    uint8_t  pad;
} SvtHipTfParams;
typedef struct SvtHipTfBlock { /* per 32x32 luma block (idx_32x32 = tf_block_col + 2 * tf_block_row) and reference picture */
    uint64_t block_error[4]; /* split: tf_16x16_block_error[4 idx + i]; else [0] = tf_32x32_block_error[idx] (me_context.h:477,485) */
    int16_t  mv_x[4];        /* split: tf_16x16_mv_x[4 idx + i]; else [0] = tf_32x32_mv_x[idx] */
    int16_t  mv_y[4];
    uint8_t  split;          /* tf_32x32_block_split_flag[idx] */
    uint8_t  pad[7];
} SvtHipTfBlock;
/* svt_av1_apply_temporal_filter_planewise_medium{,_hbd} -> _c (temporal_filtering.c:1145-1196, 1332-1400), aom_dsp_rtcd.h:813-826 */
void svt_av1_apply_temporal_filter_planewise_medium_hip(const SvtHipTfParams *params, const SvtHipTfBlock *block, const uint8_t *y_src,
                                                        int y_src_stride, const uint8_t *y_pre, int y_pre_stride, const uint8_t *u_src,
                                                        const uint8_t *v_src, int uv_src_stride, const uint8_t *u_pre, const uint8_t *v_pre,
                                                        int uv_pre_stride, unsigned int block_width, unsigned int block_height, int ss_x, int ss_y,
                                                        uint32_t *y_accum, uint16_t *y_count, uint32_t *u_accum, uint16_t *u_count, uint32_t *v_accum,
                                                        uint16_t *v_count);
void svt_av1_apply_temporal_filter_planewise_medium_hbd_hip(const SvtHipTfParams *params, const SvtHipTfBlock *block, const uint16_t *y_src,
                                                            int y_src_stride, const uint16_t *y_pre, int y_pre_stride, const uint16_t *u_src,
                                                            const uint16_t *v_src, int uv_src_stride, const uint16_t *u_pre, const uint16_t *v_pre,
                                                            int uv_pre_stride, unsigned int block_width, unsigned int block_height, int ss_x,
                                                            int ss_y, uint32_t *y_accum, uint16_t *y_count, uint32_t *u_accum, uint16_t *u_count,
                                                            uint32_t *v_accum, uint16_t *v_count, uint32_t encoder_bit_depth);
/* svt_av1_apply_zz_based_temporal_filter_planewise_medium{,_hbd} -> _c (temporal_filtering.c:867-903, 970-1013), aom_dsp_rtcd.h:798-811 */
void svt_av1_apply_zz_based_temporal_filter_planewise_medium_hip(const SvtHipTfParams *params, const SvtHipTfBlock *block, const uint8_t *y_pre,
                                                                 int y_pre_stride, const uint8_t *u_pre, const uint8_t *v_pre, int uv_pre_stride,
                                                                 unsigned int block_width, unsigned int block_height, int ss_x, int ss_y,
                                                                 uint32_t *y_accum, uint16_t *y_count, uint32_t *u_accum, uint16_t *u_count,
                                                                 uint32_t *v_accum, uint16_t *v_count);
void svt_av1_apply_zz_based_temporal_filter_planewise_medium_hbd_hip(const SvtHipTfParams *params, const SvtHipTfBlock *block,
                                                                     const uint16_t *y_pre, int y_pre_stride, const uint16_t *u_pre,
                                                                     const uint16_t *v_pre, int uv_pre_stride, unsigned int block_width,
                                                                     unsigned int block_height, int ss_x, int ss_y, uint32_t *y_accum,
                                                                     uint16_t *y_count, uint32_t *u_accum, uint16_t *u_count, uint32_t *v_accum,
                                                                     uint16_t *v_count, uint32_t encoder_bit_depth);
/* svt_estimate_noise_fp16 / svt_estimate_noise_highbd_fp16 -> _c (temporal_filtering.c:3847-3920), aom_dsp_rtcd.h:876-879; RTCD-hooked */
int32_t svt_estimate_noise_fp16_hip(const uint8_t *src, uint16_t width, uint16_t height, uint16_t stride_y);
int32_t svt_estimate_noise_highbd_fp16_hip(const uint16_t *src, int width, int height, int stride, int bd);
/* device form: plane in HBM (stride in samples), result written to noise_out[0]; workspace: svt_hip_estimate_noise_workspace() bytes, any content */
size_t svt_hip_estimate_noise_workspace(uint32_t width, uint32_t height);
void svt_hip_estimate_noise_batch(const void *plane, uint32_t width, uint32_t height, uint32_t stride, int bit_depth, int32_t *noise_out,
                                  void *workspace, void *stream);
/* Whole-picture temporal filtering given the motion-compensated predictions: for every 32x32 luma block (and its chroma blocks) the
 * accumulators start from the central picture with weight 1000 (svt_aom_apply_filtering_central{,_highbd}_c, :350-425), each reference adds
 * its plane-wise term (the functions above, chosen by use_zz_based_filter / encoder_bit_depth), and (accum + count / 2) / count is written
 * out (svt_aom_get_final_filtered_pixels_c, :2608-2672) -- produce_temporally_filtered_pic's steps 2-3 (:3360-3400) in one launch, accum and
 * count never leaving registers.  Planes are device pointers, strides in samples; out may alias central.  blocks: device,
 * [n_refs][nby][nbx]; nbx x nby 32x32 blocks are processed (the planes must cover them, as the reference's padded pictures do). */
#define SVT_HIP_TF_MAX_REFS 12   /* reference frames one launch of the frame kernel takes */
#define SVT_HIP_TF_MAX_FRAMES 32 /* = ALTREF_MAX_NFRAMES - 1 (definitions.h:304): every frame count the reference's temporal filter can use (chunked form, picture stage) */
typedef struct SvtHipTfPlanes {
    void    *y, *u, *v;
    uint32_t y_stride, uv_stride;
} SvtHipTfPlanes;
void svt_hip_tf_filter_frame(const SvtHipTfParams *params, const SvtHipTfPlanes *central, const SvtHipTfPlanes *preds, uint32_t n_refs,
                             const SvtHipTfBlock *blocks, uint32_t nbx, uint32_t nby, const SvtHipTfPlanes *out, void *stream);
/* The same for up to SVT_HIP_TF_MAX_FRAMES reference frames: chunks of SVT_HIP_TF_MAX_REFS, the accumulators of the frames so far travelling between the launches through
 * `workspace` (svt_hip_tf_filter_frame_workspace bytes, any content); only the last launch writes pixels, so out may still alias central. */
size_t svt_hip_tf_filter_frame_workspace(const SvtHipTfParams *params, uint32_t nbx, uint32_t nby);
void   svt_hip_tf_filter_frame_chunked(const SvtHipTfParams *params, const SvtHipTfPlanes *central, const SvtHipTfPlanes *preds, uint32_t n_refs,
                                       const SvtHipTfBlock *blocks, uint32_t nbx, uint32_t nby, const SvtHipTfPlanes *out, void *workspace, void *stream);

/* The temporal filter's sub-pel motion refinement, batched: replaces tf_subpel_search + svt_check_position (temporal_filtering.c:1560-1790) as
 * tf_64x64_sub_pel_search / tf_32x32_ / tf_16x16_ / tf_8x8_sub_pel_search (:1793-2250) call it for each block of a (central picture, reference
 * picture) pair.  Per block: the centre (predicted on the sub-sampled rows), then the half-, quarter- and eighth-pel rings around the running
 * best MV; each candidate = luma motion compensation (svt_aom_simple_luma_unipred: MV clamped to picture + border, EIGHTTAP_REGULAR or
 * bilinear kernels, the svt_av1_[highbd_]convolve_*_sr_c roundings) + svt_aom_mefn_ptr[bsize].vf / vf_hbd_10 against the source on every
 * (1 << subsampling_shift)-th row, << subsampling_shift; the reference's skip rules (mode >= 2: no diagonals; best == 0; early_exit_th) apply
 * in its order (x offset outer loop).  Blocks start from distortion INT_MAX like the callers (:1866, :1980, :2117).  All pointers device;
 * samples u8 (bit_depth 8) or u16 (10).  One wave per block; the result is bit-exact with the reference's (dist, mv_x, mv_y). */
typedef struct SvtHipTfSubpelParams {
    uint8_t  half_pel_mode, quarter_pel_mode, eight_pel_mode; /* pcs->tf_ctrls.*: 0 = off, 1 = all 8 neighbours, >= 2 = no diagonals */
    uint8_t  subsampling_shift;                               /* pcs->tf_ctrls.sub_sampling_shift */
    uint8_t  bit_depth;                                       /* 8 or 10 */
    uint8_t  pad[3];
    uint32_t early_exit_th;                                   /* me_ctx->tf_subpel_early_exit_th (0 = off) */
    uint32_t mi_rows, mi_cols;                                /* pcs->av1_cm->mi_rows / mi_cols (MV clamp) */
    uint32_t ref_org_x, ref_org_y, ref_stride;                /* reference pictures: padding origin and stride (samples) */
} SvtHipTfSubpelParams;
typedef struct SvtHipTfSubpelDesc {
    uint64_t src_off;    /* samples from src_base to the block's top-left source sample */
    uint64_t ref_off;    /* samples from ref_base to this block's reference picture's buffer_y (the padded plane's first sample) */
    uint32_t src_stride; /* samples */
    uint16_t pu_x, pu_y; /* block origin in the picture (luma samples) */
    uint8_t  bsize;      /* 8, 16, 32 or 64 (square) */
    uint8_t  bilinear;   /* me_ctx->tf_ctrls.use_2tap for the 64x64 / 32x32 searches (:1801-1804, :1911-1914): BILINEAR; 16x16 and 8x8 always EIGHTTAP_REGULAR */
    int16_t  mv_x, mv_y; /* starting MV, 1/8 pel (the integer ME vector << 3) */
    uint16_t pad;
} SvtHipTfSubpelDesc;
typedef struct SvtHipTfSubpelResult {
    uint64_t dist;       /* best distortion (tf_*_block_error) */
    int16_t  mv_x, mv_y; /* best MV, 1/8 pel */
    uint32_t pad;
} SvtHipTfSubpelResult;
void svt_hip_tf_subpel_search_batch(const SvtHipTfSubpelParams *params, const void *src_base, const void *ref_base, const SvtHipTfSubpelDesc *descs,
                                    uint32_t n, SvtHipTfSubpelResult *results, void *stream);
/* The same from HOST memory (a seam around tf_subpel_search, temporal_filtering.c:1670, calls it once per (central picture, reference picture) pair for every
 * block the reference may ask for): src_buf / ref_buf = the two pictures' whole padded luma buffers, the descs' offsets relative to them.  Synchronous. */
int svt_hip_tf_subpel_search_host(const SvtHipTfSubpelParams *params, const void *src_buf, size_t src_samples, const void *ref_buf, size_t ref_samples,
                                   const SvtHipTfSubpelDesc *descs, uint32_t n, SvtHipTfSubpelResult *results);

/* The temporal filter's final motion compensation, batched: replaces tf_64x64_inter_prediction / tf_32x32_ / tf_16x16_ / tf_8x8_inter_prediction
 * (temporal_filtering.c:2256-2620) for the blocks of a (central picture, reference picture) pair -- svt_aom_inter_prediction's uni-directional
 * SIMPLE_TRANSLATION path (enc_inter_prediction.c:4102) with MULTITAP_SHARP kernels (the 4-tap regular kernel for a chroma dimension <= 4), the MV
 * clamped per plane, luma + (chroma != 0: me_ctx->tf_chroma) both 4:2:0 chroma blocks at ((pu >> 3) << 3) / 2.  The prediction of a block lands at the
 * block's position in picture-sized planes -- the `preds` svt_hip_tf_filter_frame reads.  params: bit_depth, mi_rows / mi_cols, ref_org_x / ref_org_y (the
 * LUMA padding origin of the reference pictures; chroma: half); the strides come from `planes`.  All pointers device. */
typedef struct SvtHipTfMcPlanes {
    const void *ref[3];  /* base of the reference pictures' Y / U / V buffers */
    void       *pred[3]; /* base of the prediction planes */
    uint32_t    ref_stride[3], pred_stride[3]; /* samples */
} SvtHipTfMcPlanes;
typedef struct SvtHipTfMcDesc {
    uint64_t ref_off[3];  /* samples from ref[pl] to this block's reference picture's padded buffer of plane pl */
    uint64_t pred_off[3]; /* samples from pred[pl] to sample (0, 0) of this block's prediction plane pl */
    uint16_t pu_x, pu_y;  /* block origin (luma samples) */
    uint8_t  bsize, pad;  /* 8, 16, 32 or 64 */
    int16_t  mv_x, mv_y;  /* 1/8 pel (me_ctx->tf_*_mv_x / _y) */
    uint16_t pad2[3];
} SvtHipTfMcDesc;
void svt_hip_tf_inter_pred_batch(const SvtHipTfSubpelParams *params, const SvtHipTfMcPlanes *planes, const SvtHipTfMcDesc *descs, uint32_t n, int chroma, void *stream);
/* the same over a list whose length another kernel left in device memory (n_dev[0] <= max_n): the waves walk it with a grid stride, no host round trip for the count */
void svt_hip_tf_inter_pred_list(const SvtHipTfSubpelParams *params, const SvtHipTfMcPlanes *planes, const SvtHipTfMcDesc *descs, uint32_t max_n, const uint32_t *n_dev,
                                int chroma, void *stream);

/* The temporal filter of ONE central picture as a single device stage: produce_temporally_filtered_pic's per-block loop (temporal_filtering.c:3037-3400) for every
 * 64x64 block and every reference picture the caller kept (the picture-level skips of :3105-3131 -- ahd error, brightness change -- are the caller's), given the
 * ME results of each (central, reference) pair:
 *   1. the sub-pel refinements of every block size the reference may search (tf_64x64_ / tf_32x32_ / tf_16x16_ / tf_8x8_sub_pel_search, :1793-2250), batched;
 *   2. per (block, reference) the reference's decision tree on those results: 64x64 only when the ME exited early (motion_estimation.c:3110) or
 *      tf_use_64x64_pred says so (:2676, :3188-3190); else 64x64 when its error beats the four 32x32 (:3263-3270); else per 32x32 block one prediction
 *      (error below pred_error_32x32_th, :3292) or the 16x16 / 8x8 split of derive_tf_32x32_block_split_flag (:237-286);
 *   3. the final motion compensation of the chosen blocks (svt_hip_tf_inter_pred_batch) into picture-sized prediction planes;
 *   4. the 32x32 block errors of the 64x64 predictions (convert_64x64_info_to_32x32_info, :2691-2758);
 *   5. svt_hip_tf_filter_frame.
 * Host form: whole padded buffers in, the filtered 64x64 blocks written to out_* (which may be the central picture's own buffers, like the reference's in-place
 * result).  All pictures share one geometry (sp.ref_org_x / ref_org_y / ref_stride for luma, uv_stride and the halved origin for chroma; 4:2:0).
 * High bit depth (sp.bit_depth 10): the planes are the packed 16-bit pictures the reference filters (altref_buffer_highbd); with subpel_8bit the searches of step 1 read the 8-bit
 * luma of the same pictures instead, as the reference does with tf_ctrls.use_8bit_subpel.
 * Returns 0, or -1 for parameters outside what is built (n_refs > SVT_HIP_TF_MAX_FRAMES, n_refs == 0, not 4:2:0). */
typedef struct SvtHipTfPictureParams {
    SvtHipTfSubpelParams sp;  /* sub-pel controls, bit depth, mi_rows / mi_cols, the LUMA padding origin and stride of every picture */
    SvtHipTfParams       tf;  /* the filter's MeContext fields (frame form: all of them) */
    uint32_t pic_w_sb, pic_h_sb;      /* 64x64 blocks per row / column: blk_cols, blk_rows (:2823-2826); the ME tables hold pic_w_sb * pic_h_sb entries */
    uint32_t uv_stride;               /* chroma stride (samples) */
    uint32_t me_exit_th;              /* tf_ctrls.me_exit_th */
    uint64_t pred_error_32x32_th;     /* tf_ctrls.pred_error_32x32_th */
    uint8_t  use_2tap;                /* tf_ctrls.use_2tap: bilinear 64x64 / 32x32 searches */
    uint8_t  enable_8x8_pred;         /* tf_ctrls.enable_8x8_pred */
    uint8_t  use_pred_64x64_only_th;  /* tf_ctrls.use_pred_64x64_only_th */
    uint8_t  subpel_8bit;             /* high bit depth only: tf_ctrls.use_8bit_subpel -- the sub-pel searches run on the pictures' 8-bit luma (`y8`), everything else on the 16-bit planes (:3203, :3242) */
    uint8_t  zero_motion;             /* the low-delay form, produce_temporally_filtered_pic_ld (:3415-3846): no ME and no refinement, every block is predicted 64x64 at vector (0, 0)
                                       * (:3711-3726); the ME tables are not read (me may be NULL) */
    uint8_t  pad[3];
} SvtHipTfPictureParams;
typedef struct SvtHipTfHostPicture {
    const void *y, *u, *v;            /* the padded planes' first samples: buffer_y / buffer_cb / buffer_cr (8 bit), altref_buffer_highbd[C_Y / C_U / C_V] (sp.bit_depth 10: uint16) */
    size_t      y_samples, uv_samples;/* samples per plane (luma_size / chroma_size) */
    const void *y8;                   /* subpel_8bit: the picture's 8-bit luma buffer (buffer_y of the 10-bit picture: its 8 MSBs), same geometry; else NULL */
} SvtHipTfHostPicture;
typedef struct SvtHipTfMeTables { /* of one (central, reference) pair, [sb] = 64x64 block in raster order; the 85 entries in the ME order (64, 4 x 32, 16 x 16 z-order, 64 x 8 z-order) */
    const uint32_t *best_sad, *best_mv; /* [n_sb][85]: p_best_sad_* / p_best_mv* ((y << 16) | x, full pel) */
    const int16_t  *hme_sc;             /* [n_sb][2]: search_results[0][0].hme_sc_x / _y */
    const uint64_t *hme_sad;            /* [n_sb] */
} SvtHipTfMeTables;
typedef struct SvtHipTfPictureStats { /* what the decisions were (sums over blocks and references) */
    uint32_t blocks_64x64, blocks_32x32, blocks_16x16, blocks_8x8; /* predictions made per size */
    uint32_t early_exit_blocks;                                     /* (block, reference) pairs whose ME exited early */
    uint32_t pad[3];
} SvtHipTfPictureStats;
int svt_hip_tf_picture_host(const SvtHipTfPictureParams *params, const SvtHipTfHostPicture *central, const SvtHipTfHostPicture *refs, const SvtHipTfMeTables *me,
                            uint32_t n_refs, void *out_y, void *out_u, void *out_v, SvtHipTfPictureStats *stats /* or NULL */);
/* Device-resident form (for a caller whose pictures and ME tables are in HBM already, e.g. right after the ME stage): the reference frames' planes lie back to back
 * (ref_pitch / ref_uv_pitch samples apart), the ME tables are device arrays over all frames ([n_refs][n_sb][85] ...: `me` points to ONE SvtHipTfMeTables of device
 * pointers), the filtered 64x64 blocks replace the central picture in place, stats_dev (or NULL) receives the counts; asynchronous on `stream`. */
typedef struct SvtHipTfDevicePictures {
    void    *central[3];               /* the central picture's padded planes (first samples) */
    void    *refs[3];                  /* first reference frame's planes; frame r at + r * ref_pitch / ref_uv_pitch samples */
    uint64_t ref_pitch, ref_uv_pitch;  /* samples */
    void    *central_y8, *refs_y8;     /* subpel_8bit: the 8-bit luma copies (refs_y8: frames ref_y8_pitch samples apart); else NULL */
    uint64_t ref_y8_pitch;
} SvtHipTfDevicePictures;
size_t svt_hip_tf_picture_workspace(const SvtHipTfPictureParams *params, uint32_t n_refs); /* bytes; 0 for parameters outside what is built */
int    svt_hip_tf_picture(const SvtHipTfPictureParams *params, const SvtHipTfDevicePictures *pictures, const SvtHipTfMeTables *me, uint32_t n_refs, void *workspace,
                          SvtHipTfPictureStats *stats_dev, void *stream);

/* The whole open-loop ME stage from a HOST picture: upload -> quarter / sixteenth planes (made once per picture on the device, kept in the ring
 * with the full plane) -> HME levels 0-2 -> final search centre + integer_search_b64 geometry + full-pel search -> MeSbResults (+ raw tables on
 * request) -> download, on the submission's own stream like svt_hip_me_session_submit.  svt_hip_me_session_enable_stage sizes the extra
 * buffers once (decimated-plane padding as the reference's 32 / 16; max_regions = num_hme_sa_w * num_hme_sa_h; the largest ME area the stage
 * parameters can produce).  ref_ids: list-0 references first; results.num_of_ref_pic_to_search[0] + [1] must equal n_refs.  Every picture that
 * will serve as a reference must have been submitted after enable_stage (its decimated planes are made at upload time).  -5: not enabled, too
 * many regions, or areas beyond the enabled maximum. */
typedef struct SvtHipMeStageParams {
    uint8_t  num_hme_sa_w, num_hme_sa_h; /* me_ctx->num_hme_sa_w / num_hme_sa_h */
    uint8_t  hme_sub_sampled;            /* hme_search_method != FULL_SAD_SEARCH */
    uint8_t  me_sub_sad;                 /* me_search_method == SUB_SAD_SEARCH */
    int16_t  hme_sa_width[3], hme_sa_height[3]; /* per level: get_hme_l0_search_area, hme_l1_sa, hme_l2_sa */
    int16_t  me_sa_min_width, me_sa_min_height, me_sa_max_width, me_sa_max_height;
    uint8_t  mv_adj_enabled, mv_adj_nearest_ref_only;
    uint16_t mv_adj_mv_size_th, mv_adj_sa_multiplier;
    uint16_t dist[8];                    /* as SvtHipMeIntegerSearchParams */
    uint8_t  ref_pic_index[8];
    uint8_t  hme_l0_per_ref;             /* level-0 areas per reference (get_hme_l0_search_area) instead of hme_sa_width/height[0] */
    uint8_t  hme_prune_enabled, sr_adjustment; /* as SvtHipMeIntegerSearchParams */
    uint8_t  me_type_mctf;               /* me_type == ME_MCTF (the temporal filter's ME): dist[] holds the UNSCALED picture distances -- the integer search uses them as
                                          * they are (:1300-1302), pre-HME scales them itself -- and tf_me_exit_th applies */
    int16_t  hme_l0_sa_width_ref[8], hme_l0_sa_height_ref[8];
    uint16_t prune_ref_if_hme_sad_dev_bigger_than_th;
    uint16_t reduce_me_sr_based_on_mv_length_th, stationary_hme_sad_abs_th, stationary_me_sr_divisor, reduce_me_sr_based_on_hme_sad_abs_th,
             me_sr_divisor_for_low_hme_sad;
    uint32_t me_early_exit_th;           /* 0 = off */
    uint8_t  is_ref, me_8x8_var_enabled; /* as SvtHipMeIntegerSearchParams */
    uint8_t  hme_levels;                 /* 3 (or 0): enable_hme_level0/1/2_flag all set; 1: level 0 only (the temporal filter at hme_me_level 3 / 4); 2: enable_hme_level2_flag = 0 (M7 and above): the integer search starts from level 1 */
    uint8_t  pad1;
    uint32_t me_sr_div4_th, me_sr_div2_th, me_sr_mult2_th;
    uint8_t  temporal_layer_gt0;         /* me_ctx->temporal_layer_index > 0 (list 1 takes part in pre-HME / HME, reference gating is active); 0 = base layer: list 1's
                                          * references skip HME, start the integer search at (0, 0) and inherit the last list-0 slot's HME SAD for the pruning steps */
    uint8_t  prehme_enabled, prehme_skip_search_line, prehme_l1_early_exit; /* me_ctx->prehme_ctrl */
    uint16_t prehme_sa_min_width[2], prehme_sa_min_height[2], prehme_sa_max_width[2], prehme_sa_max_height[2];
    uint32_t zz_sad_th, phme_sad_th;     /* me_hme_prune_ctrls (0 = off) */
    uint16_t zz_sad_pct, phme_sad_pct;
    uint32_t prev_me_stage_based_exit_th; /* as SvtHipHmeChainInputs (0 = off; the RTC screen-content and the temporal-filter ME settings use 64 * 64 * 4) */
    uint32_t me_safe_limit_zz_th;         /* me_ctx->me_safe_limit_zz_th when the picture qualifies (see svt_hip_me_ref_safe_limit_batch), else 0 */
    uint32_t tf_me_exit_th;               /* the temporal filter's ME (me_type == ME_MCTF): as SvtHipMeIntegerSearchParams; that caller also passes dist[] unscaled
                                           * (:1300-1302), leaves hme_prune_enabled / sr_adjustment / results.prune_ref at 0 and takes the raw tables (out->best_sad /
                                           * best_mv, total_me_candidate_index = NULL: no MeSbResults are formatted) */
    SvtHipMeResultsParams results;       /* formatting parameters (n_sb is filled in by the session) */
    uint16_t reduce_hme_l0_sr_th_min, reduce_hme_l0_sr_th_max; /* me_ctx->reduce_hme_l0_sr_th_* when distance_based_hme_resizing is on (else 0): see SvtHipHmeLevelParams */
    int16_t  hme_l0_sa_width_ref2[8], hme_l0_sa_height_ref2[8]; /* the level-0 areas with the (2 + ref index) divisors */
    int16_t  hme_l0_sa_width_ref4[8], hme_l0_sa_height_ref4[8]; /* ... with the (4 + ref index) divisors: read with sr_adjustment == 2 only (SvtHipHmeLevelParams.l0_still_rule) */
} SvtHipMeStageParams;
int svt_hip_me_session_enable_stage(void *session, uint32_t quarter_pad, uint32_t sixteenth_pad, uint32_t max_regions, uint32_t max_me_area_width,
                                    uint32_t max_me_area_height);
int svt_hip_me_session_submit_stage(void *session, int64_t pic_id, const uint8_t *plane_host, const int64_t *ref_ids, uint32_t n_refs,
                                    const SvtHipMeStageParams *stage, const SvtHipMeResultsHost *out);

/* ---------------------------------------------------------------- transforms (a10, a12, a13, a14) --- */
/* TxSize / TxType numbering = Source/Lib/Codec/definitions.h (TX_4X4=0 .. TX_64X16=18; DCT_DCT=0 .. H_FLIPADST=15). */
typedef struct SvtHipFwdTxfmDesc {
    uint64_t in_off;    /* int16 elements from residual_base to the block's top-left residual */
    uint32_t in_stride; /* int16 elements */
    uint8_t  tx_type;
    uint8_t  pad[3];
} SvtHipFwdTxfmDesc;
/* n forward 2-D transforms of ONE tx_size (svt_av1_fwd_txfm2d_WxH -> av1_tranform_two_d_core_c, transforms.c:2259-2324).
 * pf_shape 0 = DEFAULT, 1 = N2, 2 = N4 (transforms.c:5202-5273: only the top-left 1/2, 1/4 corner is produced, rest 0).
 * coeff_out: n blocks of W*H int32, row-major, full W x H also for 64-point sizes (as the reference's output). */
void svt_hip_fwd_txfm2d_batch(const int16_t *residual_base, const SvtHipFwdTxfmDesc *descs, uint32_t n, int tx_size, int bit_depth,
                              int pf_shape, int32_t *coeff_out, void *stream);
typedef struct SvtHipInvTxfmDesc {
    uint64_t coeff_off;  /* int32 elements from coeff_base; block = min(W,32) x min(H,32) packed (inv_transforms.c:2567-2580) */
    uint64_t pred_off;   /* pixels from pred_base  (output_r) */
    uint64_t recon_off;  /* pixels from recon_base (output_w; may alias the prediction) */
    uint32_t pred_stride, recon_stride;
    uint8_t  tx_type;
    uint8_t  wht_full; /* svt_hip_iwht4x4_add_batch only: 1 = eob > 1 (16-coefficient form), 0 = DC-only form (inv_transforms.c:2826-2831) */
    uint8_t  pad[6];
} SvtHipInvTxfmDesc;
/* n inverse 2-D transforms + reconstruction (svt_av1_inv_txfm2d_add_WxH -> inv_txfm2d_add_c, inv_transforms.c:2459-2535) */
void svt_hip_inv_txfm2d_add_batch(const int32_t *coeff_base, const uint16_t *pred_base, uint16_t *recon_base,
                                  const SvtHipInvTxfmDesc *descs, uint32_t n, int tx_size, int bd, void *stream);
/* 8-bit pixel form (svt_av1_inv_txfm_add -> svt_av1_inv_txfm_add_c, inv_transforms.c:3177-3192) */
void svt_hip_inv_txfm2d_add_batch_u8(const int32_t *coeff_base, const uint8_t *pred_base, uint8_t *recon_base,
                                     const SvtHipInvTxfmDesc *descs, uint32_t n, int tx_size, void *stream);
/* The two entry points above compute the transform types AV1 allows for the size (test/TxfmCommon.h:160-209: DCT and identity only where a dimension is 32 or 64).
 * The reference's `_c` functions accept more: av1_iadst32_new (inv_transforms.c:1119-1552) gives every ADST type a result on the seven sizes with a 32-point
 * dimension, and the reference's own InvTxfm2dAddTest feeds them (test/InvTxfm2dAsmTest.cc:755-775).  The `_any_type` forms compute every type the `_c` function of
 * the size computes; the single-call `_hip` symbols go through them.  They are separate kernels because the 32-point ADST costs the 32x32 kernel 3 of its 7
 * waves per SIMD (71 -> 125 VGPRs); no bitstream reaches them. */
void svt_hip_inv_txfm2d_add_batch_any_type(const int32_t *coeff_base, const uint16_t *pred_base, uint16_t *recon_base,
                                           const SvtHipInvTxfmDesc *descs, uint32_t n, int tx_size, int bd, void *stream);
void svt_hip_inv_txfm2d_add_batch_any_type_u8(const int32_t *coeff_base, const uint8_t *pred_base, uint8_t *recon_base,
                                              const SvtHipInvTxfmDesc *descs, uint32_t n, int tx_size, void *stream);
/* single-call generic forms; the fixed-size RTCD symbols svt_av1_fwd_txfm2d_WxH[_N2|_N4]_hip (57) and
 * svt_av1_inv_txfm2d_add_WxH_hip (19, three signature shapes, common_dsp_rtcd.h:106-116) are generated from these. */
void svt_av1_fwd_txfm2d_hip(int16_t *input, int32_t *output, uint32_t input_stride, int tx_type, int tx_size, uint8_t bit_depth, int pf);
void svt_av1_inv_txfm2d_add_hip(const int32_t *input, uint16_t *output_r, int32_t stride_r, uint16_t *output_w, int32_t stride_w,
                                int tx_type, int tx_size, int32_t bd);
void svt_av1_inv_txfm_add_u8_hip(const int32_t *dqcoeff, uint8_t *dst_r, int32_t stride_r, uint8_t *dst_w, int32_t stride_w, int tx_type,
                                 int tx_size, int lossless, int eob);
/* svt_av1_inv_txfm_add (common_dsp_rtcd.h:144) -> svt_av1_inv_txfm_add_c (inv_transforms.c:3177-3192), exact prototype.
 * SvtHipTxfmParam is TxfmParam (definitions.h:1043-1055; TxType / TxSize are packed one-byte enums). */
#ifdef SVT_HIP_REFERENCE_TYPES
typedef TxfmParam SvtHipTxfmParam;
#else
typedef struct SvtHipTxfmParam {
    uint8_t tx_type, tx_size;
    int32_t lossless, bd, is_hbd;
    uint8_t tx_set_type; /* TxSetType, a packed one-byte enum (definitions.h:1027-1041) */
    int32_t eob;
} SvtHipTxfmParam;
#endif
void svt_av1_inv_txfm_add_hip(const int32_t *dqcoeff, uint8_t *dst_r, int32_t stride_r, uint8_t *dst_w, int32_t stride_w,
                              const SvtHipTxfmParam *txfm_param);
/* Lossless mode: 4x4 Walsh-Hadamard.  svt_av1_fwht4x4 (aom_dsp_rtcd.h:208) -> svt_av1_fwht4x4_c (transforms.c:3099-3152);
 * inverse = svt_av1_highbd_iwht4x4_16_add_c / _1_add_c (inv_transforms.c:2735-2825; reached through svt_av1_inv_txfm_add and
 * svt_av1_highbd_inv_txfm_add_4x4 when TxfmParam.lossless is set).  Batched: one 4x4 block per descriptor; coeff_out[n][16]. */
void svt_av1_fwht4x4_hip(int16_t *input, int32_t *output, uint32_t stride);
void svt_hip_fwht4x4_batch(const int16_t *residual_base, const SvtHipFwdTxfmDesc *descs, uint32_t n, int32_t *coeff_out, void *stream);
void svt_hip_iwht4x4_add_batch(const int32_t *coeff_base, const uint16_t *pred_base, uint16_t *recon_base, const SvtHipInvTxfmDesc *descs,
                               uint32_t n, int bd, void *stream);
void svt_hip_iwht4x4_add_batch_u8(const int32_t *coeff_base, const uint8_t *pred_base, uint8_t *recon_base, const SvtHipInvTxfmDesc *descs,
                                  uint32_t n, void *stream);

/* ---------------------------------------------------------------- quantization (a11, a15, a16) ------- */
typedef struct SvtHipQuantParams { /* DC/AC pairs of MacroblockPlane's tables (md_config_process.c:111-189) */
    int16_t zbin[2], round[2], quant[2], quant_shift[2], dequant[2];
    int32_t log_scale; /* 0, 1 (32x32-class), 2 (64-point) -- full_loop.c:1690 */
} SvtHipQuantParams;
typedef struct SvtHipQuantDesc {
    uint32_t qparam_idx; /* row of qparams[]                                  */
    uint32_t iscan_idx;  /* row of iscan_tables[][n_coeffs] (ScanOrder.iscan) */
    uint32_t qm_idx;     /* row of qm_tables / iqm_tables (ignored when those are NULL) */
    uint32_t reserved;
} SvtHipQuantDesc;
/* n blocks of n_coeffs (16..1024, power of two) coefficients, blocks contiguous.  mode 0 = svt_aom_quantize_b_c_ii,
 * 1 = svt_aom_highbd_quantize_b_c, 2 = quantize_fp_helper_c, 3 = highbd_quantize_fp_helper_c (full_loop.c:29-453).
 * Outputs: qcoeff/dqcoeff [n][n_coeffs] (every element written), eob[n]. */
void svt_hip_quantize_batch(int mode, const int32_t *coeff, uint32_t n, uint32_t n_coeffs, const SvtHipQuantParams *qparams,
                            const int16_t *iscan_tables, const uint8_t *qm_tables, const uint8_t *iqm_tables,
                            const SvtHipQuantDesc *descs, int32_t *qcoeff, int32_t *dqcoeff, uint16_t *eob, void *stream);
/* svt_handle_transform{64x64,32x64,64x32,16x64,64x16}[_N2_N4]_c (transforms.c:2374-2542), in place on n blocks of W*H:
 * energy[n] of the discarded high-frequency area + repack of 64-wide rows to stride 32. */
void svt_hip_handle_transform_batch(int32_t *coeff, uint32_t n, int tx_size, int n2_n4, uint64_t *energy, void *stream);
/* single-call forms; the ten RTCD quantizer symbols and the ten svt_handle_transformWxH[_N2_N4]_hip are thin aliases */
void     svt_quantize_hip(int mode, const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr,
                          const int16_t *quant_ptr, const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr,
                          const int16_t *dequant_ptr, uint16_t *eob_ptr, const int16_t *scan, const int16_t *iscan,
                          const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int log_scale);
uint64_t svt_handle_transform_hip(int32_t *output, int tx_size, int n2_n4);

/* BASELINE config 3 in ONE launch: forward transform -> (the 32x32 corner svt_handle_transform keeps of a 64-point block) -> quantize / dequantize -> inverse
 * transform + reconstruction; the coefficients stay in LDS between the stages (10 B/px of HBM traffic instead of 26; + 4 B/px when dqcoeff != NULL).
 * Bit-identical to svt_hip_fwd_txfm2d_batch(pf 0) -> svt_hip_handle_transform_batch -> svt_hip_quantize_batch -> svt_hip_inv_txfm2d_add_batch.
 * quant_mode as svt_hip_quantize_batch (0 / 2 with 8-bit pixels, 1 / 3 with 16-bit pixels); qcoeff / dqcoeff [n][min(W,32) * min(H,32)]; qm tables may be NULL. */
typedef struct SvtHipRoundtripDesc {
    uint64_t in_off;      /* int16 elements from residual_base */
    uint64_t pred_off;    /* pixels from pred_base */
    uint64_t recon_off;   /* pixels from recon_base (may alias the prediction) */
    uint32_t in_stride, pred_stride, recon_stride;
    uint32_t qparam_idx, iscan_idx, qm_idx; /* as SvtHipQuantDesc */
    uint8_t  tx_type;
    uint8_t  pad[7];
} SvtHipRoundtripDesc;
void svt_hip_txfm_quant_roundtrip_batch(const int16_t *residual_base, const void *pred_base, void *recon_base, const SvtHipRoundtripDesc *descs, uint32_t n,
                                        int tx_size, int bd, int quant_mode, const SvtHipQuantParams *qparams, const int16_t *iscan_tables,
                                        const uint8_t *qm_tables, const uint8_t *iqm_tables, int32_t *qcoeff, int32_t *dqcoeff, uint16_t *eob, void *stream);

/* ------------------------------------------- picture preparation for ME ------------------------- */
/* downsample_2d -> svt_aom_downsample_2d_c (aom_dsp_rtcd.h:841, pic_analysis_process.c:130-160): out(x, y) = (2x2 box at the centre of cell
 * (x, y) of decim_step x decim_step input pixels + 2) >> 2; host pointers (RTCD form). */
void svt_aom_downsample_2d_hip(uint8_t *input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height,
                               uint8_t *decim_samples, uint32_t decim_stride, uint32_t decim_step);
/* Device-resident form of one level of svt_aom_downsample_filtering_input_picture (pic_analysis_process.c:2138-2200): decimate the picture
 * whose first pixel is `in_origin` into the interior of the padded plane `out_base` (interior origin at (pad_x, pad_y)) AND replicate the
 * pad_x / pad_y wide borders (svt_aom_generate_padding, pic_operators.c:397-441) in the same launch. */
void svt_hip_downsample_2d_padded(const uint8_t *in_origin, uint32_t in_stride, uint32_t in_width, uint32_t in_height, uint8_t *out_base,
                                  uint32_t out_stride, uint32_t pad_x, uint32_t pad_y, uint32_t step, void *stream);
/* svt_aom_generate_padding on a device plane, in place: `base` = top-left of the padded plane */
void svt_hip_generate_padding(uint8_t *base, uint32_t stride, uint32_t width, uint32_t height, uint32_t pad_x, uint32_t pad_y, void *stream);

/* --------------------------------------------------- deblocking edge filters ---------------------- */
/* svt_aom_lpf_{horizontal,vertical}_{4,6,8,14} / svt_aom_highbd_lpf_* -> `_c` (common_dsp_rtcd.h:1037-1067, Codec/deblocking_common.c:141-865).
 * Batched form: n edge segments of 4 samples over a device plane; (x, y) = the q0 sample of the segment's first position; vertical = 1 for a
 * column boundary (filtering along x); length in {4, 6, 8, 14}; blimit / limit / thresh as the reference's per-edge LoopFilterThresh bytes.
 * Segments of one launch must not overlap (the reference filters all vertical edges of a picture before the horizontal ones).  One thread per
 * segment: list the segments in raster order (x fastest) and grouped by length, so that neighbouring threads touch neighbouring addresses and
 * take the same filter path; a vertical edge reads up to 8 samples either side of x (4 for lengths < 14), which must lie inside the allocation. */
typedef struct SvtHipLpfEdge {
    uint32_t x, y;
    uint8_t  vertical, length, blimit, limit, thresh, pad[3];
} SvtHipLpfEdge; /* 16 bytes */
void svt_hip_lpf_edges_batch(void *plane, uint32_t stride, int is_16bit, int bd, const SvtHipLpfEdge *edges, uint32_t n, void *stream);
/* One plane of svt_av1_loop_filter_frame (deblocking_filter.c:625) from HOST memory: all vertical-edge segments, then all horizontal-edge segments (the order
 * the standard defines; the reference's SB-by-SB schedule with its one-SB lag for horizontal edges gives the same picture), in place.  The plane must be
 * readable 16 samples left / right of its rows (the reference's picture padding).  A seam records the segments by running the reference's own driver with
 * recording leaf functions.  Synchronous.  The WHOLE width x height plane is downloaded in place: nothing else may write the plane while the call runs (the
 * frame-level seams call it from the one thread that owns the picture at that stage). */
int svt_hip_lpf_plane_host(void *plane, uint32_t stride, uint32_t width, uint32_t height, int is_16bit, int bd, const SvtHipLpfEdge *vert, uint32_t n_vert,
                            const SvtHipLpfEdge *horz, uint32_t n_horz);
#define SVT_HIP_LPF_DECL(LEN)                                                                                                              \
    void svt_aom_lpf_horizontal_##LEN##_hip(uint8_t *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh); \
    void svt_aom_lpf_vertical_##LEN##_hip(uint8_t *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh);   \
    void svt_aom_highbd_lpf_horizontal_##LEN##_hip(uint16_t *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit,                 \
                                                   const uint8_t *thresh, int32_t bd);                                                     \
    void svt_aom_highbd_lpf_vertical_##LEN##_hip(uint16_t *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit,                   \
                                                 const uint8_t *thresh, int32_t bd);
SVT_HIP_LPF_DECL(4)
SVT_HIP_LPF_DECL(6)
SVT_HIP_LPF_DECL(8)
SVT_HIP_LPF_DECL(14)
#undef SVT_HIP_LPF_DECL

/* ---------------------------------------------------------------- CDEF (a17-a20) --------------------- */
/* One plane of one frame.  Filter-block grid = ceil(width / (64 >> xdec)) x ceil(height / (64 >> ydec)); tile
 * construction as cdef_seg_search (cdef_process.c:208-228).  Call once per plane, luma first (it produces dir/var). */
typedef struct SvtHipCdefParams {
    const void *recon;   /* deblocked reconstruction (input), PIX = uint8_t or uint16_t            */
    const void *source;  /* original picture (search mode only)                                    */
    void       *out;     /* apply mode: filtered plane, OUT OF PLACE; caller pre-copies recon->out
                            (skipped units / zero-strength blocks are not touched)                  */
    uint32_t recon_stride, source_stride, out_stride; /* pixels */
    uint32_t width, height;                           /* plane size in pixels */
    uint8_t  xdec, ydec, pli, is_16bit;
    uint8_t  coeff_shift, pri_damping, sec_damping, subsampling; /* dampings before the per-plane adjustment of cdef.c:349-350 */
    uint32_t ncand;       /* search: number of (pri, sec) candidates */
    const uint8_t *skip;  /* [(fb rows * 8)][(fb cols * 8)] 1 = 8x8 luma unit is skipped (svt_sb_compute_cdef_list) */
    const int32_t *pri;   /* search: [ncand] primary levels; apply: [nfb]                           */
    const int32_t *sec;   /* search: [ncand] secondary strengths in {0,1,2,4}; apply: [nfb]         */
    uint8_t       *dir;   /* [nfb][64] written when pli == 0, read otherwise                        */
    int32_t       *var;   /* [nfb][64]                                                              */
    uint64_t      *mse;   /* search: [nfb][ncand] = svt_compute_cdef_dist of every candidate        */
} SvtHipCdefParams;
/* mode 0 = apply (svt_cdef_filter_fb over the frame, enc_cdef.c:284), 1 = strength search (cdef_process.c:106), 2 = apply with the luma directions /
 * variances taken from dir / var (what the search pass over the same reconstruction wrote; the reference recomputes identical values, cdef.c:367-386).
 * All pointers inside `params` are DEVICE pointers; the struct itself is read on the host. */
void svt_hip_cdef_frame(int mode, const SvtHipCdefParams *params, void *stream);
/* the same over the filter-block rows [fb_row_begin, fb_row_end) of the plane only (fb_row_end < 0: to the last row) -- one GPU's strip when a picture is split
 * over several devices; halos above / below the strip are read from the full input plane, outputs outside the strip are not touched */
void svt_hip_cdef_frame_rows(int mode, const SvtHipCdefParams *params, int fb_row_begin, int fb_row_end, void *stream);
/* svt_av1_cdef_frame (enc_cdef.c:284-560) for a 4:2:0 picture from HOST memory -- what a seam at cdef_process.c:458 calls: planes filtered IN PLACE (the
 * reference keeps the neighbours' unfiltered samples in line / column buffers, which is what filtering out of place on the device gives), skip = the 8x8
 * units svt_sb_compute_cdef_list leaves out (and every unit of a filter block the reference skips: all four strengths zero), pri / sec per filter block from
 * frm_hdr->cdef_params.cdef_y_strength / cdef_uv_strength[mbmi.cdef_strength] (sec 3 -> 4).  width / height = mi_cols * 4, mi_rows * 4.  Synchronous. */
typedef struct SvtHipCdefApplyHost {
    void          *plane[3];
    uint32_t       stride[3]; /* samples */
    uint32_t       width, height;
    uint8_t        num_planes, is_16bit, coeff_shift, damping; /* scs encoder_bit_depth - 8; frm_hdr->cdef_params.cdef_damping */
    const uint8_t *skip;      /* [(fb rows * 8)][(fb cols * 8)] */
    const int32_t *pri_y, *sec_y, *pri_uv, *sec_uv; /* [fb rows * fb cols] */
} SvtHipCdefApplyHost;
int svt_hip_cdef_apply_host(const SvtHipCdefApplyHost *params);
/* cdef_seg_search (cdef_process.c:106-345) for ALL filter blocks of a 4:2:0 picture from HOST memory: for each plane the distortion
 * (svt_compute_cdef_dist of the filtered block, NOT yet multiplied by the sub-sampling factor) of every candidate (pri, sec) and every filter block, plus the
 * luma directions / variances (pcs->cdef_dir_data).  Candidates: cdef_ctrls->default_first_pass_fs[] then default_second_pass_fs[] as (fs / 4, fs % 4 with 3 -> 4);
 * the chroma list leaves out the entries whose *_fs_uv is -1.  subsampling[] = cdef_ctrls->subsampling_factor capped as :215-219 (luma 4, chroma 1).  Synchronous. */
typedef struct SvtHipCdefSearchHost {
    const void    *recon[3], *source[3]; /* pcs->cdef_input_recon / cdef_input_source */
    uint32_t       recon_stride[3], source_stride[3];
    uint32_t       width, height; /* mi_cols * 4, mi_rows * 4 */
    uint8_t        is_16bit, coeff_shift, damping, subsampling[2], pad[3];
    const uint8_t *skip;          /* [(fb rows * 8)][(fb cols * 8)] from svt_sb_compute_cdef_list */
    uint32_t       ncand_y, ncand_uv;
    const int32_t *pri_y, *sec_y, *pri_uv, *sec_uv;
    uint64_t      *mse_y, *mse_u, *mse_v; /* [nfb][ncand] */
    uint8_t       *dir;                   /* [nfb][64] */
    int32_t       *var;                   /* [nfb][64] */
} SvtHipCdefSearchHost;
int svt_hip_cdef_search_host(const SvtHipCdefSearchHost *params);
/* Strength selection over the search output: svt_search_one_dual -> svt_search_one_dual_c (aom_dsp_rtcd.h:242,
 * enc_cdef.c:627-683).  mse0 / mse1 = [sb_count][64] luma / chroma distortion tables (device; svt_hip_cdef_frame(mode 1) writes exactly this
 * layout), lev0 / lev1 = device arrays holding the nb_strengths pairs selected so far, entry [nb_strengths] receives the new pair,
 * best_tot_mse[1] the frame total.  workspace: (4096 + sb_count) * 8 bytes of device scratch. */
void svt_hip_cdef_search_one_dual(const uint64_t *mse0, const uint64_t *mse1, int *lev0, int *lev1, int nb_strengths, int sb_count, int start_gi,
                                  int end_gi, uint64_t *best_tot_mse, void *workspace, void *stream);
uint64_t svt_search_one_dual_hip(int *lev0, int *lev1, int nb_strengths, uint64_t **mse[2], int sb_count, int start_gi, int end_gi);
/* joint_strength_search_dual (enc_cdef.c:697-726): nb_strengths greedy additions followed by 4 * nb_strengths refinement rounds, every step a
 * svt_search_one_dual on the device; lev0 / lev1 (device, >= nb_strengths entries) receive the selected pairs, best_tot_mse[1] the frame total. */
void svt_hip_cdef_joint_strength_search(const uint64_t *mse0, const uint64_t *mse1, int *lev0, int *lev1, int nb_strengths, int sb_count,
                                        int start_gi, int end_gi, uint64_t *best_tot_mse, void *workspace, void *stream);
/* per filter block: index of the selected pair with the smallest luma + chroma distortion (finish_cdef_search, enc_cdef.c:916-931) */
void svt_hip_cdef_assign_fb_strengths(const uint64_t *mse0, const uint64_t *mse1, const int *lev0, const int *lev1, int nb_strengths, int sb_count,
                                      int8_t *best_gi, void *stream);
/* RTCD-signature single-call forms (common_dsp_rtcd.h:1010-1029, aom_dsp_rtcd.h:208-209) */
uint8_t  svt_aom_cdef_find_dir_hip(const uint16_t *img, int32_t stride, int32_t *var, int32_t coeff_shift);
void     svt_aom_cdef_find_dir_dual_hip(const uint16_t *img1, const uint16_t *img2, int stride, int32_t *var1, int32_t *var2,
                                        int32_t coeff_shift, uint8_t *out1, uint8_t *out2);
void     svt_cdef_filter_block_hip(uint8_t *dst8, uint16_t *dst16, int32_t dstride, const uint16_t *in, int32_t pri_strength,
                                   int32_t sec_strength, int32_t dir, int32_t pri_damping, int32_t sec_damping, int32_t bsize,
                                   int32_t coeff_shift, uint8_t subsampling_factor);
/* SvtHipCdefList is CdefList (definitions.h:256-259): the (by, bx) 8x8 position of one non-skip block inside its 64x64 filter block */
#ifdef SVT_HIP_REFERENCE_TYPES
typedef CdefList SvtHipCdefList;
#else
typedef struct SvtHipCdefList { uint8_t by, bx; } SvtHipCdefList;
#endif
uint64_t svt_compute_cdef_dist_16bit_hip(const uint16_t *dst, int32_t dstride, const uint16_t *src, const SvtHipCdefList *dlist, int32_t cdef_count,
                                         SvtHipBlockSize bsize, int32_t coeff_shift, int32_t pli, uint8_t subsampling_factor);
uint64_t svt_compute_cdef_dist_8bit_hip(const uint8_t *dst8, int32_t dstride, const uint8_t *src8, const SvtHipCdefList *dlist, int32_t cdef_count,
                                        SvtHipBlockSize bsize, int32_t coeff_shift, int32_t pli, uint8_t subsampling_factor);
void     svt_aom_copy_rect8_8bit_to_16bit_hip(uint16_t *dst, int32_t dstride, const uint8_t *src, int32_t sstride, int32_t v, int32_t h);

/* ---------------------------------------------------------------- loop restoration (a21-a23) ---------- */
typedef struct SvtHipLrUnit {          /* RestorationUnitInfo (restoration.h:184-188) without its alignment padding */
    int32_t rtype;                     /* RestorationType: 0 RESTORE_NONE, 1 RESTORE_WIENER, 2 RESTORE_SGRPROJ */
    int16_t vfilter[8], hfilter[8];    /* WienerInfo: 7 symmetric taps + a trailing 0 */
    int32_t ep, xqd[2];                /* SgrprojInfo */
} SvtHipLrUnit;
/* One plane of svt_av1_loop_restoration_filter_frame (restoration.c:1179): restoration units of unit_size pixels
 * (shifted up by 8 >> ss_y, the last one absorbing a remainder < 3/2 unit), 64 >> ss_y row stripes, 64 >> ss_x column
 * processing units, stripe boundaries taken from the saved deblocked lines (2 rows per stripe, restoration.c:288-332),
 * frame edges replicated (svt_extend_frame).  Out of place.  All pointers are device pointers.  unit_size >= 64 >> ss_x, as in AV1 (RESTORATION_UNITSIZE
 * 64 / 128 / 256 on luma, halved on 4:2:0 chroma): a 64 >> ss_x column processing unit takes the coefficients of the unit its first column lies in. */
typedef struct SvtHipLrParams {
    const void *data;            /* CDEF output plane */
    const void *boundary_above;  /* rsb->stripe_boundary_above: rows 2*stripe, 2*stripe+1 */
    const void *boundary_below;
    void       *dst;
    uint32_t stride, boundary_stride, dst_stride; /* pixels */
    uint32_t width, height, unit_size;
    uint8_t  ss_x, ss_y, highbd, bit_depth;
    const SvtHipLrUnit *units;   /* [vert units][horz units] */
} SvtHipLrParams;
void svt_hip_lr_filter_frame(const SvtHipLrParams *params, void *stream);
/* the same over the 64-row stripes [stripe_begin, stripe_end) only (stripe_end < 0: to the last one): one GPU's strip of a picture split over several devices */
void svt_hip_lr_filter_frame_stripes(const SvtHipLrParams *params, int stripe_begin, int stripe_end, void *stream);
/* The same from HOST memory (a seam around svt_av1_loop_restoration_filter_frame, rest_process.c:632, calls it per restored plane): every pointer of params is
 * a host pointer, boundary_above / below point at frame column 0 (past the reference's RESTORATION_EXTRA_HORZ margin), dst may equal data; synchronous. */
int svt_hip_lr_filter_frame_host(const SvtHipLrParams *params);
/* The per-unit half of the loop-restoration SEARCH of one plane (restoration_seg_search, restoration_pick.c:1448-1527) as one resident device stage:
 * for every restoration unit the SSE of the unrestored unit (search_norestore_seg :1409), the Wiener solve + refinement (search_wiener_seg :1281:
 * svt_av1_compute_stats -> wiener_decompose_sep_sym -> finalize_sym_filter -> compute_score -> finer_tile_search_wiener_seg) and the self-guided
 * parameter search (search_sgrproj_seg :1205: search_selfguided_restoration + the SSE with the winner) -- everything the picture-level decisions
 * (search_*_finish / rest_finish_search, serial rate decisions against the previous unit's coefficients) read from RestUnitSearchInfo.
 * scs->use_boundaries_in_rest_search is 0 (enc_handle.c:4129): trials filter the plain plane.  dgd = the plane being restored, origin at sample (0, 0),
 * edges extended by >= 3 samples (+ 1 more on the right) as restoration_seg_search leaves it (:1480-1496); units in svt_aom_foreach_rest_unit_in_frame
 * order ([vert][horz]).  All pointers device.  Results are bit-exact with the reference's functions. */
typedef struct SvtHipLrSearchParams {
    const void *dgd, *src;
    uint32_t    dgd_stride, src_stride, width, height, unit_size; /* samples */
    uint8_t     ss_y, highbd, bit_depth;
    uint8_t     wn_enabled, wiener_win, wn_use_refinement, wn_max_one_refinement_step; /* cm->wn_filter_ctrls; wiener_win resolved (7, 5 or 3: :1286-1290) */
    uint8_t     sg_enabled, sg_start_ep, sg_end_ep, sg_ep_inc, sg_refine; /* the parameter-set loop of search_selfguided_restoration, resolved (:560-579) */
    uint8_t     pad[3];
} SvtHipLrSearchParams;
typedef struct SvtHipLrSearchUnit {
    int64_t sse[3];                 /* rusi->sse[RESTORE_NONE], [RESTORE_WIENER] (INT64_MAX: filter rejected by compute_score), [RESTORE_SGRPROJ] */
    int16_t vfilter[8], hfilter[8]; /* rusi->wiener */
    int32_t ep, xqd[2];             /* rusi->sgrproj */
    int32_t pad;
} SvtHipLrSearchUnit;
typedef struct SvtHipLrPrevUnit { /* wn_filter_ctrls.use_prev_frame_coeffs (:1297-1302): the co-located unit of the previous frame was RESTORE_WIENER */
    int32_t use;
    int16_t vfilter[8], hfilter[8];
} SvtHipLrPrevUnit;
size_t svt_hip_lr_search_workspace(const SvtHipLrSearchParams *params); /* bytes (includes 8 B per sample and self-guided parameter set searched) */
/* prev: device, [units] or NULL.  Waits on the device internally (the lock-step Wiener refinement reads the number of units still searching back every
 * few rounds, through an event on its own side stream).  Returns 0, -1 for parameters outside the reference's ranges, or -2 when the lock-step Wiener refinement hit its 4096-step cap with units
 * still searching (results of the plane must then not be used; the reference's own loops are bounded far below that).  sse[1] / sse[2] of a tool that is
 * disabled (wn_enabled / sg_enabled == 0) are 0 and carry no meaning: callers read them only for enabled tools (as integration/rest_process_seam.c does). */
int svt_hip_lr_search_plane(const SvtHipLrSearchParams *params, const SvtHipLrPrevUnit *prev, SvtHipLrSearchUnit *units, void *workspace, void *stream);
/* The same from HOST planes (what a seam around restoration_seg_search, rest_process.c:612, calls once per plane): dgd / src / prev / units are host pointers,
 * dgd readable 3 rows above / below and 3 (left) / 4 (right) samples beside the plane; uploads, runs on the calling thread's stream, downloads. */
int svt_hip_lr_search_plane_host(const SvtHipLrSearchParams *params, const SvtHipLrPrevUnit *prev, SvtHipLrSearchUnit *units);

/* RTCD-signature single-call forms (common_dsp_rtcd.h:144-181); highbd pointers use the CONVERT_TO_BYTEPTR convention.
 * SvtHipConvolveParams is ConvolveParams (definitions.h:572-585); the Wiener path reads round_0 / round_1 only (convolve.c:100-147). */
#ifdef SVT_HIP_REFERENCE_TYPES
typedef ConvolveParams SvtHipConvolveParams;
#else
typedef struct SvtHipConvolveParams {
    int32_t   ref, do_average;
    uint16_t *dst; /* ConvBufType */
    int32_t   dst_stride, round_0, round_1, plane, is_compound, use_jnt_comp_avg, fwd_offset, bck_offset, use_dist_wtd_comp_avg;
} SvtHipConvolveParams;
#endif
void svt_av1_wiener_convolve_add_src_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *filter_x,
                                         const int16_t *filter_y, int32_t w, int32_t h, const SvtHipConvolveParams *conv_params);
void svt_av1_highbd_wiener_convolve_add_src_hip(const uint8_t *src8, ptrdiff_t src_stride, uint8_t *dst8, ptrdiff_t dst_stride,
                                                const int16_t *filter_x, const int16_t *filter_y, int32_t w, int32_t h,
                                                const SvtHipConvolveParams *conv_params, int32_t bd);
void svt_av1_selfguided_restoration_hip(const uint8_t *dgd8, int32_t width, int32_t height, int32_t dgd_stride, int32_t *flt0, int32_t *flt1,
                                        int32_t flt_stride, int32_t sgr_params_idx, int32_t bit_depth, int32_t highbd);
void svt_apply_selfguided_restoration_hip(const uint8_t *dat8, int32_t width, int32_t height, int32_t stride, int32_t eps, const int32_t *xqd,
                                          uint8_t *dst8, int32_t dst_stride, int32_t *tmpbuf, int32_t bit_depth, int32_t highbd);

/* ---------------------------------------------------------------- SATD / Hadamard (a8, a9), LR search statistics (a24) ------ */
typedef struct SvtHipSatdDesc {
    uint64_t in_off, pred_off;       /* elements from input_base / pred_base */
    uint32_t in_stride, pred_stride;
} SvtHipSatdDesc;
/* n transform blocks of tx_n x tx_n (4, 8, 16, 32) 8-bit pixels: residual -> svt_aom_hadamard_NxN -> svt_aom_satd, i.e. the body
 * of hadamard_path_c (enc_mode_config.c:2147-2215).  satd_out[n]; coeff_out (optional) [n][tx_n^2] in the reference's order. */
void svt_hip_hadamard_satd_batch(const uint8_t *input_base, const uint8_t *pred_base, const SvtHipSatdDesc *descs, uint32_t n, int tx_n,
                                 uint32_t *satd_out, int32_t *coeff_out, void *stream);
typedef struct SvtHipRect { int32_t h_start, h_end, v_start, v_end; } SvtHipRect;
/* n restoration units: Wiener auto/cross-correlation M[n][49], H[n][49*49] (only win^2 / win^4 entries used),
 * svt_av1_compute_stats_c / _highbd_c (restoration_pick.c:659-745).  dgd needs a (win/2)-pixel readable border.  max_rect_width / height =
 * the largest unit of the batch (sizes the launch; the rects themselves stay on the device).  Runs on the matrix cores (int8 digit split,
 * int32 / int64 accumulation: exact). */
void svt_hip_lr_compute_stats_batch(const void *dgd, const void *src, const SvtHipRect *rects, uint32_t n, int max_rect_width, int max_rect_height,
                                    int dgd_stride, int src_stride, int wiener_win, int bit_depth, int64_t *M, int64_t *H, void *stream);
/* the same with the sample size stated (1 or 2 bytes) instead of derived from the bit depth: the reference's highbd function takes 16-bit pictures at every bit depth,
 * 8 included (an 8-bit encode in the 16-bit pipeline; av1_compute_stats_test_hbd runs it at EB_EIGHT_BIT, test/RestorationPickTest.cc:576-583).  The form above is
 * this one with sample_bytes = bit_depth > 8 ? 2 : 1. */
void svt_hip_lr_compute_stats_batch_samples(const void *dgd, const void *src, const SvtHipRect *rects, uint32_t n, int max_rect_width, int max_rect_height,
                                            int dgd_stride, int src_stride, int wiener_win, int bit_depth, int sample_bytes, int64_t *M, int64_t *H, void *stream);
/* single-call forms (aom_dsp_rtcd.h:62-81,210-215,276; common_dsp_rtcd.h:1075-1087) */
int      svt_aom_satd_hip(const int32_t *coeff, int length);
void     svt_aom_hadamard_nxn_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff, int n);
void     svt_aom_hadamard_4x4_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff);
void     svt_aom_hadamard_8x8_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff);
void     svt_aom_hadamard_16x16_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff);
void     svt_aom_hadamard_32x32_hip(const int16_t *src_diff, ptrdiff_t src_stride, int32_t *coeff);
uint32_t svt_hadamard_path_hip(const uint8_t *input, uint32_t in_stride, const uint8_t *pred, uint32_t pred_stride, int block_size);
/* hadamard_path (aom_dsp_rtcd.h:582) -> hadamard_path_c (enc_mode_config.c:2147-2215), exact prototype: four Buf2D BY VALUE (definitions.h:243-249; residual and
 * coeff are scratch there) and the one-byte packed enum BlockSize (definitions.h:764-791).  Unpacks and calls svt_hadamard_path_hip. */
#ifdef SVT_HIP_REFERENCE_TYPES
typedef Buf2D SvtHipBuf2D;
#else
typedef struct SvtHipBuf2D { uint8_t *buf, *buf0; int width, height, stride; } SvtHipBuf2D;
#endif
uint32_t hadamard_path_hip(SvtHipBuf2D residual, SvtHipBuf2D coeff, SvtHipBuf2D input, SvtHipBuf2D pred, SvtHipBlockSize bsize);
void     svt_residual_kernel8bit_hip(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride, int16_t *residual,
                                     uint32_t residual_stride, uint32_t area_width, uint32_t area_height);
void     svt_residual_kernel16bit_hip(uint16_t *input, uint32_t input_stride, uint16_t *pred, uint32_t pred_stride, int16_t *residual,
                                      uint32_t residual_stride, uint32_t area_width, uint32_t area_height);
/* SvtHipSgrParams is SgrParamsType (definitions.h:1750-1753).  `bit_depth` of svt_av1_compute_stats_highbd is the enum EbBitDepth (EbSvtAv1Formats.h:101-108), which
 * gcc and clang give the type unsigned int. */
#ifdef SVT_HIP_REFERENCE_TYPES
typedef SgrParamsType SvtHipSgrParams;
#else
typedef struct SvtHipSgrParams { int32_t r[2], s[2]; } SvtHipSgrParams;
#endif
void     svt_av1_compute_stats_hip(int32_t wiener_win, const uint8_t *dgd, const uint8_t *src, int32_t h_start, int32_t h_end, int32_t v_start,
                                   int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H);
void     svt_av1_compute_stats_highbd_hip(int32_t wiener_win, const uint8_t *dgd8, const uint8_t *src8, int32_t h_start, int32_t h_end,
                                          int32_t v_start, int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H,
                                          unsigned int bit_depth);
int64_t  svt_av1_lowbd_pixel_proj_error_hip(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t *dat8,
                                            int32_t dat_stride, int32_t *flt0, int32_t flt0_stride, int32_t *flt1, int32_t flt1_stride,
                                            int32_t xq[2], const SvtHipSgrParams *params);
int64_t  svt_av1_highbd_pixel_proj_error_hip(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t *dat8,
                                             int32_t dat_stride, int32_t *flt0, int32_t flt0_stride, int32_t *flt1, int32_t flt1_stride,
                                             int32_t xq[2], const SvtHipSgrParams *params);
void     svt_get_proj_subspace_hip(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t *dat8, int32_t dat_stride,
                                   int32_t use_highbitdepth, int32_t *flt0, int32_t flt0_stride, int32_t *flt1, int32_t flt1_stride, int32_t *xq,
                                   const SvtHipSgrParams *params);

/* ------------------------------------------------ TPL dispenser, source-based half ---------------------------------------------
 * tpl_mc_flow_dispenser_sb_generic (Codec/src_ops_process.c:519-969), the part that runs when pcs->tpl_src_data_ready == 0: per 16x16 (dispenser level 0) or
 * 32x32 (level 1; complete 64x64 SBs only, :2048-2051) block of a picture the DC intra cost from SOURCE neighbours (:620-657), the SAD of every
 * uni-directional ME candidate against its reference's source picture at the clamped full-pel vector (:761-890), the winner (:892) and, for NEWMV, the forward
 * transform (N2 / N4 shape, rows subsampled by 1 << subsample_tx) + svt_av1_quantize_fp reconstruction error of its residual (get_quantize_error, :224-247).
 * Output = exactly the TplSrcStats the reference stores per 16x16 cell (:958-967), which its own "else" branch (:969-977) consumes: a seam around
 * tpl_mc_flow_dispenser (:1848) fills pa_me_data->tpl_src_stats_buffer from one call per picture and lets the reference run the reconstruction half.
 * Covered option sets: tpl levels 4 and 5 of set_tpl_params (initial_rc_process.c:343-378: every preset from M3 up) -- use_sad_in_src_search = 1, intra_mode_end = DC_PRED,
 * subpel_depth = FULL_PEL, compute_rate = 0 -- with intra_mode_end = search_flags = 0, and tpl levels 0-3 (:301-342; level 1 = presets M0-M2: every intra mode, transform +
 * SATD costs, quarter-pel vectors, rate; 16x16 blocks only) through those two fields (csrc/tpl_full.hip).  scs->in_loop_ois = 1; the caller must run the reference's C for
 * anything else (the host forms return -1).  8-bit pictures (the reference's TPL is 8-bit only: "10BIT not supported", :464). */
typedef struct SvtHipTplRef {           /* one per rf_idx = list * 4 + ref_idx (= svt_get_ref_frame_type(list, ref_idx) - 1, :783) */
    uint64_t plane_off;                 /* bytes from ref_base to the reference picture's buffer_y (tpl_ref_ds_ptr_array[list][ref].picture_ptr: input_padded_pic, :141) */
    uint64_t picture_number;            /* tpl_ref_ds_ptr_array[list][ref].picture_number */
    uint32_t stride, org_x, org_y;      /* stride_y, org_x, org_y of that picture */
    uint16_t max_width, max_height;     /* its max_width / max_height (the vector clamp, :791-801) */
    uint8_t  valid;                     /* 0: no such reference, or excluded: ref_tpl_group_idx > 0 && !base_pcs->tpl_valid_pic[idx] (:779-781) */
    uint8_t  pad[3];
} SvtHipTplRef;
typedef struct SvtHipTplSrcParams {
    uint32_t width, height;             /* pcs->enhanced_pic->width / height */
    uint32_t aligned_width;             /* pcs->aligned_width: the statistics grid is ((aligned_width + 15) >> 4) cells wide (:552) */
    uint32_t sbs_x, n_sb;               /* 64x64 SB grid of scs->b64_geom (raster); SBs cut by the aligned picture run at dispenser level 0 */
    uint32_t src_stride;                /* enhanced_pic->stride_y */
    uint64_t src_off;                   /* bytes from src_base to picture sample (0, 0) of enhanced_pic */
    uint8_t  dispenser_search_level;    /* tpl_ctrls: 0 = 16x16 blocks, 1 = 32x32 */
    uint8_t  subsample_tx;              /* 0, 1, 2: the transform sees every (1 << subsample_tx)-th row */
    uint8_t  pf_shape;                  /* EB_TRANS_COEFF_SHAPE: 0 DEFAULT, 1 N2, 2 N4 */
    uint8_t  disable_intra_pred;        /* tpl_ctrls.disable_intra_pred_nref && temporal_layer_index == hierarchical_levels (:557) */
    uint8_t  i_slice;                   /* pcs->slice_type == I_SLICE: no ME candidates (:767) */
    uint8_t  enable_me_16x16, enable_me_8x8; /* PU count of the ME tables: 85 / 21 / 5 */
    uint8_t  max_cand, max_refs, max_l0; /* pa_me_data */
    uint8_t  intra_mode_end;            /* tpl_ctrls.intra_mode_end: DC_PRED (0) .. PAETH_PRED (12) */
    uint8_t  search_flags;              /* bit 0: !use_sad_in_src_search (transform + SATD costs), bit 1: compute_rate, bits 2-3: sub-pel rounds of tpl_subpel_search
                                           (0 FULL_PEL, 1 HALF_PEL, 2 QUARTER_PEL), bit 4: subpel_diag_refinement == 4.  Both 0 = tpl levels 4 / 5 (the fast kernels);
                                           anything else = the option set of tpl levels 0-3 (csrc/tpl_full.hip; 16x16 blocks, subsample_tx 0) */
    int16_t  quant_fp[2], round_fp[2], dequant[2]; /* quants_8bit.y_quant_fp / y_round_fp[qIndex], deq_8bit.y_dequant_qtx[qIndex] (DC, AC; :541-547) */
    SvtHipTplRef refs[8];
} SvtHipTplSrcParams;
typedef struct SvtHipTplSrcStats {      /* TplSrcStats (coding_unit.h:323-331) with explicit layout */
    int64_t  srcrf_dist, srcrf_rate;
    uint64_t ref_frame_poc;
    int16_t  mv_row, mv_col;            /* MV in 1/8 sample units */
    int32_t  best_rf_idx;               /* -1: no inter candidate */
    uint8_t  best_mode;                 /* PredictionMode: the best intra mode (DC_PRED 0 .. PAETH_PRED 12) or NEWMV (16) */
    uint8_t  best_intra_mode;           /* DC_PRED .. intra_mode_end */
    uint8_t  written;                   /* 1 for cells the reference writes (at least half of the block inside the picture, :580), else the cell is left untouched */
    uint8_t  pad[5];
} SvtHipTplSrcStats;
/* Device form: total_me_candidate_index / me_mv_array / me_candidate_array as svt_hip_me_results_batch writes them ([n_sb][n_pus], [n_sb][n_pus * max_refs],
 * [n_sb][n_pus * max_cand]); stats [rows16][cols16] with cols16 = (aligned_width + 15) >> 4 -- only the cells of processed blocks are written. */
void svt_hip_tpl_src_stage(const SvtHipTplSrcParams *params, const uint8_t *src_base, const uint8_t *ref_base, const uint8_t *total_me_candidate_index,
                           const uint32_t *me_mv_array, const uint8_t *me_candidate_array, SvtHipTplSrcStats *stats, void *stream);
/* Host form (what the seam calls once per picture): every pointer is a host pointer.  src_buf / ref_buf[rf_idx] = start of each picture's luma buffer, *_rows the
 * rows it holds (stride x rows bytes are uploaded; references that share a buffer are uploaded once); params->src_off and refs[].plane_off are offsets INSIDE
 * those buffers.  stats: host, [rows16 * cols16], zero-filled for cells no block writes.  Returns 0, or -1 for an option set outside the covered one. */
typedef struct SvtHipTplHostPlanes {
    const uint8_t *src_buf;
    uint32_t       src_rows;
    uint32_t       ref_rows[8];
    const uint8_t *ref_buf[8];
} SvtHipTplHostPlanes;
int svt_hip_tpl_src_stage_host(const SvtHipTplSrcParams *params, const SvtHipTplHostPlanes *planes, const uint8_t *total_me_candidate_index,
                               const uint32_t *me_mv_array, const uint8_t *me_candidate_array, SvtHipTplSrcStats *stats);

/* ---- TPL dispenser, RECONSTRUCTION half (tpl_mc_flow_dispenser_sb_generic, src_ops_process.c:979-1198; same option sets) ----
 * Per block, from the statistics of the source-based half: the prediction into the picture's TPL reconstruction plane (mc_flow_rec_picture_buffer[frame_idx]) --
 * NEWMV = the block at the full-pel vector of rec_refs[best_rf_idx] (:1016-1019: the TPL reconstruction of a frame inside the sliding window, else the
 * reference's source picture), else DC from the RECONSTRUCTED neighbours (:1038-1070) --, residual against the source on the rows the transform sees -> forward
 * DCT_DCT -> svt_av1_quantize_fp -> recon_error (:1112-1131), the inverse transform onto the prediction when (is_ref or intra prediction on) and a coefficient
 * survived, skipped rows copied from the row above (:1135-1167), and the block's four statistics (:1170-1180) BEFORE result_model_store (:266), which stays the
 * caller's (it depends on synth_blk_size only).  An intra block reads its left and upper neighbours' reconstruction: the stage walks the picture's blocks by
 * anti-diagonals, one launch per diagonal (a valid order for the mixed 32x32 / 16x16 grid too: DESIGN 4.16). */
typedef struct SvtHipTplReconParams {
    SvtHipTplSrcParams src;        /* geometry, option set, quantizer row as for the source-based half (src.refs is not read) */
    SvtHipTplRef       rec_refs[8]; /* per list * 4 + ref: the plane NEWMV blocks copy from, relative to rec_ref_base (valid / picture_number / max_* are not read) */
    uint64_t           recon_off;  /* bytes from recon_base to picture sample (0, 0) of the reconstruction plane */
    uint32_t           recon_stride;
    uint8_t            is_ref;     /* pcs->tpl_data.is_ref */
    uint8_t            pad[3];
} SvtHipTplReconParams;
typedef struct SvtHipTplReconStats { /* at the block's top-left 16x16 cell */
    int64_t srcrf_dist, recrf_dist, srcrf_rate, recrf_rate;
    uint8_t written;               /* as SvtHipTplSrcStats.written */
    uint8_t coded;                 /* a coefficient survived the quantizer (eob != 0) */
    uint8_t pad[2];                /* pad[0] of a row's first cell: 0xEE = the row-wavefront form (SVT_HIP_TPL_RECON_FORM=1) gave up waiting for the row above */
    uint32_t reserved;             /* the stage's own (row progress of the row-wavefront form); meaningless afterwards */
} SvtHipTplReconStats;
/* Device form: src_stats / out [rows16][cols16] as the source-based stage writes them; recon_base is read (neighbours) and written (every processed block). */
void svt_hip_tpl_recon_stage(const SvtHipTplReconParams *params, const uint8_t *src_base, const uint8_t *rec_ref_base, const SvtHipTplSrcStats *src_stats,
                             uint8_t *recon_base, SvtHipTplReconStats *out, void *stream);
/* Host form: planes->src_buf / ref_buf[] as above with ref_buf[rf] = the buffer rec_refs[rf] lives in (only the references the statistics name are uploaded);
 * recon_buf = start of the reconstruction picture's luma buffer (recon_rows rows of recon_stride bytes: uploaded, updated, downloaded). */
int svt_hip_tpl_recon_stage_host(const SvtHipTplReconParams *params, const SvtHipTplHostPlanes *planes, const SvtHipTplSrcStats *src_stats, uint8_t *recon_buf,
                                 uint32_t recon_rows, SvtHipTplReconStats *out);
/* Both halves of tpl_mc_flow_dispenser_sb_generic (src_ops_process.c:519-1198) for one picture in ONE host call: src_planes as svt_hip_tpl_src_stage_host takes them,
 * rec_planes->ref_buf[rf] = the buffer params->rec_refs[rf] lives in (every valid reference; src_buf unused).  Every distinct buffer is uploaded once, the source-based
 * statistics stay on the device between the halves and come back in src_stats (for the caller's TplSrcStats buffer), the written rectangle of the reconstruction and
 * `out` as svt_hip_tpl_recon_stage_host returns them; one synchronisation.  Returns 0, -1 (option set not covered), -3 (a reference buffer is missing), -4 (a block
 * gave up waiting for its neighbours). */
int svt_hip_tpl_stage_host(const SvtHipTplReconParams *params, const SvtHipTplHostPlanes *src_planes, const SvtHipTplHostPlanes *rec_planes,
                           const uint8_t *total_me_candidate_index, const uint32_t *me_mv_array, const uint8_t *me_candidate_array, SvtHipTplSrcStats *src_stats,
                           uint8_t *recon_buf, uint32_t recon_rows, SvtHipTplReconStats *out);
/* The same with the planes kept RESIDENT on the device across calls.  ids names the CONTENT of every buffer (0 = do not keep): a buffer found under its (pointer, id)
 * is not uploaded again, an uploaded one stays for later calls, and the reconstruction this call produces stays under (recon_buf, ids->recon) with its borders
 * replicated as svt_aom_generate_padding replicates the host copy's after the dispenser (recon_width / height / org: the reconstruction picture's geometry).  A picture
 * of a TPL group is the source of one call and a reference of several others: with ids a call uploads one new plane instead of up to nine.
 * total_me_candidate_index == NULL: the source-based statistics are the CALLER's (src_stats is an input: an earlier TPL group's, src_ops_process.c:969-977) and only
 * the reconstruction half runs.  svt_hip_tpl_plane_drop(buffer): the host rewrote that buffer by other means -- forget what the device holds of it. */
typedef struct SvtHipTplPlaneIds {
    uint64_t src, src_ref[8], rec_ref[8], recon;
    uint32_t recon_width, recon_height, recon_org_x, recon_org_y;
} SvtHipTplPlaneIds;
int  svt_hip_tpl_stage_host_resident(const SvtHipTplReconParams *params, const SvtHipTplHostPlanes *src_planes, const SvtHipTplHostPlanes *rec_planes,
                                     const SvtHipTplPlaneIds *ids, const uint8_t *total_me_candidate_index, const uint32_t *me_mv_array, const uint8_t *me_candidate_array,
                                     SvtHipTplSrcStats *src_stats, uint8_t *recon_buf, uint32_t recon_rows, SvtHipTplReconStats *out);
void svt_hip_tpl_plane_drop(const void *host_buffer);
void svt_hip_tpl_plane_counts(uint64_t *hits, uint64_t *misses); /* resident-plane look-ups of the calling thread's device so far */

/* ---- ONE picture over several GPUs from a C host (the frame-partition case; csrc/partition.hip) ----
 * devices[0] = the HOME device: every pointer of the calls below lives there and `stream` belongs to it.  Each call is the batched primitive of the same name cut
 * into contiguous strips (ME: descriptor ranges = bands of SB rows; CDEF: filter-block rows; LR: 64-row stripes), strip k on devices[k]: the inputs are copied home ->
 * k (hipMemcpyPeerAsync, xGMI inside a node), the strip runs on k, its output rows are copied back into the caller's arrays exactly where the single-device call
 * writes them; the call returns with everything enqueued (the home stream waits for the peers).  Results are bit-identical to the single-device call.  A partition
 * is used by one host thread at a time.  create returns NULL for an empty / repeated / unavailable device list. */
void *svt_hip_frame_partition_create(const int *devices, int n);
void  svt_hip_frame_partition_destroy(void *partition);
int   svt_hip_frame_partition_size(const void *partition);
/* Test instruments.  svt_hip_debug_spin: a delay kernel of about `microseconds` (at most 100 ms) on `stream`.  svt_hip_frame_partition_set_jitter: from now on every
 * call of this partition puts delays of pseudo-random length (0 .. max_us, seeded) between the steps of its protocol on the home and the peer streams; 0 = off. */
void  svt_hip_debug_spin(void *stream, uint32_t microseconds);
void  svt_hip_frame_partition_set_jitter(void *partition, uint32_t seed, uint32_t max_us);
void  svt_hip_frame_partition_stats(const void *partition, uint64_t *calls, uint64_t *peer_bytes_in, uint64_t *peer_bytes_out);
/* svt_hip_me_fullpel_search_batch over items [0, n): src_bytes / ref_bytes = the extent of the plane sets behind src_base / ref_base (mirrored whole) */
int   svt_hip_frame_partition_me(void *partition, const uint8_t *src_base, size_t src_bytes, const uint8_t *ref_base, size_t ref_bytes, const SvtHipMeSearchDesc *descs,
                                 uint32_t n, uint32_t max_w, uint32_t max_h, int sub_sad, uint32_t *best_sad, uint32_t *best_mv, void *workspace, void *stream);
/* svt_hip_cdef_frame (modes 0 / 1 / 2) */
int   svt_hip_frame_partition_cdef(void *partition, int mode, const SvtHipCdefParams *params, void *stream);
/* svt_hip_lr_filter_frame */
int   svt_hip_frame_partition_lr(void *partition, const SvtHipLrParams *params, void *stream);
/* The host forms' switch: with n > 1 devices set, svt_hip_cdef_apply_host / svt_hip_cdef_search_host / svt_hip_lr_filter_frame_host run their frame launches through a
 * partition of the calling thread (home = devices[0], which must be the thread's device; other threads keep the single-device path); n <= 1 switches it off.
 * The binding of the reference encoder sets it from SVT_HIP_STRIPS=<d0,d1,...>.  Returns 0, or -1 for an invalid list. */
int   svt_hip_set_frame_partition(const int *devices, int n);
unsigned long long svt_hip_frame_partition_host_calls(void); /* frame launches of host forms that went through a partition so far */

/* ---- the fixed-size symbols of the RTCD tables (what svt_hip_setup_rtcd installs): thin aliases of the generic forms above, declared here so that a caller can
 * also bind them by name.  Prototypes as the reference's pointers (aom_dsp_rtcd.h / common_dsp_rtcd.h). ---- */
#define SVT_HIP_FOR_ALL_SAD_SIZES(X)                                                                                                              \
    X(128, 128) X(128, 64) X(64, 128) X(64, 64) X(64, 32) X(32, 64) X(32, 32) X(32, 16) X(16, 32) X(16, 16) X(16, 8) X(8, 16) X(8, 8) X(8, 4) X(4, 8) \
    X(4, 4) X(4, 16) X(16, 4) X(8, 32) X(32, 8) X(16, 64) X(64, 16)
#define SVT_HIP_SAD_DECL(W, H)                                                                                              \
    uint32_t svt_aom_sad##W##x##H##_hip(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride);            \
    void     svt_aom_sad##W##x##H##x4d_hip(const uint8_t *src, int src_stride, const uint8_t *const ref_array[], int ref_stride, uint32_t *sad_array);
SVT_HIP_FOR_ALL_SAD_SIZES(SVT_HIP_SAD_DECL)
#undef SVT_HIP_SAD_DECL
#define SVT_HIP_FOR_ALL_TX_SIZES(X)                                                                                                                \
    X(4, 4) X(8, 8) X(16, 16) X(32, 32) X(64, 64) X(4, 8) X(8, 4) X(8, 16) X(16, 8) X(16, 32) X(32, 16) X(32, 64) X(64, 32) X(4, 16) X(16, 4) X(8, 32) \
    X(32, 8) X(16, 64) X(64, 16)
#define SVT_HIP_FWD_DECL(W, H)                                                                                                        \
    void svt_av1_fwd_txfm2d_##W##x##H##_hip(int16_t *input, int32_t *output, uint32_t stride, uint8_t tx_type, uint8_t bd);           \
    void svt_av1_fwd_txfm2d_##W##x##H##_N2_hip(int16_t *input, int32_t *output, uint32_t stride, uint8_t tx_type, uint8_t bd);        \
    void svt_av1_fwd_txfm2d_##W##x##H##_N4_hip(int16_t *input, int32_t *output, uint32_t stride, uint8_t tx_type, uint8_t bd);
SVT_HIP_FOR_ALL_TX_SIZES(SVT_HIP_FWD_DECL)
#undef SVT_HIP_FWD_DECL
/* inverse: squares (input, r, stride_r, w, stride_w, tx_type, bd); 4x8 / 8x4 / 4x16 / 16x4 add tx_size; the rest add tx_size, eob (common_dsp_rtcd.h:106-116) */
#define SVT_HIP_INV_SQ_DECL(N) \
    void svt_av1_inv_txfm2d_add_##N##x##N##_hip(const int32_t *in, uint16_t *r, int32_t sr, uint16_t *w, int32_t sw, uint8_t tx_type, int32_t bd);
#define SVT_HIP_INV_R1_DECL(W, H) \
    void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *in, uint16_t *r, int32_t sr, uint16_t *w, int32_t sw, uint8_t tx_type, uint8_t tx_size, int32_t bd);
#define SVT_HIP_INV_R2_DECL(W, H) \
    void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *in, uint16_t *r, int32_t sr, uint16_t *w, int32_t sw, uint8_t tx_type, uint8_t tx_size, int32_t eob, int32_t bd);
SVT_HIP_INV_SQ_DECL(4) SVT_HIP_INV_SQ_DECL(8) SVT_HIP_INV_SQ_DECL(16) SVT_HIP_INV_SQ_DECL(32) SVT_HIP_INV_SQ_DECL(64)
SVT_HIP_INV_R1_DECL(4, 8) SVT_HIP_INV_R1_DECL(8, 4) SVT_HIP_INV_R1_DECL(4, 16) SVT_HIP_INV_R1_DECL(16, 4)
SVT_HIP_INV_R2_DECL(8, 16) SVT_HIP_INV_R2_DECL(16, 8) SVT_HIP_INV_R2_DECL(16, 32) SVT_HIP_INV_R2_DECL(32, 16) SVT_HIP_INV_R2_DECL(32, 64) SVT_HIP_INV_R2_DECL(64, 32)
SVT_HIP_INV_R2_DECL(8, 32) SVT_HIP_INV_R2_DECL(32, 8) SVT_HIP_INV_R2_DECL(16, 64) SVT_HIP_INV_R2_DECL(64, 16)
#undef SVT_HIP_INV_SQ_DECL
#undef SVT_HIP_INV_R1_DECL
#undef SVT_HIP_INV_R2_DECL
/* the ten quantizer pointers (aom_dsp_rtcd.h: svt_aom_quantize_b ... svt_av1_highbd_quantize_fp_qm) and the ten svt_handle_transformWxH[_N2_N4] */
#define SVT_HIP_QARGS                                                                                                                       \
    const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr,                \
        const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr, const int16_t *scan, \
        const int16_t *iscan
void svt_aom_quantize_b_hip(SVT_HIP_QARGS, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, const int32_t log_scale);
void svt_aom_highbd_quantize_b_hip(SVT_HIP_QARGS, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, const int32_t log_scale);
void svt_av1_quantize_fp_hip(SVT_HIP_QARGS);
void svt_av1_quantize_fp_32x32_hip(SVT_HIP_QARGS);
void svt_av1_quantize_fp_64x64_hip(SVT_HIP_QARGS);
void svt_av1_quantize_fp_qm_hip(SVT_HIP_QARGS, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int16_t log_scale);
void svt_av1_highbd_quantize_fp_hip(SVT_HIP_QARGS, int16_t log_scale);
void svt_av1_highbd_quantize_fp_qm_hip(SVT_HIP_QARGS, const uint8_t *qm_ptr, const uint8_t *iqm_ptr, int16_t log_scale);
#undef SVT_HIP_QARGS
#define SVT_HIP_HT_DECL(W, H)                                   \
    uint64_t svt_handle_transform##W##x##H##_hip(int32_t *output); \
    uint64_t svt_handle_transform##W##x##H##_N2_N4_hip(int32_t *output);
SVT_HIP_HT_DECL(64, 64) SVT_HIP_HT_DECL(32, 64) SVT_HIP_HT_DECL(64, 32) SVT_HIP_HT_DECL(16, 64) SVT_HIP_HT_DECL(64, 16)
#undef SVT_HIP_HT_DECL

#ifdef __cplusplus
}
#endif
#endif /* SVTAV1_HIP_H */
