// rtcd_hook.hip -- installs the `_hip` variants into the reference's run-time dispatch table.
//
// The reference selects a SIMD variant by assigning global function pointers once, single-threaded, inside
// svt_aom_setup_rtcd_internal() / svt_aom_setup_common_rtcd_internal() (Source/Lib/Codec/aom_dsp_rtcd.c:188,
// common_dsp_rtcd.c:466; called from Source/Lib/Globals/enc_handle.c:1444-1445 before any worker thread exists).
// Those pointers are ordinary exported globals (`RTCD_EXTERN`, aom_dsp_rtcd.h:24-29).  We reference them as WEAK
// symbols: when this library is loaded into a process that contains libSvtAv1Enc they resolve and are overwritten;
// in a stand-alone process (tests, bench) they are null and skipped.  No reference header is needed or copied: each
// declaration below restates the pointer's prototype (file:line cited) so the compiler checks our variant against it.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

#define WEAK_PTR(ret, name, args) extern "C" __attribute__((weak)) ret(*name) args

// aom_dsp_rtcd.h:842 / :779 / :846-868 (SAD family)
WEAK_PTR(uint32_t, svt_nxm_sad_kernel, (const uint8_t*, uint32_t, const uint8_t*, uint32_t, uint32_t, uint32_t));
WEAK_PTR(void, svt_sad_loop_kernel, (uint8_t*, uint32_t, uint8_t*, uint32_t, uint32_t, uint32_t, uint64_t*, int16_t*, int16_t*, uint32_t,
                                     uint8_t, int16_t, int16_t));
WEAK_PTR(void, svt_ext_all_sad_calculation_8x8_16x16, (uint8_t*, uint32_t, uint8_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*,
                                                       uint32_t*, uint32_t[16][8], uint32_t[64][8], bool));
WEAK_PTR(void, svt_ext_eight_sad_calculation_32x32_64x64, (uint32_t[16][8], uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t,
                                                           uint32_t[4][8]));
WEAK_PTR(void, svt_ext_sad_calculation_8x8_16x16, (uint8_t*, uint32_t, uint8_t*, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*,
                                                   uint32_t, uint32_t*, uint32_t*, bool));
WEAK_PTR(void, svt_ext_sad_calculation_32x32_64x64, (uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t*));
WEAK_PTR(void, svt_initialize_buffer_32bits, (uint32_t*, uint32_t, uint32_t, uint32_t));
WEAK_PTR(uint32_t, sad_16b_kernel, (uint16_t*, uint32_t, uint16_t*, uint32_t, uint32_t, uint32_t));

#define INSTALL(ptr, fn)     \
    do {                     \
        if (&ptr != nullptr) { \
            ptr = fn;        \
            n++;             \
        }                    \
    } while (0)

extern "C" int svt_hip_setup_rtcd(uint64_t flags) {
    (void)flags; // reserved: a future EB_CPU_FLAGS_HIP bit (EbSvtAv1.h:390-429 has free bits >= 17)
    svthip::ensure_device();
    int n = 0;
    INSTALL(svt_nxm_sad_kernel, svt_nxm_sad_kernel_hip);
    INSTALL(svt_sad_loop_kernel, svt_sad_loop_kernel_hip);
    INSTALL(svt_ext_all_sad_calculation_8x8_16x16, svt_ext_all_sad_calculation_8x8_16x16_hip);
    INSTALL(svt_ext_eight_sad_calculation_32x32_64x64, svt_ext_eight_sad_calculation_32x32_64x64_hip);
    INSTALL(svt_ext_sad_calculation_8x8_16x16, svt_ext_sad_calculation_8x8_16x16_hip);
    INSTALL(svt_ext_sad_calculation_32x32_64x64, svt_ext_sad_calculation_32x32_64x64_hip);
    INSTALL(svt_initialize_buffer_32bits, svt_initialize_buffer_32bits_hip);
    INSTALL(sad_16b_kernel, svt_aom_sad_16b_kernel_hip);
    return n;
}
