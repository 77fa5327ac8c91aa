// hip_TemporalFilterTestPlanewise.cc -- the reference's test/TemporalFilterTestPlanewise.cc, EstimateNoiseTestFP / EstimateNoiseTestFPHbd: svt_estimate_noise_fp16 and
// svt_estimate_noise_highbd_fp16 (the temporal filter's noise level) at seven picture sizes up to 3840x2160, against the `_c` functions.
// Its other fixtures (TemporalFilterTestPlanewiseMedium[Hbd], ...GetFinalFilteredPixels, ...ApplyFilteringCentral*) take functions whose first argument is the
// reference's whole MeContext; this library runs the temporal filter as a picture stage (svt_hip_tf_*, include/svtav1_hip.h) behind the seam in temporal_filtering.c and does
// not install variants into those pointers (they are not in csrc/rtcd_hooks.def), so there is no `_hip` function of that prototype to hand them.
#include "hip_decl.h"
#include "TemporalFilterTestPlanewise.cc"

// TemporalFilterTestPlanewise.cc:1317-1329: the 8-bit kernels take (uint8_t*, uint16_t width, uint16_t height, uint16_t stride); the fixture's function type is the 16-bit one
static int32_t hip_estimate_noise_fp16_c_wrapper(const uint16_t *src, int width, int height, int stride, int bd) {
    (void)bd;
    return svt_estimate_noise_fp16_c((const uint8_t *)src, width, height, stride);
}
static int32_t hip_estimate_noise_fp16_wrapper(const uint16_t *src, int width, int height, int stride, int bd) {
    (void)bd;
    return svt_estimate_noise_fp16_hip((const uint8_t *)src, width, height, stride);
}
// TemporalFilterTestPlanewise.cc:1420-1434 (AVX2)
INSTANTIATE_TEST_SUITE_P(HIP, EstimateNoiseTestFP,
                         ::testing::Combine(::testing::Values(hip_estimate_noise_fp16_c_wrapper), ::testing::Values(hip_estimate_noise_fp16_wrapper),
                                            ::testing::Values(3840, 1920, 1280, 800, 640, 360, 357), ::testing::Values(2160, 1080, 720, 600, 480, 240, 237),
                                            ::testing::Values(8)));
INSTANTIATE_TEST_SUITE_P(HIP, EstimateNoiseTestFPHbd,
                         ::testing::Combine(::testing::Values(svt_estimate_noise_highbd_fp16_c), ::testing::Values(svt_estimate_noise_highbd_fp16_hip),
                                            ::testing::Values(3840, 1920, 1280, 800, 640, 360, 357), ::testing::Values(2160, 1080, 720, 600, 480, 240, 237),
                                            ::testing::Values(10)));
