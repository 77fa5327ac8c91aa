// hip_MotionEstimationTest.cc -- the reference's test/MotionEstimationTest.cc (svt_aom_sadMxN / svt_aom_sadMxNx4d, all 22 block sizes) with the `_hip` tables.
// That file names the `_avx2` functions directly (test/CMakeLists.txt lists it under x86_arch_files), so this one translation unit is compiled with
// ARCH_X86_64 defined (Makefile) and its MotionEstimation_avx2 tests exist in the binary too; the pytest runs the HIP* ones.
#include "hip_decl.h"
#include "MotionEstimationTest.cc"

// the reference's size order (MotionEstimationTest.cc:29-33), as its AVX2 tables at :57-82
static AomSadFn aom_sad_hip_func_ptr_array[num_sad] = {
    svt_aom_sad4x4_hip,   svt_aom_sad4x8_hip,   svt_aom_sad4x16_hip,  svt_aom_sad8x4_hip,   svt_aom_sad8x8_hip,    svt_aom_sad8x16_hip,
    svt_aom_sad8x32_hip,  svt_aom_sad16x4_hip,  svt_aom_sad16x8_hip,  svt_aom_sad16x16_hip, svt_aom_sad16x32_hip,  svt_aom_sad16x64_hip,
    svt_aom_sad32x8_hip,  svt_aom_sad32x16_hip, svt_aom_sad32x32_hip, svt_aom_sad32x64_hip, svt_aom_sad64x16_hip,  svt_aom_sad64x32_hip,
    svt_aom_sad64x64_hip, svt_aom_sad64x128_hip, svt_aom_sad128x64_hip, svt_aom_sad128x128_hip};
static AomSadMultiDFn aom_sad_4d_hip_func_ptr_array[num_sad] = {
    svt_aom_sad4x4x4d_hip,   svt_aom_sad4x8x4d_hip,   svt_aom_sad4x16x4d_hip,  svt_aom_sad8x4x4d_hip,   svt_aom_sad8x8x4d_hip,    svt_aom_sad8x16x4d_hip,
    svt_aom_sad8x32x4d_hip,  svt_aom_sad16x4x4d_hip,  svt_aom_sad16x8x4d_hip,  svt_aom_sad16x16x4d_hip, svt_aom_sad16x32x4d_hip,  svt_aom_sad16x64x4d_hip,
    svt_aom_sad32x8x4d_hip,  svt_aom_sad32x16x4d_hip, svt_aom_sad32x32x4d_hip, svt_aom_sad32x64x4d_hip, svt_aom_sad64x16x4d_hip,  svt_aom_sad64x32x4d_hip,
    svt_aom_sad64x64x4d_hip, svt_aom_sad64x128x4d_hip, svt_aom_sad128x64x4d_hip, svt_aom_sad128x128x4d_hip};

// MotionEstimationTest.cc:281-287
TEST(HIP_MotionEstimation, sadMxN_match) {
    sadMxN_match_test(aom_sad_hip_func_ptr_array);
}
TEST(HIP_MotionEstimation, sadMxNx4d_match) {
    sadMxNx4d_match_test(aom_sad_4d_hip_func_ptr_array);
}
