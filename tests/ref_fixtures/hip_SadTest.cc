// hip_SadTest.cc -- the reference's test/SadTest.cc fixtures with the `_hip` functions as the function under test (see Makefile).
// Parameter generators are the reference's own (TEST_PATTERNS, TEST_LOOP_AREAS, TEST_SAD_PATTERNS); each block mirrors the AVX2 instantiation it is cited next to.
#include "hip_decl.h"
#include "SadTest.cc"

namespace {
// SadTest.cc:411-414 (AVX2, SADTest)
INSTANTIATE_TEST_SUITE_P(HIP, SADTest, ::testing::Combine(::testing::ValuesIn(TEST_PATTERNS), ::testing::Values(svt_nxm_sad_kernel_hip)));
// SadTest.cc:658-663 (AVX2, sad_LoopTest)
INSTANTIATE_TEST_SUITE_P(HIP, sad_LoopTest,
                         ::testing::Combine(::testing::ValuesIn(TEST_PATTERNS), ::testing::ValuesIn(TEST_LOOP_AREAS), ::testing::Values(0, 1),
                                            ::testing::Values(svt_sad_loop_kernel_hip)));
// SadTest.cc:842-847 (AVX2, Allsad8x8_CalculationTest)
INSTANTIATE_TEST_SUITE_P(HIP, Allsad8x8_CalculationTest,
                         ::testing::Combine(::testing::ValuesIn(TEST_PATTERNS), ::testing::ValuesIn(TEST_SAD_PATTERNS),
                                            ::testing::Values(svt_ext_all_sad_calculation_8x8_16x16_hip)));
// SadTest.cc:971-976 (AVX2, Allsad32x32_CalculationTest)
INSTANTIATE_TEST_SUITE_P(HIP, Allsad32x32_CalculationTest,
                         ::testing::Combine(::testing::ValuesIn(TEST_PATTERNS), ::testing::ValuesIn(TEST_SAD_PATTERNS),
                                            ::testing::Values(svt_ext_eight_sad_calculation_32x32_64x64_hip)));
// SadTest.cc:1114-1119 (AVX2, Extsad8x8_CalculationTest)
INSTANTIATE_TEST_SUITE_P(HIP, Extsad8x8_CalculationTest,
                         ::testing::Combine(::testing::ValuesIn(TEST_PATTERNS), ::testing::ValuesIn(TEST_SAD_PATTERNS),
                                            ::testing::Values(svt_ext_sad_calculation_8x8_16x16_hip)));
// SadTest.cc:1240-1245 (SSE4_1, Extsad32x32_CalculationTest)
INSTANTIATE_TEST_SUITE_P(HIP, Extsad32x32_CalculationTest,
                         ::testing::Combine(::testing::ValuesIn(TEST_PATTERNS), ::testing::ValuesIn(TEST_SAD_PATTERNS),
                                            ::testing::Values(svt_ext_sad_calculation_32x32_64x64_hip)));
// SadTest.cc:1311-1315 (SSE2, InitializeBuffer32)
INSTANTIATE_TEST_SUITE_P(HIP, InitializeBuffer32,
                         ::testing::Combine(::testing::Values(2, 3, 4), ::testing::Values(1, 2, 3), ::testing::Values(svt_initialize_buffer_32bits_hip)));
// SadTest.cc:1568-1571 (AVX2, SADTestSubSample16bit)
INSTANTIATE_TEST_SUITE_P(HIP, SADTestSubSample16bit, ::testing::Combine(::testing::ValuesIn(TEST_PATTERNS), ::testing::Values(svt_aom_sad_16b_kernel_hip)));
// SadTest.cc:1835-1839 (AVX2, PmeSadLoopTest)
INSTANTIATE_TEST_SUITE_P(HIP, PmeSadLoopTest,
                         ::testing::Combine(::testing::ValuesIn(TEST_PATTERNS), ::testing::ValuesIn(TEST_LOOP_AREAS), ::testing::Values(svt_pme_sad_loop_kernel_hip)));
}  // namespace
