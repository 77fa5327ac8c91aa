// hip_ResidualTest.cc -- the reference's test/ResidualTest.cc: svt_residual_kernel8bit / svt_residual_kernel16bit over its area sizes and min / max / random patterns.
#include "hip_decl.h"
#include "ResidualTest.cc"

namespace {
// ResidualTest.cc:239-243, :384-388 (AVX2)
INSTANTIATE_TEST_SUITE_P(HIP, ResidualKernel8BitTest,
                         ::testing::Combine(::testing::ValuesIn(TEST_AREA_SIZES), ::testing::ValuesIn(TEST_PATTERNS), ::testing::Values(svt_residual_kernel8bit_hip)));
INSTANTIATE_TEST_SUITE_P(HIP, ResidualKernel16BitTest,
                         ::testing::Combine(::testing::ValuesIn(TEST_AREA_SIZES), ::testing::ValuesIn(TEST_PATTERNS), ::testing::Values(svt_residual_kernel16bit_hip)));
}  // namespace
