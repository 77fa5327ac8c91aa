// hip_wiener_convolve_test.cc -- the reference's test/wiener_convolve_test.cc: svt_av1_wiener_convolve_add_src and the high-bit-depth form (8 / 10 / 12 bit) over its
// 22 block sizes, 7/5/3-tap random legal kernels and the extreme kernels, random and extreme pixels, against the `_c` functions.
#include "hip_decl.h"
#include "wiener_convolve_test.cc"

namespace {
// wiener_convolve_test.cc:470-474 (AVX2, AV1WienerConvolveLbdTest)
INSTANTIATE_TEST_SUITE_P(HIP, AV1WienerConvolveLbdTest,
                         ::testing::Combine(::testing::ValuesIn(test_block_size_table), ::testing::Values(svt_av1_wiener_convolve_add_src_hip)));
// wiener_convolve_test.cc:509-514 (AVX2, AV1WienerConvolveHbdTest)
INSTANTIATE_TEST_SUITE_P(HIP, AV1WienerConvolveHbdTest,
                         ::testing::Combine(::testing::ValuesIn(test_block_size_table), ::testing::Values(svt_av1_highbd_wiener_convolve_add_src_hip),
                                            testing::Values(8, 10, 12)));
}  // namespace
