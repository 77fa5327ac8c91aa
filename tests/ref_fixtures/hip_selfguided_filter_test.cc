// hip_selfguided_filter_test.cc -- the reference's test/selfguided_filter_test.cc: svt_apply_selfguided_restoration on 8-bit and (8 / 10 / 12 bit) 16-bit pictures,
// every one of the 16 parameter sets, random xqd, processed in the reference's own 64-wide units, against svt_apply_selfguided_restoration_c.
#include "hip_decl.h"
#include "selfguided_filter_test.cc"

namespace {
// selfguided_filter_test.cc:260-262 (AVX2, AV1SelfguidedFilterTest)
INSTANTIATE_TEST_SUITE_P(HIP, AV1SelfguidedFilterTest, ::testing::Values(make_tuple(svt_apply_selfguided_restoration_hip)));
// selfguided_filter_test.cc:509-512 (AVX2, AV1HighbdSelfguidedFilterTest)
const int32_t highbd_params_hip[] = {8, 10, 12};
INSTANTIATE_TEST_SUITE_P(HIP, AV1HighbdSelfguidedFilterTest,
                         ::testing::Combine(::testing::Values(svt_apply_selfguided_restoration_hip), ::testing::ValuesIn(highbd_params_hip)));
}  // namespace
