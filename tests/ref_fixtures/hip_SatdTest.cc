// hip_SatdTest.cc -- the reference's test/SatdTest.cc (svt_aom_satd: constant-answer min / max fills and random match) with svt_aom_satd_hip.
#include "hip_decl.h"
#include "SatdTest.cc"

namespace {
// SatdTest.cc:142-145 (AVX2)
INSTANTIATE_TEST_SUITE_P(HIP, SatdTest, ::testing::Combine(::testing::Values(16, 64, 256, 1024), ::testing::Values(svt_aom_satd_hip)));
}  // namespace
