// hip_SelfGuidedUtilTest.cc -- the reference's test/SelfGuidedUtilTest.cc: svt_av1_lowbd_pixel_proj_error / svt_av1_highbd_pixel_proj_error (random values, random
// sizes, extreme values) and svt_get_proj_subspace (8-bit and 16-bit pictures), against the `_c` functions.
#include "hip_decl.h"
#include "SelfGuidedUtilTest.cc"

namespace {
// SelfGuidedUtilTest.cc:338-341, :408-411 (AVX2)
INSTANTIATE_TEST_SUITE_P(HIP, PixelProjErrorLbdTest, ::testing::Values(make_tuple(svt_av1_lowbd_pixel_proj_error_hip, svt_av1_lowbd_pixel_proj_error_c)));
INSTANTIATE_TEST_SUITE_P(HIP, PixelProjErrorHbdTest, ::testing::Values(make_tuple(svt_av1_highbd_pixel_proj_error_hip, svt_av1_highbd_pixel_proj_error_c)));
// SelfGuidedUtilTest.cc:585-589 (AVX2)
INSTANTIATE_TEST_SUITE_P(HIP, GetProjSubspaceTestLbd, ::testing::Values(svt_get_proj_subspace_hip));
INSTANTIATE_TEST_SUITE_P(HIP, GetProjSubspaceTestHbd, ::testing::Values(svt_get_proj_subspace_hip));
}  // namespace
