// hip_PictureOperatorTest.cc -- the reference's test/PictureOperatorTest.cc, Downsample2DTest: svt_aom_downsample_2d (the 1/4 and 1/16 decimation that feeds hierarchical ME,
// SURVEY 8f rank 1) at 1920x1080 ... 88x72 and steps 2 / 4 / 8, against svt_aom_downsample_2d_c.  (Its PictureOperatorTest is the bi-prediction averaging kernel of mode
// decision -- outside SURVEY 8 -- and is not instantiated.)
#include "hip_decl.h"
#include "PictureOperatorTest.cc"

namespace {
// PictureOperatorTest.cc:319-323 (AVX2, Downsample2DTest)
INSTANTIATE_TEST_SUITE_P(HIP, Downsample2DTest,
                         ::testing::Combine(::testing::ValuesIn(DOWNSAMPLE_SIZES), ::testing::ValuesIn(DECIM_STEPS), ::testing::Values(svt_aom_downsample_2d_hip)));
}  // namespace
