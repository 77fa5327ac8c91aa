// hip_DeblockTest.cc -- the reference's test/DeblockTest.cc: the sixteen loop-filter kernels svt_aom_lpf_{horizontal,vertical}_{4,6,8,14} and their high-bit-depth forms
// (8 / 10 / 12 bit) on random pictures with random thresholds, against the `_c` functions (the deblocking step before CDEF).
#include "hip_decl.h"
#include "DeblockTest.cc"

namespace {
// DeblockTest.cc:285-351 (the SSE2 tables)
#define HBD_ROW(bd)                                                                                                                                         \
    make_tuple(&svt_aom_highbd_lpf_horizontal_4_hip, &svt_aom_highbd_lpf_horizontal_4_c, bd), make_tuple(&svt_aom_highbd_lpf_horizontal_6_hip, &svt_aom_highbd_lpf_horizontal_6_c, bd),   \
    make_tuple(&svt_aom_highbd_lpf_horizontal_8_hip, &svt_aom_highbd_lpf_horizontal_8_c, bd), make_tuple(&svt_aom_highbd_lpf_horizontal_14_hip, &svt_aom_highbd_lpf_horizontal_14_c, bd), \
    make_tuple(&svt_aom_highbd_lpf_vertical_4_hip, &svt_aom_highbd_lpf_vertical_4_c, bd), make_tuple(&svt_aom_highbd_lpf_vertical_6_hip, &svt_aom_highbd_lpf_vertical_6_c, bd),           \
    make_tuple(&svt_aom_highbd_lpf_vertical_8_hip, &svt_aom_highbd_lpf_vertical_8_c, bd), make_tuple(&svt_aom_highbd_lpf_vertical_14_hip, &svt_aom_highbd_lpf_vertical_14_c, bd)
const HbdLpfTestParam kHbdLoop8TestHip[] = {HBD_ROW(8), HBD_ROW(10), HBD_ROW(12)};
const LdbLpfTestParam kLoop8TestHip[] = {
    make_tuple(&svt_aom_lpf_horizontal_4_hip, &svt_aom_lpf_horizontal_4_c, 8),   make_tuple(&svt_aom_lpf_vertical_4_hip, &svt_aom_lpf_vertical_4_c, 8),
    make_tuple(&svt_aom_lpf_horizontal_6_hip, &svt_aom_lpf_horizontal_6_c, 8),   make_tuple(&svt_aom_lpf_vertical_6_hip, &svt_aom_lpf_vertical_6_c, 8),
    make_tuple(&svt_aom_lpf_horizontal_8_hip, &svt_aom_lpf_horizontal_8_c, 8),   make_tuple(&svt_aom_lpf_vertical_8_hip, &svt_aom_lpf_vertical_8_c, 8),
    make_tuple(&svt_aom_lpf_horizontal_14_hip, &svt_aom_lpf_horizontal_14_c, 8), make_tuple(&svt_aom_lpf_vertical_14_hip, &svt_aom_lpf_vertical_14_c, 8)};
INSTANTIATE_TEST_SUITE_P(HIP, LbdLoopFilterTest, ::testing::ValuesIn(kLoop8TestHip));
INSTANTIATE_TEST_SUITE_P(HIP, HbdLoopFilterTest, ::testing::ValuesIn(kHbdLoop8TestHip));
}  // namespace
