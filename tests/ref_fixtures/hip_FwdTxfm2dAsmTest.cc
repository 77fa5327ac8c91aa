// hip_FwdTxfm2dAsmTest.cc -- the reference's test/FwdTxfm2dAsmTest.cc: all 19 transform sizes x every allowed transform type x {8, 10} bit x {full, N2, N4} coefficient
// shapes, random residuals, against svt_av1_transform_two_d_*_c / svt_av1_fwd_txfm2d_*_c (the tables of test/TxfmCommon.h:101-139).
#include "hip_decl.h"
#include "FwdTxfm2dAsmTest.cc"

namespace {
// index = TxSize (definitions.h:849-878), as the reference's tables at FwdTxfm2dAsmTest.cc:50-153; unlike its AVX2 table no entry is NULL
#define HIP_FWD_TABLE(sfx)                                                                                                                              \
    {svt_av1_fwd_txfm2d_4x4##sfx##_hip,   svt_av1_fwd_txfm2d_8x8##sfx##_hip,   svt_av1_fwd_txfm2d_16x16##sfx##_hip, svt_av1_fwd_txfm2d_32x32##sfx##_hip, \
     svt_av1_fwd_txfm2d_64x64##sfx##_hip, svt_av1_fwd_txfm2d_4x8##sfx##_hip,   svt_av1_fwd_txfm2d_8x4##sfx##_hip,   svt_av1_fwd_txfm2d_8x16##sfx##_hip,  \
     svt_av1_fwd_txfm2d_16x8##sfx##_hip,  svt_av1_fwd_txfm2d_16x32##sfx##_hip, svt_av1_fwd_txfm2d_32x16##sfx##_hip, svt_av1_fwd_txfm2d_32x64##sfx##_hip, \
     svt_av1_fwd_txfm2d_64x32##sfx##_hip, svt_av1_fwd_txfm2d_4x16##sfx##_hip,  svt_av1_fwd_txfm2d_16x4##sfx##_hip,  svt_av1_fwd_txfm2d_8x32##sfx##_hip,  \
     svt_av1_fwd_txfm2d_32x8##sfx##_hip,  svt_av1_fwd_txfm2d_16x64##sfx##_hip, svt_av1_fwd_txfm2d_64x16##sfx##_hip}
static const FwdTxfm2dFunc fwd_txfm_2d_hip_func[TX_SIZES_ALL]    = HIP_FWD_TABLE();
static const FwdTxfm2dFunc fwd_txfm_2d_N2_hip_func[TX_SIZES_ALL] = HIP_FWD_TABLE(_N2);
static const FwdTxfm2dFunc fwd_txfm_2d_N4_hip_func[TX_SIZES_ALL] = HIP_FWD_TABLE(_N4);

// FwdTxfm2dAsmTest.cc:519-547 (AVX2, N2_AVX2, N4_AVX2)
INSTANTIATE_TEST_SUITE_P(HIP, FwdTxfm2dAsmTest,
                         ::testing::Combine(::testing::Range(static_cast<int>(TX_4X4), static_cast<int>(TX_SIZES_ALL), 1),
                                            ::testing::Values(static_cast<int>(EB_EIGHT_BIT), static_cast<int>(EB_TEN_BIT)), ::testing::Values(DEFAULT_SHAPE),
                                            ::testing::Values(fwd_txfm_2d_c_func), ::testing::Values(fwd_txfm_2d_hip_func)));
INSTANTIATE_TEST_SUITE_P(HIP_N2, FwdTxfm2dAsmTest,
                         ::testing::Combine(::testing::Range(static_cast<int>(TX_4X4), static_cast<int>(TX_SIZES_ALL), 1),
                                            ::testing::Values(static_cast<int>(EB_EIGHT_BIT), static_cast<int>(EB_TEN_BIT)), ::testing::Values(N2_SHAPE),
                                            ::testing::Values(fwd_txfm_2d_N2_c_func), ::testing::Values(fwd_txfm_2d_N2_hip_func)));
INSTANTIATE_TEST_SUITE_P(HIP_N4, FwdTxfm2dAsmTest,
                         ::testing::Combine(::testing::Range(static_cast<int>(TX_4X4), static_cast<int>(TX_SIZES_ALL), 1),
                                            ::testing::Values(static_cast<int>(EB_EIGHT_BIT), static_cast<int>(EB_TEN_BIT)), ::testing::Values(N4_SHAPE),
                                            ::testing::Values(fwd_txfm_2d_N4_c_func), ::testing::Values(fwd_txfm_2d_N4_hip_func)));
}  // namespace
