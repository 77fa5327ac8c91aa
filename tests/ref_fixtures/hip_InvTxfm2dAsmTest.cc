// hip_InvTxfm2dAsmTest.cc -- the reference's test/InvTxfm2dAsmTest.cc: square (5 sizes), rectangular "type 1" (10 sizes, with tx_size and eob arguments) and "type 2"
// (4 sizes, with tx_size) inverse transforms, the 8-bit-pixel wrapper svt_av1_inv_txfm_add, and svt_handle_transform*, each at 8 and 10 bit against the `_c` function.
// The square suites get all_txtype_imp for EVERY size: where the reference's SIMD tables skip transform types a variant does not implement
// (InvTxfm2dAsmTest.cc:36-48, :247-271), the `_hip` functions are checked on every type the size allows.
#include "hip_decl.h"
#include "InvTxfm2dAsmTest.cc"

namespace {
// InvTxfm2dAsmTest.cc:262-274 (AVX2, InvTxfm2dAsmSqrTest)
static const InvSqrTxfmTestParam sqr_inv_txfm_c_hip_func_pairs[10] = {
    SQR_FUNC_PAIRS(svt_av1_inv_txfm2d_add_4x4, hip, TX_4X4, all_txtype_imp),       SQR_FUNC_PAIRS(svt_av1_inv_txfm2d_add_8x8, hip, TX_8X8, all_txtype_imp),
    SQR_FUNC_PAIRS(svt_av1_inv_txfm2d_add_16x16, hip, TX_16X16, all_txtype_imp), SQR_FUNC_PAIRS(svt_av1_inv_txfm2d_add_32x32, hip, TX_32X32, all_txtype_imp),
    SQR_FUNC_PAIRS(svt_av1_inv_txfm2d_add_64x64, hip, TX_64X64, all_txtype_imp),
};
INSTANTIATE_TEST_SUITE_P(HIP, InvTxfm2dAsmSqrTest, ::testing::ValuesIn(sqr_inv_txfm_c_hip_func_pairs));

// InvTxfm2dAsmTest.cc:447-493 (AVX2, InvTxfm2dAsmType1Test): the reference passes one generic dav1d entry for all ten sizes; here every size has its own `_hip` function
#define T1(w, h, sz) {svt_av1_inv_txfm2d_add_##w##x##h##_c, svt_av1_inv_txfm2d_add_##w##x##h##_hip, sz, 8}, {svt_av1_inv_txfm2d_add_##w##x##h##_c, svt_av1_inv_txfm2d_add_##w##x##h##_hip, sz, 10}
static const InvRectTxfmType1TestParam rect_type1_ref_funcs_hip[20] = {T1(8, 16, TX_8X16),   T1(8, 32, TX_8X32),   T1(16, 8, TX_16X8),   T1(16, 32, TX_16X32),
                                                                        T1(16, 64, TX_16X64), T1(32, 8, TX_32X8),   T1(32, 16, TX_32X16), T1(32, 64, TX_32X64),
                                                                        T1(64, 16, TX_64X16), T1(64, 32, TX_64X32)};
INSTANTIATE_TEST_SUITE_P(HIP, InvTxfm2dAsmType1Test, ::testing::ValuesIn(rect_type1_ref_funcs_hip));

// InvTxfm2dAsmTest.cc:674-696 (AVX2, InvTxfm2dAsmType2Test)
static const InvRectTxfmType2TestParam rect_type2_ref_funcs_hip[8] = {T1(4, 8, TX_4X8), T1(8, 4, TX_8X4), T1(4, 16, TX_4X16), T1(16, 4, TX_16X4)};
INSTANTIATE_TEST_SUITE_P(HIP, InvTxfm2dAsmType2Test, ::testing::ValuesIn(rect_type2_ref_funcs_hip));

// InvTxfm2dAsmTest.cc:841-845 (AVX2, InvTxfm2dAddTest)
INSTANTIATE_TEST_SUITE_P(HIP, InvTxfm2dAddTest,
                         ::testing::Combine(::testing::Values(svt_av1_inv_txfm_add_hip), ::testing::Values(static_cast<int>(EB_EIGHT_BIT), static_cast<int>(EB_TEN_BIT))));

// InvTxfm2dAsmTest.cc:980-1001 (AVX2, HandleTransformTest)
#define HT(w, h, sz) {svt_handle_transform##w##x##h##_c, svt_handle_transform##w##x##h##_hip, sz}, {svt_handle_transform##w##x##h##_N2_N4_c, svt_handle_transform##w##x##h##_N2_N4_hip, sz}
static const HandleTransformParam HandleTransformArrHIP[10] = {HT(16, 64, TX_16X64), HT(32, 64, TX_32X64), HT(64, 16, TX_64X16), HT(64, 32, TX_64X32), HT(64, 64, TX_64X64)};
INSTANTIATE_TEST_SUITE_P(HIP, HandleTransformTest, ::testing::ValuesIn(HandleTransformArrHIP));
}  // namespace
