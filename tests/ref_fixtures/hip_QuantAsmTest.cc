// hip_QuantAsmTest.cc -- the reference's test/QuantAsmTest.cc: svt_aom_quantize_b / svt_aom_highbd_quantize_b and their quantisation-matrix forms at TX_16X16 /
// 32X32 / 64X64 (log_scale 0 / 1 / 2), zero input, DC/AC extremes at q 0 and 255, DC only, and random input over every q index, against the `_c` functions.
#include "hip_decl.h"
#include "QuantAsmTest.cc"

namespace QuantizeAsmTest {
#define HIP_QUANT_B(prefix, fixture, bd, fn)                                                                                                             \
    INSTANTIATE_TEST_SUITE_P(prefix, fixture,                                                                                                            \
                             ::testing::Combine(::testing::Values(static_cast<int>(TX_16X16), static_cast<int>(TX_32X32), static_cast<int>(TX_64X64)), \
                                                ::testing::Values(static_cast<int>(bd)), ::testing::Values(fn)))
// QuantAsmTest.cc:318-340 (LBD_AVX2, HBD_AVX2)
HIP_QUANT_B(HIP_LBD, QuantizeBTest, EB_EIGHT_BIT, svt_aom_quantize_b_hip);
HIP_QUANT_B(HIP_HBD, QuantizeBTest, EB_TEN_BIT, svt_aom_highbd_quantize_b_hip);
// QuantAsmTest.cc:550-564: the `_qm` dispatch pointers receive the same `_hip` functions (the C table does the same with its own, aom_dsp_rtcd.c:218-219)
HIP_QUANT_B(HIP_LBD, QuantizeBQmTest, EB_EIGHT_BIT, svt_aom_quantize_b_hip);
HIP_QUANT_B(HIP_HBD, QuantizeBQmTest, EB_TEN_BIT, svt_aom_highbd_quantize_b_hip);
}  // namespace QuantizeAsmTest
