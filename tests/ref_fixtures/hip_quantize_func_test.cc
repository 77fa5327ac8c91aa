// hip_quantize_func_test.cc -- the reference's test/quantize_func_test.cc: the "fast path" quantisers (svt_av1_quantize_fp, _fp_32x32, _fp_64x64,
// svt_av1_highbd_quantize_fp, the two _fp_qm forms) over zero, large negative, DC-only, random, all-q and half-dequant inputs, against the `_c` functions.
// (Its ComputeCulLevelTest is entropy-coder context derivation -- svt_av1_compute_cul_level, outside the block-DSP path of SURVEY 8 -- and is not instantiated.)
#include "hip_decl.h"
#include "quantize_func_test.cc"

namespace {
#define LBD(c_fn, hip_fn, sz) make_tuple(&c_fn, &hip_fn, static_cast<TxSize>(sz), TYPE_FP, EB_EIGHT_BIT)
// quantize_func_test.cc:806-842 (kQParamArrayAvx2: every size the AVX2 / SSE4.1 rows name, once)
const QuantizeParam kQParamArrayHip[] = {
    LBD(svt_av1_quantize_fp_c, svt_av1_quantize_fp_hip, TX_16X16),           LBD(svt_av1_quantize_fp_c, svt_av1_quantize_fp_hip, TX_4X16),
    LBD(svt_av1_quantize_fp_c, svt_av1_quantize_fp_hip, TX_16X4),            LBD(svt_av1_quantize_fp_c, svt_av1_quantize_fp_hip, TX_32X8),
    LBD(svt_av1_quantize_fp_c, svt_av1_quantize_fp_hip, TX_8X32),            LBD(svt_av1_quantize_fp_32x32_c, svt_av1_quantize_fp_32x32_hip, TX_32X32),
    LBD(svt_av1_quantize_fp_32x32_c, svt_av1_quantize_fp_32x32_hip, TX_16X64), LBD(svt_av1_quantize_fp_32x32_c, svt_av1_quantize_fp_32x32_hip, TX_64X16),
    LBD(svt_av1_quantize_fp_64x64_c, svt_av1_quantize_fp_64x64_hip, TX_64X64)};
INSTANTIATE_TEST_SUITE_P(HIP, QuantizeLbdTest, ::testing::ValuesIn(kQParamArrayHip));

// quantize_func_test.cc:900-936 (kQHbdParamArrayAvx2; its TX_64X64 row at ten bit says EB_EIGHT_BIT -- here all three depths carry all six sizes, as the SSE4.1 array at :844-898)
#define HBD(sz, bd) make_tuple(&svt_av1_highbd_quantize_fp_c, &svt_av1_highbd_quantize_fp_hip, static_cast<TxSize>(sz), TYPE_FP, bd)
#define HBD_ALL(bd) HBD(TX_16X16, bd), HBD(TX_4X16, bd), HBD(TX_16X4, bd), HBD(TX_32X8, bd), HBD(TX_8X32, bd), HBD(TX_64X64, bd)
const QuantizeHbdParam kQHbdParamArrayHip[] = {HBD_ALL(EB_EIGHT_BIT), HBD_ALL(EB_TEN_BIT), HBD_ALL(EB_TWELVE_BIT)};
INSTANTIATE_TEST_SUITE_P(HIP, QuantizeHbdTest, ::testing::ValuesIn(kQHbdParamArrayHip));

// quantize_func_test.cc:938-965 (kQmParamArrayAvx2, kQmParamHbdArrayAvx2)
#define QM(c_fn, hip_fn, sz, bd) make_tuple(&c_fn, &hip_fn, static_cast<TxSize>(sz), TYPE_FP, bd)
#define QM_ALL(c_fn, hip_fn, bd) QM(c_fn, hip_fn, TX_16X16, bd), QM(c_fn, hip_fn, TX_4X16, bd), QM(c_fn, hip_fn, TX_16X4, bd), QM(c_fn, hip_fn, TX_32X8, bd), QM(c_fn, hip_fn, TX_8X32, bd)
const QuantizeQmParam kQmParamArrayHip[]    = {QM_ALL(svt_av1_quantize_fp_qm_c, svt_av1_quantize_fp_qm_hip, EB_EIGHT_BIT)};
const QuantizeQmParam kQmParamHbdArrayHip[] = {QM_ALL(svt_av1_highbd_quantize_fp_qm_c, svt_av1_highbd_quantize_fp_qm_hip, EB_TEN_BIT)};
INSTANTIATE_TEST_SUITE_P(HIP, QuantizeQmTest, ::testing::ValuesIn(kQmParamArrayHip));
INSTANTIATE_TEST_SUITE_P(HIP, QuantizeQmHbdTest, ::testing::ValuesIn(kQmParamHbdArrayHip));

// The reference's MultipleQ test (quantize_func_test.cc:285-289 and its three siblings) calls the function 256 q indices x kTestNum = 1000 random blocks = 256 000 times per
// parameter set, 47 sets: twelve million PCIe round trips (45 of the 60 CPU-minutes of the whole fixture run, profiles/r06_reference_fixtures.txt: all green).  The default run of
// tests/test_ref_fixtures.py leaves `*.MultipleQ/*` to SVT_HIP_FIXTURES=full and takes these derived fixtures instead: the fixture's own QuantizeRun() -- generator, reference
// call and element-wise checks -- over EVERY q index with 24 random blocks each.
#define HIP_MULTIPLE_Q_FEWER_BLOCKS(Derived, Base)         \
    class Derived : public Base {};                        \
    TEST_P(Derived, EveryQ) {                              \
        for (int q = 0; q < QINDEX_RANGE; ++q) {           \
            QuantizeRun(true, q, 24);                      \
            if (::testing::Test::HasFatalFailure())        \
                return;                                    \
        }                                                  \
    }
HIP_MULTIPLE_Q_FEWER_BLOCKS(QuantizeLbdFewerBlocksTest, QuantizeLbdTest)
HIP_MULTIPLE_Q_FEWER_BLOCKS(QuantizeHbdFewerBlocksTest, QuantizeHbdTest)
HIP_MULTIPLE_Q_FEWER_BLOCKS(QuantizeQmFewerBlocksTest, QuantizeQmTest)
INSTANTIATE_TEST_SUITE_P(HIP, QuantizeLbdFewerBlocksTest, ::testing::ValuesIn(kQParamArrayHip));
INSTANTIATE_TEST_SUITE_P(HIP, QuantizeHbdFewerBlocksTest, ::testing::ValuesIn(kQHbdParamArrayHip));
INSTANTIATE_TEST_SUITE_P(HIP, QuantizeQmFewerBlocksTest, ::testing::ValuesIn(kQmParamArrayHip));
INSTANTIATE_TEST_SUITE_P(HIP_HBD, QuantizeQmFewerBlocksTest, ::testing::ValuesIn(kQmParamHbdArrayHip));
}  // namespace
