// hip_hadamard_test.cc -- the reference's test/hadamard_test.cc (HadamardLowbdTest: random blocks and strides 8..56 against its own butterfly reference).
#include "hip_decl.h"
extern "C" decltype(svt_aom_hadamard_4x4_c) svt_aom_hadamard_4x4_hip;  // not a dispatch pointer of its own (aom_dsp_rtcd.h); the C function is what the C suite at :270 takes
#include "hadamard_test.cc"

namespace {
// hadamard_test.cc:270-283 (C: 4, 8, 16, 32; AVX2: 8, 16, 32)
INSTANTIATE_TEST_SUITE_P(HIP, HadamardLowbdTest,
                         ::testing::Values(HadamardFuncWithSize(&svt_aom_hadamard_4x4_hip, 4), HadamardFuncWithSize(&svt_aom_hadamard_8x8_hip, 8),
                                           HadamardFuncWithSize(&svt_aom_hadamard_16x16_hip, 16), HadamardFuncWithSize(&svt_aom_hadamard_32x32_hip, 32)));
}  // namespace
