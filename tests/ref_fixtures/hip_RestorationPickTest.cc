// hip_RestorationPickTest.cc -- the reference's test/RestorationPickTest.cc: svt_av1_compute_stats and svt_av1_compute_stats_highbd (the Wiener normal equations M, H)
// over every block size (+ the two unit sizes past BlockSizeS_ALL), its 6 / 8 content patterns, windows 7 / 5 / 3 and 8 / 10 / 12 bit, against the `_c` functions.
#include "hip_decl.h"
#include "RestorationPickTest.cc"

// RestorationPickTest.cc:266-273 (AVX2, av1_compute_stats_test)
INSTANTIATE_TEST_SUITE_P(HIP, av1_compute_stats_test,
                         ::testing::Combine(::testing::Range(BLOCK_4X4, (BlockSize)(BlockSizeS_ALL + 2)), ::testing::Values(svt_av1_compute_stats_hip),
                                            ::testing::Range(0, 6), ::testing::Values(WIENER_WIN_CHROMA, WIENER_WIN, WIENER_WIN_3TAP)));
// RestorationPickTest.cc:576-583 (AVX2, av1_compute_stats_test_hbd)
INSTANTIATE_TEST_SUITE_P(HIP, av1_compute_stats_test_hbd,
                         ::testing::Combine(::testing::Range(BLOCK_4X4, (BlockSize)(BlockSizeS_ALL + 2)), ::testing::Values(svt_av1_compute_stats_highbd_hip),
                                            ::testing::Range(0, 8), ::testing::Values(WIENER_WIN_CHROMA, WIENER_WIN, WIENER_WIN_3TAP),
                                            ::testing::Values(EB_EIGHT_BIT, EB_TEN_BIT, EB_TWELVE_BIT)));
