// hip_CdefTest.cc -- the reference's test/CdefTest.cc: svt_cdef_filter_block (4 block sizes x 16 boundary masks x 8/10/12 bit, every direction, strength and damping,
// dst8 and dst16 forms), svt_aom_cdef_find_dir and its dual form, svt_aom_copy_rect8_8bit_to_16bit, svt_compute_cdef_dist_16bit / _8bit and svt_search_one_dual.
// (svt_cdef_filter_block_8xn_16 -- the last tuple element of CDEFBlockTest -- is an internal pointer of the AVX2 / AVX-512 kernels, CdefTest.cc:42-45; it has no
// meaning for the `_hip` function and stays 0 = "not used".)
#include "hip_decl.h"
#include "CdefTest.cc"

namespace {
// CdefTest.cc:378-386 (AVX2, CDEFBlockTest), boundary masks 1..15: the reference's test as it is (180 of its 192 parameter sets).
// With boundary mask 0 its loops make 8.9 (8 bit) to 50 (12 bit) MILLION calls per parameter set (16 damping pairs x 128 levels x bd noise widths x 8 directions x
// 17 primary x 4 secondary strengths x 2 sub-samplings, CdefTest.cc:133-245) -- sized for a 50 ns SIMD call, not for a function that crosses PCIe twice per call.
// Those 12 sets run below through the fixture's own prepare_data() / run_test() with the three OUTER loops thinned; the inner ones (every direction, strength pair
// and sub-sampling, dst8 and dst16) are the reference's.  (Interior blocks at full density are what tests/test_cdef.py covers against the checker pinned on
// svt_cdef_filter_block_c: whole 4K planes, every strength.)
INSTANTIATE_TEST_SUITE_P(HIP, CDEFBlockTest,
                         ::testing::Combine(::testing::Values(&svt_cdef_filter_block_hip), ::testing::Values(&svt_cdef_filter_block_c),
                                            ::testing::Values(BLOCK_4X4, BLOCK_4X8, BLOCK_8X4, BLOCK_8X8), ::testing::Range(1, 16), ::testing::Range(8, 13, 2),
                                            ::testing::Values(0)));

class CDEFBlockInteriorTest : public CDEFBlockTest {
  public:
    // test_cdef(1) of CdefTest.cc:227-265 with: damping pairs (min, min) (min, max) (max, min) instead of all 16, three levels (0, the middle, the top of the range)
    // instead of 128, noise widths 1, bd / 2 and bd instead of every one -- 27 pictures x 544 (x 2 sub-samplings) calls per parameter set
    void test_cdef_thinned() {
        const int lo = 3 + bd_ - 8, hi = 6 + bd_ - 8;
        const int damp[3][2] = {{lo, lo}, {lo, hi}, {hi, lo}};
        const int top = (1 << bd_) - 1, levels[3] = {0, top / 2, top - (2 << (bd_ - 8)) + 1}, widths[3] = {1, bd_ / 2, bd_};
        for (const auto &d : damp)
            for (const int level : levels)
                for (const int bits : widths) {
                    prepare_data(level, bits);
                    run_test(d[0], d[1], 1);
                    if (bsize_ > BLOCK_4X4)  // (CdefTest.cc:250-258: the 2x sub-sampled 4x4 AVX2 kernel differs from C by design; kept as there)
                        run_test(d[0], d[1], 2);
                    if (::testing::Test::HasFatalFailure())
                        return;
                }
    }
};
TEST_P(CDEFBlockInteriorTest, MatchTest) {
    test_cdef_thinned();
}
INSTANTIATE_TEST_SUITE_P(HIP, CDEFBlockInteriorTest,
                         ::testing::Combine(::testing::Values(&svt_cdef_filter_block_hip), ::testing::Values(&svt_cdef_filter_block_c),
                                            ::testing::Values(BLOCK_4X4, BLOCK_4X8, BLOCK_8X4, BLOCK_8X8), ::testing::Values(0), ::testing::Range(8, 13, 2),
                                            ::testing::Values(0)));
// CdefTest.cc:507-510 / :652-655 (AVX2, CDEFFindDirTest / CDEFFindDirDualTest).  The reference's loops call the function 512 x (256 levels x 8..12 noise widths) x 3 depths
// = 3.9 million times (:459-481) -- 9 and 12 MINUTES of PCIe round trips for the two tests (profiles/r06_reference_fixtures.txt: both green).  They are instantiated under
// the prefix HIPFULL, which tests/test_ref_fixtures.py runs only with SVT_HIP_FIXTURES=full; the default run takes the derived fixtures below: the same generator and the
// same checks over 12 of the 512 repetitions (every depth, level and noise width kept).
INSTANTIATE_TEST_SUITE_P(HIPFULL, CDEFFindDirTest, ::testing::Values(make_tuple(&svt_aom_cdef_find_dir_hip, &svt_aom_cdef_find_dir_c)));
INSTANTIATE_TEST_SUITE_P(HIPFULL, CDEFFindDirDualTest, ::testing::Values(make_tuple(&svt_aom_cdef_find_dir_dual_hip, &svt_aom_cdef_find_dir_dual_c)));

class CDEFFindDirFewerRepeatsTest : public CDEFFindDirTest {
  public:
    void test_finddir_repeats(const int repeats) {  // test_finddir() of CdefTest.cc:455-485 with `count < repeats`
        for (int depth = 8; depth <= 12; depth += 2)
            for (int count = 0; count < repeats; count++)
                for (int level = 0, shift = depth - 8; level < (1 << depth); level += 1 << shift)
                    for (int bits = 1; bits <= depth; bits++) {
                        prepare_data(depth, bits, level);
                        int32_t       var_ref = 0, var_tst = 0;
                        const uint8_t res_ref = func_ref_(src_, size_, &var_ref, shift), res_tst = func_tst_(src_, size_, &var_tst, shift);
                        ASSERT_EQ(res_tst, res_ref) << "direction, depth " << depth << " level " << level << " bits " << bits;
                        ASSERT_EQ(var_tst, var_ref) << "variance, depth " << depth << " level " << level << " bits " << bits;
                    }
    }
};
TEST_P(CDEFFindDirFewerRepeatsTest, MatchTest) {
    test_finddir_repeats(12);
}
INSTANTIATE_TEST_SUITE_P(HIP, CDEFFindDirFewerRepeatsTest, ::testing::Values(make_tuple(&svt_aom_cdef_find_dir_hip, &svt_aom_cdef_find_dir_c)));

class CDEFFindDirDualFewerRepeatsTest : public CDEFFindDirDualTest {
  public:
    void test_finddir_repeats(const int repeats) {  // test_finddir() of CdefTest.cc:575-633 with `count < repeats`
        for (int depth = 8; depth <= 12; depth += 2)
            for (int count = 0; count < repeats; count++)
                for (int level = 0, shift = depth - 8; level < (1 << depth); level += 1 << shift)
                    for (int bits = 1; bits <= depth; bits++) {
                        prepare_data(depth, bits, level);
                        uint8_t r1 = 0, r2 = 0, t1 = 0, t2 = 0;
                        int32_t vr1 = 0, vr2 = 0, vt1 = 0, vt2 = 0;
                        func_ref_(src_, src2_, size_, &vr1, &vr2, shift, &r1, &r2);
                        func_tst_(src_, src2_, size_, &vt1, &vt2, shift, &t1, &t2);
                        ASSERT_EQ(t1, r1) << "direction 1, depth " << depth << " level " << level << " bits " << bits;
                        ASSERT_EQ(t2, r2) << "direction 2, depth " << depth << " level " << level << " bits " << bits;
                        ASSERT_EQ(vt1, vr1) << "variance 1, depth " << depth << " level " << level << " bits " << bits;
                        ASSERT_EQ(vt2, vr2) << "variance 2, depth " << depth << " level " << level << " bits " << bits;
                    }
    }
};
TEST_P(CDEFFindDirDualFewerRepeatsTest, MatchTest) {
    test_finddir_repeats(12);
}
INSTANTIATE_TEST_SUITE_P(HIP, CDEFFindDirDualFewerRepeatsTest, ::testing::Values(make_tuple(&svt_aom_cdef_find_dir_dual_hip, &svt_aom_cdef_find_dir_dual_c)));
}  // namespace
// CdefTest.cc:752-754, :878-880, :985-987, :1156-1157 (AVX2)
INSTANTIATE_TEST_SUITE_P(HIP, CDEFCopyRectTest, ::testing::Values(svt_aom_copy_rect8_8bit_to_16bit_hip));
INSTANTIATE_TEST_SUITE_P(HIP, CDEFComputeCdefDist16Bit, ::testing::Values(svt_compute_cdef_dist_16bit_hip));
INSTANTIATE_TEST_SUITE_P(HIP, CDEFComputeCdefDist8BitTest, ::testing::Values(svt_compute_cdef_dist_8bit_hip));
INSTANTIATE_TEST_SUITE_P(HIP, CDEFSearchOneDualTest, ::testing::Values(svt_search_one_dual_hip));
