// fixtures_main.cc -- TEST INFRASTRUCTURE: entry point of the reference-fixture binary (tests/ref_fixtures/Makefile).
// Same shape as the reference's test/svt_av1_test.cc (InitGoogleTest + RUN_ALL_TESTS) plus the two things this binary needs that that one does not:
// the device is bound before the first fixture runs (a missing GPU is an error here, never a skip), and the reference's dispatch tables are set to the C variants
// (test/TestEnv.c: setup_test_env with no CPU flags), as several fixtures call through the pointers for their reference side.
#include <cstdio>

#include "gtest/gtest.h"
#include "svtav1_hip.h"

extern "C" void reset_test_env();

int main(int argc, char** argv) {
    ::testing::InitGoogleTest(&argc, argv);
    reset_test_env();
    if (svt_hip_init(0) != 0) {
        fprintf(stderr, "SvtAv1HipFixtures: no usable HIP device\n");
        return 3;
    }
    const int rc = RUN_ALL_TESTS();
    if (svt_hip_failed()) {
        fprintf(stderr, "SvtAv1HipFixtures: the device path switched itself off: %s\n", svt_hip_last_error());
        return 4;
    }
    if (svt_hip_debug_commit_violations()) { // (include/svtav1_hip.h: a `_hip` wrapper touched the device after writing caller memory)
        fprintf(stderr, "SvtAv1HipFixtures: %llu HIP operations were issued after a host form had written its caller's memory\n",
                (unsigned long long)svt_hip_debug_commit_violations());
        return 5;
    }
    return rc;
}
