"""The silicon's cross-lane / packed-byte instructions must agree with the C models the CPU interpreter uses
(tests/emu/hipemu.h).  A mismatch here names the misunderstood instruction directly."""
import numpy as np
import pytest

from conftest import EmuBackend, GpuBackend

ROWS = ["v_sad_u8", "v_qsad_pk_u16_u8.lo", "v_qsad_pk_u16_u8.hi", "v_alignbyte_b32", "dpp quad_perm[1,0,3,2]", "dpp quad_perm[2,3,0,1]",
        "dpp row_ror:4", "dpp row_ror:8", "shfl_xor 16", "shfl_xor 32", "dpp row_shr:1", "dpp row_shl:1", "v_permlane16_swap.vdst", "v_permlane16_swap.src0",
        "v_permlane32_swap.vdst", "v_permlane32_swap.src0", "v_dot4_u32_u8"]


def selftest(be):
    out = be.empty(64 * len(ROWS), np.uint32)
    be.lib.svt_hip_selftest(be.ptr(out), be.stream)
    return be.host(out).reshape(len(ROWS), 64)


def test_selftest_emu_runs():
    r = selftest(EmuBackend())
    assert r.shape == (len(ROWS), 64) and r.any()


@pytest.mark.gpu
def test_isa_primitives_match_c_model():
    got, want = selftest(GpuBackend()), selftest(EmuBackend())
    for i, name in enumerate(ROWS):
        assert np.array_equal(got[i], want[i]), "%s differs from the C model: gpu=%s model=%s" % (name, got[i][:8], want[i][:8])
