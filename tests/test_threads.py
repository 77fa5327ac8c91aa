"""Drop-in boundary, threading: the RTCD pointers are read lock-free by many encoder threads at once, so the `_hip` symbols must be
re-entrant -- per-thread stream, device arena and pinned staging, the device bound per thread.  Worker threads call different symbols
concurrently (ctypes drops the GIL for the duration of a foreign call) and every result is compared with the oracle's."""
import ctypes as C
import threading

import numpy as np
import pytest

from conftest import p, rng
from test_txfm import oracle_fwd

pytestmark = pytest.mark.gpu  # the CPU SIMT interpreter keeps global fiber state: single-threaded by construction


def test_symbols_from_concurrent_threads(be, oracle):
    if not be.is_gpu:
        pytest.skip("needs the real device")
    lib, errors = be.lib, []
    oracle.oracle_sad_nxm.restype = C.c_uint32
    lib.svt_nxm_sad_kernel_hip.restype = C.c_uint32

    def sad_worker(seed):
        g = rng(seed)
        for it in range(40):
            w, h = [(64, 64), (32, 16), (16, 16), (8, 8), (128, 128)][it % 5]
            src, ref = g.integers(0, 256, (h, w + 7), dtype=np.uint8), g.integers(0, 256, (h, w + 3), dtype=np.uint8)
            got = lib.svt_nxm_sad_kernel_hip(p(src), w + 7, p(ref), w + 3, h, w)
            want = oracle.oracle_sad_nxm(p(src), w + 7, p(ref), w + 3, h, w)
            if got != want:
                errors.append(("sad", seed, it, got, want))

    def txfm_worker(seed):
        g = rng(seed)
        for it in range(25):
            res = g.integers(-255, 256, 16 * 18).astype(np.int16)
            out = np.zeros(256, np.int32)
            lib.svt_av1_fwd_txfm2d_16x16_hip(p(res), p(out), 18, it % 4, 8)
            if not np.array_equal(out, oracle_fwd(oracle, res, 18, it % 4, 2, 8)):
                errors.append(("txfm", seed, it))

    def satd_worker(seed):
        g = rng(seed)
        for it in range(40):
            n = [16, 64, 256, 1024][it % 4]
            a = g.integers(-32640, 32641, n).astype(np.int32)
            if lib.svt_aom_satd_hip(p(a), n) != oracle.oracle_satd(p(a), n):
                errors.append(("satd", seed, it))

    def noise_worker(seed):
        g = rng(seed)
        for it in range(15):
            a = np.clip(100 + g.integers(-6, 7, (40, 72)), 0, 255).astype(np.uint8)
            if lib.svt_estimate_noise_fp16_hip(p(a), 64, 40, 72) != oracle.oracle_estimate_noise_fp16(p(a), 64, 40, 72, 8):
                errors.append(("noise", seed, it))

    workers = [sad_worker, txfm_worker, satd_worker, noise_worker, sad_worker, txfm_worker, satd_worker, noise_worker]
    threads = [threading.Thread(target=w, args=(4000 + i,)) for i, w in enumerate(workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive(), "worker hung"
    assert not errors, errors[:5]
