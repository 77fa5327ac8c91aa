"""N > 1 path on CPU (gloo, world size 2): every rank searches its own shard of the (SB, ref) work list -- no data-path
collective -- and only the barrier / max-over-ranks timing reduction and a result gather (for the check) use the process group.
The union of the shards must equal the single-process result."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent('''
    import os, sys, time
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
    from conftest import EmuBackend
    from test_sad import run_me_batch
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
    rank, world = dist.get_rank(), dist.get_world_size()
    be = EmuBackend()
    g = np.random.default_rng(7)
    n, stride, rows = 6, 64 * 6 + 120, 64 + 24
    src = g.integers(0, 256, (rows, stride), dtype=np.uint8)
    ref = g.integers(0, 256, (rows, stride), dtype=np.uint8)
    descs = np.zeros(n, dtype=be.pkg.MeSearchDesc)
    for i in range(n):
        descs[i] = (i * 64, i * 64 + 2, stride, stride, -8, -4, 16, 9)
    lo, hi = be.pkg.shard_range(n, rank, world)
    dist.barrier()
    t0 = time.perf_counter()
    bs, bm = run_me_batch(be, src, ref, descs[lo:hi], 16, 9, 0)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # the bench's max-over-ranks timing
    full = [None, None]
    dist.all_gather_object(full, (lo, hi, bs, bm))              # test-only gather
    if rank == 0:
        all_bs, all_bm = run_me_batch(be, src, ref, descs, 16, 9, 0)
        for (a, b, s, m) in full:
            assert np.array_equal(all_bs[a:b], s) and np.array_equal(all_bm[a:b], m)
        assert sorted((a, b) for (a, b, _, _) in full) == [(0, 3), (3, 6)] and t.item() > 0
        print("DIST_OK")
    dist.destroy_process_group()
''')


def test_frame_sharding_world_size_2():
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(o[1][-2000:] for o in outs)
    assert "DIST_OK" in outs[0][0]
