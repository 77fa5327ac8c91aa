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
    n, stride, rows = 4, 64 * 4 + 120, 64 + 24
    src = g.integers(0, 256, (rows, stride), dtype=np.uint8)
    ref = g.integers(0, 256, (rows, stride), dtype=np.uint8)
    descs = np.zeros(n, dtype=be.pkg.MeSearchDesc)
    for i in range(n):
        descs[i] = (i * 64, i * 64 + 2, stride, stride, -4, -1, 8, 3)
    lo, hi = be.pkg.shard_range(n, rank, world)
    dist.barrier()
    t0 = time.perf_counter()
    bs, bm = run_me_batch(be, src, ref, descs[lo:hi], 8, 3, 0)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # the bench's max-over-ranks timing
    full = [None, None]
    dist.all_gather_object(full, (lo, hi, bs, bm))              # test-only gather
    if rank == 0:
        all_bs, all_bm = run_me_batch(be, src, ref, descs, 8, 3, 0)
        for (a, b, s, m) in full:
            assert np.array_equal(all_bs[a:b], s) and np.array_equal(all_bm[a:b], m)
        assert sorted((a, b) for (a, b, _, _) in full) == [(0, 2), (2, 4)] and t.item() > 0
        print("DIST_OK")
    dist.destroy_process_group()
''')


def test_frame_sharding_world_size_2():
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(o[1][-2000:] for o in outs)
    assert "DIST_OK" in outs[0][0]


STRIP_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
    from conftest import EmuBackend
    from test_sad import run_me_batch, oracle_me
    import ctypes as C, bench
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
    rank, world = dist.get_rank(), dist.get_world_size()
    be = EmuBackend()
    W, H, PAD, refs, aw, ah = 192, 200, 40, 2, 16, 9                    # 3 x 4 SBs: strips of 2 and 2 SB rows
    stride, rows = W + 2 * PAD, H + 2 * PAD + 64
    plane = stride * rows
    planes = torch.zeros((1 + refs) * plane, dtype=torch.uint8)
    if rank == 0:
        planes = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (1 + refs) * plane, dtype=np.uint8))
    dist.broadcast(planes, src=0)                                        # source + reference planes to every rank, once
    host = planes.numpy()
    full = be.pkg.me_descs_for_frame(W, H, stride, PAD, PAD, aw, ah, plane, n_refs=refs, src_plane=0, ref_plane0=1)
    sbs_x, sbs_y = (W + 63) // 64, (H + 63) // 64
    r0, r1 = bench.strip_rows(sbs_y, rank, world)
    max_rows = bench.strip_rows(sbs_y, 0, world)[1]
    mine = np.concatenate([full[r * sbs_x * sbs_y + r0 * sbs_x:r * sbs_x * sbs_y + r1 * sbs_x] for r in range(refs)])
    bs, bm = run_me_batch(be, host, host, mine, aw, ah, 0)
    n_pad = refs * max_rows * sbs_x
    local = torch.zeros(2, n_pad, 85, dtype=torch.int32)
    local[0, :len(mine)] = torch.from_numpy(bs.view(np.int32)); local[1, :len(mine)] = torch.from_numpy(bm.view(np.int32))
    gathered = torch.zeros(world * 2 * n_pad * 85, dtype=torch.int32)    # flat, like bench.py's device buffers
    dist.all_gather_into_tensor(gathered, local.reshape(-1))             # the data-path collective of the frame-partition mode
    # every rank now holds the picture-wide tables: reassemble [ref][sb] and compare with the CPU checker
    oracle = C.CDLL(os.path.join(sys.argv[1], "oracle", "liboracle.so"))
    got = gathered.numpy().view(np.uint32).reshape(world, 2, n_pad, 85)
    for rk in range(world):
        q0, q1 = bench.strip_rows(sbs_y, rk, world)
        for rf in range(refs):
            for row in range(q0, q1):
                for col in range(sbs_x):
                    k = rf * (q1 - q0) * sbs_x + (row - q0) * sbs_x + col
                    ws, wm = oracle_me(oracle, host, host, full[rf * sbs_x * sbs_y + row * sbs_x + col], 0)
                    assert np.array_equal(got[rk, 0, k], ws) and np.array_equal(got[rk, 1, k], wm), (rk, rf, row, col)
    assert [bench.strip_rows(17, k, 8) for k in range(8)][:3] == [(0, 3), (3, 5), (5, 7)] and bench.strip_rows(17, 7, 8) == (15, 17)
    print("STRIPS_OK", rank)
    dist.destroy_process_group()
''')


def test_frame_partition_strips_world_size_2():
    """north_star's frame-partition case on CPU (gloo): one picture, SB-row strips per rank, planes broadcast once, all-gather of the per-strip tables."""
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, "-c", STRIP_WORKER, ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-2000:] for o in outs)
    assert "STRIPS_OK" in outs[0][0] and "STRIPS_OK" in outs[1][0]


FILTER_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
    from conftest import EmuBackend, p
    from test_cdef import synth_plane
    from test_oracle_pin_restoration import make_units, unit_grid
    import ctypes as C, bench
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
    rank, world = dist.get_rank(), dist.get_world_size()
    be = EmuBackend()
    oracle = C.CDLL(os.path.join(sys.argv[1], "oracle", "liboracle.so"))
    bd, W, H, us = 10, 136, 200, 64                                       # 3 x 4 filter blocks: strips of 2 and 2 rows; 4 stripes: 2 and 2
    g = np.random.default_rng(5)
    plane = torch.zeros(H * W * 2, dtype=torch.uint8)                     # (gloo moves bytes: the u16 planes travel as uint8 views)
    if rank == 0:
        plane = torch.from_numpy(synth_plane(g, W, H, bd).astype(np.uint16).reshape(-1).view(np.uint8).copy())
    dist.broadcast(plane, src=0)                                          # the deblocked plane on every rank, once
    rec = plane.numpy().view(np.uint16).reshape(H, W).copy()
    nhfb, nvfb = (W + 63) // 64, (H + 63) // 64
    nfb = nhfb * nvfb
    g2 = np.random.default_rng(6)                                         # identical side data on both ranks
    skip = (g2.random((nvfb * 8, nhfb * 8)) < 0.2).astype(np.uint8)
    apri, asec = g2.choice(np.array([0, 4, 9], np.int32), nfb).astype(np.int32), g2.choice(np.array([0, 1, 2, 4], np.int32), nfb).astype(np.int32)
    nstripes = (H + 8 + 63) // 64
    above, below = g2.integers(0, 1 << bd, (2 * nstripes, W)).astype(np.uint16), g2.integers(0, 1 << bd, (2 * nstripes, W)).astype(np.uint16)
    nvu, nhu = unit_grid(W, H, us)
    units = make_units(g2, nvu, nhu, be.pkg.LrUnit)
    # ---- CDEF apply over this rank's strip of filter-block rows, all-gather of the strips
    cdef = rec.copy()
    dirs, var = np.zeros(nfb * 64, np.uint8), np.zeros(nfb * 64, np.int32)
    P = be.pkg.CdefParams(rec.ctypes.data, rec.ctypes.data, cdef.ctypes.data, W, W, W, W, H, 0, 0, 0, 1, bd - 8, 5, 5, 1, 0, skip.ctypes.data, apri.ctypes.data, asec.ctypes.data,
                          dirs.ctypes.data, var.ctypes.data, None)
    r0, r1 = bench.strip_rows(nvfb, rank, world)
    be.lib.svt_hip_cdef_frame_rows(0, C.byref(P), r0, r1, None)
    max_rows = bench.strip_rows(nvfb, 0, world)[1] * 64
    loc = torch.zeros(max_rows * W * 2, dtype=torch.uint8)
    y0, y1 = r0 * 64, min(r1 * 64, H)
    loc[:(y1 - y0) * W * 2] = torch.from_numpy(cdef[y0:y1].reshape(-1).view(np.uint8).copy())
    gat = torch.zeros(world * max_rows * W * 2, dtype=torch.uint8)
    dist.all_gather_into_tensor(gat, loc)
    for k in range(world):
        q0, q1 = bench.strip_rows(nvfb, k, world)
        a0, a1 = q0 * 64, min(q1 * 64, H)
        cdef[a0:a1] = gat.numpy().view(np.uint16)[k * max_rows * W:k * max_rows * W + (a1 - a0) * W].reshape(a1 - a0, W)
    want_c, o_dir, o_var, o_mse = rec.copy(), np.zeros(nfb * 64, np.uint8), np.zeros(nfb * 64, np.int32), np.zeros(1, np.uint64)
    oracle.oracle_cdef_frame(0, p(rec), W, p(rec), W, p(want_c), W, W, H, 0, 0, 0, 1, bd - 8, 5, 5, 1, p(skip), p(apri), p(asec), 0, p(o_dir), p(o_var), p(o_mse))
    assert np.array_equal(cdef, want_c), np.argwhere(cdef != want_c)[:5]
    # ---- loop restoration over this rank's range of stripes (it reads the ASSEMBLED CDEF plane: 3 rows beyond its stripes), all-gather of the rows
    out = np.zeros((H, W), np.uint16)
    L = be.pkg.LrParams(cdef.ctypes.data, above.ctypes.data, below.ctypes.data, out.ctypes.data, W, W, W, W, H, us, 0, 0, 1, bd, units.ctypes.data)
    s0, s1 = bench.strip_rows(nstripes, rank, world)
    be.lib.svt_hip_lr_filter_frame_stripes(C.byref(L), s0, s1, None)
    rows = lambda k: (max(bench.strip_rows(nstripes, k, world)[0] * 64 - 8, 0), min(bench.strip_rows(nstripes, k, world)[1] * 64 - 8, H))
    mx = max(rows(k)[1] - rows(k)[0] for k in range(world))
    loc = torch.zeros(mx * W * 2, dtype=torch.uint8)
    y0, y1 = rows(rank)
    loc[:(y1 - y0) * W * 2] = torch.from_numpy(out[y0:y1].reshape(-1).view(np.uint8).copy())
    gat = torch.zeros(world * mx * W * 2, dtype=torch.uint8)
    dist.all_gather_into_tensor(gat, loc)
    for k in range(world):
        a0, a1 = rows(k)
        out[a0:a1] = gat.numpy().view(np.uint16)[k * mx * W:k * mx * W + (a1 - a0) * W].reshape(a1 - a0, W)
    want_l = np.zeros((H, W), np.uint16)
    oracle.oracle_lr_filter_frame(p(want_c), W, p(above), p(below), W, p(want_l), W, W, H, 0, us, p(units), bd, 1)
    assert np.array_equal(out, want_l), np.argwhere(out != want_l)[:5]
    print("FILTER_STRIPS_OK", rank)
    dist.destroy_process_group()
''')


def test_in_loop_filter_strips_world_size_2():
    """The in-loop filter half of the frame-partition case on CPU (gloo): one plane, CDEF over strips of filter-block rows and loop restoration over ranges of stripes,
    halos read from the replicated input, the filtered strips all-gathered; the assembled planes equal the CPU checker's whole-frame results on every rank."""
    port = str(33500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, "-c", FILTER_WORKER, ROOT, port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-2000:] for o in outs)
    assert "FILTER_STRIPS_OK" in outs[0][0] and "FILTER_STRIPS_OK" in outs[1][0]
