"""The N > 1 path on CPU: two gloo ranks shard a batch, each solves its shard (with the oracle standing
in for the device), results are gathered on rank 0 exactly as bench.py does over RCCL, and the gathered
result equals the unsharded solve."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ragged, result_file):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from mrs_uav_trajectory_generation_amd import problem as pr, shard
    from oracle import pyoracle as po
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_paths = 13
    batch = pr.random_batch(n_paths, "ragged" if ragged else 6, seed0=100)
    counts = np.diff(batch.seg_offsets)
    if ragged:
        mine = shard.balanced_shard(counts, world)[rank]
    else:
        a, b = shard.contiguous_shard(n_paths, rank, world)
        mine = np.arange(a, b)
    sub = batch.select(mine)
    out = po.solve_batch(sub.seg_offsets, sub.waypoints, sub.fixed_mask, sub.fixed_values, sub.limits,
                         np.zeros(sub.n_segments), estimate_times=True)
    g_c = shard.gather_ragged_to_root(torch.from_numpy(out["coeffs"]), dist)
    g_t = shard.gather_ragged_to_root(torch.from_numpy(out["times"]), dist)
    g_s = shard.gather_ragged_to_root(torch.from_numpy(out["status"]), dist)
    g_i = shard.gather_ragged_to_root(torch.from_numpy(np.asarray(mine, dtype=np.int64)), dist)
    if rank == 0:
        full = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                              np.zeros(batch.n_segments), estimate_times=True)
        ok = True
        seen = []
        for r in range(world):
            idx = g_i[r].numpy()
            seen += idx.tolist()
            off = 0
            for j, p in enumerate(idx):
                a, b = batch.seg_offsets[p], batch.seg_offsets[p + 1]
                n = b - a
                ok &= np.array_equal(g_c[r].numpy()[off:off + n], full["coeffs"][a:b])
                ok &= np.array_equal(g_t[r].numpy()[off:off + n], full["times"][a:b])
                ok &= int(g_s[r][j]) == int(full["status"][p])
                off += n
        ok &= sorted(seen) == list(range(n_paths))
        with open(result_file, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ragged", [False, True])
def test_two_rank_shard_and_gather(tmp_path, ragged):
    import torch.multiprocessing as mp
    result = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), ragged, str(result)), nprocs=2, join=True)
    assert result.read_text() == "ok"


def _worker_config3(rank, world, port, total, n_seg, result_file):
    """bench.py's configs[3] job on CPU: ONE fixed batch cut into contiguous shards over the ranks (an UNEVEN cut), every rank
    packs coefficients | times | status into one buffer, pads it to the largest shard and the padded buffers travel in one
    gather (shard.gather_packed_shards -- the function bench.py calls); rank 0 unpacks and compares with the unsharded solve."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from mrs_uav_trajectory_generation_amd import problem as pr, shard
    from oracle import pyoracle as po
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard.contiguous_shard(total, rank, world)
    n = b - a
    batch = pr.random_batch(n, n_seg, seed0=a)          # path p of the job is seeded with p, whatever the number of ranks
    out = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                         np.zeros(batch.n_segments), estimate_times=True, n_threads=2)
    nS = batch.n_segments
    packed = torch.zeros(shard.packed_doubles(n, nS), dtype=torch.float64)
    c, t, s = shard.packed_views(packed, n, nS)
    c.copy_(torch.from_numpy(out["coeffs"]))
    t.copy_(torch.from_numpy(out["times"]))
    s.copy_(torch.from_numpy(out["status"]).to(torch.float64))
    cap = shard.shard_capacity(total, world)
    pad = torch.zeros(shard.packed_doubles(cap, cap * n_seg), dtype=torch.float64)
    bufs = shard.gather_packed_shards(packed, pad, dist)
    sizes = [shard.contiguous_shard(total, r, world) for r in range(world)]
    if rank == 0:
        ok = len({e - b0 for b0, e in sizes}) == 2 and sizes[-1][1] == total and sum(e - b0 for b0, e in sizes) == total   # uneven, complete
        gc, gt, gs = shard.unpack_gathered_shards(bufs, total, world, n_seg)
        # the unsharded job, rank by rank's seeds: the same paths in the same order
        full = pr.random_batch(total, n_seg, seed0=0)
        ref = po.solve_batch(full.seg_offsets, full.waypoints, full.fixed_mask, full.fixed_values, full.limits,
                             np.zeros(full.n_segments), estimate_times=True, n_threads=8)
        ok &= gc.shape[0] == total * n_seg and gs.shape[0] == total
        ok &= np.array_equal(gc.numpy(), ref["coeffs"]) and np.array_equal(gt.numpy(), ref["times"])
        ok &= np.array_equal(gs.numpy(), ref["status"]) and bool((gs == 1).all())
        with open(result_file, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_four_rank_uneven_contiguous_shards_with_the_packed_gather(tmp_path):
    """65536 divides by 4; 65535 does not: ranks 0-2 own 16384 paths, rank 3 owns 16383 -- the remainder logic of
    shard.contiguous_shard together with the padded, packed gather of bench.py's configs[3]"""
    import torch.multiprocessing as mp
    result = tmp_path / "result.txt"
    mp.spawn(_worker_config3, args=(4, _free_port(), 65535, 3, str(result)), nprocs=4, join=True)
    assert result.read_text() == "ok"


class _NoGpuCuda:
    """stand-in for torch.cuda inside bench.time_regions on a CPU box: the synchronize is the step function's own return"""
    @staticmethod
    def synchronize():
        return None


class _TorchOnCpu:
    def __init__(self, torch):
        self._torch = torch
        self.cuda = _NoGpuCuda()

    def __getattr__(self, name):
        return getattr(self._torch, name)


def _worker_regions(rank, world, port, result_file):
    """bench.py's repeated timed regions on four gloo ranks: the log of every rank must read
    (barrier -> K steps -> own clock) x R -> closing gather, the figures returned are the MAX over ranks per region, and
    every rank returns the same figures (round-5 VERDICT item 9)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import torch
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []
    real_barrier = dist.barrier

    class _Dist:
        """torch.distributed with the barrier logged"""
        def __getattr__(self, name):
            return getattr(dist, name)

        @staticmethod
        def barrier():
            log.append("barrier")
            real_barrier()

    K, R, W = 5, 7, 2
    # rank r's step takes (1 + r) ms, and region 3 of rank 1 is disturbed (+ 30 ms): the MAX over ranks must carry both
    region_no = [-1]
    n_in_region = [0]

    def step():
        log.append("step")
        time.sleep(1e-3 * (1 + rank))
        n_in_region[0] += 1
        if rank == 1 and len([x for x in log if x == "barrier"]) == 4 and n_in_region[0] % K == 1:
            time.sleep(0.030)

    def gather():
        log.append("gather")
        t = torch.full((4,), float(rank), dtype=torch.float64)
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, bufs, dst=0)

    regions, gather_s = bench.time_regions(step, K, W, _Dist(), _TorchOnCpu(torch), regions=R, final_fn=gather)
    own = list(bench.LAST_OWN_REGIONS[0])
    # the order on this rank: W warm-up steps, the gather's warm-up, then R x (barrier, K steps), the timed gather, a closing barrier
    expect = ["step"] * W + ["gather"]
    for _ in range(R):
        expect += ["barrier"] + ["step"] * K
    expect += ["gather", "barrier"]
    ok = log == expect
    ok &= len(regions) == R and len(own) == R and gather_s > 0
    # MAX over ranks: every region at least the slowest rank's K steps (4 ms each), and no rank's own figure above it
    ok &= all(r >= K * 4e-3 * 0.95 for r in regions)
    ok &= all(o <= r + 1e-9 for o, r in zip(own, regions))
    # the disturbed region is visible in the per-region figures and absent from the median
    ok &= regions[3] >= 0.030 and bench.median_of(regions) < 0.030 + K * 4e-3
    allr = [None] * world
    dist.all_gather_object(allr, (regions, gather_s))
    ok &= all(a == allr[0] for a in allr)       # one figure for the job, on every rank
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        with open(result_file, "w") as f:
            f.write("ok" if all(flags) else "mismatch %r %r" % (flags, regions))
    real_barrier()
    dist.destroy_process_group()


def test_four_rank_repeated_timed_regions_keep_the_barrier_steps_clock_gather_order(tmp_path):
    import torch.multiprocessing as mp
    result = tmp_path / "result.txt"
    mp.spawn(_worker_regions, args=(4, _free_port(), str(result)), nprocs=4, join=True)
    assert result.read_text() == "ok"


def test_stream_pair_choice_takes_the_shortest_round_and_reports_every_pair():
    """bench.py's calibration of the grouped issue path's two streams (round 6): every pair of the candidates is measured once,
    the shortest two-dispatch round wins, ties go to the first pair"""
    sys.path.insert(0, ROOT)
    import bench
    us = {(0, 1): 67.0, (0, 2): 60.5, (0, 3): 49.2, (1, 2): 49.2, (1, 3): 57.0, (2, 3): 66.0}
    seen = []

    def measure(a, b):
        seen.append((a, b))
        return us[(a, b)]
    best, table = bench.pick_stream_pair(4, measure)
    assert best == (0, 3) and table == us and seen == sorted(us)
    best2, table2 = bench.pick_stream_pair(2, lambda a, b: 55.0)
    assert best2 == (0, 1) and table2 == {(0, 1): 55.0}
