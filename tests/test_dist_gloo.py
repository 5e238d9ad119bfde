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
