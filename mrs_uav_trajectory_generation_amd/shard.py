"""Multi-GPU sharding of a batch of independent paths, and the one collective of the job.

Paths are independent units (one QP + outer loop each; the reference itself handles one path per
request, /root/reference/src/mrs_trajectory_generation.cpp:1064-1083), so a batch shards with no
data-path collective: every rank solves its own contiguous range (uniform batches) or a greedily balanced
subset (ragged batches), and the only exchange is the final gather of coefficients / times / status to
rank 0.  Works with any torch.distributed backend ("nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).
"""
import numpy as np


def contiguous_shard(n_paths, rank, world):
    """[begin, end) of the paths owned by `rank`: sizes differ by at most one."""
    base, rem = divmod(int(n_paths), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def balanced_shard(seg_counts, world):
    """Ragged batches: longest-processing-time-first greedy on the segment count.
    Returns a list (one entry per rank) of sorted path-index arrays."""
    seg_counts = np.asarray(seg_counts)
    order = np.argsort(-seg_counts, kind="stable")
    load = np.zeros(world, dtype=np.int64)
    owner = [[] for _ in range(world)]
    for p in order:
        r = int(np.argmin(load))
        owner[r].append(int(p))
        load[r] += int(seg_counts[p])
    return [np.array(sorted(o), dtype=np.int64) for o in owner]


def gather_to_root(tensor, dist, dst=0, bufs=None):
    """Gather equally-shaped per-rank tensors on `dst`; returns the list there, None elsewhere.
    `bufs` lets the caller reuse receive buffers across calls (root only)."""
    world = dist.get_world_size()
    if dist.get_rank() == dst:
        if bufs is None:
            import torch
            bufs = [torch.empty_like(tensor) for _ in range(world)]
    else:
        bufs = None
    dist.gather(tensor, bufs, dst=dst)
    return bufs


def gather_ragged_to_root(tensor, dist, dst=0):
    """Gather per-rank tensors whose leading dimension differs (ragged shards): sizes first, then padded
    payloads; returns the list of correctly sized tensors on `dst`, None elsewhere."""
    import torch
    world = dist.get_world_size()
    n = torch.tensor([tensor.shape[0]], dtype=torch.int64, device=tensor.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    padded = torch.zeros((cap,) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
    padded[:tensor.shape[0]] = tensor
    bufs = gather_to_root(padded, dist, dst)
    if bufs is None:
        return None
    return [b[:s] for b, s in zip(bufs, sizes)]


# ---- the packed result of a shard and its gather (bench.py's configs[3] job; tests/test_dist_gloo.py runs the same code on gloo)

def packed_doubles(n_paths, n_segments):
    """doubles of one shard's packed result: coefficients [n_segments][4][10] | segment times [n_segments] | status [n_paths]
    (int32 carried as doubles so that the job's one collective moves one buffer)"""
    return int(n_segments) * 41 + int(n_paths)


def packed_views(packed, n_paths, n_segments):
    """(coeffs [n_segments][4][10], times [n_segments], status-as-double [n_paths]) views into a packed buffer"""
    nS = int(n_segments)
    return packed[:nS * 40].view(nS, 4, 10), packed[nS * 40:nS * 41], packed[nS * 41:nS * 41 + int(n_paths)]


def shard_capacity(total_paths, world):
    """paths of the largest contiguous shard: what every rank pads its packed result to, so that an uneven cut (65535 paths
    over 4 ranks: 16384, 16384, 16384, 16383) still travels in ONE equally-shaped gather"""
    return (int(total_paths) + int(world) - 1) // int(world)


def gather_packed_shards(packed, pad, dist, bufs=None, via_host=False):
    """Copy a rank's packed result into its padded send buffer `pad` (sized for the largest shard) and gather the padded
    buffers on rank 0.  Returns the receive list on rank 0, None elsewhere.  via_host: the collective runs on host copies
    (gloo ranks that share one GPU)."""
    pad[:packed.numel()].copy_(packed)
    if via_host:
        return gather_to_root(pad.cpu(), dist)
    return gather_to_root(pad, dist, bufs=bufs)


def unpack_gathered_shards(bufs, total_paths, world, segments_per_path):
    """rank 0: the per-rank receive buffers of gather_packed_shards -> (coeffs, times, status int32) of the WHOLE batch in path
    order (uniform batches, contiguous shards: rank r's slice holds the paths contiguous_shard(total, r, world))"""
    import torch
    cs, ts, ss = [], [], []
    for r in range(world):
        a, b = contiguous_shard(total_paths, r, world)
        c, t, s = packed_views(bufs[r], b - a, (b - a) * segments_per_path)
        cs.append(c)
        ts.append(t)
        ss.append(s.to(torch.int32))
    return torch.cat(cs), torch.cat(ts), torch.cat(ss)
