"""Host-side problem construction (no GPU, no oracle needed): vertex layout, heading unwrap, CSR batches."""
import math

import numpy as np

from mrs_uav_trajectory_generation_amd import problem as pr, shard


def test_build_vertices_min_snap_layout():
    wp, m, v = pr.build_vertices(pr.CONFIG1_WAYPOINTS, pr.SNAP)
    # S = 3 -> 12 fixed, 8 free
    assert m.sum() == 12 and m.size - m.sum() == 8
    assert m[0].tolist() == [1, 1, 1, 1, 1] and m[1].tolist() == [1, 0, 0, 0, 0] and m[3].tolist() == [1, 1, 1, 1, 1]
    assert np.array_equal(v[:, 0], wp) and np.all(v[:, 1:] == 0)


def test_build_vertices_variants():
    wp = pr.random_box_waypoints(6, 21)
    init = dict(heading=0.3, velocity=[0.5, -0.2, 0.1, 0.05], acceleration=[0.1, 0.0, -0.1, 0.0], jerk=[0.0, 0.2, 0.0, 0.0])
    stop = [False, False, False, True, False, False, False]
    free = {}
    for d in (2, 3, 4):
        _, m, v = pr.build_vertices(wp, d, stop_at=stop, initial_state=init)
        free[d] = (5 - m.sum(axis=1)).tolist()
        assert np.allclose(v[0, 1], init["velocity"]) and np.allclose(v[0, 3], init["jerk"])
    # SURVEY.md A.4 probe: free-per-vertex for d = 2 is [1,4,4,1,4,4,2]; ends shrink with d
    assert free[2] == [1, 4, 4, 1, 4, 4, 2]
    assert free[3] == [1, 4, 4, 1, 4, 4, 1]
    assert free[4] == [0, 4, 4, 1, 4, 4, 0]


def test_heading_unwrap_is_sequential():
    wp = np.array([[0, 0, 0, 3.0], [1, 0, 0, -3.0], [2, 0, 0, 3.1], [3, 0, 0, -3.1]], dtype=float)
    out, _, _ = pr.build_vertices(wp, pr.SNAP)
    assert np.all(np.abs(np.diff(out[:, 3])) < math.pi)
    assert abs(out[1, 3] - (-3.0 + 2 * math.pi)) < 1e-12
    assert abs(pr.unwrap_heading(3.5, -3.0) - (3.5 - 2 * math.pi)) < 1e-12


def test_generators_are_deterministic_and_in_range():
    a, b = pr.random_box_waypoints(10, 5), pr.random_box_waypoints(10, 5)
    assert np.array_equal(a, b) and a.shape == (11, 4)
    assert np.all(np.abs(a[:, :2]) <= 10) and np.all((a[:, 2] >= 1) & (a[:, 2] <= 10))
    assert np.all(np.linalg.norm(np.diff(a, axis=0), axis=1) > 0.2)
    w = pr.random_walk_waypoints(10, 5)
    steps = np.linalg.norm(np.diff(w[:, :2], axis=0), axis=1)
    assert np.all((steps >= 0.5 - 1e-12) & (steps <= 2.0 + 1e-12)) and np.all(np.abs(w[:, 2] - 5.0) <= 0.1)
    counts = [pr.ragged_segment_count(p) for p in range(2000)]
    assert min(counts) == 3 and max(counts) == 30


def test_csr_batch_layout_and_select():
    b = pr.random_batch(5, "ragged", seed0=1)
    assert b.seg_offsets[0] == 0 and b.n_segments == b.seg_offsets[-1]
    assert b.waypoints.shape[0] == b.n_segments + b.n_paths
    for p in range(5):
        v0, v1 = b.vertex_range(p)
        assert v1 - v0 == b.seg_offsets[p + 1] - b.seg_offsets[p] + 1
    sub = b.select([3, 1])
    assert np.array_equal(sub.path(0)[0], b.path(3)[0]) and np.array_equal(sub.path(1)[2], b.path(1)[2])


def test_shard_partitions():
    for n, w in ((1024, 8), (10, 3), (7, 8), (0, 2)):
        ranges = [shard.contiguous_shard(n, r, w) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1
    counts = np.array([pr.ragged_segment_count(p) for p in range(8192)])
    parts = shard.balanced_shard(counts, 8)
    assert sorted(np.concatenate(parts).tolist()) == list(range(8192))
    loads = [counts[p].sum() for p in parts]
    assert max(loads) - min(loads) <= 30
