"""Host-side problem construction: waypoints -> per-vertex constraints, synthetic batches.

Mirrors the input side of the reference's solver adapter (all citations relative to /root/reference/):
  * vertex construction with sequential heading unwrap, makeStartOrEnd end vertices, position-only
    interior vertices, stop_at vertices and the optional initial state:
    src/mrs_trajectory_generation.cpp:923-977, src/eth_trajectory_generation/vertex.cpp:134-163
  * the box-uniform random-vertex generator: src/eth_trajectory_generation/vertex.cpp:33-84
  * the random-walk path generator of PathRandomFlier: src/path_random_flier.cpp:317-351
  * the integration tests' 4-waypoint path: test/service_fly_now/test.cpp:29-32

Pure numpy; no GPU, no oracle.  The arrays produced here are exactly the C-ABI's inputs
(include/mrs_tg.h).
"""
from dataclasses import dataclass
import math

import numpy as np

N_COEFF = 10
N_DIM = 4
N_SLOT = 5

ACCELERATION, JERK, SNAP = 2, 3, 4

# v, a, j x {horizontal, vertical, heading}.  2.0 m/s and 2.0 m/s^2 are the only numeric limits the
# reference mentions (config/private/path_random_flier.yaml:30-33); the rest is this project's choice
# (SURVEY.md section 8d).
DEFAULT_LIMITS = np.array([2.0, 2.0, 1.0, 2.0, 2.0, 2.0, 20.0, 20.0, 20.0])

CONFIG1_WAYPOINTS = np.array([[-5.0, -5.0, 5.0, 1.0], [-5.0, 5.0, 5.0, 2.0],
                              [5.0, -5.0, 5.0, 3.0], [5.0, 5.0, 5.0, 4.0]])


class SplitMix64:
    """Explicit PRNG -> double mapping so every consumer (numpy, C, HIP) can regenerate the inputs."""

    MASK = (1 << 64) - 1

    def __init__(self, seed):
        self.state = seed & self.MASK

    def next_u64(self):
        self.state = (self.state + 0x9E3779B97F4A7C15) & self.MASK
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.MASK
        return z ^ (z >> 31)

    def uniform(self, lo, hi):
        u = (self.next_u64() >> 11) * (1.0 / 9007199254740992.0)  # 53 bits -> [0, 1)
        return lo + (hi - lo) * u


def _wrap_pi(a):
    r = math.fmod(a + math.pi, 2.0 * math.pi)
    if r < 0:
        r += 2.0 * math.pi
    return r - math.pi


def unwrap_heading(what, frm):
    """mrs_lib sradians::unwrap as used at src/mrs_trajectory_generation.cpp:935."""
    d = _wrap_pi(what) - _wrap_pi(frm)
    if d < -math.pi:
        d += 2.0 * math.pi
    elif d >= math.pi:
        d -= 2.0 * math.pi
    return frm + d


def build_vertices(waypoints, derivative_to_optimize=SNAP, stop_at=None, initial_state=None):
    """Waypoints [V][4] -> (unwrapped waypoints, fixed_mask [V][5] u8, fixed_values [V][5][4]).

    initial_state: optional dict(heading=, velocity=(4,), acceleration=(4,), jerk=(4,)) where the
    4th component is the heading rate / acceleration / jerk (src/mrs_trajectory_generation.cpp:946-957).
    """
    wp = np.array(waypoints, dtype=np.float64).reshape(-1, N_DIM).copy()
    V = wp.shape[0]
    assert V >= 2
    d = int(derivative_to_optimize)
    mask = np.zeros((V, N_SLOT), dtype=np.uint8)
    vals = np.zeros((V, N_SLOT, N_DIM))
    last = initial_state["heading"] if initial_state is not None else wp[0, 3]
    for i in range(V):
        wp[i, 3] = unwrap_heading(wp[i, 3], last)
        last = wp[i, 3]
        mask[i, 0] = 1
        vals[i, 0] = wp[i]
        if i == 0 or i == V - 1:
            mask[i, 1:d + 1] = 1  # makeStartOrEnd: derivatives 1..d = 0
            if i == 0 and initial_state is not None:
                for k, key in ((1, "velocity"), (2, "acceleration"), (3, "jerk")):
                    mask[i, k] = 1
                    vals[i, k] = np.asarray(initial_state[key], dtype=np.float64)
        elif stop_at is not None and stop_at[i]:
            mask[i, 1:4] = 1  # vel = acc = jerk = 0
    return wp, mask, vals


@dataclass
class Batch:
    """A batch of paths in the C-ABI's CSR layout."""
    seg_offsets: np.ndarray   # int32 [P+1]
    waypoints: np.ndarray     # f64 [sum V][4]  (headings already unwrapped)
    fixed_mask: np.ndarray    # u8  [sum V][5]
    fixed_values: np.ndarray  # f64 [sum V][5][4]
    limits: np.ndarray        # f64 [P][9]
    derivative_to_optimize: int = SNAP

    @property
    def n_paths(self):
        return self.seg_offsets.size - 1

    @property
    def n_segments(self):
        return int(self.seg_offsets[-1])

    def vertex_range(self, p):
        s0 = int(self.seg_offsets[p])
        s1 = int(self.seg_offsets[p + 1])
        return s0 + p, s1 + p + 1

    def path(self, p):
        v0, v1 = self.vertex_range(p)
        return self.waypoints[v0:v1], self.fixed_mask[v0:v1], self.fixed_values[v0:v1]

    def select(self, idx):
        idx = list(idx)
        parts = [self.path(p) for p in idx]
        return assemble_batch(parts, self.limits[idx], self.derivative_to_optimize)


def assemble_batch(parts, limits, derivative_to_optimize=SNAP):
    segs = [p[0].shape[0] - 1 for p in parts]
    so = np.zeros(len(parts) + 1, dtype=np.int32)
    so[1:] = np.cumsum(segs)
    return Batch(so, np.ascontiguousarray(np.concatenate([p[0] for p in parts])),
                 np.ascontiguousarray(np.concatenate([p[1] for p in parts])),
                 np.ascontiguousarray(np.concatenate([p[2] for p in parts])),
                 np.ascontiguousarray(np.asarray(limits, dtype=np.float64).reshape(len(parts), 9)),
                 int(derivative_to_optimize))


def random_box_waypoints(n_seg, seed):
    """Generator G1: uniform in x,y in [-10,10], z in [1,10], heading in [-pi,pi];
    re-draw while the 4-D distance to the previous vertex is <= 0.2 (vertex.cpp:62-76)."""
    rng = SplitMix64(seed)
    lo = (-10.0, -10.0, 1.0, -math.pi)
    hi = (10.0, 10.0, 10.0, math.pi)
    pts = []
    while len(pts) < n_seg + 1:
        cand = [rng.uniform(lo[k], hi[k]) for k in range(N_DIM)]
        if pts and math.sqrt(sum((a - b) ** 2 for a, b in zip(cand, pts[-1]))) <= 0.2:
            continue
        pts.append(cand)
    return np.array(pts)


def random_walk_waypoints(n_seg, seed):
    """Generator G2: PathRandomFlier's random walk (src/path_random_flier.cpp:317-351,
    tmux/dynamic_test/config/path_random_flier.yaml): bearing += U(-0.4,0.4), step U(0.5,2.0) m,
    z = 5 +- 0.1, heading = bearing."""
    rng = SplitMix64(seed ^ 0x5A5A5A5A)
    x = y = 0.0
    bearing = rng.uniform(-math.pi, math.pi)
    pts = [[x, y, 5.0, bearing]]
    for _ in range(n_seg):
        bearing += rng.uniform(-0.4, 0.4)
        dist = rng.uniform(0.5, 2.0)
        x += math.cos(bearing) * dist
        y += math.sin(bearing) * dist
        pts.append([x, y, 5.0 + rng.uniform(-0.1, 0.1), bearing])
    return np.array(pts)


def ragged_segment_count(p):
    """Config 5: S_p = 3 + (hash(p) mod 28) in [3, 30]."""
    return 3 + SplitMix64(p * 2654435761 + 12345).next_u64() % 28


def random_batch(n_paths, n_seg=10, *, seed0=0, derivative_to_optimize=SNAP, generator="box", limits=None):
    """Path p uses its own stream seeded seed0 + p.  n_seg: int, or 'ragged' (config 5)."""
    gen = random_box_waypoints if generator == "box" else random_walk_waypoints
    parts = []
    for p in range(n_paths):
        S = ragged_segment_count(seed0 + p) if n_seg == "ragged" else int(n_seg)
        parts.append(build_vertices(gen(S, seed0 + p), derivative_to_optimize))
    lim = np.tile(DEFAULT_LIMITS if limits is None else np.asarray(limits, dtype=np.float64), (n_paths, 1))
    return assemble_batch(parts, lim, derivative_to_optimize)


def random_mixed_batch(n_paths, derivative_to_optimize=SNAP, seed0=0, max_segments=30):
    """Every constraint pattern the adapter produces, mixed in one batch: 1..max_segments segments, both waypoint generators,
    stop_at interior vertices (src/mrs_trajectory_generation.cpp:959-966), non-zero initial states (:946-957), limits
    scaled by 0.3..3 per path."""
    parts, lims = [], []
    for p in range(seed0, seed0 + n_paths):
        rng = SplitMix64(0xABCDEF + p)
        S = 1 + rng.next_u64() % max_segments
        wp = (random_box_waypoints if rng.next_u64() % 2 else random_walk_waypoints)(S, p)
        stop = [rng.next_u64() % 4 == 0 for _ in range(S + 1)]
        init = None
        if rng.next_u64() % 2:
            init = dict(heading=wp[0, 3] + rng.uniform(-0.5, 0.5),
                        velocity=[rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-1, 1), rng.uniform(-0.5, 0.5)],
                        acceleration=[rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-0.5, 0.5), rng.uniform(-0.3, 0.3)],
                        jerk=[rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-0.5, 0.5), rng.uniform(-0.3, 0.3)])
        parts.append(build_vertices(wp, derivative_to_optimize, stop_at=stop, initial_state=init))
        lims.append(DEFAULT_LIMITS * rng.uniform(0.3, 3.0))
    return assemble_batch(parts, np.array(lims), derivative_to_optimize)


def config1_batch(derivative_to_optimize=SNAP):
    """The reference tests' 4-waypoint path (BASELINE.json configs[0])."""
    return assemble_batch([build_vertices(CONFIG1_WAYPOINTS, derivative_to_optimize)], DEFAULT_LIMITS[None],
                          derivative_to_optimize)
