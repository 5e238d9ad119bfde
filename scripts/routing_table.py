#!/usr/bin/env python3
"""The routing table of the library, asked of the routers themselves (mrs_tg_plan_explain: the launch functions run dry and
note the kernels they would launch).  Prints a markdown table of batch shape x options -> kernels; DESIGN.md section 4 holds its
output for an MI355X, tests/test_gpu_routing.py pins the rows of the BASELINE configs and of the nodelet's defaults.

    python scripts/routing_table.py > profiles/round6_routing_table.md      (needs the GPU: a plan lives on a device)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrs_uav_trajectory_generation_amd import api  # noqa: E402


def uniform(n, S):
    return (np.arange(n + 1, dtype=np.int64) * S).astype(np.int32)


def ragged(n):
    # BASELINE configs[4]: S_p = 3 + (hash(p) mod 28) in [3, 30]; any spread over 3..30 gives the same route
    S = 3 + (np.arange(n) * 2654435761 % 2 ** 32 >> 7) % 28
    return np.concatenate([[0], np.cumsum(S)]).astype(np.int32)


SHAPES = [("1 x 3 (configs[0])", lambda: uniform(1, 3)), ("1 x 10 (one request)", lambda: uniform(1, 10)),
          ("1 x 80 (one subdivided request)", lambda: uniform(1, 80)),
          ("1024 x 10 (configs[1], [2])", lambda: uniform(1024, 10)), ("1024 x 14", lambda: uniform(1024, 14)),
          ("4096 x 10", lambda: uniform(4096, 10)),
          ("8192 x 10 (a shard of configs[3])", lambda: uniform(8192, 10)), ("8192 ragged 3..30 (configs[4])", lambda: ragged(8192)),
          ("65536 x 10 (configs[3])", lambda: uniform(65536, 10)), ("300 x 80", lambda: uniform(300, 80)),
          ("4 x 200", lambda: uniform(4, 200))]

MEL = dict(time_alloc_method=api.TIME_ALLOC_MELLINGER, estimate_times=1, sampling_dt=0.2, sample_capacity=512)
OPTIONS = [("fixed times, min-snap", dict(derivative_to_optimize=4)),
           ("fixed times, min-snap, materialised blocks", dict(derivative_to_optimize=4, flags=api.FLAG_MATERIALIZED_BLOCKS)),
           ("Mellinger + sampling, min-snap", dict(derivative_to_optimize=4, **MEL)),
           ("Mellinger + sampling, min-acceleration (the nodelet's default)", dict(derivative_to_optimize=2, **MEL)),
           ("Mellinger + sampling, min-snap, stop_at / moving-start hint", dict(derivative_to_optimize=4, flags=api.FLAG_CONSTRAINED_SLOTS, **MEL)),
           ("squared-time search (mode 0) + sampling", dict(derivative_to_optimize=4, time_alloc_method=0, estimate_times=1, sampling_dt=0.2,
                                                           sample_capacity=512))]


def routes(ctx):
    out = []
    for sname, make in SHAPES:
        plan = api.Plan(ctx, make())
        for oname, kw in OPTIONS:
            out.append((sname, oname, plan.explain(api.default_options(**kw))))
        if plan.max_segments <= 15:
            opt = api.default_options(derivative_to_optimize=4)
            out.append((sname, "fixed times, min-snap, grouped dispatch of 10 batches", plan.explain(opt, group_size=10)))
        plan.close()
    return out


def compress(names):
    """kernel names in first-launch order; a kernel launched several times (the iterations of a gradient-free search) once, with its count"""
    seen, order = {}, []
    for n in names:
        if n not in seen:
            order.append(n)
        seen[n] = seen.get(n, 0) + 1
    return ["`%s`%s" % (n, " (x %d%s)" % (seen[n], "+" if len(names) >= 32 else "") if seen[n] > 1 else "") for n in order]


def main():
    ctx = api.Context(0)
    print("| batch | options | kernels, in launch order |")
    print("|---|---|---|")
    for sname, oname, names in routes(ctx):
        print("| %s | %s | %s |" % (sname, oname, " → ".join(compress(names))))
    ctx.close()


if __name__ == "__main__":
    main()
