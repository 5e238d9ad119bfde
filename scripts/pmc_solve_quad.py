"""Launch the library's saturated-device solve (solve_quad_kernel, 65536 x 10 random paths, fixed times) a few times for a
rocprofv3 counter pass:

    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmcqw -- python3 scripts/pmc_solve_quad.py
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmcqf -- python3 scripts/pmc_solve_quad.py

(separate passes: the two counters do not fit one TCC pass, MI355X_MICROARCH.md "rocprofv3 PMC slots")."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr

P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = api.Context(0)
ctx.use_torch_stream()
batch = pr.random_batch(P, 10, seed0=0)
plan = api.Plan(ctx, batch.seg_offsets)
db = api.DeviceBatch(batch, "cuda:0")
est = api.default_options(derivative_to_optimize=4, estimate_times=1)
plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints, limits=db.limits)
# vertex positions from the value array (solve_quad_kernel<false>) and, under MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS, from the compact
# waypoint array (solve_quad_kernel<true>): the counter summaries keep the two kernels apart by name
for flags in (0, api.FLAG_POSITIONS_ARE_WAYPOINTS):
    opt = api.default_options(derivative_to_optimize=4, flags=flags)
    for _ in range(5):
        plan.solve(opt, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost, waypoints=db.waypoints)
    torch.cuda.synchronize()
plan.close()
