"""Whole-batch parity sweep against the oracle (all host cores): for every path of a large batch compare stopping
reason, segment times, coefficients and sample count of the nonlinear pipeline; print the agreement statistics and the
worst paths.  usage: parity_sweep.py [n_paths] [n_seg|ragged] [deriv] [generator] [mode]
ORACLE_ARITH=1: the oracle's per-segment matrices from exactly rounded unit-time tables (oracle/mto_linear.c) instead of
the reference's numerically inverted mapping matrix; ORACLE_ARITH=2: its whole linear solve in 113-bit arithmetic -- the
same algorithm without the rounding noise of the reference's route (about 50 times slower)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po

P = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n_seg = sys.argv[2] if len(sys.argv) > 2 else "10"
n_seg = n_seg if n_seg == "ragged" else int(n_seg)
deriv = int(sys.argv[3]) if len(sys.argv) > 3 else 4
gen = sys.argv[4] if len(sys.argv) > 4 else "box"
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 2
ctx = api.Context(0)
arith = int(os.environ.get("ORACLE_ARITH", "0"))
po.lib().mto_set_arithmetic(arith)

seed0 = int(os.environ.get("SEED0", "0"))  # other seeds: other paths
max_seg = int(os.environ.get("MAX_SEGMENTS", "30"))  # mixed batches: 12 = what optimize_wave_kernel takes (with P <= 2560)
batch = pr.random_mixed_batch(P, deriv, seed0=seed0, max_segments=max_seg) if gen == "mixed" else pr.random_batch(P, n_seg, seed0=seed0, derivative_to_optimize=deriv, generator=gen)
cap = 256
flags = api.FLAG_CAREFUL_COST if os.environ.get("CAREFUL") == "1" else 0  # CAREFUL=1: with the careful re-run of guarded paths
out = ctx.solve_batch(batch, None, time_alloc_method=mode, sampling_dt=0.2, sample_capacity=cap, flags=flags)
t0 = time.time()
cache = os.environ.get("PARITY_CACHE")  # oracle results of an identical earlier invocation (threshold experiments)
if cache and os.path.exists(cache):
    ref = dict(np.load(cache))
else:
    ref = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits,
                         np.zeros(batch.n_segments), deriv=deriv, time_alloc_method=mode, runaway_rule=True, estimate_times=True, sampling_dt=0.2,
                         sample_capacity=cap, n_threads=os.cpu_count() or 8)
    if cache:
        np.savez(cache, **{k: v for k, v in ref.items() if k in ("status", "times", "coeffs", "n_samples")})
print("oracle: %.1f s on %d threads" % (time.time() - t0, os.cpu_count() or 8))
so = batch.seg_offsets
same_status = out["status"] == ref["status"]
dt = np.array([np.max(np.abs(out["times"][a:b] - ref["times"][a:b]) / ref["times"][a:b]) for a, b in zip(so[:-1], so[1:])])
dc = np.array([np.max(np.abs(out["coeffs"][a:b] - ref["coeffs"][a:b])) / np.max(np.abs(ref["coeffs"][a:b]))
               for a, b in zip(so[:-1], so[1:])])
ns_same = out["n_samples"] == np.minimum(ref["n_samples"], cap + 1)
print("paths %d  segments %s  d=%d  generator %s  mode %d%s%s" % (P, ("1..%d" % max_seg) if gen == "mixed" else n_seg, deriv, gen, mode, "  careful re-run" if flags else "",
                                                                 ("", "  ORACLE: exact unit-time constants", "  ORACLE: linear solve in 113-bit arithmetic")[arith]))
print("status equal: %.4f %%   statuses gpu %s" % (100 * same_status.mean(), dict(zip(*np.unique(out["status"], return_counts=True)))))
for tol in (1e-9, 1e-6, 1e-3):
    print("  times within %.0e: %.4f %%   coeffs within %.0e: %.4f %%" % (tol, 100 * (dt < tol).mean(), tol, 100 * (dc < tol).mean()))
print("sample counts equal: %.4f %%" % (100 * ns_same.mean()))
bad_gpu = sum(1 for a, b in zip(so[:-1], so[1:]) if not np.isfinite(out["coeffs"][a:b]).all())
bad_ref = sum(1 for a, b in zip(so[:-1], so[1:]) if not np.isfinite(ref["coeffs"][a:b]).all())
print("non-finite gpu paths:", bad_gpu, " oracle:", bad_ref)
worst = np.argsort(-dt)[:8]
for p in worst:
    a, b = so[p], so[p + 1]
    print("  path %d S=%d: status gpu %d oracle %d, max dt %.2e, gpu times max %.3g oracle max %.3g" %
          (p, b - a, out["status"][p], ref["status"][p], dt[p], out["times"][a:b].max(), ref["times"][a:b].max()))
