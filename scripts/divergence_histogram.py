"""Why do the GPU and the oracle end on different segment times on a few paths of a large batch?  

For every path of a batch whose outer-loop result (no feasibility scaling: limits far away, so the times that come back are
the search's last evaluated point) differs between the GPU and the oracle at 1e-6, find the FIRST objective evaluation at
which the two searches stand on different points -- both are run with an evaluation budget of 1, 2, ..., 10 and return their
k-th trial point -- and look at the decision that produced that point in the oracle's own trace (mto_set_optimizer_trace):
how far was it from flipping?

  * margin = min(|Armijo margin|, |ftol margin|, |xtol margin|) of the decision (all relative to |f| or |x_i|): a margin
    below ~1e-7 means the two implementations, whose J differ by ~1e-9 relative, sat on the boundary of a comparison;
  * |dJ| / J at the point both evaluated last together (GPU's cost_gradient kernel vs the oracle);
  * 'smooth': the two trial points differ by less than 1e-3 relative at the first divergent evaluation: no comparison
    flipped, the difference is rounding noise in the forward-difference gradient amplified by the L-BFGS update.

A second pass runs the WHOLE pipeline (the batch's own limits: outer loop, solve, per-segment feasibility scaling, re-solve)
and sorts the paths whose final times differ at 1e-6 by where the difference entered: in the search (above), or behind it --
the two searches ended on the same point to 1e-6 and the scaling / re-solve at that point told them apart.  For the latter
the shortest segment of the shared point and (T_max / T_min)^7, the growth of the linear system's condition number, are
printed: those are the points with a segment on the 0.01 s bound.

usage: divergence_histogram.py <generator: box|mixed|ragged> <n_paths> <deriv> [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mrs_uav_trajectory_generation_amd import api, problem as pr
from oracle import pyoracle as po
from tests import util

gen, P, deriv = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
out_path = sys.argv[4] if len(sys.argv) > 4 else None
def make_batch():
    if gen == "mixed":
        return pr.random_mixed_batch(P, deriv)
    if gen == "ragged":
        return pr.random_batch(P, "ragged", seed0=0)
    return pr.random_batch(P, 10, seed0=0)


full = make_batch()
with_limits = make_batch()
full.limits[:] = 1e12   # no feasibility scaling: what comes back is the outer loop's last evaluated point
so = full.seg_offsets
torch.zeros(1, device="cuda")
ctx = api.Context(0)
t0_all = util.oracle_times(with_limits)


def run(batch, t0, budget):
    ref = po.solve_batch(batch.seg_offsets, batch.waypoints, batch.fixed_mask, batch.fixed_values, batch.limits, t0.copy(),
                         deriv=deriv, time_alloc_method=2, runaway_rule=True, estimate_times=False, n_threads=os.cpu_count() or 8, max_iterations=budget)
    out = ctx.solve_batch(batch, t0.copy(), time_alloc_method=2, max_iterations=budget)
    return ref, out


def rel_diff(a, b, offsets):
    return np.array([np.max(np.abs(a[x:y] - b[x:y]) / b[x:y]) for x, y in zip(offsets[:-1], offsets[1:])])


ref, out = run(full, t0_all, 10)
d_final = rel_diff(out["times"], ref["times"], so)
bad = np.nonzero((d_final > 1e-6) | (out["status"] != ref["status"]))[0]
print("%s x %d, d = %d: %d paths differ at 1e-6 (%.3f %%), %d at 1e-9" % (gen, P, deriv, bad.size, 100.0 * bad.size / P, int((d_final > 1e-9).sum())))
sub = full.select(list(bad)) if bad.size else None
records = []
if sub is not None:
    sso = sub.seg_offsets
    t0_sub = np.concatenate([t0_all[so[p]:so[p + 1]] for p in bad])
    first = np.full(bad.size, -1)
    gap_at_first = np.zeros(bad.size)
    prev_ref = None
    for k in range(1, 11):
        r, o = run(sub, t0_sub, k)
        d = rel_diff(o["times"], r["times"], sso)
        newly = (first < 0) & (d > 1e-9)
        first[newly] = k
        gap_at_first[newly] = d[newly]
        if k == 1:
            prev_points = {}
        for j in np.nonzero(newly)[0]:
            prev_points[int(j)] = None if prev_ref is None else prev_ref["times"][sso[j]:sso[j + 1]].copy()
        prev_ref = r
    plan_cache = {}
    for j, p in enumerate(bad):
        _, m, v = sub.path(j)
        t0 = t0_sub[sso[j]:sso[j + 1]]
        rc, t, ne, tr = po.optimize_times_traced(deriv, m, v, t0, po.default_nlopt(10))
        k = int(first[j])
        rec = dict(path=int(p), segments=int(sso[j + 1] - sso[j]), first_divergent_evaluation=k, gap_there=float(gap_at_first[j]),
                   final_gap=float(d_final[p]), status_gpu=int(out["status"][p]), status_oracle=int(ref["status"][p]))
        if k >= 2:
            # the decision that produced trial point k was taken after evaluation k - 1 (trace row of evaluation k - 1; the
            # first evaluation has no row: nothing is compared there)
            rows = [row for row in tr if int(row[0]) == k - 1]
            if rows:
                row = rows[0]
                margins = [abs(row[5])]
                if row[5] >= 0:
                    margins += [abs(row[6]), abs(row[7])]
                rec["decision_margin"] = float(min(margins))
                rec["armijo_margin"], rec["ftol_margin"], rec["xtol_margin"] = float(row[5]), float(row[6]), float(row[7])
            xp = prev_points.get(j)
            if xp is not None:   # J of both at the last point they shared
                one = sub.select([j])
                plan = api.Plan(ctx, one.seg_offsets)
                cost = torch.zeros(1, dtype=torch.float64, device="cuda")
                grad = torch.zeros(len(xp), dtype=torch.float64, device="cuda")
                dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
                plan.cost_gradient(deriv, dv(one.fixed_mask), dv(one.fixed_values), dv(xp), cost, grad)
                torch.cuda.synchronize()
                Jg, gg = float(cost.cpu()[0]), grad.cpu().numpy()
                plan.close()
                Jo, go = po.cost_and_gradient(deriv, m, v, xp)
                rec["rel_dJ_at_shared_point"] = abs(Jg - Jo) / abs(Jo)
                rec["rel_dgrad_at_shared_point"] = float(np.max(np.abs(gg - go)) / np.max(np.abs(go)))
                rec["min_time_at_shared_point"] = float(xp.min())
        rec["kind"] = ("never (final only)" if k < 0 else "start" if k == 1 else
                       "smooth (gradient noise amplified)" if gap_at_first[j] < 1e-3 else "branch flip")
        records.append(rec)
kinds = {}
for r in records:
    kinds[r["kind"]] = kinds.get(r["kind"], 0) + 1
print("kinds:", kinds)
flips = [r for r in records if r["kind"] == "branch flip" and "decision_margin" in r]
if flips:
    mg = np.array([r["decision_margin"] for r in flips])
    edges = [0, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-4, 1e-2, np.inf]
    hist = np.histogram(mg, bins=edges)[0]
    print("branch flips: margin of the flipped comparison (relative), histogram over", edges)
    print("   ", hist.tolist())
    dj = np.array([r.get("rel_dJ_at_shared_point", np.nan) for r in flips])
    print("    |dJ|/J at the shared point: median %.2e  max %.2e" % (np.nanmedian(dj), np.nanmax(dj)))
    unexplained = [r for r in flips if r["decision_margin"] > 100 * max(r.get("rel_dJ_at_shared_point", 0), r.get("rel_dgrad_at_shared_point", 0), 1e-12)]
    print("    flips whose margin exceeds 100 x the J / gradient difference at the shared point: %d" % len(unexplained))
    for r in unexplained[:10]:
        print("      ", r)
smooth = [r for r in records if r["kind"].startswith("smooth")]
if smooth:
    print("smooth divergences: first gap median %.2e, final gap median %.2e; gradient difference at the shared point median %.2e" % (
        np.median([r["gap_there"] for r in smooth]), np.median([r["final_gap"] for r in smooth]),
        np.nanmedian([r.get("rel_dgrad_at_shared_point", np.nan) for r in smooth])))
# ---- the whole pipeline, with the batch's own limits
ref_p, out_p = run(with_limits, t0_all, 10)
d_pipe = rel_diff(out_p["times"], ref_p["times"], so)
differ = np.nonzero((d_pipe > 1e-6) | (out_p["status"] != ref_p["status"]))[0]
in_search = [p for p in differ if d_final[p] > 1e-6 or out["status"][p] != ref["status"][p]]
behind = [p for p in differ if p not in set(in_search)]
print("whole pipeline: %d paths differ at 1e-6 (%.3f %%): %d already in the search, %d behind it (scaling + re-solve at a shared point)"
      % (differ.size, 100.0 * differ.size / P, len(in_search), len(behind)))
pipeline = dict(differing_at_1e6=int(differ.size), in_search=len(in_search), behind_search=len(behind))
if behind:
    tmin = np.array([ref["times"][so[p]:so[p + 1]].min() for p in behind])
    spread = np.array([(ref["times"][so[p]:so[p + 1]].max() / ref["times"][so[p]:so[p + 1]].min()) ** 7 for p in behind])
    gap_search = np.array([d_final[p] for p in behind])
    on_bound = int((tmin <= 0.0100001).sum())
    print("  behind the search: shortest segment of the shared point: %d of %d on the 0.01 s bound, median %.3f s; (T_max / T_min)^7 median %.1e, "
          "smallest %.1e; gap of the searches there median %.1e; final gap median %.1e"
          % (on_bound, len(behind), np.median(tmin), np.median(spread), spread.min(), np.median(gap_search), np.median(d_pipe[behind])))
    runaway = int(sum(1 for p in behind if out_p["status"][p] == -4 or ref_p["status"][p] == -4))
    print("  of these, flagged ROUNDOFF_LIMITED (runaway) by either side: %d" % runaway)
    pipeline.update(on_bound=on_bound, median_min_segment=float(np.median(tmin)), median_spread7=float(np.median(spread)),
                    min_spread7=float(spread.min()), flagged_runaway=runaway)
if out_path:
    with open(out_path, "w") as f:
        json.dump(dict(generator=gen, paths=P, derivative=deriv, differing_at_1e6=int(bad.size), kinds=kinds, pipeline=pipeline,
                       records=records), f, indent=0)
