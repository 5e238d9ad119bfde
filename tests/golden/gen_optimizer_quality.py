#!/usr/bin/env python3
"""Independent check of the outer loop's QUALITY.

The reference minimises J_d(T) over T >= 0.01 with NLopt's LD_LBFGS (Luksan PLIS) and at most `max_iterations` = 10
objective evaluations (/root/reference/include/eth_trajectory_generation/impl/polynomial_optimization_nonlinear_impl.h:160-234,
src/mrs_trajectory_generation.cpp:884-896).  NLopt is not vendored and its iterates are not reproduced: the kernel and the
oracle run the project's own projected L-BFGS.  This script pins how good that search is against an
optimiser neither of them shares any code with: scipy's L-BFGS-B (Byrd, Lu, Nocedal, Zhu) on the SAME objective and the SAME
h = 0.1 forward-difference gradient (the oracle's restatement of getCostAndGradientMellinger), same bounds, and the same
budget: the search is cut off after its 10th objective evaluation (scipy's own maxfun only stops between iterations) and
the best point it has evaluated by then is what counts.

A second run per path puts scipy under the REFERENCE'S STOPPING RULE as well (round 3: like for like): NLopt's relative
tolerances ftol_rel 0.05 on the objective and xtol_rel 0.1 on every component, tested on successive accepted iterates
(scipy's callback), next to the 10-evaluation budget -- the point it stands on when the rule fires (or the best point of
its 10 evaluations when the budget ends first) is `J_scipy_same_rule`.

Output: tests/golden/optimizer_quality.json -- per path the start cost and the best cost scipy reached within 10 evaluations.
The inputs are regenerated from the seeds by the tests (problem.random_batch / random_walk generator).

Run in the build container:   python tests/golden/gen_optimizer_quality.py
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import minimize

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mrs_uav_trajectory_generation_amd import problem as pr  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

N_PATHS, N_SEG, SEED0, BUDGET = 200, 10, 31000, 10


def relstop(vold, vnew, reltol):
    """NLopt's scalar stopping rule (nlopt/util/stop.c), absolute tolerance off"""
    if np.isinf(vold):
        return False
    dv = abs(vnew - vold)
    return dv < reltol * (abs(vnew) + abs(vold)) * 0.5 or (reltol > 0 and vnew == vold)


def same_rule(m, v, t0, f_rel=0.05, x_rel=0.1):
    """scipy's L-BFGS-B under the reference's stopping rule (src/mrs_trajectory_generation.cpp:884-885, 896)"""
    calls = [0]
    best = [np.inf, None]
    seen = {}
    state = dict(x=np.array(t0), f=None, stop=None)

    class Stop(Exception):
        pass

    def fun(t):
        if calls[0] >= BUDGET:
            state["stop"] = "maxeval"
            raise Stop()
        calls[0] += 1
        tc = np.maximum(t, 0.01)
        J, g = po.cost_and_gradient(4, m, v, tc)
        seen[tc.tobytes()] = float(J)
        if J < best[0]:
            best[0], best[1] = float(J), np.array(tc)
        if state["f"] is None:
            state["f"] = float(J)
        return J, g

    def callback(xk):
        xk = np.maximum(xk, 0.01)
        fk = seen.get(xk.tobytes())
        if fk is None:
            fk = float(po.cost_and_gradient(4, m, v, xk)[0])      # (not counted: scipy has evaluated this point itself)
        if relstop(state["f"], fk, f_rel):
            state["stop"] = "ftol"
        elif all(relstop(a, b, x_rel) for a, b in zip(state["x"], xk)):
            state["stop"] = "xtol"
        state["x"], state["f"] = np.array(xk), fk
        if state["stop"]:
            raise Stop()

    try:
        minimize(fun, t0, jac=True, method="L-BFGS-B", bounds=[(0.01, None)] * len(t0), callback=callback,
                 options=dict(maxfun=BUDGET, maxiter=100, ftol=0.0, gtol=0.0))
    except Stop:
        pass
    if state["stop"] in ("ftol", "xtol"):
        J, T = state["f"], state["x"]
    else:
        J, T = best[0], best[1]
    return dict(J_scipy_same_rule=float(J), sum_T_scipy_same_rule=float(np.sum(T)), evaluations_same_rule=int(calls[0]),
                stop_same_rule=state["stop"] or "converged")


def main():
    out = dict(generator="tests/golden/gen_optimizer_quality.py", optimiser="scipy.optimize.minimize(method='L-BFGS-B'), best point of the first 10 objective evaluations",
               n_paths=N_PATHS, n_segments=N_SEG, seed0=SEED0, derivative_to_optimize=4, sets=[])
    for gen in ("box", "walk"):
        batch = pr.random_batch(N_PATHS, N_SEG, seed0=SEED0, generator=gen)
        rows = []
        for p in range(batch.n_paths):
            wp, m, v = batch.path(p)
            t0 = po.estimate_times(wp, batch.limits[p])
            calls = [0]
            best = [np.inf, None]

            class BudgetSpent(Exception):
                pass

            def fun(t):
                if calls[0] >= BUDGET:
                    raise BudgetSpent()
                calls[0] += 1
                J, g = po.cost_and_gradient(4, m, v, np.maximum(t, 0.01))
                if J < best[0]:
                    best[0], best[1] = float(J), np.array(t)
                return J, g
            J0, _ = po.cost_and_gradient(4, m, v, t0)
            try:
                minimize(fun, t0, jac=True, method="L-BFGS-B", bounds=[(0.01, None)] * N_SEG,
                         options=dict(maxfun=BUDGET, maxiter=100, ftol=0.0, gtol=0.0))
            except BudgetSpent:
                pass
            row = dict(J_start=float(J0), J_scipy=best[0], evaluations=int(calls[0]),
                       sum_T_start=float(np.sum(t0)), sum_T_scipy=float(np.sum(best[1])))
            row.update(same_rule(m, v, t0))
            rows.append(row)
        out["sets"].append(dict(generator=gen, paths=rows))
        r = np.array([row["J_scipy"] / row["J_start"] for row in rows])
        print(gen, "scipy J_end / J_start: median %.3f  mean evals %.1f" % (np.median(r), np.mean([row["evaluations"] for row in rows])))
        print(gen, "same rule: mean evals %.1f, stops %s" % (np.mean([row["evaluations_same_rule"] for row in rows]),
                                                             {k: sum(1 for row in rows if row["stop_same_rule"] == k) for k in ("ftol", "xtol", "maxeval", "converged")}))
    with open(os.path.join(ROOT, "tests", "golden", "optimizer_quality.json"), "w") as f:
        json.dump(out, f, indent=0)


if __name__ == "__main__":
    main()
