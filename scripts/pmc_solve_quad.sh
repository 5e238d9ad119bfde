#!/bin/bash
# HBM traffic of solve_quad_kernel at 65536 x 10 (separate --pmc passes, --kernel-trace only): usage scripts/pmc_solve_quad.sh <tag>
tag=${1:-round}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf gpurun_out/pmcq_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcq_$c -- python3 scripts/pmc_solve_quad.py > gpurun_out/pmcq_$c.log 2>&1
done
python3 scripts/summarize_pmc.py gpurun_out/${tag}_pmc_solve_quad_hbm_traffic.csv gpurun_out/pmcq_WRITE_SIZE gpurun_out/pmcq_FETCH_SIZE > /dev/null
python3 - gpurun_out/${tag}_pmc_solve_quad_hbm_traffic.csv gpurun_out/${tag}_pmc_solve_quad <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(l for l in open(sys.argv[1]) if not l.startswith("#"))]
P, S = 65536, 10
for inst, suffix, what in (("solve_quad_kernel<false>", "_hbm_traffic.json", "vertex positions read from fixed_values"),
                           ("solve_quad_kernel<true>", "_wp_hbm_traffic.json", "vertex positions read from the waypoint array (MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS)")):
    w = [float(r["mean_value"]) for r in rows if inst in r["kernel"] and r["counter"] == "WRITE_SIZE"]
    f = [float(r["mean_value"]) for r in rows if inst in r["kernel"] and r["counter"] == "FETCH_SIZE"]
    if not w or not f:
        continue
    e = dict(kernel=inst, positions=what, paths=P, segments=S, write_size_kib=w[0], fetch_size_kib=f[0],
             hbm_bytes_per_launch=int((w[0] + 2.0 * f[0]) * 1024), compulsory_bytes_per_launch=P * (40 * S + 288 + 328 * S),
             note="WRITE_SIZE + 2 x FETCH_SIZE KiB (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md); compulsory = SURVEY 8d's (40 S + 288) in + 328 S out per path")
    e["traffic_over_compulsory"] = e["hbm_bytes_per_launch"] / e["compulsory_bytes_per_launch"]
    json.dump(e, open(sys.argv[2] + suffix, "w"), indent=1)
    print(e)
PY
