#!/bin/bash
# Counters of the kernel bench.py's timed region runs (since round 6 solve_duo_group_kernel, 8 paths per wavefront; until round 5
# solve_quad_group_kernel, 16: KERNEL=solve_quad PPW=16 with MRS_TG_DUO=0 in the environment; 10 x 1024 paths per dispatch, the frozen issue
# policy): HBM traffic (WRITE_SIZE, FETCH_SIZE: separate passes, FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md says) and
# the SQ issue / wait counters, each group in its own rocprofv3 --pmc pass with --kernel-trace only.
#   usage: scripts/pmc_headline.sh <tag>   ->  gpurun_out/<tag>_pmc_solve_duo_group_hbm_traffic.{csv,json}, <tag>_pmc_sq_solve_duo_group.{csv,json}
tag=${1:-round}
K=${KERNEL:-solve_duo}
export K PPW=${PPW:-8}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
B="--no-cpu-baseline --no-extras --steps 20 --warmup 3"
dirs=""
for c in WRITE_SIZE FETCH_SIZE; do
  d=gpurun_out/pmch_$c
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py $B > $d.log 2>&1
  dirs="$dirs $d"
done
python3 scripts/summarize_pmc.py gpurun_out/${tag}_pmc_${K}_group_hbm_traffic.csv $dirs > /dev/null
python3 - gpurun_out/${tag}_pmc_${K}_group_hbm_traffic.csv gpurun_out/${tag}_pmc_${K}_group_hbm_traffic.json <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(l for l in open(sys.argv[1]) if not l.startswith("#"))]
import os
K, PPW = os.environ["K"], int(os.environ["PPW"])
sel = [r for r in rows if K + "_group_kernel" in r["kernel"]]
grid = max(int(r["grid_size"]) for r in sel)               # the full groups (a trailing partial group has a smaller grid)
w = [float(r["mean_value"]) for r in sel if int(r["grid_size"]) == grid and r["counter"] == "WRITE_SIZE"]
f = [float(r["mean_value"]) for r in sel if int(r["grid_size"]) == grid and r["counter"] == "FETCH_SIZE"]
S = 10
paths = grid // 64 * PPW                                   # PPW paths per 64-lane workgroup
e = dict(kernel=K + "_group_kernel", paths_per_dispatch=paths, segments=S, grid_size=grid, write_size_kib=w[0], fetch_size_kib=f[0],
         hbm_bytes_per_dispatch=int((w[0] + 2.0 * f[0]) * 1024), compulsory_bytes_per_dispatch=paths * (40 * S + 288 + 328 * S),
         note="WRITE_SIZE + 2 x FETCH_SIZE KiB (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md); compulsory = SURVEY 8d's "
              "(40 S + 288) in + 328 S out per path; counters serialise the dispatches, the bytes per dispatch do not change")
e["traffic_over_compulsory"] = e["hbm_bytes_per_dispatch"] / e["compulsory_bytes_per_dispatch"]
json.dump(e, open(sys.argv[2], "w"), indent=1)
print(e)
PY
groups=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
dirs=""
i=0
for g in "${groups[@]}"; do
  d=gpurun_out/pmchsq_$i
  rm -rf $d
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -- python3 bench.py $B > $d.log 2>&1
  dirs="$dirs $d"
  i=$((i+1))
done
python3 scripts/summarize_pmc.py gpurun_out/${tag}_pmc_sq_${K}_group.csv $dirs > /dev/null
grid=$(python3 -c "import json;print(json.load(open('gpurun_out/${tag}_pmc_${K}_group_hbm_traffic.json'))['grid_size'])")
python3 scripts/pmc_sq_json.py gpurun_out/${tag}_pmc_sq_${K}_group.csv ${K}_group_kernel $grid 10240 10 gpurun_out/${tag}_pmc_sq_${K}_group.json
grep ${K}_group gpurun_out/${tag}_pmc_sq_${K}_group.csv | head -40
