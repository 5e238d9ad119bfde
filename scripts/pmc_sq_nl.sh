#!/bin/bash
# SQ counters of the nonlinear pipeline only (one batch in flight), separate rocprofv3 --pmc passes with --kernel-trace only.
# usage: scripts/pmc_sq_nl.sh <tag> [paths]
tag=${1:-round}
paths=${2:-1024}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
B="--no-cpu-baseline --no-extras --in-flight 1 --steps 20 --warmup 3 --workload nonlinear --paths $paths"
groups=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
dirs=""
i=0
for g in "${groups[@]}"; do
  d=gpurun_out/pmcsq_nl${paths}_$i
  rm -rf $d
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -- python3 bench.py $B > $d.log 2>&1
  dirs="$dirs $d"
  i=$((i+1))
done
python3 scripts/summarize_pmc.py gpurun_out/${tag}_pmc_sq_nonlinear_${paths}.csv $dirs > /dev/null
grep -E "optimize_|solve_rows|segment_maxima|sample_kernel" gpurun_out/${tag}_pmc_sq_nonlinear_${paths}.csv | head -120
