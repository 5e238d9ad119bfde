#!/bin/bash
# SQ counters of the linear step and of the nonlinear pipeline (one batch in flight), separate rocprofv3 --pmc passes with
# --kernel-trace only.  usage: scripts/pmc_sq.sh <tag>
tag=${1:-round}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
B="--no-cpu-baseline --no-extras --in-flight 1 --steps 20 --warmup 3"
groups=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
for wl in linear nonlinear; do
  dirs=""
  i=0
  for g in "${groups[@]}"; do
    d=gpurun_out/pmcsq_${wl}_$i
    rm -rf $d
    rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -- python3 bench.py --workload $wl $B > $d.log 2>&1
    dirs="$dirs $d"
    i=$((i+1))
  done
  python3 scripts/summarize_pmc.py gpurun_out/${tag}_pmc_sq_${wl}.csv $dirs > /dev/null
  grep -E "solve_rows|optimize_|sample_kernel|segment_maxima|assemble_blocks_uniform" gpurun_out/${tag}_pmc_sq_${wl}.csv | grep -E "102400|,1024,|65536|,64,|131072|92160|,655360" | head -80
done
