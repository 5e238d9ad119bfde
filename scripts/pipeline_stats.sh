#!/bin/bash
# rocprofv3 kernel statistics of the nonlinear pipeline at the three large batch shapes (one batch in flight)
# usage: scripts/pipeline_stats.sh <tag>
tag=${1:-round}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in config4 config5 config6; do
  rm -rf gpurun_out/prof_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$cfg -- python3 scripts/measure_configs.py x $cfg > gpurun_out/prof_$cfg.log 2>&1
  f=$(find gpurun_out/prof_$cfg -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_${cfg}_kernel_stats.csv
  tail -1 gpurun_out/prof_$cfg.log
  python3 scripts/kstats.py gpurun_out/${tag}_${cfg}_kernel_stats.csv
done
