#!/bin/bash
# Per-kernel time (rocprofv3 --kernel-trace --stats) of the Mellinger pipelines a nodelet's defaults produce: min-acceleration
# (DERIV=2), moving starts (MOVING=1), stop_at waypoints (STOP=1), long paths.  Writes gpurun_out/r5_pipeline_stats_nodelet.txt.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp && mkdir -p gpurun_out
out=gpurun_out/r5_pipeline_stats_nodelet.txt
: > $out
run() {  # label, env..., config
  label=$1; shift
  rm -rf gpurun_out/prof_nd
  ( export "${@:1:$#-1}"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nd -- python3 scripts/measure_configs.py x "${@: -1}" > gpurun_out/prof_nd.log 2>&1 )
  echo "== $label: $(grep -o "ms_per_step.: [0-9.]*" gpurun_out/prof_nd.log)" >> $out
  python3 scripts/kstats.py $(find gpurun_out/prof_nd -name "*kernel_stats.csv" | head -1) | grep -v "at::\|rocclr\|kernel_stats.csv" | head -7 >> $out
}
run "65536 x 10, min-acceleration, at rest" DERIV=2 uniform65536x10
run "8192 x 10, min-acceleration, moving starts" DERIV=2 MOVING=1 uniform8192x10
run "8192 x 10, min-acceleration, every interior waypoint a stop" DERIV=2 STOP=1 uniform8192x10
run "8192 ragged, min-acceleration, moving starts" DERIV=2 MOVING=1 ragged8192
run "one 80-segment path, min-acceleration, moving start" DERIV=2 MOVING=1 uniform1x80
run "1024 x 10, min-acceleration, moving starts (one wavefront per path)" DERIV=2 MOVING=1 uniform1024x10
cat $out
