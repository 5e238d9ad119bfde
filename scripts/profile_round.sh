#!/bin/bash
# rocprofv3 kernel statistics behind the numbers in DESIGN.md / profiles/ (run on the GPU box; copies land in gpurun_out/)
# usage: scripts/profile_round.sh <tag>   e.g. round1
tag=${1:-round}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
run() {  # name, bench arguments...
  name=$1; shift
  rm -rf gpurun_out/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python3 bench.py "$@" > gpurun_out/prof_$name.log 2>&1
  f=$(find gpurun_out/prof_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_${name}_kernel_stats.csv
}
run bench_linear --steps 200 --warmup 20 --no-cpu-baseline --no-extras   # the bench command (four batches in flight: kernels of the streams overlap)
run bench_linear_one_in_flight --in-flight 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras   # per-kernel durations without overlap
run bench_nonlinear_1024 --workload nonlinear --in-flight 1 --steps 50 --warmup 5 --no-cpu-baseline --no-extras
run bench_nonlinear_8192 --workload nonlinear --in-flight 1 --paths 8192 --steps 30 --warmup 5 --no-cpu-baseline --no-extras
python3 scripts/kstats.py gpurun_out/${tag}_*_kernel_stats.csv
