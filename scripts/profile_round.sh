#!/bin/bash
# rocprofv3 kernel statistics and PMC traffic behind the numbers in DESIGN.md / profiles/ (run on the GPU box; the copies
# land in gpurun_out/, the ones to be judged are then committed under profiles/).
# usage: scripts/profile_round.sh <tag>   e.g. round2
tag=${1:-round}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # name, program, arguments...   (the program itself follows `--`: no wrapper between rocprofv3 and it)
  name=$1; shift
  rm -rf gpurun_out/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- "$@" > gpurun_out/prof_$name.log 2>&1
  f=$(find gpurun_out/prof_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_${name}_kernel_stats.csv
}
B="--no-cpu-baseline --no-extras"
run bench_linear python3 bench.py --steps 200 --warmup 20 $B                       # the bench command (four batches in flight: kernels of the streams overlap)
run bench_linear_one_in_flight python3 bench.py --in-flight 1 --steps 200 --warmup 20 $B   # per-kernel durations without overlap
run bench_nonlinear_1024 python3 bench.py --workload nonlinear --in-flight 1 --steps 50 --warmup 5 $B
run bench_nonlinear_8192 python3 bench.py --workload nonlinear --in-flight 1 --paths 8192 --steps 30 --warmup 5 $B
run k1_fill_ceiling ./scripts/k1_variants.bin 1024 10 50                            # pure fills of the same bytes, empty kernel: per-dispatch durations
# HBM traffic of the assembly kernel: separate counter passes, --kernel-trace only (MI355X_MICROARCH.md, "HBM traffic")
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 scripts/pmc_assemble.py > gpurun_out/pmc_$c.log 2>&1
done
python3 scripts/summarize_pmc.py gpurun_out/${tag}_pmc_assemble_hbm_traffic.csv gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_FETCH_SIZE > /dev/null
python3 scripts/pmc_traffic_json.py gpurun_out/${tag}_pmc_assemble_hbm_traffic.csv gpurun_out/${tag}_pmc_assemble_hbm_traffic.json
python3 scripts/kstats.py gpurun_out/${tag}_*_kernel_stats.csv
