#!/bin/bash
# Round 5: objective orders below snap (d = 2 min-acceleration -- the nodelet's default config --, d = 3 min-jerk) on the large
# batches' kernels: whole Mellinger pipeline, ms per call (scripts/measure_configs.py), d = 4 beside them.
#   before : one-sided masked sweeps (optimize_lean_masked_kernel) and the quad solve's general step  (MRS_TG_LEAN_SHARED=0 MRS_TG_QUAD_ENDS=0)
#   after  : shared half sweeps with free end slots (optimize_lean_shared_ends_kernel) + solve_quad_kernel<., true>  (shipped)
# Writes gpurun_out/r5_objective_orders_ab.txt; a copy lives in profiles/.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
ms() { python3 scripts/measure_configs.py x "$1" 2>&1 | grep -o "ms_per_step.: [0-9.]*" | cut -d' ' -f2; }
{
  echo "# config             d   before    after    (d = 4: shipped only)"
  for c in uniform1x30 ragged1024 uniform8192x10 ragged8192 uniform65536x10; do
    for d in 2 3; do
      a=$(DERIV=$d MRS_TG_LEAN_SHARED=0 MRS_TG_QUAD_ENDS=0 ms $c); b=$(DERIV=$d ms $c)
      printf "%-18s %2d %8.4f %8.4f\n" $c $d $a $b
    done
    printf "%-18s %2d %8s %8.4f\n" $c 4 "-" $(DERIV=4 ms $c)
  done
} | tee gpurun_out/r5_objective_orders_ab.txt
