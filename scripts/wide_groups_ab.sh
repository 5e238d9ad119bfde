#!/bin/bash
# Round 5: (1) wide lane groups of the plain-path outer loop (13-15 and 29-30 segments get the S + 4 lanes of the shared half
# sweeps; launch_nonlinear picks them up to 8 residency rounds) against the narrow groups, and (2) the plan's lane-per-dimension
# rule (dim_split_for: not for paths of 16 segments or more) against the old rule (every batch of <= 2560 paths), same box,
# whole Mellinger pipeline (scripts/measure_configs.py, ms per call).  Writes gpurun_out/r5_wide_groups_ab.txt and
# gpurun_out/r5_dim_split_crossover.txt; copies live in profiles/.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
ms() { python3 scripts/measure_configs.py x "$1" 2>&1 | grep -o "ms_per_step.: [0-9.]*" | cut -d' ' -f2; }
{
  echo "# whole pipeline, ms per call: narrow groups (MRS_TG_LEAN_WIDE=0) | wide groups (=1) | shipped rule (unset)"
  for c in ragged4096 ragged8192 ragged16384 ragged32768 ragged65536 uniform4096x14 uniform8192x14 uniform16384x14 \
           uniform65536x14 uniform8192x6 uniform65536x6 uniform8192x30 uniform32768x30 config4; do
    a=$(MRS_TG_LEAN_WIDE=0 ms $c); b=$(MRS_TG_LEAN_WIDE=1 ms $c); d=$(ms $c)
    printf "%-18s %8.4f %8.4f %8.4f\n" $c $a $b $d
  done
} | tee gpurun_out/r5_wide_groups_ab.txt
{
  echo "# whole pipeline, ms per call: old rule (MRS_TG_DIM_SPLIT_MAX_PATHS=2560) | lean kernel (=0) | shipped rule (unset)"
  for c in ragged128 ragged512 ragged1024 ragged2048 uniform256x13 uniform1024x13 uniform2048x13 uniform256x16 uniform1024x16 \
           uniform2048x16 uniform256x30 uniform1024x30 uniform2048x30 uniform256x4 uniform2048x4 config3; do
    a=$(MRS_TG_DIM_SPLIT_MAX_PATHS=2560 ms $c); b=$(MRS_TG_DIM_SPLIT_MAX_PATHS=0 ms $c); d=$(ms $c)
    printf "%-18s %8.4f %8.4f %8.4f\n" $c $a $b $d
  done
} | tee gpurun_out/r5_dim_split_crossover.txt
