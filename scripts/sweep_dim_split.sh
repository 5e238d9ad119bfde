#!/bin/bash
# nonlinear step time vs batch size for the two lane mappings of the outer loop (MRS_TG_DIM_SPLIT_MAX_PATHS forces one)
for P in 512 1024 1536 2048 3072 4096 6144; do
  for T in 0 1000000; do
    r=$(MRS_TG_DIM_SPLIT_MAX_PATHS=$T python bench.py --workload nonlinear --paths $P --no-cpu-baseline --no-extras --in-flight 1 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; print('%.1f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
    echo "P=$P split_max=$T us_per_step=$r"
  done
done
