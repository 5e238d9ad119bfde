#!/bin/bash
# linear step time vs batch size with the tile solve kernel forced on / off (MRS_TG_TILE_MAX_PATHS)
for P in 1024 2048 3072 4096 6144 8192; do
  for T in 0 1000000; do
    r=$(MRS_TG_TILE_MAX_PATHS=$T python bench.py --paths $P --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; print('%.1f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
    echo "P=$P tile_max=$T us_per_step=$r"
  done
done
