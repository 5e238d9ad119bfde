#!/bin/bash
# step time vs batch size with the tile solve kernel forced on / off (MRS_TG_TILE_MAX_PATHS); one batch in flight
# usage: scripts/sweep_tile.sh [linear|nonlinear] [P...]
w=${1:-linear}; shift
sizes=${@:-1024 2048 4096 6144 8192 16384 32768 65536}
for P in $sizes; do
  for T in 0 10000000; do
    r=$(MRS_TG_TILE_MAX_PATHS=$T python bench.py --workload $w --paths $P --no-cpu-baseline --no-extras --in-flight 1 --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; print('%.1f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
    echo "$w P=$P tile_max=$T us_per_step=$r"
  done
done
