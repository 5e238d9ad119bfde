#!/bin/bash
# linear step time (one batch in flight) for a list of batch sizes: scripts/step_times.sh [workload] P...
w=${1:-linear}; shift
for P in "$@"; do
  r=$(python bench.py --workload $w --paths $P --no-cpu-baseline --no-extras --in-flight 1 --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; print('%.1f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  echo "$w P=$P us_per_step=$r"
done
