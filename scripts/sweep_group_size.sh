#!/bin/bash
# grouped launches of the headline step: batches in flight / batches per dispatch / streams, at the driver's 20 steps (three
# repeats) and at 200 steps.   usage (GPU box): scripts/sweep_group_size.sh
for cfg in "10 5 4" "16 8 4" "16 8 2" "20 10 2" "32 16 2" "24 8 3" "20 10 4" "30 10 3"; do
  set -- $cfg
  for steps in 20 20 20 200; do
    v=$(python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-extras --in-flight $1 --group-size $2 --streams $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f M/s  %.2f us/step' % (d['value']/1e6, d['ms_per_step']*1e3))")
    echo "in-flight $1 group $2 streams $3 steps $steps: $v"
  done
done
