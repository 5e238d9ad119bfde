#!/bin/bash
# experiment: cost-cancellation guard thresholds (base,perturbed) against the oracle on the mixed and the plain batch
mkdir -p gpurun_out
for g in "0.5e-9,0.5e-9" "1e-11,1e-12" "1e-12,1e-13" "1e-13,1e-14" "1e-14,1e-15"; do
  for cfg in "16384 10 4 mixed" "16384 10 2 mixed" "65536 10 4 box"; do
    tag=$(echo $cfg | tr ' ' '_')
    echo "=== guard $g  batch $cfg"
    MRS_TG_GUARD=$g PARITY_CACHE=/tmp/oracle_$tag.npz timeout 300 python scripts/parity_sweep.py $cfg 2>&1 | grep -v amdgpu.ids | head -9
  done
done
