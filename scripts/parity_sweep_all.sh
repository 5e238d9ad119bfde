#!/bin/bash
# every section of profiles/roundN_parity_sweep.txt (about 2.5 minutes on the GPU box, mostly the oracle on all host cores)
out=${1:-gpurun_out/parity_sweep_all.txt}
: > $out
for cfg in "65536 10 4 box 2" "8192 ragged 4 box 2" "8192 10 2 walk 2" "4096 10 3 box 2" "4096 10 4 box 0" "2048 5 2 box 3" \
           "16384 10 4 mixed 2" "16384 10 3 mixed 2" "16384 10 2 mixed 2" "2048 10 4 mixed 1" "2048 10 4 mixed 4"; do
  timeout 300 python scripts/parity_sweep.py $cfg 2>&1 | grep -v amdgpu.ids >> $out
done
