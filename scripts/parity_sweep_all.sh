#!/bin/bash
# every section of profiles/roundN_parity_sweep.txt (about 2.5 minutes on the GPU box, mostly the oracle on all host cores)
out=${1:-gpurun_out/parity_sweep_all.txt}
: > $out
for cfg in "65536 10 4 box 2" "8192 ragged 4 box 2" "8192 10 2 walk 2" "4096 10 3 box 2" "4096 10 4 box 0" "2048 5 2 box 3" \
           "16384 10 4 mixed 2" "16384 10 3 mixed 2" "16384 10 2 mixed 2" "2048 10 4 mixed 1" "2048 10 4 mixed 4" \
           "2048 10 4 box 2" "2048 8 2 walk 2" "1024 12 3 box 2"; do   # (the last three: optimize_wave_kernel)
  timeout 300 python scripts/parity_sweep.py $cfg 2>&1 | grep -v amdgpu.ids >> $out
done
# mixed constraint patterns of at most 12 segments per path, 2048 paths: optimize_wave_kernel's special evaluations
for cfg in "2048 12 4 mixed 2" "2048 12 3 mixed 2" "2048 12 2 mixed 2"; do
  MAX_SEGMENTS=12 timeout 300 python scripts/parity_sweep.py $cfg 2>&1 | grep -v amdgpu.ids >> $out
done
# the Mellinger batches again, against the oracle with its linear solve in 113-bit arithmetic (ORACLE_ARITH=2,
# oracle/mto_linear.c): the reference's algorithm without the rounding noise of the reference's arithmetic route -- how much of
# the disagreement above belongs to the double-precision oracle.  (The oracle runs ~50 times slower: smaller batches.)
for cfg in "8192 10 4 box 2" "4096 ragged 4 box 2" "4096 10 2 walk 2" "4096 10 3 box 2" "4096 10 4 mixed 2" "4096 10 3 mixed 2" \
           "4096 10 2 mixed 2"; do
  ORACLE_ARITH=2 timeout 900 python scripts/parity_sweep.py $cfg 2>&1 | grep -v amdgpu.ids >> $out
done
