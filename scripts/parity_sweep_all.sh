#!/bin/bash
# every section of profiles/roundN_parity_sweep.txt (about 2.5 minutes on the GPU box, mostly the oracle on all host cores)
out=${1:-gpurun_out/parity_sweep_all.txt}
: > $out
for cfg in "65536 10 4 box 2" "8192 ragged 4 box 2" "8192 10 2 walk 2" "4096 10 3 box 2" "4096 10 4 box 0" "2048 5 2 box 3" \
           "16384 10 4 mixed 2" "16384 10 3 mixed 2" "16384 10 2 mixed 2" "2048 10 4 mixed 1" "2048 10 4 mixed 4"; do
  timeout 300 python scripts/parity_sweep.py $cfg 2>&1 | grep -v amdgpu.ids >> $out
done
# the Mellinger batches again, against the oracle with its linear solve in 113-bit arithmetic (ORACLE_ARITH=2,
# oracle/mto_linear.c): the reference's algorithm without the rounding noise of the reference's arithmetic route -- how much of
# the disagreement above belongs to the double-precision oracle.  (The oracle runs ~50 times slower: smaller batches.)
for cfg in "8192 10 4 box 2" "4096 ragged 4 box 2" "4096 10 2 walk 2" "4096 10 3 box 2" "4096 10 4 mixed 2" "4096 10 3 mixed 2" \
           "4096 10 2 mixed 2"; do
  ORACLE_ARITH=2 timeout 900 python scripts/parity_sweep.py $cfg 2>&1 | grep -v amdgpu.ids >> $out
done
