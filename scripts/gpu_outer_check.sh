#!/bin/bash
# outer-loop kernel check on the GPU box: phase clocks and per-dispatch times of the one-wavefront kernel and of the
# two-wavefront kernel (MRS_TG_WAVE_KERNEL=0), then the parity tests of the Mellinger path.
#   usage: scripts/gpu_outer_check.sh [tag]      SKIP_TESTS=1 leaves the tests out; PATHS="1024 2048", SEGS=10
tag=${1:-outer}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
[ -x scripts/outer_phases.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mrs_uav_trajectory_generation_amd/csrc -I include scripts/outer_phases.hip -o scripts/outer_phases.bin
for P in ${PATHS:-1024 2048 256}; do
  echo "== wave kernel P=$P"; scripts/outer_phases.bin $P ${SEGS:-10}
  [ -n "$NO_SPLIT" ] || { echo "== split kernel P=$P"; MRS_TG_WAVE_KERNEL=0 scripts/outer_phases.bin $P ${SEGS:-10}; }
done > $O/${tag}_phases.txt 2>&1
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests/test_gpu_nonlinear.py tests/test_gpu_baseline_sizes.py tests/test_optimizer_quality.py tests/test_gpu_reference_scenarios.py tests/test_gpu_boundary_round2.py tests/test_gpu_careful_cost.py tests/test_gpu_policy.py -m gpu -x -q > $O/${tag}_tests.txt 2>&1
  tail -15 $O/${tag}_tests.txt
fi
grep -v "tick [3-5]:\|cost\[" $O/${tag}_phases.txt
