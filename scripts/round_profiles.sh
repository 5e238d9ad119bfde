#!/bin/bash
# Everything under profiles/<tag>_* that comes from the GPU box, in one call (about ten minutes, plus the parity sweeps).  The results land in
# gpurun_out/; the ones to be judged are then copied into profiles/.   usage: scripts/round_profiles.sh <tag>
tag=${1:-round}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
# the micro-benchmark binaries are not tracked: build what is missing (hipcc is on the GPU box as well)
[ -x scripts/tile_phases.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mrs_uav_trajectory_generation_amd/csrc -I include scripts/tile_phases.hip -o scripts/tile_phases.bin
[ -x scripts/k1_variants.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mrs_uav_trajectory_generation_amd/csrc -I include scripts/k1_variants.hip -o scripts/k1_variants.bin
python3 bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{"metric' > $O/${tag}_bench_line_20_steps.json
python3 bench.py --steps 200 --warmup 20 2>/dev/null | grep '^{"metric' > $O/${tag}_bench_line.json
scripts/profile_round.sh $tag > $O/${tag}_profile_round.log 2>&1
scripts/pmc_sq.sh $tag > $O/${tag}_pmc_sq.log 2>&1
python3 scripts/pmc_sq_json.py $O/${tag}_pmc_sq_linear.csv solve_rows_kernel 65536 1024 10 $O/${tag}_pmc_sq_solve_rows.json
python3 scripts/pmc_sq_json.py $O/${tag}_pmc_sq_nonlinear.csv optimize_wave_kernel 65536 1024 10 $O/${tag}_pmc_sq_outer_loop.json
scripts/pmc_sq_nl.sh $tag 8192 > $O/${tag}_pmc_sq_nl_8192.log 2>&1
scripts/pmc_solve_quad.sh $tag > $O/${tag}_pmc_solve_quad.log 2>&1
scripts/pmc_headline.sh $tag > $O/${tag}_pmc_headline.log 2>&1            # the kernel the headline's timed region runs (grouped dispatch)
scripts/pmc_sq_nl.sh $tag 65536 > $O/${tag}_pmc_sq_nl_65536.log 2>&1     # the lean outer loop with the device saturated
python3 scripts/quad_ab.py 65536 8192 2>&1 | grep -v amdgpu.ids > $O/${tag}_quad_positions_ab.txt
python3 scripts/scratch_sites.py > $O/${tag}_scratch_sites.txt 2>&1       # static: which functions hold the scratch instructions
python3 scripts/measure_configs.py $tag > $O/${tag}_configs.txt 2>&1
scripts/pipeline_stats.sh $tag > $O/${tag}_pipeline_stats.txt 2>&1
python3 scripts/host_call_rate.py > $O/${tag}_host_call_rate.txt 2>&1
python3 scripts/host_call_rate.py 64 >> $O/${tag}_host_call_rate.txt 2>&1
python3 scripts/host_call_rate.py 8192 >> $O/${tag}_host_call_rate.txt 2>&1
scripts/tile_phases.bin 1024 10 > $O/${tag}_phase_clocks_1024x10.txt 2>&1
[ -x scripts/pipeline_phases.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mrs_uav_trajectory_generation_amd/csrc -I include scripts/pipeline_phases.hip -o scripts/pipeline_phases.bin
(scripts/pipeline_phases.bin 1024 10; scripts/pipeline_phases.bin 1 10) 2>&1 | grep -v amdgpu.ids > $O/${tag}_phase_clocks_closing_stages.txt
(scripts/request_latency.sh; scripts/request_timeline.sh) 2>&1 | grep -v amdgpu.ids > $O/${tag}_request_latency.txt
python3 scripts/lean_ab.py MRS_TG_MAXIMA_BOUNDS 0,1 config4,config5,config6 2>&1 | grep -v amdgpu.ids > $O/${tag}_maxima_bounds_ab.txt
[ -n "$SKIP_PARITY" ] || scripts/parity_sweep_all.sh $O/${tag}_parity_sweep.txt   # ~17 minutes (the 113-bit oracle); SKIP_PARITY=1 leaves it out
ls -la $O/${tag}_*
