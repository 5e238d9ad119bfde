#!/bin/bash
# one request through mrs_tg_find_trajectory from a g++-built host (examples/request_latency_host.cpp): the call's latency
# (median of 300), with the closing stages in one launch and in separate launches, for 10 and 4 segments
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
L=$PWD/mrs_uav_trajectory_generation_amd
g++ -std=c++17 -O2 -I include examples/request_latency_host.cpp -o /tmp/request_latency_host -L $L -lmrs_tg -Wl,-rpath,$L || exit 1
echo "== default"; /tmp/request_latency_host 11 300; /tmp/request_latency_host 5 300
echo "== separate closing launches (MRS_TG_ROWS_PIPELINE=0)"; MRS_TG_ROWS_PIPELINE=0 /tmp/request_latency_host 11 300; MRS_TG_ROWS_PIPELINE=0 /tmp/request_latency_host 5 300
# long paths (20 and 30 segments): the lane-group outer loop that takes them since round 5 (dim_split_for: no lane-per-dimension
# kernel from 16 segments on) against the old rule (MRS_TG_DIM_SPLIT_MAX_PATHS=2560: every batch of <= 2560 paths)
echo "== long paths, shipped rule"; /tmp/request_latency_host 21 300; /tmp/request_latency_host 31 300
echo "== long paths, lane-per-dimension kernel (MRS_TG_DIM_SPLIT_MAX_PATHS=2560)"
MRS_TG_DIM_SPLIT_MAX_PATHS=2560 /tmp/request_latency_host 21 300; MRS_TG_DIM_SPLIT_MAX_PATHS=2560 /tmp/request_latency_host 31 300
