#!/bin/bash
# A / B of library variants (python -m mrs_uav_trajectory_generation_amd.build --variant NAME -D...) on the saturated-device solve:
#   scripts/quad_variants.sh NAME [NAME ...]     ("default" = the shipped library)
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ "$v" = default ]; then lib=mrs_uav_trajectory_generation_amd/libmrs_tg.so; else lib=mrs_uav_trajectory_generation_amd/libmrs_tg_$v.so; fi
  echo "== $v"
  MRS_TG_LIB_PATH=$PWD/$lib python3 scripts/quad_ab.py 65536 8192 2>&1 | grep -v amdgpu.ids
done
