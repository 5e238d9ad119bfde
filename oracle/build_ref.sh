#!/bin/bash
# Build the Eigen-only part of the REFERENCE ITSELF into oracle/_ref/libmrs_tg_ref.so, from the sources where they lie under
# /root/reference (nothing is copied), plus our own C harness (oracle/ref_harness.cpp).  Dormant in the image this was
# written in: Eigen3 is not installed there and may not be stubbed -- the script then says so and exits 0, and
# tests/test_oracle_vs_reference_build.py skips.  On an image that carries Eigen3 the same command turns "parity unpinned"
# into a test run (oracle/REF_BUILD.md).        usage: oracle/build_ref.sh [reference root = /root/reference]
REF=${1:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
[ -d "$REF/src/eth_trajectory_generation" ] || { echo "build_ref: no reference tree at $REF (GPU box: prebuilt files only)"; exit 0; }
EIGEN=""
for d in /usr/include/eigen3 /usr/local/include/eigen3 /opt/conda/include/eigen3 "$EIGEN3_INCLUDE_DIR"; do
  [ -n "$d" ] && [ -f "$d/Eigen/Core" ] && EIGEN="$d" && break
done
if [ -z "$EIGEN" ]; then
  echo "build_ref: Eigen3 headers not found (searched /usr/include/eigen3, /usr/local/include/eigen3, /opt/conda/include/eigen3, \$EIGEN3_INCLUDE_DIR): the reference is unbuildable here, oracle/_ref stays empty"
  exit 0
fi
mkdir -p "$HERE/_ref"
S="$REF/src/eth_trajectory_generation"
# polynomial.cpp, rpoly/rpoly_ak1.cpp: Eigen only (their logging macros are the reference's own misc.h).  vertex.cpp needs
# mrs_lib/geometry/cyclic.h, the nonlinear layer <nlopt.hpp>: not part of this build.
g++ -O3 -std=c++17 -fPIC -shared -I "$EIGEN" -I "$REF/include" -o "$HERE/_ref/libmrs_tg_ref.so" \
    "$HERE/ref_harness.cpp" "$S/polynomial.cpp" "$S/rpoly/rpoly_ak1.cpp" && echo "build_ref: built $HERE/_ref/libmrs_tg_ref.so (Eigen at $EIGEN)"
