#!/bin/bash
# the kernels of one mrs_tg_find_trajectory call on the GPU's clock: start (relative to the call's first kernel) and duration
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
L=$PWD/mrs_uav_trajectory_generation_amd
mkdir -p gpurun_out
g++ -std=c++17 -O2 -I include examples/request_latency_host.cpp -o /tmp/request_latency_host -L $L -lmrs_tg -Wl,-rpath,$L || exit 1
for route in 1 0; do
  rm -rf /tmp/req_prof
  MRS_TG_ROWS_PIPELINE=$route rocprofv3 --kernel-trace --output-format csv -d /tmp/req_prof -o req -- /tmp/request_latency_host 11 40 > /dev/null 2>&1
  f=$(find /tmp/req_prof -name 'req_kernel_trace.csv' | head -1)
  echo "== MRS_TG_ROWS_PIPELINE=$route"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("mrs_tg::", "").replace("void ", "") for r in rows]
# calls start with the first kernel of the repeating pattern: take the last complete call
first = names[-1]
# find period
per = None
for p in range(1, 12):
    if names[-p:] == names[-2 * p:-p]:
        per = p
        break
call = rows[-per:]
t0 = int(call[0]["Start_Timestamp"])
for r in call:
    print("  %-40s start %7.2f us  duration %6.2f us" % (r["Kernel_Name"].split("(")[0].replace("mrs_tg::", "")[:40],
          (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("  first kernel start -> last kernel end: %.2f us" % ((int(call[-1]["End_Timestamp"]) - t0) / 1e3))
PY
done
