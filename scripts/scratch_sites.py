#!/usr/bin/env python3
"""Where the scratch (private-segment) traffic of the library's kernels sits, statically: every translation unit is compiled
to gfx950 assembly and the scratch_load / scratch_store / buffer_*_dword ... offen instructions are counted per FUNCTION
(kernel body vs the non-inlined device functions it calls).  A kernel whose resource summary reports scratch bytes inherits
them from its callees: `optimize_wave_kernel` is listed with 472 B per lane, all of it inside `wave_evaluate_generic` (the
one-sided sweeps of paths without a specialised step, behind a real call) -- the kernel's own body holds no scratch
instruction, so the common path touches none.  Runs without a GPU.   python scripts/scratch_sites.py > profiles/roundN_scratch_sites.txt"""
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRATCH = re.compile(r"^\s*(scratch_(load|store)\S*|buffer_(load|store)_dword\S*\s.*\boffen\b)")


def main():
    print("# scratch instructions per function (hipcc --offload-arch=gfx950 -O3 -S), functions with a private segment only")
    print("# %-64s %-22s %9s %14s %14s" % ("function", "file", "scratch B", "scratch loads", "scratch stores"))
    for src in sorted(glob.glob(os.path.join(ROOT, "mrs_uav_trajectory_generation_amd", "csrc", "*.hip"))):
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                            "-S", "--cuda-device-only", "-o", tmp.name, src], check=True, capture_output=True)
            text = open(tmp.name).read().splitlines()
        cur, rows = None, {}
        for line in text:
            m = re.match(r"^(_Z\w+):\s", line)
            if m:
                cur = m.group(1)
                rows[cur] = dict(loads=0, stores=0, scratch=None)
                continue
            if cur is None:
                continue
            if SCRATCH.match(line):
                rows[cur]["stores" if "store" in line.split()[0] else "loads"] += 1
            m = re.match(r"^; ScratchSize: (\d+)", line)
            if m:
                rows[cur]["scratch"] = int(m.group(1))
        for name, r in rows.items():
            if not r["scratch"] and not r["loads"] and not r["stores"]:
                continue
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem).replace("mrs_tg::", "").replace("void ", "")
            print("  %-64s %-22s %9d %14d %14d" % (dem, os.path.basename(src), r["scratch"] or 0, r["loads"], r["stores"]))


if __name__ == "__main__":
    main()
