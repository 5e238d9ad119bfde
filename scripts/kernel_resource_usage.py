"""Registers, scratch and occupancy of every kernel of the library as the compiler reports them
(-Rpass-analysis=kernel-resource-usage): python scripts/kernel_resource_usage.py > profiles/roundN_kernel_resource_usage.txt
Runs without a GPU (hipcc cross-compiles gfx950)."""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = (("vgpr", r" VGPRs: (\d+)"), ("agpr", r" AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"),
        ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("waves", r"Occupancy \[waves/SIMD\]: (\d+)"),
        ("sgpr_spill", r"SGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"))


def main():
    rows = []
    for src in sorted(glob.glob(os.path.join(ROOT, "mrs_uav_trajectory_generation_amd", "csrc", "*.hip"))):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                            "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull], capture_output=True, text=True)
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1), "file": os.path.basename(src)}
                rows.append(cur)
            for key, pat in KEYS:
                m = re.search(pat, line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
    print("# hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage, the library's kernels")
    print("# %-58s %-22s %5s %5s %5s %16s %11s %11s %10s" % ("kernel", "file", "VGPR", "AGPR", "SGPR", "scratch[B/lane]", "waves/SIMD",
                                                            "SGPR spills", "static LDS"))
    for r in sorted(rows, key=lambda r: (r["file"], r["name"])):
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("mrs_tg::", "").replace("void ", "")
        print("  %-58s %-22s %5d %5d %5d %16d %11d %11d %10d" % (name, r["file"], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"],
                                                               r["waves"], r["sgpr_spill"], r["lds"]))


if __name__ == "__main__":
    main()
