"""Turn the condensed PMC CSV of the assembly kernel (scripts/summarize_pmc.py) into the small JSON bench.py reads:
HBM bytes per launch = WRITE_SIZE + 2 x FETCH_SIZE KiB (FETCH_SIZE is doubled on gfx950, MI355X_MICROARCH.md "HBM";
WRITE_SIZE was calibrated on this kernel's 8-byte-per-lane store pattern with a pure fill of known size in round 1,
profiles/round1_pmc_assemble_hbm_traffic.csv).     python scripts/pmc_traffic_json.py in.csv out.json"""
import csv
import json
import sys

rows = [r for r in csv.DictReader(l for l in open(sys.argv[1]) if not l.startswith("#"))]
out = []
for grid, paths in (("102400", 1024), ("6553600", 65536)):
    w = [float(r["mean_value"]) for r in rows if "assemble_blocks_uniform" in r["kernel"] and r["grid_size"] == grid and r["counter"] == "WRITE_SIZE"]
    f = [float(r["mean_value"]) for r in rows if "assemble_blocks_uniform" in r["kernel"] and r["grid_size"] == grid and r["counter"] == "FETCH_SIZE"]
    if w and f:
        out.append(dict(kernel="assemble_blocks_uniform_kernel", paths=paths, segments=10, write_size_kib=w[0], fetch_size_kib=f[0],
                        hbm_bytes_per_launch=int((w[0] + 2.0 * f[0]) * 1024), algorithmic_bytes_per_launch=1608 * paths * 10))
# bench.py looks for the entry of its own batch shape: one file per shape
base = sys.argv[2]
for e in out:
    name = base if e["paths"] == 1024 else base.replace(".json", "_%d.json" % e["paths"])
    json.dump(e, open(name, "w"), indent=1)
    print(name, e)
