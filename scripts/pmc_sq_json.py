"""SQ counter summary (scripts/summarize_pmc.py CSV) -> the JSON bench.py reads for one kernel at one launch shape.
usage: pmc_sq_json.py in.csv kernel_substring grid_size paths segments out.json"""
import csv
import json
import sys

src, kernel, grid, paths, segments, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
counters, name = {}, None
for r in csv.DictReader(open(src)):
    if kernel in r["kernel"] and r["grid_size"] == grid:
        counters[r["counter"]] = float(r["mean_value"])
        name = r["kernel"].replace("void ", "").replace("mrs_tg::", "")
if not counters:
    sys.exit("no rows for %s at grid %s in %s" % (kernel, grid, src))
json.dump(dict(kernel=name, paths=paths, segments=segments, grid_size=int(grid), counters=counters, source="profiles/" + src.split("/")[-1]),
          open(out, "w"), indent=1)
print(out, len(counters), "counters")
