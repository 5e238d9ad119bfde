"""Condense rocprofv3 --pmc output (counter_collection.csv files under the given directories) into one small CSV:
kernel, grid size, counter, mean value per dispatch.   python scripts/summarize_pmc.py out.csv dir [dir...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(list)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0]
                acc[(name, row["Grid_Size"], row["Counter_Name"])].append(float(row["Counter_Value"]))
with open(out, "w") as fh:
    fh.write("kernel,grid_size,counter,dispatches,mean_value\n")
    for (k, g, c), v in sorted(acc.items()):
        fh.write('"%s",%s,%s,%d,%.3f\n' % (k, g, c, len(v), sum(v) / len(v)))
print(open(out).read())
