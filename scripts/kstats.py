"""Print a rocprofv3 kernel_stats.csv compactly: name (truncated), calls, average us, share."""
import csv
import sys

for path in sys.argv[1:]:
    print(path)
    for r in csv.DictReader(open(path)):
        print("  %-64s %6s calls %10.2f us avg %7s %%" % (r["Name"].replace("void ", "").replace("mrs_tg::", "")[:64], r["Calls"],
                                                        float(r["AverageNs"]) / 1e3, r["Percentage"]))
