"""Print a rocprofv3 kernel_stats.csv (or the newest one under a directory) as a table."""
import csv, glob, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(path)):
    print("%-64s calls=%4s avg=%9.1f us min=%9.1f max=%9.1f %6.2f%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3,
          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
