"""Print the top rows of a rocprofv3 kernel_stats.csv:  python tools/kstats.py <dir or csv> [n]"""
import csv, glob, os, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print("%-80s calls=%6s avg=%9.1fus pct=%5.2f" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
