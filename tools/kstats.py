#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 *_kernel_stats.csv (name, calls, average us, share).   python tools/kstats.py <csv> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'][:64]:64s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['Percentage']):5.2f} %")
