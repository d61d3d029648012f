#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as ms/step (usage: show_ks.py file.csv [steps_in_trace])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
tot = 0.0
for r in rows[:28]:
    ms = float(r['TotalDurationNs']) / 1e6 / steps
    tot += ms
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(6), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(9), ('%.2f' % ms).rjust(8), r['Percentage'])
print('sum(top) ms/step', '%.2f' % tot)
