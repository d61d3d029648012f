#!/usr/bin/env python3
"""print the kernel dispatches of a rocprofv3 kernel trace csv in order: name, grid, duration (us)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    print(r["Kernel_Name"][:48].ljust(48), r.get("Grid_Size_X", r.get("Grid_Size", "")).rjust(8), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1))
