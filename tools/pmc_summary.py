#!/usr/bin/env python3
"""Print mean counter values per fltx kernel from rocprofv3 counter_collection CSVs (any number of files)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(list)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "fltx" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-50s %-24s %.4g" % (k, c, sum(v) / len(v)))
