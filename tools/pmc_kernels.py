#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc run: tools/pmc_kernels.py <dir> -> kernel, launches, counter averages."""
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(rows.items()):
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print(k, {c: (len(v), sum(v[skip:]) / max(1, len(v[skip:]))) for c, v in cs.items()})
