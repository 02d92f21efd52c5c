#!/usr/bin/env python3
"""Collect rocprofv3 --pmc CSVs under a directory: per counter, mean value per recon_kernel dispatch."""
import csv
import collections
import sys
from pathlib import Path

root = Path(sys.argv[1])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(root.rglob("*counter_collection.csv")):
    variant = "-"
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            key = next((k for k in ("recon_wide_kernel", "recon_kernel", "rgba", "audio_kernel") if k in name), None)
            if key is None:
                continue
            agg[variant + " " + key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for variant, d in agg.items():
    print("== variant", variant)
    for k, v in sorted(d.items()):
        print("  %-32s n=%3d mean=%.6g" % (k, len(v), sum(v) / len(v)))
