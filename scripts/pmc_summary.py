"""Aggregates rocprofv3 counter_collection CSVs (one per pass) into per-kernel averages per launch."""
import csv
import collections
import glob
import os
import sys

directory = sys.argv[1]
per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(directory, "*.csv"))):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            short = name.split("(")[0].replace("void ", "")
            per_kernel[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("kernel,counter,launches,mean_per_launch,total")
for kernel in sorted(per_kernel):
    if not (kernel.startswith("k_") and not kernel.startswith("k_build")):
        continue
    for counter, values in sorted(per_kernel[kernel].items()):
        print("%s,%s,%d,%.6g,%.6g" % (kernel, counter, len(values), sum(values) / len(values), sum(values)))
