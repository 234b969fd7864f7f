#!/usr/bin/env python3
"""Averages the rocprofv3 counter CSVs of tools/collect_sq.sh per kernel and prints counter = value (and per 64-line
tile for the tile kernel of a 10 M-line batch)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for path in glob.glob(os.path.join(out, "sq*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "k_extract" not in k and "k_jsonl" not in k and "k_split" not in k:
            continue
        short = k.split("(")[0].split("::")[-1][:60]
        if "k_jsonl_tile" in k:
            short = "k_jsonl_tile<write>" if "k_jsonl_tile<unsigned int, true" in k or "k_jsonl_tile<unsigned long, true" in k else "k_jsonl_tile<sizes>"
        acc[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-62s %-26s %16.0f  (n=%d)" % (k, c, sum(v) / len(v), len(v)))
