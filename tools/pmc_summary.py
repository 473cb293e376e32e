"""Summarise rocprofv3 --pmc CSV output (counter_collection) per kernel: mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
for d in sorted(glob.glob(f"{out}/pmc_*")):
    files = glob.glob(f"{d}/**/*counter_collection*.csv", recursive=True)
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")[:60]
                acc[k][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0) or 0))
    print(f"--- {d}")
    for k, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))[:14]:
        print(f"{k:60s} " + "  ".join(f"{c}={sum(v) / len(v):.4g} (n={len(v)})" for c, v in sorted(cs.items())))
