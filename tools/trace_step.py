"""Print the kernel timeline of the last step of a `rocprofv3 --kernel-trace` run of bench.py.
usage: python tools/trace_step.py <dir with *_kernel_trace.csv> [anchor-kernel-substring | tail:N]"""
import csv
import glob
import sys

d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "project_fused_fwd"
f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
if anchor.startswith("tail:"):           # the last N launches of the run
    a, b = max(0, len(rows) - int(anchor[5:]) - 1), len(rows) - 1
else:
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev = None
ksum = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) if prev else 0
    ksum += e - s
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-58:]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  grid {r['Grid_Size_X']:>9} {name}")
    prev = e
span = int(rows[b]["Start_Timestamp"]) - t0
print(f"step span {span / 1e3:.1f} us, kernels {ksum / 1e3:.1f} us, idle {(span - ksum) / 1e3:.1f} us, launches {b - a}")
