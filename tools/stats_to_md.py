"""rocprofv3 kernel_stats CSV -> the markdown table kept under profiles/.  usage: stats_to_md.py <csv> <title line> [note]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
print(f"# {sys.argv[2]}\n")
if len(sys.argv) > 3:
    print(sys.argv[3] + "\n")
print("| kernel | calls | total ms | avg us | % | min us | max us |\n|---|---|---|---|---|---|---|")
for r in rows[:40]:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    name = name if len(name) < 90 else name[:87] + "..."
    print(f"| `{name}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | "
          f"{float(r['Percentage']):.2f} | {int(r['MinNs']) / 1e3:.1f} | {int(r['MaxNs']) / 1e3:.1f} |")
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"\nkernel time of the run: {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches")
