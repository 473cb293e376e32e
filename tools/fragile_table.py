"""Print tests/test_gpu_parity.py's FRAGILE_OBSERVED table from a `pytest -s` log of the GPU suite
(lines "[fragile <tag>] <fraction> (bound ...)"): python tools/fragile_table.py gpurun_out/<visit>/pytest_gpu.log"""
import re
import sys

seen = {}
for ln in open(sys.argv[1], errors="replace"):
    m = re.search(r"\[fragile (.+?)\] ([0-9.]+) \(bound", ln)
    if m:
        seen[m.group(1)] = max(seen.get(m.group(1), 0.0), float(m.group(2)))
for k in sorted(seen):
    print(f"    {k!r}: {seen[k]:.5f},")
