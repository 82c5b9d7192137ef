"""Top kernels of a rocprofv3 *_kernel_stats.csv (diagnostic helper)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    name = r["Name"]
    if "(anonymous namespace)::" in name:
        name = name.split("(anonymous namespace)::")[1]
    print(f"{name[:44]:46s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Percentage']}%")
