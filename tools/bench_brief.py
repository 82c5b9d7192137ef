"""Print the headline numbers of a bench.py JSON line read from stdin (diagnostic helper)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read() if len(sys.argv) > 1 and sys.argv[1] != "/dev/stdin" else sys.stdin.read())
r = d["roofline"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "ms/step", round(d["ms_per_step"], 3), "xH",
      d["config"]["global_xH"], "r_loop ms", round(r["r_loop"]["ms"], 3), "frac",
      round(r["r_loop"]["frac"], 4), "dominant", r.get("kernel", "")[:27], "frac",
      round(r.get("frac", 0), 4), "ms", round(r.get("ms_per_launch", 0), 4),
      "others", [round(k["ms"], 4) for k in r.get("other_kernels", [])])
