"""Print the headline and the per-kernel times of a bench.py JSON line (development aid)."""
import json
import sys

for path in sys.argv[1:]:
    d = json.load(open(path))
    print(path, round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 4))
    print("  ", {k: round(v["ms_per_step"], 4) for k, v in d.get("kernels", {}).items()})
    if "roofline" in d:
        print("   roofline frac", round(d["roofline"]["frac"], 4))
