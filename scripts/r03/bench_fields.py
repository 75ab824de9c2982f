"""Prints a few fields of a bench.py line read from stdin: `python bench.py ... | python scripts/r03/bench_fields.py label`."""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c, r = d["config"], d.get("roofline") or {}
mv = c.get("kernel_ms_alone_moving")
print(" ".join(sys.argv[1:]), "| Mrays/s", round(d["value"]), "| ms/frame", round(d["ms_per_step"], 4), "| alone ms", round(c["kernel_ms_alone"], 4),
      "| alone, moving ms", round(mv, 4) if mv else None, "| frac", round(r["frac"], 3) if r else None, "| frac_moving", round(r["frac_moving"], 3) if r and r.get("frac_moving") else None, flush=True)
