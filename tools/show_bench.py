#!/usr/bin/env python3
"""One line per bench.py JSON file: value, ms per step, parity, roofline fractions.  python tools/show_bench.py FILE..."""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(path, d["value"], d["unit"], d["ms_per_step"], "ms/step", "parity", (d.get("parity") or {}).get("ok"), "frac", r.get("frac"),
          "valu_issue", (r.get("valu_issue") or {}).get("frac"), "kernel ms", r.get("avg_kernel_ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"),
          "stages", {k[3:]: round(v, 3) for k, v in (r.get("stage_ms") or {}).items()})
