#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py DIR [DIR...]"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-42:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, v in sorted(agg.items()):
    if k.startswith("__amd"):
        continue
    print(k)
    for c, val in sorted(v.items()):
        n = len(calls[(k, c)])
        print(f"    {c:34s} {val:14.5g}  /dispatch {val / n:12.5g}  (n={n})")
    g = v.get
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        print(f"    -> VALU active/wave-cycle {g('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}  wait_any {g('SQ_WAIT_ANY', 0) / wc:.3f}  wait_inst {g('SQ_WAIT_INST_ANY', 0) / wc:.3f}")
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        print(f"    -> lanes active per VALU instruction {g('SQ_THREAD_CYCLES_VALU') / g('SQ_INSTS_VALU', 1):.1f} / 64")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
        print(f"    -> L2 hit rate {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}")
