import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["parity"]["ok"], d["roofline"]["frac"], d["roofline"]["valu_issue"]["frac"], d["roofline"]["avg_kernel_ms"], d["cpu_baseline"]["value"])
