#!/usr/bin/env python3
"""Static vector-instruction mix of the traversal kernels (what bench.py's roofline.valu_issue prices): compiles
raytracing_engine_amd/csrc/path_b.hip to gfx950 assembly with the library's flags and counts, per kernel, the wave-level vector
instructions of the fast issue class (v_fma / v_fmac / v_mul / v_add / v_sub f32, v_mov_b32: 2.65-2.87 cycles per SIMD,
profiles/r02_valu_issue_rates.txt) against all others (4.3-4.8 cycles).   python tools/valu_mix.py [kernel-substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-rdc",
         "--cuda-device-only", "-S"]
FAST = re.compile(r"^v_(fma_f32|fmac_f32|mul_f32|add_f32|sub_f32|subrev_f32|mov_b32)(_e32|_e64)?$")

want = sys.argv[1:] or ["pt_trace_fused", "pt_trace<", "pt_trace_packet"]
with tempfile.TemporaryDirectory() as d:
    asm = os.path.join(d, "path_b.s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(ROOT, "raytracing_engine_amd", "csrc", "path_b.hip"), "-o", asm], check=True,
                   stderr=subprocess.DEVNULL)
    cur, stats = None, {}
    for line in open(asm):
        m = re.match(r"^(_ZN2rt\w+):", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            stats[cur] = [0, 0]
            continue
        if cur and "s_endpgm" in line:
            cur = None
            continue
        t = line.split()
        if cur and t and t[0].startswith("v_"):
            stats[cur][1] += 1
            if FAST.match(t[0]):
                stats[cur][0] += 1
for k, (fast, total) in stats.items():
    if any(w in k for w in want) and total:
        print(f"{k:60s} vector instructions {total:5d}  fast class {fast:4d} = {fast / total:.2f}  -> {fast / total * 2.75 + (1 - fast / total) * 4.7:.2f} cycles / instruction")
