#!/usr/bin/env python3
"""Per-rank cost of a 1/N tile share of the headline frame on ONE GPU (rank 0 of N emulated with
rt_set_partition), for 1 .. --lanes frames in flight (contexts taking the frames round-robin, as bench.py
does): what strong scaling can reach before the exchange.   python tools/partition_scaling.py [--lanes 6]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import raytracing_engine_amd as R  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=6)
ap.add_argument("--frames", type=int, default=24)
a = ap.parse_args()
mesh = R.scenes.soup_scene(1_000_000, seed=1, edge=0.08)
rs, bufs = [], []
for _ in range(a.lanes):
    r = R.Renderer(0)
    r.set_mesh(*mesh)
    r.resize(1920, 1080)
    rs.append(r)
    bufs.append(torch.empty(1920 * 1088 * 3 + 64 * 64 * 3 * 600, dtype=torch.float32, device="cuda"))
prm = rs[0].pt_params(spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25))


def run(lanes, n_ranks):
    for r in rs:
        r.set_partition(0, n_ranks)
    for i in range(2 * lanes):
        rs[i % lanes].render_pt_device((0, 0, 0, 1), (0, 0, 0), prm, bufs[i % lanes].data_ptr(), True)
    for r in rs:
        r.synchronize()
    t0 = time.perf_counter()
    for i in range(a.frames):
        rs[i % lanes].render_pt_device((0, 0, 0, 1), (0, 0, 0), prm, bufs[i % lanes].data_ptr(), True)
    for r in rs:
        r.synchronize()
    return (time.perf_counter() - t0) / a.frames * 1e3


base = run(1, 1)
print(f"whole frame, one lane: {base:.3f} ms")
for n in (1, 2, 4, 8):
    row = [run(lanes, n) for lanes in range(1, a.lanes + 1)]
    print(f"1/{n} of the frame (ideal {base / n:6.3f} ms): " + "  ".join(f"{lanes} lane{'s' if lanes > 1 else ' '} {t:6.3f} ({base / n / t:.2f})" for lanes, t in enumerate(row, 1)), flush=True)
