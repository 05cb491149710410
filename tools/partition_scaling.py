#!/usr/bin/env python3
"""Per-rank cost of a 1/N tile share of the headline frame on ONE GPU (rank 0 of N emulated with
rt_set_partition), for 1 .. --lanes frames in flight (contexts taking the frames round-robin, as bench.py
does), INCLUDING rank 0's de-tile of a full (N, tiles_per_rank, 64, 64, 3) gather buffer on the lane's stream after
every frame: what strong scaling can reach before the xGMI transfer itself (25 MB per frame into rank 0, which one
GPU cannot emulate).   python tools/partition_scaling.py [--lanes 6] [--no-detile]"""
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
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--no-detile", action="store_true")
ap.add_argument("--tune", default="", help="k=v,k=v tuning knobs of rt_pt_params")
a = ap.parse_args()
mesh = R.scenes.soup_scene(1_000_000, seed=1, edge=0.08)
rs, bufs = [], []
for _ in range(a.lanes):
    r = R.Renderer(0)
    r.set_mesh(*mesh)
    r.resize(1920, 1080)
    rs.append(r)
    bufs.append(torch.empty(1920 * 1088 * 3 + 64 * 64 * 3 * 600, dtype=torch.float32, device="cuda"))
tiles = 30 * 17
gathered = [torch.zeros(8 * 64 * 64 * 64 * 3, dtype=torch.float32, device="cuda") for _ in range(a.lanes)]  # room for (N, ceil(510 / N), 64, 64, 3), N <= 8
frames = [torch.empty(1920 * 1080 * 3, dtype=torch.float32, device="cuda") for _ in range(a.lanes)]
tune = dict((k, int(v, 0)) for k, v in (kv.split("=") for kv in a.tune.split(",") if kv))
prm = rs[0].pt_params(spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25), **tune)


def frame(i, lanes, n_ranks):
    k = i % lanes
    rs[k].render_pt_device((0, 0, 0, 1), (0, 0, 0), prm, bufs[k].data_ptr(), True)
    if not a.no_detile:
        rs[k].detile_device(gathered[k].data_ptr(), n_ranks, -(-tiles // n_ranks), frames[k].data_ptr())


def run(lanes, n_ranks):
    for r in rs:
        r.set_partition(0, n_ranks)
    for i in range(2 * lanes):
        frame(i, lanes, n_ranks)
    for r in rs:
        r.synchronize()
    best = None
    for _ in range(a.reps):  # the best of a few repetitions: a single host-side stall of 20-30 ms inside one 24-frame window otherwise doubles a row's entry
        t0 = time.perf_counter()
        for i in range(a.frames):
            frame(i, lanes, n_ranks)
        for r in rs:
            r.synchronize()
        t = (time.perf_counter() - t0) / a.frames * 1e3
        best = t if best is None else min(best, t)
    return best


base = run(1, 1)
print(f"whole frame, one lane: {base:.3f} ms")
for n in (1, 2, 4, 8):
    row = [run(lanes, n) for lanes in range(1, a.lanes + 1)]
    print(f"1/{n} of the frame (ideal {base / n:6.3f} ms): " + "  ".join(f"{lanes} lane{'s' if lanes > 1 else ' '} {t:6.3f} ({base / n / t:.2f})" for lanes, t in enumerate(row, 1)), flush=True)
