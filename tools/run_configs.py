#!/usr/bin/env python3
"""Runs the five BASELINE.json configs on ONE MI355X (+ the CPU oracle on the box's host cores) and
prints the table BASELINE.md asks for:  python tools/run_configs.py [--out profiles/r01_configs.json]

GPU numbers: HIP-event time around the stage loop (rt_get_stats / rt_get_pt_stats ms_total), median
of `--reps` after one warm-up, inputs resident in HBM, no read-back in the timed region.
CPU numbers: the oracle (same source that defines parity), 1 thread and all granted cores, on the
bounded sample stated per row.  Parity column: max-abs RGB error of the GPU frame against the oracle
(full frame for path A, a 16-row band of the full-size frame for path B).
Configs 4 and 5 name 8 GPUs; this script reports their single-GPU rate (the driver measures 2/4/8)."""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402  (CPU baseline + parity column only)
import raytracing_engine_amd as R  # noqa: E402
from bench import host_threads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--out", default="")
a = ap.parse_args()
threads = host_threads()
r = R.Renderer(0)
rows = []


def timed(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        ts.append(fn())
    return statistics.median(ts)


def path_a(name, w, h, spp, gpu=True):
    scene = R.cornell_scene()
    osc = O.scene_from_bytes(bytes(scene))
    row = {"config": name, "oracle": "A", "alg_bytes": "27 B/pixel"}
    n = int(round(spp ** 0.5))

    def cpu(th):
        t0 = time.perf_counter()
        rays = 0
        acc = None
        for s in range(spp):
            i, j = s % n, s // n
            jit = (((np.float32(2 * i + 1) / np.float32(n)) - np.float32(1)) / np.float32(w), ((np.float32(2 * j + 1) / np.float32(n)) - np.float32(1)) / np.float32(h))
            o = O.render_a(osc, w, h, jitter=jit, want_levels=False, threads=th)
            rays += w * h + o["counters"]["shadow_rays"]
            acc = o["rgb"] if acc is None else acc + o["rgb"]
        return rays / (time.perf_counter() - t0) / 1e6, acc / np.float32(spp)

    row["cpu_1core_mrays"], _ = cpu(1)
    row["cpu_ncore_mrays"], ref = cpu(threads)
    row["cpu_cores"] = threads
    row["cpu_sample"] = "full workload"
    if gpu:
        r.set_scene(scene)
        r.resize(w, h)

        def run():
            r.render(spp=spp)
            return r.stats()["ms_total"]

        ms = timed(run, a.reps)
        st = r.stats()
        rays = st["primary_rays"] + st["shadow_rays"]
        row["gpu_ms"], row["gpu_mrays"] = ms, rays / ms / 1e3
        row["max_abs_err"] = float(np.abs(r.render(spp=spp) - ref).max())
        row["algorithmic_over_hbm_peak"] = (w * h * spp * 27.0) / (ms * 1e-3) / 8e12
    rows.append(row)


def path_b(name, n_tris, edge, w, h, spp, bounces, cpu_rows, cpu_spp, levels=1, cpu=True):
    mesh = R.scenes.soup_scene(n_tris, seed=1, edge=edge)
    r.set_mesh(*mesh, bvh_levels=levels)
    r.resize(w, h)
    sky = (0.2, 0.2, 0.25)
    prm = r.pt_params(spp=spp, bounces=bounces, seed=1, sky=sky)

    def run():
        r.render_pt(params=prm)
        return r.pt_stats()["ms_total"]

    ms = timed(run, max(1, a.reps if w * h * spp < 4e8 else 1))
    st = r.pt_stats()
    rays = st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
    r.render_pt(params=r.pt_params(spp=min(spp, 4), bounces=bounces, seed=1, sky=sky, count_traversal=True))
    ct = r.pt_stats()
    nr = ct["camera_rays"] + ct["bounce_rays"] + ct["shadow_rays"]
    # records fetched: once per ray by the per-lane kernels, once per wave of 64 camera rays by the packet kernel
    bytes_per_ray = ((ct["nodes_visited"] + ct["shadow_nodes_visited"] + ct["packet_nodes_fetched"]) * 80.0
                     + (ct["tris_tested"] + ct["shadow_tris_tested"] + ct["packet_tris_fetched"]) * 48.0) / nr + 45.0
    row = {"config": name, "oracle": "B", "gpu_ms": ms, "gpu_mrays": rays / ms / 1e3, "rays": rays,
           "alg_bytes": f"{bytes_per_ray:.0f} B/ray", "algorithmic_over_hbm_peak": rays * bytes_per_ray / (ms * 1e-3) / 8e12,
           "bvh_nodes": st["n_nodes"], "bvh_levels": st["bvh_levels"], "bvh_build_ms": st["bvh_build_ms"]}
    # parity on a band of the full-size frame
    gpu = r.render_pt(params=prm)
    osc = O.TriScene(*mesh)
    y0 = h // 2
    band, _ = osc.render(w, h, spp=spp, bounces=bounces, seed=1, sky=sky, rows=(y0, y0 + 4), threads=threads)
    row["max_abs_err"] = float(np.abs(gpu[y0:y0 + 4] - band).max())

    if not cpu:
        row["cpu_cores"] = threads
        rows.append(row)
        return

    def cpu(th):
        t0 = time.perf_counter()
        _, c = osc.render(w, h, spp=cpu_spp, bounces=bounces, seed=1, sky=sky, rows=cpu_rows, threads=th)
        return (c["camera_rays"] + c["bounce_rays"] + c["shadow_rays"]) / (time.perf_counter() - t0) / 1e6

    one_rows = (cpu_rows[0], cpu_rows[0] + max(1, (cpu_rows[1] - cpu_rows[0]) // 8))
    t0 = time.perf_counter()
    _, c = osc.render(w, h, spp=cpu_spp, bounces=bounces, seed=1, sky=sky, rows=one_rows, threads=1)
    row["cpu_1core_mrays"] = (c["camera_rays"] + c["bounce_rays"] + c["shadow_rays"]) / (time.perf_counter() - t0) / 1e6
    row["cpu_ncore_mrays"] = cpu(threads)
    row["cpu_cores"] = threads
    row["cpu_sample"] = f"rows {cpu_rows[0]}..{cpu_rows[1]} of the frame at {cpu_spp} spp"
    rows.append(row)


path_a("1: 8 spheres + 1 light, 256x256, 1 spp (CPU only)", 256, 256, 1, gpu=False)
path_a("2: same scene, 1920x1080, 4 spp", 1920, 1080, 4)
path_b("3: 100 k triangles, 2-level BVH (top level over 64 SAH-cut chunks), 1920x1080, 4 spp, 1 bounce", 100_000, 0.25, 1920, 1080, 4, 1, (0, 1080), 1, levels=2)
path_b("3 (single-level BVH8, for comparison): as 3", 100_000, 0.25, 1920, 1080, 4, 1, (0, 1080), 1, cpu=False)
path_b("4: 1 M triangles, 1920x1080, 8 spp, 1 bounce (1 GPU of 8)", 1_000_000, 0.08, 1920, 1080, 8, 1, (0, 1080), 1)
path_b("5: 1 M triangles, 3840x2160, 64 spp, 8 bounces (1 GPU of 8)", 1_000_000, 0.08, 3840, 2160, 64, 8, (1000, 1128), 1)

print("| # | config | oracle | CPU Mrays/s (1 core) | CPU Mrays/s (N cores) | N | 1 GPU Mrays/s | ms/frame | alg. bytes | alg. bytes/s over HBM peak | max-abs RGB err |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for x in rows:
    f = lambda k, fmt="{:.1f}": fmt.format(x[k]) if x.get(k) is not None else "n/a"
    print(f"| {x['config'][:1]} | {x['config'][3:]} | {x['oracle']} | {f('cpu_1core_mrays', '{:.2f}')} | {f('cpu_ncore_mrays', '{:.2f}')} | {x['cpu_cores']} | "
          f"{f('gpu_mrays')} | {f('gpu_ms', '{:.3f}')} | {x['alg_bytes']} | {f('algorithmic_over_hbm_peak', '{:.3f}')} | {f('max_abs_err', '{:.2e}')} |")
if a.out:
    json.dump(rows, open(a.out, "w"), indent=1)
