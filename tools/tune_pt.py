#!/usr/bin/env python3
"""Tuning harness for the path-B trace kernels (run on the GPU box):
    python tools/tune_pt.py [--tris 1000000] [--refill 8,16,32] [--blocks 0,4]
Prints per-stage HIP-event times for each parameter combination on the bench scene."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_engine_amd as R  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--edge", type=float, default=0.08)
ap.add_argument("--refill", default="16")
ap.add_argument("--blocks", default="0")
ap.add_argument("--lds", default="0")
ap.add_argument("--k", default="0")
ap.add_argument("--gate", default="0")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--scene", default="soup")
ap.add_argument("--ranks", type=int, default=1, help="render rank 0's share of an N-rank tile split")
ap.add_argument("--count", action="store_true", help="also print traversal counters per configuration")
a = ap.parse_args()

r = R.Renderer(0)
mesh = (R.scenes.soup_scene(a.tris, seed=1, edge=a.edge) if a.scene == "soup" else R.scenes.terrain_scene(708, seed=1) if a.scene == "terrain"
        else R.scenes.cornell_tri_scene())
r.set_mesh(*mesh)
r.resize(1920, 1080)
r.set_partition(0, a.ranks)
cfg = r.default_config()
cfg.profile_stages = 1
r.set_config(cfg)
pos = (0, 0, 0) if a.scene == "soup" else (0, 0, 4) if a.scene == "terrain" else (0, 1, 0)
rot = R.camera_quat(0.0, -0.25) if a.scene == "terrain" else (0, 0, 0, 1)
print("bvh", {k: r.pt_stats()[k] for k in ("n_nodes", "bvh_depth", "stack_need", "bvh_build_ms")})
r.render_pt(rot=rot, pos=pos, params=r.pt_params(spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25), count_traversal=True))
_c = r.pt_stats()
_cr = _c["camera_rays"] + _c["bounce_rays"]
print(f"per closest ray: nodes {_c['nodes_visited'] / _cr:.1f} tris {_c['tris_tested'] / _cr:.1f}; per shadow ray: nodes "
      f"{_c['shadow_nodes_visited'] / max(_c['shadow_rays'], 1):.1f} tris {_c['shadow_tris_tested'] / max(_c['shadow_rays'], 1):.1f}; "
      f"rays cam {_c['camera_rays']} bounce {_c['bounce_rays']} shadow {_c['shadow_rays']}")
print(f"closest kernels: wave-rounds {_c['wave_rounds']}, alive lanes/round {_c['alive_lane_rounds'] / max(_c['wave_rounds'], 1):.1f}, "
      f"lanes in node phase/round {_c['nodes_visited'] / max(_c['wave_rounds'], 1):.1f}, lanes in tri phase/round {_c['tris_tested'] / max(_c['wave_rounds'], 1):.1f}")
import itertools

ints = lambda v: [int(x) for x in v.split(",")]
for refill, kk, gate, lds, blocks in itertools.product(ints(a.refill), ints(a.k), ints(a.gate), ints(a.lds), ints(a.blocks)):
    prm = r.pt_params(spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25), tune_refill_min=refill | (kk << 8) | (gate << 16),  # gate byte = local refill threshold
                      tune_blocks_per_cu=blocks, tune_lds_stack=lds)
    r.render_pt(rot=rot, pos=pos, params=prm)
    acc = {}
    for _ in range(a.reps):
        r.render_pt(rot=rot, pos=pos, params=prm)
        st = r.pt_stats()
        for k in ("ms_total", "ms_generate", "ms_trace_closest", "ms_shade", "ms_trace_shadow", "ms_resolve"):
            acc[k] = acc.get(k, 0.0) + st[k] / a.reps
    rays = st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
    print(f"refill={refill:2d} k={kk} gate={gate:2d} lds={lds:2d} blocks={blocks} Mrays/s={rays / acc['ms_total'] / 1e3:8.1f} "
          + " ".join(f"{k[3:]}={v:7.3f}" for k, v in acc.items()), flush=True)
    if a.count:
        r.render_pt(rot=rot, pos=pos, params=r.pt_params(spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25), count_traversal=True, tune_refill_min=refill | (kk << 8) | (gate << 16),
                                                tune_blocks_per_cu=blocks, tune_lds_stack=lds))
        c = r.pt_stats()
        cr = c["camera_rays"] + c["bounce_rays"]
        print(f"    nodes/ray {c['nodes_visited'] / cr:.1f} tris/ray {c['tris_tested'] / cr:.1f} wave-rounds {c['wave_rounds']} alive/round "
              f"{c['alive_lane_rounds'] / max(c['wave_rounds'], 1):.1f} node lanes/round {c['nodes_visited'] / max(c['wave_rounds'], 1):.1f}", flush=True)
