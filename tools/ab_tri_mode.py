#!/usr/bin/env python3
"""A/B of the triangle-test schedules of the per-lane traversal kernels (rt_pt_params.tune_tri_mode) on the headline workload,
inside ONE process (devices of the pool differ by a few per cent): inline triangle phase against wave-pooled tests, with the pool's
flush parameters, the LDS share of the traversal stack and the resident workgroups swept.  Per-stage HIP-event times, one frame at
a time, plus the traversal counters (nodes per ray move with how late a ray learns its tmax).
    python tools/ab_tri_mode.py [--quick] [--scene soup|terrain] [--tris N]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_engine_amd as R  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
ap.add_argument("--scene", default="soup")
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--edge", type=float, default=0.08)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--variants", default="", help="semicolon-separated k=v,k=v lists to run instead of the built-in sweep")
a = ap.parse_args()

r = R.Renderer(0)
mesh = R.scenes.soup_scene(a.tris, seed=1, edge=a.edge) if a.scene == "soup" else R.scenes.terrain_scene(708, seed=1)
r.set_mesh(*mesh)
r.resize(1920, 1080)
cfg = r.default_config()
cfg.profile_stages = 1
r.set_config(cfg)
pos = (0, 0, 0) if a.scene == "soup" else (0, 0, 4)
rot = R.camera_quat(0.0, -0.25) if a.scene == "terrain" else (0, 0, 0, 1)
base = dict(spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25))


def pool(flush=0, wait=0, mode=2):
    return mode | (flush << 8) | (wait << 16)


def defer(hold=0, stuck=0):
    return pool(hold, stuck, 3)


if a.variants:
    variants = [dict((k, int(v, 0)) for k, v in (kv.split("=") for kv in item.split(",") if kv)) for item in a.variants.split(";")]
elif a.quick:
    variants = [dict(tune_tri_mode=1), dict(tune_tri_mode=4), dict(tune_tri_mode=4, tune_refill_min=4), dict(tune_tri_mode=4, tune_refill_min=16), dict(tune_tri_mode=4, tune_lds_stack=10),
                dict(tune_tri_mode=4, tune_blocks_per_cu=6), dict(tune_tri_mode=4, tune_blocks_per_cu=5), dict(tune_tri_mode=1)]
else:
    variants = [dict(tune_tri_mode=1)]
    variants += [dict(tune_tri_mode=pool(f, w)) for f in (16, 32, 48) for w in (3, 6)]
    variants += [dict(tune_tri_mode=pool(), tune_lds_stack=l) for l in (6, 10)]
    variants += [dict(tune_tri_mode=pool(), tune_blocks_per_cu=b) for b in (4, 6)]
    variants += [dict(tune_tri_mode=defer(h, u)) for h in (8, 16, 24, 32, 40, 48) for u in (2, 4, 8, 16)]
    variants += [dict(tune_tri_mode=defer(), tune_refill_min=m) for m in (8, 16, 32)]
    variants += [dict(tune_tri_mode=defer(), tune_blocks_per_cu=b) for b in (4, 6, 8)]
    variants += [dict(tune_tri_mode=1)]

keys = ("ms_total", "ms_trace_packet", "ms_trace_fused", "ms_shade", "ms_trace_shadow", "ms_trace_closest")
ref = None
for kw in variants:
    prm = r.pt_params(**base, **kw)
    img = r.render_pt(rot=rot, pos=pos, params=prm)
    if ref is None:
        ref = img
    same = bool((img == ref).all())
    acc = {}
    for _ in range(a.reps):
        r.render_pt(rot=rot, pos=pos, params=prm)
        st = r.pt_stats()
        for k in keys:
            acc[k] = acc.get(k, 0.0) + st[k] / a.reps
    rays = st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
    r.render_pt(rot=rot, pos=pos, params=r.pt_params(**base, count_traversal=True, **kw))
    c = r.pt_stats()
    cr = max(c["bounce_rays"], 1)
    desc = " ".join(f"{k[5:]}={v:#x}" if k == "tune_tri_mode" else f"{k[5:]}={v}" for k, v in kw.items())
    print(f"{desc:42s} same={same} Mrays/s={rays / acc['ms_total'] / 1e3:7.1f} " + " ".join(f"{k[3:]}={v:6.3f}" for k, v in acc.items())
          + f" | closest nodes/ray {c['nodes_visited'] / cr:.1f} tris/ray {c['tris_tested'] / cr:.2f}; shadow nodes/ray {c['shadow_nodes_visited'] / max(c['shadow_rays'], 1):.1f}"
          f" tris/ray {c['shadow_tris_tested'] / max(c['shadow_rays'], 1):.2f}; rounds {c['wave_rounds']} alive/round {c['alive_lane_rounds'] / max(c['wave_rounds'], 1):.1f}"
          f" all rounds {c['wave_rounds_all']} flushes {c['pool_flushes']} tris/flush {(c['tris_tested'] + c['shadow_tris_tested']) / max(c['pool_flushes'], 1):.1f}"
          f" rays b {c['bounce_rays']} s {c['shadow_rays']}; packet nodes/wave {c['packet_nodes_fetched'] / max(c['packets'], 1):.1f} tris/wave {c['packet_tris_fetched'] / max(c['packets'], 1):.1f}", flush=True)
