#!/usr/bin/env python3
"""A/B of the path-B scheduling variants on the headline workload inside ONE process (devices differ by a few per cent, so variants
are only comparable within a run): default, rays sorted in LDS, camera rays through the per-lane kernel, default again.  Per-stage
HIP-event times, one frame at a time.   python tools/ab_pt_variants.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_engine_amd as R
r = R.Renderer(0)
r.set_mesh(*R.scenes.soup_scene(1_000_000, seed=1, edge=0.08))
r.resize(1920, 1080)
cfg = r.default_config(); cfg.profile_stages = 1; r.set_config(cfg)
variants = [dict(), dict(tune_sort_rays=1), dict(tune_no_packet=1), dict(tune_no_overlap=1), dict()]
for kw in variants:
    prm = r.pt_params(spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25), **kw)
    r.render_pt(params=prm)
    acc = {}
    for _ in range(6):
        r.render_pt(params=prm); st = r.pt_stats()
        for k in ("ms_total", "ms_generate", "ms_trace_packet", "ms_trace_closest", "ms_trace_fused", "ms_shade", "ms_trace_shadow", "ms_resolve"):
            acc[k] = acc.get(k, 0) + st[k] / 6
    print(kw, {k: round(v, 3) for k, v in acc.items()}, flush=True)
