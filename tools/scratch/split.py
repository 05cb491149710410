import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracing_engine_amd as R
r = R.Renderer(0)
r.set_mesh(*R.scenes.soup_scene(1_000_000, seed=1, edge=0.08))
r.resize(1920, 1080)
cfg = r.default_config(); cfg.profile_stages = 1; r.set_config(cfg)
for b in (0, 1):
    prm = r.pt_params(spp=4, bounces=b, seed=1, sky=(0.2, 0.2, 0.25), count_traversal=True)
    r.render_pt(params=prm); c = r.pt_stats()
    prm = r.pt_params(spp=4, bounces=b, seed=1, sky=(0.2, 0.2, 0.25))
    acc = {}
    for _ in range(5):
        r.render_pt(params=prm); st = r.pt_stats()
        for k in ("ms_total", "ms_generate", "ms_trace_closest", "ms_shade", "ms_trace_shadow", "ms_resolve"):
            acc[k] = acc.get(k, 0) + st[k] / 5
    cr = c["camera_rays"] + c["bounce_rays"]
    print("bounces", b, {k: round(v, 3) for k, v in acc.items()}, "rays", c["camera_rays"], c["bounce_rays"], c["shadow_rays"],
          "nodes/ray", round(c["nodes_visited"] / cr, 2), "tris/ray", round(c["tris_tested"] / cr, 2), "shadow nodes/ray", round(c["shadow_nodes_visited"] / max(1, c["shadow_rays"]), 2),
          "rounds", c["wave_rounds"], "alive/round", round(c["alive_lane_rounds"] / max(1, c["wave_rounds"]), 1), flush=True)
