import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import raytracing_engine_amd as R
r = R.Renderer(0)
for name, lights in (("cornell 1 light", 1), ("cornell 0 lights", 0)):
    sc = R.cornell_scene()
    sc.lightCount = lights
    r.set_scene(sc)
    r.resize(1920, 1080)
    cfg = r.default_config(); cfg.profile_stages = 1; r.set_config(cfg)
    acc = None
    for _ in range(6):
        r.render(spp=4); st = r.stats()
        v = np.array([st["ms_total"], st["ms_cone"], st["ms_shade"]] + list(st["ms_level"]))
        acc = v if acc is None else acc + v
    acc /= 6
    print(name, "total %.3f cone %.3f shade %.3f" % tuple(acc[:3]), "levels", np.round(acc[3:], 3), "hit px", st["hit_pixels"], "shadow rays", st["shadow_rays"], flush=True)
