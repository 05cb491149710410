#!/usr/bin/env python3
"""Single-level BVH8 against the two-level build (top level over 64 bottom-level chunks, rt_set_mesh_ex) on the 100 k and 1 M
triangle soups: build times, frame time, node visits, and the cost of rebuilding one chunk.  python tools/two_level_bvh.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_engine_amd as R
CHUNKS = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["64"])]  # python tools/two_level_bvh.py 8,64,512
r = R.Renderer(0)
for n, edge in ((100_000, 0.25), (1_000_000, 0.08)):
    mesh = R.scenes.soup_scene(n, seed=1, edge=edge)
    r.resize(1920, 1080)
    for levels, chunks in [(1, 0)] + [(2, c) for c in CHUNKS]:
        r.set_mesh(*mesh, bvh_levels=levels, blas_chunks=chunks)
        st = r.pt_stats()
        info = {k: st[k] for k in ("n_nodes", "bvh_depth", "bvh_levels", "blas_chunks", "tlas_nodes")}
        info.update({k: round(st[k], 2) for k in ("bvh_build_ms", "ms_build_blas", "ms_build_tlas", "ms_build_flatten")})
        prm = r.pt_params(spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25), count_traversal=True)
        r.render_pt(params=prm); c = r.pt_stats()
        prm = r.pt_params(spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25))
        ms = []
        for _ in range(6):
            r.render_pt(params=prm); ms.append(r.pt_stats()["ms_total"])
        rays = c["camera_rays"] + c["bounce_rays"] + c["shadow_rays"]
        print(n, "levels", levels, info, "ms/frame %.3f" % np.median(ms), "Mrays/s %.0f" % (rays / np.median(ms) / 1e3),
              "shadow nodes/ray %.2f" % (c["shadow_nodes_visited"] / c["shadow_rays"]), "closest fetches", c["nodes_visited"], flush=True)
        if levels == 2:
            sizes = [len(r.mesh_chunk(c)) for c in range(st["blas_chunks"])]
            print("   chunk sizes: min %d max %d" % (min(sizes), max(sizes)), flush=True)
            ids = r.mesh_chunk(7)
            v2 = mesh[0][ids] * np.float32(0.999)  # moved, and inside the coordinate range the mesh was padded for
            t0 = time.perf_counter(); r.update_mesh_chunk(7, v2); dt = (time.perf_counter() - t0) * 1e3
            st = r.pt_stats()
            print("   chunk rebuild: wall %.2f ms (blas %.2f, tlas %.3f, flatten %.2f)" % (dt, st["ms_build_blas"], st["ms_build_tlas"], st["ms_build_flatten"]), flush=True)
