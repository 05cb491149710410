#!/usr/bin/env python3
"""Per-rank cost of a 1/N tile share of the headline frame on ONE GPU (rank 0 and rank N-1 of N emulated
with rt_set_partition): what strong scaling can reach before the exchange.  python tools/partition_scaling.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import raytracing_engine_amd as R  # noqa: E402

r = R.Renderer(0)
r.set_mesh(*R.scenes.soup_scene(1_000_000, seed=1, edge=0.08))
r.resize(1920, 1080)
prm = r.pt_params(spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25))
buf = torch.empty(1920 * 1088 * 3 + 64 * 64 * 3 * 600, dtype=torch.float32, device="cuda")
base = None
for n in (1, 2, 4, 8):
    for rank in ((0,) if n == 1 else (0, n - 1)):
        r.set_partition(rank, n)
        for _ in range(3):
            r.render_pt_device((0, 0, 0, 1), (0, 0, 0), prm, buf.data_ptr(), True)
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            r.render_pt_device((0, 0, 0, 1), (0, 0, 0), prm, buf.data_ptr(), True)
        r.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        base = base or dt
        cfg = r.default_config()
        cfg.profile_stages = 1
        r.set_config(cfg)
        r.render_pt((0, 0, 0, 1), (0, 0, 0), params=prm)
        st = r.pt_stats()
        cfg.profile_stages = 0
        r.set_config(cfg)
        stages = " ".join(f"{k[3:]}={st[k]:.3f}" for k in ("ms_generate", "ms_trace_closest", "ms_shade", "ms_trace_shadow", "ms_resolve"))
        print(f"ranks {n} rank {rank}: {dt:7.3f} ms/step (ideal {base / n:6.3f}, efficiency {base / n / dt:.2f})   stages, serialised: {stages}", flush=True)
