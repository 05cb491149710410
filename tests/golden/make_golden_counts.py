"""Generates tests/golden/path_b_tri1m_counts.json: oracle B's ray and traversal counts for the WHOLE
1 M-triangle workloads (BASELINE.json metric config and configs[3]), which take the oracle tens of
seconds and therefore are not recomputed inside the GPU tests (run from the repo root, ~1 min on 8 cores).

Path B has NO reference counterpart: these numbers pin the oracle against drift and give
tests/test_gpu_configs.py and bench.py whole-frame ray counts to compare the kernels' queue counters
with — "parity unpinned by the reference".  nodes_visited / tris_tested are those of the oracle's own
median-split BVH2 with <= 4-triangle leaves (SURVEY.md section 8d's N_node / N_tri), closest-hit and any-hit
traversals together."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O  # noqa: E402
from raytracing_engine_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    sc = O.TriScene(*scenes.soup_scene(1_000_000, seed=1, edge=0.08))
    out = {}
    for name, (w, h, spp, bounces) in {"tri1m_1080p_4spp": (1920, 1080, 4, 1), "tri1m_1080p_8spp": (1920, 1080, 8, 1)}.items():
        _, ct = sc.render(w, h, spp=spp, bounces=bounces, seed=1, sky=(0.2, 0.2, 0.25))
        out[name] = ct
        print(name, ct, flush=True)
    json.dump(out, open(os.path.join(HERE, "path_b_tri1m_counts.json"), "w"), indent=1)
