"""Generates tests/golden/path_b_tri1m_counts.json: oracle B's ray and traversal counts for the WHOLE
1 M-triangle workloads (BASELINE.json metric config and configs[3]) and for configs[2]'s 100 k-triangle workload, which
take the oracle tens of seconds and therefore are not recomputed inside the GPU tests (run from the repo root, ~2 min on
8 cores;  `python tests/golden/make_golden_counts.py tri100k_1080p_4spp` regenerates one entry and keeps the others).

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

WORKLOADS = {  # name: (triangles, edge, width, height, spp, bounces)
    "tri1m_1080p_4spp": (1_000_000, 0.08, 1920, 1080, 4, 1),
    "tri1m_1080p_8spp": (1_000_000, 0.08, 1920, 1080, 8, 1),
    "tri100k_1080p_4spp": (100_000, 0.25, 1920, 1080, 4, 1),  # configs[2]
}

if __name__ == "__main__":
    path = os.path.join(HERE, "path_b_tri1m_counts.json")
    only = sys.argv[1:]
    out = json.load(open(path)) if only and os.path.exists(path) else {}
    scenes_built = {}
    for name, (n, edge, w, h, spp, bounces) in WORKLOADS.items():
        if only and name not in only:
            continue
        if (n, edge) not in scenes_built:
            scenes_built[(n, edge)] = O.TriScene(*scenes.soup_scene(n, seed=1, edge=edge))
        _, ct = scenes_built[(n, edge)].render(w, h, spp=spp, bounces=bounces, seed=1, sky=(0.2, 0.2, 0.25))
        out[name] = ct
        print(name, ct, flush=True)
    json.dump(out, open(path, "w"), indent=1)
