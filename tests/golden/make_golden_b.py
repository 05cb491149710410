"""Generates tests/golden/path_b_*.npz from oracle B (run from the repo root).

Path B has NO reference counterpart: these fixtures pin the oracle against drift,
nothing more — "parity unpinned by the reference"."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O  # noqa: E402
from raytracing_engine_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    v, a, e = scenes.cornell_tri_scene()
    rgb, ct = O.TriScene(v, a, e).render(64, 64, spp=4, bounces=2, seed=7, pos=(0, 1, 0))
    np.savez_compressed(os.path.join(HERE, "path_b_cornell_64.npz"), rgb=rgb, counters=np.array([ct["camera_rays"], ct["bounce_rays"], ct["shadow_rays"]], np.uint64))
    print(ct)
    v, a, e = scenes.soup_scene(2000, seed=3, edge=1.5)
    rgb, ct = O.TriScene(v, a, e).render(96, 54, spp=2, bounces=1, seed=5, sky=(0.3, 0.3, 0.4))
    np.savez_compressed(os.path.join(HERE, "path_b_soup2k_96x54.npz"), rgb=rgb, counters=np.array([ct["camera_rays"], ct["bounce_rays"], ct["shadow_rays"]], np.uint64))
    print(ct)
