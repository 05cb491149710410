"""Generates tests/golden/path_a_*.npz from oracle A (run from the repo root:
`python tests/golden/make_golden_a.py`).

The reference ships no golden vectors and cannot be built here, so these fixtures
pin the ORACLE (against accidental edits / compiler or libm drift on another host), not the
reference: "parity unpinned by the reference".  Inputs are the reference's own start-up scene
(src/main.rs:524-591) and the BASELINE.json configs[0] scene.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O  # noqa: E402
from raytracing_engine_amd import host  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def case(name, scene_bytes, w, h, rot, pos):
    scene = O.scene_from_bytes(scene_bytes)
    r = O.render_a(scene, w, h, rot=rot, pos=pos)
    out = {"scene": np.frombuffer(scene_bytes, np.uint8), "width": w, "height": h,
           "rot": np.asarray(rot, np.float32), "pos": np.asarray(pos, np.float32), "rgb": r["rgb"],
           "counters": np.array(list(r["counters"].values()), np.uint64)}
    for i, lv in enumerate(r["levels"]):
        out[f"level{i}"] = lv
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, r["counters"])


if __name__ == "__main__":
    case("path_a_default_64.npz", bytes(host.default_scene()), 64, 64, (0, 0, 0, 1), (0, 0, 0))
    case("path_a_default_turn_96x64.npz", bytes(host.default_scene()), 96, 64, host.camera_quat(0.6, -0.2), (1.0, -2.0, 0.5))
    case("path_a_cornell_256.npz", bytes(host.cornell_scene()), 256, 256, (0, 0, 0, 1), (0, 0, 0))
