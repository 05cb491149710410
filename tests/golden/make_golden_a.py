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


def case(name, scene_bytes, w, h, rot, pos, march_algorithm=0, repeat=(0.0, 0.0, 0.0), max_steps=None, reflections=0, reflectivity=0.5,
         transmissions=0, transparency=0.5, refraction_index=1.0):
    scene = O.scene_from_bytes(scene_bytes)
    cfg = O.default_config()
    cfg.march_algorithm = march_algorithm
    cfg.repeat[:] = repeat
    cfg.reflections, cfg.reflectivity = reflections, reflectivity
    cfg.transmissions, cfg.transparency, cfg.refraction_index = transmissions, transparency, refraction_index
    if max_steps:
        cfg.max_steps = max_steps
    r = O.render_a(scene, w, h, rot=rot, pos=pos, cfg=cfg)
    out = {"scene": np.frombuffer(scene_bytes, np.uint8), "width": w, "height": h,
           "rot": np.asarray(rot, np.float32), "pos": np.asarray(pos, np.float32), "rgb": r["rgb"],
           "march_algorithm": march_algorithm, "repeat": np.asarray(repeat, np.float32), "max_steps": int(cfg.max_steps),
           "reflections": reflections, "reflectivity": np.float32(reflectivity),
           "transmissions": transmissions, "transparency": np.float32(transparency), "refraction_index": np.float32(refraction_index),
           "counters": np.array(list(r["counters"].values()), np.uint64)}
    for i, lv in enumerate(r["levels"]):
        out[f"level{i}"] = lv
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, r["counters"])


if __name__ == "__main__":
    only = sys.argv[1:]  # names to (re)generate; none = all
    _case = case

    def case(name, *a, **kw):  # noqa: F811
        if not only or name in only:
            _case(name, *a, **kw)

    case("path_a_default_64.npz", bytes(host.default_scene()), 64, 64, (0, 0, 0, 1), (0, 0, 0))
    case("path_a_default_turn_96x64.npz", bytes(host.default_scene()), 96, 64, host.camera_quat(0.6, -0.2), (1.0, -2.0, 0.5))
    case("path_a_cornell_256.npz", bytes(host.cornell_scene()), 256, 256, (0, 0, 0, 1), (0, 0, 0))
    # SDF feature growth: march algorithms 1 and 2, domain repetition
    case("path_a_alg1_96x64.npz", bytes(host.default_scene()), 96, 64, host.camera_quat(0.3, -0.1), (0.5, -1.0, 0.2), march_algorithm=1, max_steps=4096)
    case("path_a_alg2_96x64.npz", bytes(host.default_scene()), 96, 64, host.camera_quat(0.3, -0.1), (0.5, -1.0, 0.2), march_algorithm=2, max_steps=4096)
    case("path_a_repeat_96x64.npz", bytes(host.default_scene()), 96, 64, host.camera_quat(0.3, -0.1), (0.5, -1.0, 0.2), repeat=(40.0, 0.0, 40.0), max_steps=4096)
    # mirror reflections (fragment.glsl:125 TODO; build-defined): two bounces between the spheres of the start-up scene
    case("path_a_mirror_96x64.npz", bytes(host.default_scene()), 96, 64, host.camera_quat(0.3, -0.1), (0.5, -1.0, 0.2), reflections=2, reflectivity=0.6)
    # transmission (fragment.glsl:124 / :126 TODOs; build-defined): straight through (transparency) and bent by Snell's law (refraction)
    case("path_a_transparency_96x64.npz", bytes(host.default_scene()), 96, 64, host.camera_quat(0.3, -0.1), (0.5, -1.0, 0.2), transmissions=2, transparency=0.6)
    case("path_a_refraction_96x64.npz", bytes(host.default_scene()), 96, 64, host.camera_quat(0.3, -0.1), (0.5, -1.0, 0.2), transmissions=2, transparency=0.6,
         refraction_index=1.5)
