"""CPU tests of oracle A (the restatement of compute.glsl / fragment.glsl / host launch logic).

The reference holds no tests or golden vectors, so the oracle is pinned by
analytic known-answer tests (SURVEY.md §8c i-vii), by the independent brute-force marcher of
shaders/tracing_algorithms.txt:2-13, and by committed fixtures that detect drift of the oracle
itself.  "parity unpinned by the reference".
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

import oracle as O
from raytracing_engine_amd import host


def one_sphere(center=(0, 10, 0), radius=2.0, lights=()):
    return O.scene_from_bytes(bytes(host.make_scene([(*center, radius)], [(1, 1, 1, 1, 0.05)], list(lights))))


def ray_sphere_t(origin, direction, center, radius):
    o, d, c = (np.asarray(a, np.float64) for a in (origin, direction, center))
    oc = o - c
    b = np.dot(oc, d)
    disc = b * b - (np.dot(oc, oc) - radius * radius)
    return None if disc < 0 else -b - math.sqrt(disc)


# (v) std140 layout: sizes and offsets of SURVEY.md §8b
def test_std140_layout():
    assert C.sizeof(O.Scene) == 656 and C.sizeof(O.Material) == 32 and C.sizeof(O.Object) == 16 and C.sizeof(O.Light) == 32
    assert O.Scene.mats.offset == 16 and O.Scene.objs.offset == 272 and O.Scene.lights.offset == 400
    assert O.Material.shine.offset == 20 and O.Material.ambient.offset == 24 and O.Light.color.offset == 16


# (vi) pyramid table of SURVEY.md §2c (derived from src/main.rs:209-213, 639)
PYRAMIDS = {
    (256, 256): [(8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (256, 256)],
    (1920, 1080): [(16, 16), (32, 24), (64, 40), (120, 72), (240, 136), (480, 272), (960, 544), (1920, 1080)],
    (3840, 2160): [(16, 16), (32, 24), (64, 40), (120, 72), (240, 136), (480, 272), (960, 544), (1920, 1080), (3840, 2160)],
}


@pytest.mark.parametrize("res", list(PYRAMIDS))
def test_pyramid_dims(res):
    w, h = res
    count = O.level_count(w)
    assert count == len(PYRAMIDS[res])
    assert [O.level_dims(w, h, count, i) for i in range(count)] == PYRAMIDS[res]


def test_level_count_edges():
    assert O.level_count(1) == 1 and O.level_count(8) == 1 and O.level_count(15) == 1 and O.level_count(16) == 2
    assert O.level_count(2048) == 9 and O.level_count(4096) == 9 and O.level_count(16384) == 9  # cap, main.rs:359
    # f32 formula of src/main.rs:639 gives the same floor(log2(w/8)) + 1
    for w in list(range(8, 600)) + [1000, 1920, 2047, 2048, 3840]:
        ref = min(int(np.log2(np.float32(w) / np.float32(8.0))) + 1, 9)
        assert O.level_count(w) == ref, w


# (iv) quaternion rotate vs rotation-matrix form
def test_rotate_matches_matrix():
    rng = np.random.default_rng(1)
    for _ in range(50):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        x, y, z, w = q
        m = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        v = rng.normal(size=3)
        np.testing.assert_allclose(O.rotate(q, v), m @ v, atol=2e-6)


def test_camera_quat_matches_glam_definition():
    for yaw, pitch in [(0, 0), (0.3, 0.2), (-1.2, 1.0), (3.0, -1.5)]:
        q = O.camera_quat(yaw, pitch)
        qz = np.array([0, 0, math.sin(-yaw / 2), math.cos(-yaw / 2)])
        qx = np.array([math.sin(pitch / 2), 0, 0, math.cos(pitch / 2)])
        # Hamilton product qz * qx
        x1, y1, z1, w1 = qz
        x2, y2, z2, w2 = qx
        ref = [w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
               w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2]
        np.testing.assert_allclose(q, ref, atol=1e-6)
        np.testing.assert_allclose(host.camera_quat(yaw, pitch), q, atol=1e-6)
    # yaw > 0 turns the forward axis (+Y) towards +X (src/main.rs:402: from_rotation_z(-yaw))
    f = O.rotate(O.camera_quat(0.5, 0), (0, 1, 0))
    assert f[0] > 0 and abs(f[2]) < 1e-6


# (i) cone-march distance vs closed-form ray/sphere distance: conservative and within O(radius)
@pytest.mark.parametrize("threshold", [0.75, 0.05, 0.006])
def test_trace_cone_is_conservative(threshold):
    sc = one_sphere()
    rng = np.random.default_rng(2)
    n_hit = 0
    for _ in range(300):
        d = np.array([rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.3)])
        d /= np.linalg.norm(d)
        t = O.trace_cone(sc, (0, 0, 0), d, threshold)
        tstar = ray_sphere_t((0, 0, 0), d, (0, 10, 0), 2.0)
        if tstar is not None:
            n_hit += 1
            assert t <= tstar + 1e-3  # never passes the surface
            # stops where the cone touches the sphere: within radius/sin-like slack of t*
            assert tstar - t <= 3.0 * (tstar + 1.0) * threshold + 2.0 * math.sqrt(2.0 * 2.0 * (tstar + 1) * threshold) + 1e-3
        if t >= 1000.0:
            assert tstar is None or True
    assert n_hit > 50


# (ii) a ray whose cone misses every sphere returns >= RENDER_DIST
def test_trace_cone_miss():
    sc = one_sphere()
    for d in [(0, -1, 0), (1, 0, 0), (0.0, 0.6, 0.8)]:
        assert O.trace_cone(sc, (0, 0, 0), d, 0.006) >= 1000.0


def test_trace_cone_vs_bruteforce_algorithm1():
    """tracing_algorithms.txt:2-13 re-evaluates every SDF every step.  The shipped lazy variant
    (algorithm 3) samples the ray at other positions, so the two may disagree on rays that graze a
    sphere's cone-inflated silhouette, and nowhere else; where both hit they stop within a few
    cone radii of each other."""
    sc = O.default_scene()
    rng = np.random.default_rng(3)
    both = disagree = 0
    n = 1000
    for _ in range(n):
        d = np.array([rng.uniform(-1, 1), 1.0, rng.uniform(-0.6, 0.6)])
        d /= np.linalg.norm(d)
        thr = rng.choice([0.75, 0.09, 0.0059])
        a = O.trace_cone(sc, (0, 0, 0), d, thr)
        b = O.trace_cone(sc, (0, 0, 0), d, thr, brute=True)
        if (a >= 1000.0) != (b >= 1000.0):
            disagree += 1
        elif a < 1000.0:
            both += 1
            assert abs(a - b) <= 3.0 * (max(a, b) + 1.0) * thr + 1e-3
    assert both > 100 and disagree <= 0.02 * n, (both, disagree)


# (iii) pixel-centre mapping: level-(count-1) rays go through pixel centres (2x+1)/W - 1
def test_pixel_centre_mapping():
    # The finest-level cone is sqrt(2)*8/W NDC wide (compute.glsl:75), i.e. a tiny sphere is seen by
    # a disc of pixels ~5.7 px in radius; that disc must be centred on the pixel whose centre ray
    # points at the sphere, and that pixel's depth is the distance to the cone-inflated surface.
    w = h = 64
    ratio = host.default_ratio(w, h)
    for px, py in [(20, 20), (31, 32), (50, 12)]:
        nx, ny = ((2 * px + 1) / w - 1) * ratio[0], ((2 * py + 1) / h - 1) * ratio[1]
        d = np.array([nx, 1.0, ny])
        d /= np.linalg.norm(d)
        sc = one_sphere(center=tuple(d * 20.0), radius=0.05)
        depth = O.render_a(sc, w, h, want_rgb=False)["levels"][-1]
        hit = np.argwhere(depth < 1000.0)
        assert len(hit) >= 1 and [py, px] in hit.tolist()
        np.testing.assert_allclose(hit.mean(axis=0), [py, px], atol=0.75)
        assert np.abs(hit - [py, px]).max() <= 8
        thr = math.sqrt(2) * 8 / w
        assert 20.0 - 0.05 - 2 * 21 * thr <= depth[py, px] <= 20.0


# (vii) shadowRay: 0 when a sphere blocks the segment, 1 in empty space, penumbra in between
def test_shadow_ray():
    sc = one_sphere(center=(0, 10, 0), radius=2.0)
    assert O.shadow_ray(sc, (0, 0, 0), (0, 1, 0), 20.0) == 0.0
    assert O.shadow_ray(sc, (50, 0, 0), (0, 1, 0), 20.0) == 1.0
    pen = O.shadow_ray(sc, (2.5, 0, 0), (0, 1, 0), 20.0)
    assert 0.0 < pen < 1.0 and abs(pen - 0.5) < 0.05  # min SDF seen = 0.5
    assert O.shadow_ray(sc, (0, 0, 0), (0, 1, 0), 5.0) == 1.0  # segment ends before the sphere


def test_shading_known_answer():
    """One sphere, one light at the camera: centre pixel colour from fragment.glsl:162-185 by hand."""
    sc = one_sphere(center=(0, 10, 0), radius=2.0, lights=[((0, 0, 0), (1, 1, 1))])
    r = O.render_a(sc, 64, 64)
    depth, rgb = r["levels"][-1], r["rgb"]
    px = py = 32
    d = float(depth[py, px])
    thr = math.sqrt(2) * 8 / 64
    assert 8.0 - 2 * 9 * thr <= d <= 8.0  # stops where the cone (radius (len+1)*thr) touches the sphere
    # float64 evaluation of fragment.glsl:129-185 for this pixel (light at the camera, no occluder)
    ratio = host.default_ratio(64, 64)
    step = np.array([((px + 0.5) * 2 / 64 - 1) * ratio[0], 1.0, ((py + 0.5) * 2 / 64 - 1) * ratio[1]])
    step /= np.linalg.norm(step)
    position = step * d
    normal = (position - [0, 10, 0]) / np.linalg.norm(position - [0, 10, 0])
    cam_fall = max(0.01 * (d * d + 1), 1.0)
    normal_fall = max(np.dot(normal, -step), 0.0)
    light_dir, light_dist = -step, d
    light_fall = max(0.01 * light_dist * light_dist, 1.0)
    diffuse = max(np.dot(normal, light_dir), 0.0)
    refl = -light_dir - 2 * np.dot(normal, -light_dir) * normal
    spec = max(diffuse * np.dot(refl, -step) ** 1.0, 0.0)
    expect = (0.05 + max(diffuse + spec, 0) * 1.0 / light_fall * 1.0) / cam_fall * normal_fall * 1.0
    assert abs(rgb[py, px, 0] - expect) < 1e-4 and np.allclose(rgb[py, px], rgb[py, px, 0])
    # misses are black (fragment.glsl:137-140)
    assert np.all(rgb[depth >= 1000.0] == 0.0)


def test_spp_jitter_zero_is_reference_sample():
    sc = O.default_scene()
    a = O.render_a(sc, 64, 64)
    b = O.render_a(sc, 64, 64, jitter=(0.0, 0.0))
    assert np.array_equal(a["rgb"], b["rgb"])
    c = O.render_a(sc, 64, 64, jitter=(0.5 / 64, -0.5 / 64))
    assert not np.array_equal(a["rgb"], c["rgb"])


def test_threads_do_not_change_results():
    sc = O.default_scene()
    a = O.render_a(sc, 96, 64, threads=1)
    b = O.render_a(sc, 96, 64, threads=4)
    assert np.array_equal(a["rgb"], b["rgb"]) and a["counters"] == b["counters"]
    assert all(np.array_equal(x, y) for x, y in zip(a["levels"], b["levels"]))


def test_unorm8():
    rgb = np.array([[[-1.0, 0.0, 0.5], [1.0, 2.0, 0.999]]], np.float32)
    out = O.to_unorm8(rgb)
    assert out.tolist() == [[[0, 0, 128, 255], [255, 255, 255, 255]]]


FIXTURES_A = ["path_a_default_64.npz", "path_a_default_turn_96x64.npz", "path_a_cornell_256.npz",
              "path_a_alg1_96x64.npz", "path_a_alg2_96x64.npz", "path_a_repeat_96x64.npz", "path_a_mirror_96x64.npz",
              "path_a_transparency_96x64.npz", "path_a_refraction_96x64.npz"]


def fixture_config(g):
    cfg = O.default_config()
    cfg.march_algorithm = int(g["march_algorithm"])
    cfg.repeat[:] = [float(v) for v in g["repeat"]]
    cfg.max_steps = int(g["max_steps"])
    if "reflections" in g:  # fixtures older than the mirror variant do not carry the fields
        cfg.reflections, cfg.reflectivity = int(g["reflections"]), float(g["reflectivity"])
    if "transmissions" in g:
        cfg.transmissions, cfg.transparency, cfg.refraction_index = int(g["transmissions"]), float(g["transparency"]), float(g["refraction_index"])
    return cfg


@pytest.mark.parametrize("name", FIXTURES_A)
def test_oracle_matches_committed_fixture(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    sc = O.scene_from_bytes(g["scene"].tobytes())
    r = O.render_a(sc, int(g["width"]), int(g["height"]), rot=g["rot"], pos=g["pos"], cfg=fixture_config(g))
    for i, lv in enumerate(r["levels"]):
        assert np.array_equal(lv, g[f"level{i}"]), f"level {i}"
    # powf is the only libm call in the path: allow its last-ulp variation across hosts
    np.testing.assert_allclose(r["rgb"], g["rgb"], rtol=0, atol=1e-6)
    want = g["counters"].tolist()  # fixtures written before a counter existed hold a prefix of today's list; a counter they lack is 0
    have = list(r["counters"].values())
    assert have[:len(want)] == want and not any(have[len(want):])


# ---- SDF feature growth the author sketched -----------------------------------
@pytest.mark.parametrize("alg", [1, 2])
def test_sketched_march_algorithms_known_answer(alg):
    """shaders/tracing_algorithms.txt:2-13 / :16-37 by hand for one sphere straight ahead (centre 10 away,
    radius 2): both reach the surface at len = 8 with SDF 0 and stop with len = 8 - (8 + 1) * threshold.
    Algorithm 2 needs one more iteration: its first refresh is taken at the loop-top position."""
    f = np.float32
    sc = one_sphere((0.0, 10.0, 0.0), 2.0)
    cfg = O.default_config()
    cfg.march_algorithm = alg
    t = f(0.01)
    want = f(8.0) - (f(8.0) + f(1.0)) * t
    assert f(O.trace_cone(sc, (0, 0, 0), (0, 1, 0), float(t), cfg=cfg)) == want


def test_algorithm1_is_the_bruteforce_listing():
    sc = O.scene_from_bytes(bytes(host.default_scene()))
    cfg = O.default_config()
    cfg.march_algorithm = 1
    for d in [(0, 1, 0), (0.3, 0.9, -0.1), (-0.6, 0.7, 0.2)]:
        d = np.asarray(d, np.float32) / np.float32(np.linalg.norm(d))
        assert O.trace_cone(sc, (0, 0, 0), d, 0.02, cfg=cfg) == O.trace_cone(sc, (0, 0, 0), d, 0.02, brute=True)


def test_domain_repetition_known_answer():
    """utilities.glsl:31-34: a sphere of radius 1 at the origin repeated every 8 units along y; a ray from
    y = 2 along +y meets the copy at y = 8 (surface at 7): len = 5 - radius term.  Without repetition it
    leaves the scene."""
    f = np.float32
    sc = one_sphere((0.0, 0.0, 0.0), 1.0)
    cfg = O.default_config()
    assert O.trace_cone(sc, (0, 2, 0), (0, 1, 0), 0.01, cfg=cfg) >= cfg.render_dist
    cfg.repeat[1] = 8.0
    got = f(O.trace_cone(sc, (0, 2, 0), (0, 1, 0), 0.01, cfg=cfg))
    # algorithm 3: first step 1 (SDF at y=2 in the repeated domain), ... the exact chain is arithmetic in f32;
    # the hit distance is within one cone radius of the surface of the copy
    assert 5.0 - (5.0 + 1.0) * 0.011 <= got <= 5.0
    # periodicity: starting one period later gives the same answer
    assert f(O.trace_cone(sc, (0, 10, 0), (0, 1, 0), 0.01, cfg=cfg)) == got


def test_default_scene_matches_reference_listing():
    """src/main.rs:524-591."""
    s = O.default_scene()
    assert (s.matCount, s.objCount, s.lightCount) == (4, 4, 2)
    assert [tuple(o.pos) + (o.size,) for o in s.objs[:4]] == [(5, 5, -1, 3), (5, 4, 10, 6), (-3, 3, -3, 1), (4, -1, 0, 2)]
    assert [m.shine for m in s.mats[:4]] == [1, 10, 1, 1]
    assert bytes(s) == bytes(host.default_scene())


# ---- mirror reflections (fragment.glsl:125 "TODO: reflection"; build-defined, oracle.h) ------------------------------
def test_mirror_reflection_known_answers():
    """One mirror sphere facing a lit sphere: reflections add exactly weight x (what the reflected point shows when shaded
    from the mirror point); nothing changes where the mirror ray leaves the scene; reflectivity 0 or specular 0 is the
    reference image; the second bounce adds to the first."""
    import copy

    def scene(specular):
        sc = O.Scene()
        sc.matCount = sc.objCount = 2
        sc.lightCount = 1
        for i, (pos, size, col) in enumerate([((0, 10, 0), 4.0, (0.9, 0.9, 0.9)), ((6, 6, 0), 2.0, (1.0, 0.2, 0.2))]):
            sc.objs[i].pos[:] = pos
            sc.objs[i].size = size
            sc.mats[i].color[:] = col
            sc.mats[i].diffuse = 1.0
            sc.mats[i].specular = specular
            sc.mats[i].shine = 4.0
            sc.mats[i].ambient = 0.05
        sc.lights[0].pos[:] = (0, 0, 12)
        sc.lights[0].color[:] = (2, 2, 2)
        return sc

    w, h = 96, 96
    base = O.render_a(scene(1.0), w, h, want_levels=False)
    cfg = O.default_config()
    cfg.reflections = 1
    one = O.render_a(scene(1.0), w, h, cfg=cfg, want_levels=False)
    d = one["rgb"] - base["rgb"]
    assert one["counters"]["reflection_rays"] == base["counters"]["hit_pixels"] and base["counters"]["reflection_rays"] == 0
    changed = np.abs(d).max(-1) > 0
    assert d.min() >= 0 and changed.sum() > 100  # light is only ever added: the spheres show up in each other
    on_grey = changed & (np.abs(base["rgb"][..., 0] - base["rgb"][..., 1]) <= 1e-6 * (1 + base["rgb"][..., 0]))  # primary hit = the grey sphere
    assert on_grey.sum() > 50 and d[on_grey][:, 0].sum() > 4 * d[on_grey][:, 1].sum()  # what the grey mirror shows is the red sphere
    # pixel (48, 48) looks straight at the grey sphere's pole: the mirror ray returns along -Y, past the camera, to nothing
    assert np.array_equal(one["rgb"][48, 48], base["rgb"][48, 48])
    cfg.reflectivity = 0.0
    assert np.array_equal(O.render_a(scene(1.0), w, h, cfg=cfg, want_levels=False)["rgb"], base["rgb"])
    cfg.reflectivity = 0.5
    assert np.array_equal(O.render_a(scene(0.0), w, h, cfg=cfg, want_levels=False)["rgb"], O.render_a(scene(0.0), w, h, want_levels=False)["rgb"])
    # linear in the weight: reflectivity 0.25 adds half of what 0.5 adds (one bounce, exact powers of two)
    cfg.reflectivity = 0.25
    q = O.render_a(scene(1.0), w, h, cfg=cfg, want_levels=False)["rgb"] - base["rgb"]
    np.testing.assert_allclose(q, d * np.float32(0.5), rtol=0, atol=2e-7)
    cfg.reflectivity = 0.5
    cfg.reflections = 2
    two = O.render_a(scene(1.0), w, h, cfg=cfg, want_levels=False)
    assert (two["rgb"] >= one["rgb"]).all() and two["counters"]["reflection_rays"] > one["counters"]["reflection_rays"]
    assert np.isfinite(two["rgb"]).all()


# ---- transmission (fragment.glsl:124 "TODO: transparency", :126 "TODO: refraction"; build-defined, oracle.h) ---------
def test_transmission_known_answers():
    """A clear sphere in front of a lit red one: a transmitted ray per hit pixel, light is only ever added and what shows through
    the clear sphere is the red one; transparency 0 or mat.diffuse 0 is the reference image; the added light is linear in the
    weight; at normal incidence Snell's law does not bend the ray, so the pixel that looks through the pole is the same with
    index 1 and 1.5 while the others move; a second pass crosses the red sphere too and finds nothing behind it."""
    def scene(diffuse):
        sc = O.Scene()
        sc.matCount = sc.objCount = 2
        sc.lightCount = 1
        for i, (pos, size, col) in enumerate([((0, 10, 0), 3.0, (0.9, 0.9, 0.9)), ((0, 24, 0), 5.0, (1.0, 0.2, 0.2))]):
            sc.objs[i].pos[:] = pos
            sc.objs[i].size = size
            sc.mats[i].color[:] = col
            sc.mats[i].diffuse = diffuse
            sc.mats[i].specular = 1.0
            sc.mats[i].shine = 4.0
            sc.mats[i].ambient = 0.05
        sc.lights[0].pos[:] = (0, 14, 14)
        sc.lights[0].color[:] = (2, 2, 2)
        return sc

    w, h = 97, 97  # odd: pixel (48, 48) looks exactly along the view axis, through the clear sphere's pole
    base = O.render_a(scene(1.0), w, h, want_levels=False)
    cfg = O.default_config()
    assert cfg.transmissions == 0 and cfg.transparency == 0.5 and cfg.refraction_index == 1.0
    cfg.transmissions = 1
    one = O.render_a(scene(1.0), w, h, cfg=cfg, want_levels=False)
    d = one["rgb"] - base["rgb"]
    assert one["counters"]["transmission_rays"] == base["counters"]["hit_pixels"] and base["counters"]["transmission_rays"] == 0
    assert one["counters"]["reflection_rays"] == 0
    changed = np.abs(d).max(-1) > 0
    assert d.min() >= 0 and changed.sum() > 100
    on_clear = changed & (np.abs(base["rgb"][..., 0] - base["rgb"][..., 1]) <= 1e-6 * (1 + base["rgb"][..., 0]))  # primary hit = the grey, clear sphere
    assert on_clear.sum() > 50 and d[on_clear][:, 0].sum() > 4 * d[on_clear][:, 1].sum()  # what shows through it is the red sphere
    # one shadow ray per light for every shaded point: the primary hits and the points found behind the clear sphere
    assert one["counters"]["shadow_rays"] > base["counters"]["shadow_rays"]
    cfg.transparency = 0.0
    assert np.array_equal(O.render_a(scene(1.0), w, h, cfg=cfg, want_levels=False)["rgb"], base["rgb"])
    cfg.transparency = 0.5
    assert np.array_equal(O.render_a(scene(0.0), w, h, cfg=cfg, want_levels=False)["rgb"], O.render_a(scene(0.0), w, h, want_levels=False)["rgb"])
    cfg.transparency = 0.25  # linear in the weight (one pass, exact powers of two)
    q = O.render_a(scene(1.0), w, h, cfg=cfg, want_levels=False)["rgb"] - base["rgb"]
    np.testing.assert_allclose(q, d * np.float32(0.5), rtol=0, atol=2e-7)
    cfg.transparency = 0.5
    cfg.refraction_index = 1.5
    bent = O.render_a(scene(1.0), w, h, cfg=cfg, want_levels=False)
    # a sphere entered from outside never reflects totally; the rays that end are the silhouette pixels, where only the CONE touched
    # the sphere (disc <= 0: no crossing, the bent ray points inward at its "exit")
    assert 0.7 * one["counters"]["transmission_rays"] <= bent["counters"]["transmission_rays"] <= one["counters"]["transmission_rays"]
    # the axis pixel enters and leaves at normal incidence: Snell's law does not bend it (its neighbours already move: a ball lens)
    np.testing.assert_allclose(bent["rgb"][48, 48], one["rgb"][48, 48], rtol=0, atol=1e-6)
    assert one["rgb"][48, 48, 0] > base["rgb"][48, 48, 0]
    moved = np.abs(bent["rgb"] - one["rgb"]).max(-1) > 1e-3
    assert moved.sum() > 100 and np.isfinite(bent["rgb"]).all() and (bent["rgb"] >= base["rgb"]).all()
    cfg.refraction_index = 1.0
    cfg.transmissions = 2
    two = O.render_a(scene(1.0), w, h, cfg=cfg, want_levels=False)
    assert two["counters"]["transmission_rays"] > one["counters"]["transmission_rays"] and np.array_equal(two["rgb"], one["rgb"])  # nothing behind the red sphere
