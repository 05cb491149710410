"""CPU tests of oracle B (triangles + BVH + path tracing).  Path B has NO reference counterpart
(SURVEY.md §0, §8a last row): the oracle is the executable form of DESIGN.md §6 and is pinned by
the analytic known-answer tests below — "parity unpinned by the reference"."""
import math
import os

import numpy as np
import pytest

import oracle as O
from raytracing_engine_amd import scenes


def quad(p0, p1, p2, p3):
    return [np.concatenate([p0, p1, p2]).astype(np.float32), np.concatenate([p0, p2, p3]).astype(np.float32)]


def mesh(tris, albedo, emission):
    return O.TriScene(np.array(tris, np.float32), np.array(albedo, np.float32), np.array(emission, np.float32))


def test_rng_is_a_counter_hash():
    u = np.array([O.pt_rand(p, s, d, k, 1) for p in range(40) for s in range(4) for d in range(2) for k in range(7)])
    assert (u >= 0).all() and (u < 1).all() and abs(u.mean() - 0.5) < 0.03 and len(np.unique(u)) > 0.99 * len(u)
    assert O.pt_rand(5, 1, 0, 2, 9) == O.pt_rand(5, 1, 0, 2, 9) != O.pt_rand(5, 1, 0, 2, 10)
    # independent numpy restatement of the hash chain (spec §6.2)
    def h(x):
        x &= 0xFFFFFFFF; x ^= x >> 16; x = (x * 0x7FEB352D) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846CA68B) & 0xFFFFFFFF; x ^= x >> 16
        return x
    key = h((h((123 + h(77)) & 0xFFFFFFFF) + 3) & 0xFFFFFFFF)
    expect = (h((key + (2 * 8 + 5 + 1) * 0x9E3779B9) & 0xFFFFFFFF) >> 8) * 2.0 ** -24
    assert O.pt_rand(123, 3, 2, 5, 77) == np.float32(expect)


def test_sincos_polynomial():
    us = np.linspace(0, 1, 4001, endpoint=False, dtype=np.float32)
    sc = np.array([O.sincos_2pi(float(u)) for u in us])
    np.testing.assert_allclose(sc[:, 0], np.sin(2 * np.pi * us.astype(np.float64)), atol=4e-7)
    np.testing.assert_allclose(sc[:, 1], np.cos(2 * np.pi * us.astype(np.float64)), atol=4e-7)


def test_cosine_dir_distribution():
    rng = np.random.default_rng(0)
    for n in [(0, 0, 1), (0, 0, -1), (0.6, 0.0, 0.8), (-0.267, 0.534, -0.802)]:
        n = np.array(n) / np.linalg.norm(n)
        d = np.array([O.cosine_dir(n, float(a), float(b)) for a, b in rng.random((4000, 2), dtype=np.float32)])
        np.testing.assert_allclose(np.linalg.norm(d, axis=1), 1.0, atol=2e-6)
        cos = d @ n
        assert (cos >= -1e-6).all()
        assert abs(cos.mean() - 2 / 3) < 0.02          # E[cos] under a cosine-weighted pdf
        perp = d - np.outer(cos, n)
        assert np.abs(perp.mean(axis=0)).max() < 0.03  # azimuthally symmetric


def test_single_triangle_hits():
    sc = mesh([[-1, 5, -1, 1, 5, -1, 0, 5, 1]], [[0.5, 0.5, 0.5]], [[0, 0, 0]])
    for bvh in (True, False):
        tri, t = sc.closest_hit((0, 0, 0), (0, 1, 0), bvh)
        assert tri == 0 and t == 5.0
        assert sc.closest_hit((0, 0, 0), (0, -1, 0), bvh)[0] == -1      # behind
        assert sc.closest_hit((3, 0, 0), (0, 1, 0), bvh)[0] == -1       # beside
        assert sc.closest_hit((0, 10, 0), (0, -1, 0), bvh) == (0, 5.0)  # two-sided
        assert sc.closest_hit((0, 0, 0), (0, 2, 0), bvh) == (0, 2.5)    # t is in units of |dir|
        assert sc.occluded((0, 0, 0), (0, 10, 0), bvh) and not sc.occluded((0, 0, 0), (0, 4, 0), bvh)
        assert not sc.occluded((0, 0, 0), (0, 5, 0), bvh)               # end point itself is excluded (t < 0.999)


def test_closest_is_lexicographic_min_on_coplanar_duplicates():
    t0 = [-1, 5, -1, 1, 5, -1, 0, 5, 1]
    sc = mesh([t0, t0, [-1, 7, -1, 1, 7, -1, 0, 7, 1]], [[0.5] * 3] * 3, [[0] * 3] * 3)
    assert sc.closest_hit((0, 0, 0), (0, 1, 0), True) == (0, 5.0) == sc.closest_hit((0, 0, 0), (0, 1, 0), False)


def test_bvh_equals_bruteforce_on_random_rays():
    v, a, e = scenes.soup_scene(3000, seed=2, edge=2.0)
    sc = O.TriScene(v, a, e)
    rng = np.random.default_rng(5)
    hits = 0
    for _ in range(1500):
        o = rng.uniform([-12, 0, -12], [12, 30, 12])
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        a_ = sc.closest_hit(o, d, True)
        assert a_ == sc.closest_hit(o, d, False)
        hits += a_[0] >= 0
        seg = d * rng.uniform(1, 20)
        assert sc.occluded(o, seg, True) == sc.occluded(o, seg, False)
    assert hits > 300


def test_render_bvh_equals_bruteforce_and_thread_count():
    v, a, e = scenes.cornell_tri_scene()
    sc = O.TriScene(v, a, e)
    kw = dict(spp=3, bounces=2, seed=11, pos=(0, 1, 0))
    a1, c1 = sc.render(48, 48, use_bvh=True, threads=1, **kw)
    a2, c2 = sc.render(48, 48, use_bvh=False, threads=4, **kw)
    assert np.array_equal(a1, a2)
    assert [c1[k] for k in ("camera_rays", "bounce_rays", "shadow_rays")] == [c2[k] for k in ("camera_rays", "bounce_rays", "shadow_rays")]
    assert c1["camera_rays"] == 48 * 48 * 3


def test_diffuse_plane_under_uniform_sky_is_albedo_times_sky():
    """Furnace-style KAT: camera looks at a huge diffuse plane, no lights, 1 bounce: every bounce
    ray escapes to the sky, and cosine sampling cancels cos/pi exactly -> pixel = albedo * sky."""
    f = np.float32
    big = quad(np.array([-1e4, 20, -1e4], f), np.array([1e4, 20, -1e4], f), np.array([1e4, 20, 1e4], f), np.array([-1e4, 20, 1e4], f))
    sc = mesh(big, [[0.25, 0.5, 0.75]] * 2, [[0, 0, 0]] * 2)
    rgb, ct = sc.render(16, 16, spp=4, bounces=1, sky=(2.0, 1.0, 0.5))
    np.testing.assert_allclose(rgb, np.broadcast_to(np.array([0.5, 0.5, 0.375], f), rgb.shape), rtol=1e-6)
    assert ct["shadow_rays"] == 0 and ct["bounce_rays"] == 16 * 16 * 4


def test_direct_lighting_matches_analytic_irradiance():
    """NEE KAT: small square light of radiance Le straight above a diffuse floor point:
    L = albedo/pi * Le * A * cos*cos / d^2 (small-source limit)."""
    f = np.float32
    floor = quad(np.array([-50, 0, -2], f), np.array([50, 0, -2], f), np.array([50, 100, -2], f), np.array([-50, 100, -2], f))
    s = 0.25
    light = quad(np.array([-s, 10 - s, 3], f), np.array([s, 10 - s, 3], f), np.array([s, 10 + s, 3], f), np.array([-s, 10 + s, 3], f))
    sc = mesh(floor + light, [[0.8] * 3] * 2 + [[0] * 3] * 2, [[0] * 3] * 2 + [[40, 40, 40]] * 2)
    # camera above the floor looking down at (0,10,-2): pitch -pi/2 turns +Y forward into -Z
    rot = O.camera_quat(0.0, -math.pi / 2)
    rgb, _ = sc.render(9, 9, spp=256, bounces=0, rot=rot, pos=(0, 10, 2.5), ratio=(0.002, 0.002))
    expect = 0.8 / math.pi * 40.0 * (2 * s) ** 2 / 5.0 ** 2
    assert abs(rgb[4, 4, 0] - expect) / expect < 0.02 and np.allclose(rgb[4, 4], rgb[4, 4, 0])
    # an occluder between floor and light kills it
    blocker = quad(np.array([-1, 9, 1], f), np.array([1, 9, 1], f), np.array([1, 11, 1], f), np.array([-1, 11, 1], f))
    sc2 = mesh(floor + light + blocker, [[0.8] * 3] * 2 + [[0] * 3] * 2 + [[0.5] * 3] * 2, [[0] * 3] * 2 + [[40, 40, 40]] * 2 + [[0] * 3] * 2)
    rgb2, _ = sc2.render(9, 9, spp=16, bounces=0, rot=rot, pos=(0, 10, 0.5), ratio=(0.002, 0.002))
    assert rgb2[4, 4].max() == 0.0


def test_camera_sees_light_directly_only_at_depth_zero():
    f = np.float32
    light = quad(np.array([-1, 5, -1], f), np.array([1, 5, -1], f), np.array([1, 5, 1], f), np.array([-1, 5, 1], f))
    sc = mesh(light, [[0] * 3] * 2, [[3, 2, 1]] * 2)
    rgb, ct = sc.render(8, 8, spp=2, bounces=3, ratio=(0.01, 0.01))
    np.testing.assert_array_equal(rgb, np.broadcast_to(np.array([3, 2, 1], f), rgb.shape))
    assert ct["bounce_rays"] == 0 and ct["shadow_rays"] == 0


@pytest.mark.parametrize("name,args", [("path_b_cornell_64.npz", dict(kind="cornell", w=64, h=64, spp=4, bounces=2, seed=7, pos=(0, 1, 0))),
                                       ("path_b_soup2k_96x54.npz", dict(kind="soup", w=96, h=54, spp=2, bounces=1, seed=5, sky=(0.3, 0.3, 0.4)))])
def test_oracle_matches_committed_fixture(golden_dir, name, args):
    g = np.load(os.path.join(golden_dir, name))
    kind, w, h = args.pop("kind"), args.pop("w"), args.pop("h")
    v, a, e = scenes.cornell_tri_scene() if kind == "cornell" else scenes.soup_scene(2000, seed=3, edge=1.5)
    rgb, ct = O.TriScene(v, a, e).render(w, h, **args)
    assert np.array_equal(rgb, g["rgb"])  # no libm call anywhere in path B: bit-exact on every host
    assert [ct["camera_rays"], ct["bounce_rays"], ct["shadow_rays"]] == g["counters"].tolist()


def test_soup_scene_is_reproducible():
    v, a, e = scenes.soup_scene(1000, seed=1)
    assert v.shape == (1000, 9) and (e[-2:] > 0).all() and not e[:-2].any()
    assert abs(float(v[:998, 0].mean())) < 0.5 and 14 < float(v[:998, 1].mean()) < 16
    # first triangle of seed 1 is pinned (counter hash, independent of numpy's generators)
    v2, _, _ = scenes.soup_scene(1000, seed=1)
    assert np.array_equal(v, v2) and not np.array_equal(v, scenes.soup_scene(1000, seed=2)[0])
