"""GPU parity tests for path A (reference-faithful cone marcher + shading), through the C ABI.

Bar: pyramid levels (depth) bit-exact vs oracle A; RGB within 1e-4 max-abs (north_star tolerance;
powf in the specular term is the only operation that is not bit-defined).  The oracle is pinned
by tests/test_oracle_a.py, not by the reference ("parity unpinned by the reference").
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
import raytracing_engine_amd as R
from raytracing_engine_amd import host

pytestmark = pytest.mark.gpu
RGB_TOL = 1e-4


def oracle_scene(scene):
    return O.scene_from_bytes(bytes(scene))


def on_screen_mask(shape, level, count, w, h):
    """Texels of pyramid level `level` that have a descendant inside the w x h frame.  The fused pyramid
    kernel (rt_config.fuse_levels = 1) computes exactly these; the others are never read
    by anything (the reference computes them only because its images are padded to multiples of 8)."""
    s = count - 1 - level
    gy, gx = np.mgrid[0:shape[0], 0:shape[1]]
    return ((gx << s) < w) & ((gy << s) < h)


def set_fused(r, fused):
    cfg = r.default_config()
    cfg.fuse_levels = int(fused)
    r.set_config(cfg)


def check_frame(r, scene, w, h, rot=(0, 0, 0, 1), pos=(0, 0, 0), levels=True):
    """Both launch schedules against oracle A: one launch per level + shade (the reference's own
    schedule; every texel compared) and the fused one-launch pyramid (texels with on-screen descendants)."""
    r.set_scene(scene)
    r.resize(w, h)
    ref = O.render_a(oracle_scene(scene), w, h, rot=rot, pos=pos)
    assert r.level_info() == [lv.shape[::-1] for lv in ref["levels"]]
    count = len(ref["levels"])
    try:
        for fused in (False, True):
            set_fused(r, fused)
            rgb, depth = r.render(rot, pos, want_depth=True)
            if levels:
                for i, lv in enumerate(ref["levels"]):
                    got = r.read_level(i)
                    m = on_screen_mask(lv.shape, i, count, w, h) if fused else np.ones(lv.shape, bool)
                    assert np.array_equal(got[m], lv[m]), f"fused={fused} level {i}: {np.count_nonzero(got[m] != lv[m])} texels differ"
            assert np.array_equal(depth[:h, :w], ref["levels"][-1][:h, :w])
            err = np.abs(rgb - ref["rgb"]).max()
            assert err <= RGB_TOL, (fused, err)
            st = r.stats()
            assert st["hit_pixels"] == ref["counters"]["hit_pixels"]
            assert st["shadow_rays"] == ref["counters"]["shadow_rays"]
            assert st["primary_rays"] == w * h
    finally:
        set_fused(r, False)
    return rgb, ref


@pytest.mark.parametrize("w,h", [(64, 64), (256, 256), (96, 64), (200, 120), (1000, 700)])
def test_default_scene_parity(renderer, w, h):
    check_frame(renderer, R.default_scene(), w, h)


def test_cornell_scene_parity(renderer):
    check_frame(renderer, R.cornell_scene(), 256, 256)


@pytest.mark.parametrize("yaw,pitch,pos", [(0.6, -0.2, (1.0, -2.0, 0.5)), (-2.5, 1.2, (3, 3, 3)), (3.1, -1.57, (-10, 0, 4))])
def test_camera_poses(renderer, yaw, pitch, pos):
    check_frame(renderer, R.default_scene(), 160, 96, rot=R.camera_quat(yaw, pitch), pos=pos)


@pytest.mark.parametrize("name", ["path_a_default_64.npz", "path_a_default_turn_96x64.npz", "path_a_cornell_256.npz",
                                  "path_a_alg1_96x64.npz", "path_a_alg2_96x64.npz", "path_a_repeat_96x64.npz", "path_a_mirror_96x64.npz",
                                  "path_a_transparency_96x64.npz", "path_a_refraction_96x64.npz"])
def test_against_committed_fixture(renderer, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    renderer.set_scene(g["scene"].tobytes())
    w, h = int(g["width"]), int(g["height"])
    renderer.resize(w, h)
    count = len(renderer.level_info())
    variant = int(g["march_algorithm"]) != 0 or bool(np.any(g["repeat"] > 0))
    try:
        for fused in ((False,) if variant else (False, True)):  # the sketched variants exist in the per-level schedule only
            cfg = renderer.default_config()
            cfg.fuse_levels = int(fused)
            cfg.march_algorithm = int(g["march_algorithm"])
            cfg.repeat[:] = [float(v) for v in g["repeat"]]
            cfg.max_steps = int(g["max_steps"])
            cfg.reflections, cfg.reflectivity = int(g["reflections"]), float(g["reflectivity"])
            if "transmissions" in g:
                cfg.transmissions, cfg.transparency, cfg.refraction_index = int(g["transmissions"]), float(g["transparency"]), float(g["refraction_index"])
            renderer.set_config(cfg)
            rgb = renderer.render(g["rot"], g["pos"])
            st = renderer.stats()
            assert [st["hit_pixels"], st["shadow_rays"], st["reflection_rays"]] == [int(g["counters"][3]), int(g["counters"][4]), int(g["counters"][7])]
            assert st["transmission_rays"] == (int(g["counters"][8]) if len(g["counters"]) > 8 else 0)
            for i in range(count):
                got, want = renderer.read_level(i), g[f"level{i}"]
                m = on_screen_mask(want.shape, i, count, w, h) if fused else np.ones(want.shape, bool)
                assert np.array_equal(got[m], want[m])
            assert np.abs(rgb - g["rgb"]).max() <= RGB_TOL
    finally:
        set_fused(renderer, False)


@pytest.mark.parametrize("n_obj,n_light", [(1, 0), (1, 1), (2, 3), (3, 1), (5, 2), (6, 1), (7, 8), (8, 8)])
def test_object_and_light_counts(renderer, n_obj, n_light):
    rng = np.random.default_rng(n_obj * 16 + n_light)
    spheres = [(rng.uniform(-8, 8), rng.uniform(6, 25), rng.uniform(-6, 6), rng.uniform(0.5, 3)) for _ in range(n_obj)]
    mats = [(*rng.uniform(0.1, 1, 3), float(rng.choice([1, 2, 10, 30])), 0.05) for _ in range(n_obj)]
    lights = [(tuple(rng.uniform(-10, 10, 3)), tuple(rng.uniform(0.1, 1.5, 3))) for _ in range(n_light)]
    check_frame(renderer, host.make_scene(spheres, mats, lights), 128, 72)


def test_camera_inside_sphere_and_all_miss(renderer):
    # inside a sphere: SDF negative at the origin -> len clamps to 0 (compute.glsl:86 max(len,0))
    inside = host.make_scene([(0, 0, 0, 5)], [(1, 1, 1, 1, 0.05)], [((0, 1, 0), (1, 1, 1))])
    check_frame(renderer, inside, 64, 64)
    # nothing in view: every pixel misses, image is black
    miss = host.make_scene([(0, -50, 0, 1)], [(1, 1, 1, 1, 0.05)], [((0, 1, 0), (1, 1, 1))])
    rgb, _ = check_frame(renderer, miss, 64, 64)
    assert not rgb.any()


def test_config_is_honoured(renderer):
    try:
        scene = R.default_scene()
        renderer.set_scene(scene)
        renderer.resize(96, 64)
        ocfg = O.default_config()
        ocfg.render_dist, ocfg.cam_fall_off, ocfg.light_fall_off, ocfg.ray_radius = 40.0, 0.02, 0.005, 0.02
        ref = O.render_a(oracle_scene(scene), 96, 64, cfg=ocfg)
        for fused in (0, 1):
            cfg = renderer.default_config()
            cfg.render_dist, cfg.cam_fall_off, cfg.light_fall_off, cfg.ray_radius, cfg.fuse_levels = 40.0, 0.02, 0.005, 0.02, fused
            renderer.set_config(cfg)
            rgb, depth = renderer.render(want_depth=True)
            assert np.array_equal(depth, ref["levels"][-1])
            assert np.abs(rgb - ref["rgb"]).max() <= RGB_TOL
    finally:
        renderer.set_config(renderer.default_config())


@pytest.mark.parametrize("alg", [1, 2, 3])
@pytest.mark.parametrize("repeat", [(0.0, 0.0, 0.0), (40.0, 0.0, 40.0), (9.0, 50.0, 0.0)])
@pytest.mark.parametrize("n_obj", [4, 8])
def test_sketched_variants(renderer, alg, repeat, n_obj):
    """march algorithms 1 / 2 (shaders/tracing_algorithms.txt) and repeat()
    (utilities.glsl:31-34) against the oracle's restatement: depth pyramid bit-exact, RGB <= 1e-4."""
    scene = R.default_scene() if n_obj == 4 else R.cornell_scene()
    w, h = 200, 120
    rot, pos = host.camera_quat(0.3, -0.1), (0.5, -1.0, 0.2)
    ocfg = O.default_config()
    ocfg.march_algorithm, ocfg.max_steps = alg, 4096
    ocfg.repeat[:] = repeat
    ref = O.render_a(oracle_scene(scene), w, h, rot=rot, pos=pos, cfg=ocfg)
    try:
        cfg = renderer.default_config()
        cfg.march_algorithm, cfg.max_steps = alg, 4096
        cfg.repeat[:] = repeat
        renderer.set_config(cfg)
        renderer.set_scene(scene)
        renderer.resize(w, h)
        rgb = renderer.render(rot, pos)
        for i, lv in enumerate(ref["levels"]):
            got = renderer.read_level(i)
            assert np.array_equal(got, lv), f"level {i}: {np.count_nonzero(got != lv)} texels differ"
        assert np.abs(rgb - ref["rgb"]).max() <= RGB_TOL
        assert renderer.stats()["hit_pixels"] == ref["counters"]["hit_pixels"]
        # 4 spp through the batched launches == the four jittered oracle frames in index order
        rgb4 = renderer.render(rot, pos, spp=4)
        acc = None
        for s in range(4):
            i, j = s % 2, s // 2
            jit = (((np.float32(2 * i + 1) / np.float32(2)) - np.float32(1)) / np.float32(w),
                   ((np.float32(2 * j + 1) / np.float32(2)) - np.float32(1)) / np.float32(h))
            f = O.render_a(oracle_scene(scene), w, h, rot=rot, pos=pos, jitter=jit, cfg=ocfg, want_levels=False)["rgb"]
            acc = f if acc is None else acc + f
        assert np.abs(rgb4 - acc / np.float32(4)).max() <= RGB_TOL
    finally:
        renderer.set_config(renderer.default_config())


@pytest.mark.parametrize("reflections,reflectivity", [(1, 0.5), (2, 0.8), (4, 1.0)])
@pytest.mark.parametrize("n_obj", [4, 8])
@pytest.mark.parametrize("repeat", [(0.0, 0.0, 0.0), (40.0, 0.0, 40.0)])
def test_mirror_reflections(renderer, reflections, reflectivity, n_obj, repeat):
    """SURVEY.md section 8 f.4, the remainder: fragment.glsl:125 is "TODO: reflection" in the reference; the build-defined
    mirror bounce (rt_config.reflections; specification in include/rt_abi.h and oracle/oracle.h) against the oracle's
    restatement: RGB <= 1e-4, hit / shadow-ray / mirror-ray counts equal, 1 and 4 spp, with and without repeat()."""
    scene = R.default_scene() if n_obj == 4 else R.cornell_scene()
    w, h = 200, 120
    rot, pos = host.camera_quat(0.3, -0.1), (0.5, -1.0, 0.2)
    ocfg = O.default_config()
    ocfg.reflections, ocfg.reflectivity, ocfg.max_steps = reflections, reflectivity, 4096
    ocfg.repeat[:] = repeat
    ref = O.render_a(oracle_scene(scene), w, h, rot=rot, pos=pos, cfg=ocfg)
    plain = O.render_a(oracle_scene(scene), w, h, rot=rot, pos=pos, want_levels=False)
    assert ref["counters"]["reflection_rays"] >= ref["counters"]["hit_pixels"] and (repeat[0] > 0 or np.abs(ref["rgb"] - plain["rgb"]).max() > 1e-3)
    try:
        cfg = renderer.default_config()
        cfg.reflections, cfg.reflectivity, cfg.max_steps = reflections, reflectivity, 4096
        cfg.repeat[:] = repeat
        renderer.set_config(cfg)
        renderer.set_scene(scene)
        renderer.resize(w, h)
        rgb, depth = renderer.render(rot, pos, want_depth=True)
        assert np.array_equal(depth, ref["levels"][-1])
        assert np.abs(rgb - ref["rgb"]).max() <= RGB_TOL
        st = renderer.stats()
        assert (st["hit_pixels"], st["shadow_rays"], st["reflection_rays"]) == (ref["counters"]["hit_pixels"], ref["counters"]["shadow_rays"], ref["counters"]["reflection_rays"])
        rgb4 = renderer.render(rot, pos, spp=4)
        acc = None
        for s in range(4):
            i, j = s % 2, s // 2
            jit = (((np.float32(2 * i + 1) / np.float32(2)) - np.float32(1)) / np.float32(w),
                   ((np.float32(2 * j + 1) / np.float32(2)) - np.float32(1)) / np.float32(h))
            f = O.render_a(oracle_scene(scene), w, h, rot=rot, pos=pos, jitter=jit, cfg=ocfg, want_levels=False)["rgb"]
            acc = f if acc is None else acc + f
        assert np.abs(rgb4 - acc / np.float32(4)).max() <= RGB_TOL
    finally:
        renderer.set_config(renderer.default_config())


@pytest.mark.parametrize("transmissions,transparency,index,reflections", [(1, 0.5, 1.0, 0), (2, 0.8, 1.0, 0), (1, 0.6, 1.5, 0), (3, 1.0, 1.33, 0), (2, 0.7, 1.5, 2)])
@pytest.mark.parametrize("n_obj", [4, 8])
@pytest.mark.parametrize("repeat", [(0.0, 0.0, 0.0), (40.0, 0.0, 40.0)])
def test_transmission(renderer, transmissions, transparency, index, reflections, n_obj, repeat):
    """SURVEY.md section 8 f.4, the remainder: fragment.glsl:124 "TODO: transparency" and :126 "TODO: refraction" in the reference;
    the build-defined transmission chain (rt_config.transmissions / transparency / refraction_index; specification in
    include/rt_abi.h and oracle/oracle.h) against the oracle's restatement: RGB <= 1e-4, hit / shadow-ray / mirror-ray /
    transmitted-ray counts equal, 1 and 4 spp, with and without repeat(), alone and together with the mirror chain."""
    scene = R.default_scene() if n_obj == 4 else R.cornell_scene()
    w, h = 200, 120
    rot, pos = host.camera_quat(0.3, -0.1), (0.5, -1.0, 0.2)
    ocfg = O.default_config()
    ocfg.transmissions, ocfg.transparency, ocfg.refraction_index, ocfg.max_steps = transmissions, transparency, index, 4096
    ocfg.reflections, ocfg.reflectivity = reflections, 0.6
    ocfg.repeat[:] = repeat
    ref = O.render_a(oracle_scene(scene), w, h, rot=rot, pos=pos, cfg=ocfg)
    plain = O.render_a(oracle_scene(scene), w, h, rot=rot, pos=pos, want_levels=False)
    assert ref["counters"]["transmission_rays"] >= 0.5 * ref["counters"]["hit_pixels"]
    # (the start-up scene's four spheres are ball lenses far apart: from this camera the bent rays find nothing behind them and the
    # frame is the plain one; inside the eight-sphere room they always land on a wall)
    assert repeat[0] > 0 or (index != 1.0 and n_obj == 4) or np.abs(ref["rgb"] - plain["rgb"]).max() > 1e-3
    try:
        cfg = renderer.default_config()
        cfg.transmissions, cfg.transparency, cfg.refraction_index, cfg.max_steps = transmissions, transparency, index, 4096
        cfg.reflections, cfg.reflectivity = reflections, 0.6
        cfg.repeat[:] = repeat
        renderer.set_config(cfg)
        renderer.set_scene(scene)
        renderer.resize(w, h)
        rgb, depth = renderer.render(rot, pos, want_depth=True)
        assert np.array_equal(depth, ref["levels"][-1])
        assert np.abs(rgb - ref["rgb"]).max() <= RGB_TOL
        st = renderer.stats()
        assert (st["hit_pixels"], st["shadow_rays"], st["reflection_rays"], st["transmission_rays"]) == \
            (ref["counters"]["hit_pixels"], ref["counters"]["shadow_rays"], ref["counters"]["reflection_rays"], ref["counters"]["transmission_rays"])
        rgb4 = renderer.render(rot, pos, spp=4)
        acc = None
        for s in range(4):
            i, j = s % 2, s // 2
            jit = (((np.float32(2 * i + 1) / np.float32(2)) - np.float32(1)) / np.float32(w),
                   ((np.float32(2 * j + 1) / np.float32(2)) - np.float32(1)) / np.float32(h))
            f = O.render_a(oracle_scene(scene), w, h, rot=rot, pos=pos, jitter=jit, cfg=ocfg, want_levels=False)["rgb"]
            acc = f if acc is None else acc + f
        assert np.abs(rgb4 - acc / np.float32(4)).max() <= RGB_TOL
    finally:
        renderer.set_config(renderer.default_config())


def test_sketched_variants_error_behaviour(renderer):
    cfg = renderer.default_config()
    cfg.march_algorithm = 4
    with pytest.raises(R.RtError):
        renderer.set_config(cfg)
    for bad in (dict(reflections=9), dict(reflectivity=1.5), dict(reflectivity=-0.1), dict(transmissions=9), dict(transparency=1.5), dict(transparency=-0.1),
                dict(refraction_index=0.9), dict(refraction_index=float("nan")), dict(refraction_index=5.0)):
        cfg = renderer.default_config()
        for k, v in bad.items():
            setattr(cfg, k, v)
        with pytest.raises(R.RtError):
            renderer.set_config(cfg)
    cfg = renderer.default_config()
    cfg.march_algorithm, cfg.fuse_levels = 1, 1
    with pytest.raises(R.RtError):
        renderer.set_config(cfg)
    cfg = renderer.default_config()
    cfg.repeat[0] = -1.0
    with pytest.raises(R.RtError):
        renderer.set_config(cfg)
    cfg = renderer.default_config()
    cfg.repeat[2], cfg.fuse_levels = 8.0, 1
    with pytest.raises(R.RtError):
        renderer.set_config(cfg)
    renderer.set_config(renderer.default_config())


def test_spp4_stratified(renderer):
    scene = R.default_scene()
    w, h = 128, 72
    renderer.set_scene(scene)
    renderer.resize(w, h)
    rgb = renderer.render(spp=4)
    acc = None
    n = 2
    for s in range(4):
        i, j = s % n, s // n
        jit = (((np.float32(2 * i + 1) / np.float32(n)) - np.float32(1)) / np.float32(w),
               ((np.float32(2 * j + 1) / np.float32(n)) - np.float32(1)) / np.float32(h))
        f = O.render_a(oracle_scene(scene), w, h, jitter=jit, want_levels=False)["rgb"]
        acc = f if acc is None else acc + f
    ref = acc / np.float32(4)
    assert np.abs(rgb - ref).max() <= RGB_TOL
    assert renderer.stats()["primary_rays"] == w * h * 4


def test_config1_spheres8_1080p_4spp(renderer):
    """BASELINE.json configs[1] AS NAMED: the 8-sphere room, 1920x1080, 4 spp (= 2 x 2 stratified sub-pixel centres, each one
    full frame, averaged in sample order: DESIGN.md section 5) against four oracle frames; ray counts equal."""
    scene = R.cornell_scene()
    w, h, n = 1920, 1080, 2
    renderer.set_scene(scene)
    renderer.resize(w, h)
    rgb = renderer.render(spp=4)
    st = renderer.stats()
    acc, shadow = None, 0
    for s in range(4):
        i, j = s % n, s // n
        jit = (((np.float32(2 * i + 1) / np.float32(n)) - np.float32(1)) / np.float32(w),
               ((np.float32(2 * j + 1) / np.float32(n)) - np.float32(1)) / np.float32(h))
        o = O.render_a(oracle_scene(scene), w, h, jitter=jit, want_levels=False)
        acc = o["rgb"] if acc is None else acc + o["rgb"]
        shadow += o["counters"]["shadow_rays"]
    err = np.abs(rgb - acc / np.float32(4)).max()
    assert err <= RGB_TOL, err
    assert st["primary_rays"] == w * h * 4 and st["shadow_rays"] == shadow
    renderer.resize(64, 64)


def test_deterministic_and_rgba8(renderer):
    scene = R.cornell_scene()
    renderer.set_scene(scene)
    renderer.resize(200, 120)
    a = renderer.render()
    rgba = renderer.read_rgba8()
    b = renderer.render()
    assert np.array_equal(a, b)
    assert np.array_equal(rgba, O.to_unorm8(a))
    # rt_read_rgba8 sits in a per-frame loop: its staging buffer is kept across calls and follows rt_resize
    for pos in [(0, 1, 0), (1, 0, 0.5)]:
        c = renderer.render(pos=pos)
        assert np.array_equal(renderer.read_rgba8(), O.to_unorm8(c))
    renderer.resize(96, 64)
    c = renderer.render()
    assert np.array_equal(renderer.read_rgba8(), O.to_unorm8(c)) and renderer.read_rgba8().shape == (64, 96, 4)


@pytest.mark.parametrize("n_ranks", [2, 3, 8])
def test_partition_union_equals_single(renderer, n_ranks):
    """Tile-split rendering: every rank renders only its 64x64 tiles; the union is bit-identical to
    the single-GPU frame.  Ranks are emulated by re-partitioning one context."""
    import torch

    scene = R.default_scene()
    w, h = 300, 200
    renderer.set_scene(scene)
    renderer.resize(w, h)
    renderer.set_partition(0, 1)
    full = renderer.render()
    tx, ty, _ = renderer.tile_info()
    tiles_per_rank = -(-(tx * ty) // n_ranks)
    gathered = torch.zeros((n_ranks, tiles_per_rank, 64, 64, 3), dtype=torch.float32, device="cuda")
    try:
        for rank in range(n_ranks):
            renderer.set_partition(rank, n_ranks)
            assert renderer.tile_info()[2] == len(range(rank, tx * ty, n_ranks))
            renderer.render_device((0, 0, 0, 1), (0, 0, 0), 1, gathered[rank].data_ptr(), tile_major=True)
            renderer.synchronize()
            part = renderer.render()  # full-frame layout, foreign tiles zero
            mask = np.zeros((h, w), bool)
            for t in range(rank, tx * ty, n_ranks):
                mask[(t // tx) * 64:(t // tx + 1) * 64, (t % tx) * 64:(t % tx + 1) * 64] = True
            assert np.array_equal(part[mask], full[mask]) and not part[~mask].any()
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        renderer.detile_device(gathered.data_ptr(), n_ranks, tiles_per_rank, out.data_ptr())
        renderer.synchronize()
        assert np.array_equal(out.cpu().numpy(), full)
        assert np.array_equal(host.tiles_to_frame(gathered.cpu().numpy().reshape(-1, 64, 64, 3), n_ranks, tiles_per_rank, w, h), full)
    finally:
        renderer.set_partition(0, 1)


def test_error_behaviour(renderer):
    lib = R.load()
    with pytest.raises(R.RtError) as e:
        renderer.set_scene(b"\0" * 100)
    assert e.value.code == -1
    bad = R.default_scene()
    bad.objCount = 9
    with pytest.raises(R.RtError):
        renderer.set_scene(bad)
    with pytest.raises(R.RtError):
        renderer.resize(0, 10)
    with pytest.raises(R.RtError):
        renderer.set_partition(3, 2)
    renderer.set_scene(R.default_scene())
    renderer.resize(64, 64)
    with pytest.raises(R.RtError):
        renderer.render(spp=3)
    fresh = R.Renderer(0)
    with pytest.raises(R.RtError) as e:
        fresh.render()
    assert e.value.code == -4
    fresh.close()
    ctx = C.c_void_p()
    assert lib.rt_create(C.byref(ctx), 9999) == -2 and b"ordinal" in lib.rt_last_error(None)


def test_full_hd_parity_and_properties(renderer):
    """BASELINE.json configs[1] size (1920x1080): full oracle comparison (the oracle needs < 1 s
    on a few cores) plus size-independent properties."""
    scene = R.cornell_scene()
    rgb, ref = check_frame(renderer, scene, 1920, 1080, levels=False)
    depth = renderer.read_level(7)
    assert depth.shape == (1080, 1920)
    # misses are exactly black, hits are finite and non-negative
    assert not rgb[depth >= 1000.0].any() and np.isfinite(rgb).all() and (rgb >= 0).all()
    # a child starts at its parent's depth and can only back off by its own cone radius at len 0
    # (compute.glsl:50,63: len -= (0 + 1) * threshold), so child >= parent - threshold_child
    parent = renderer.read_level(6)
    up = np.repeat(np.repeat(parent, 2, 0), 2, 1)[:1080, :1920]
    thr = np.float32(1.4142135 * 8.0) / np.float32(1920)
    assert (depth >= up - thr * 1.001).all()


@pytest.mark.parametrize("w,h", [(8, 8), (12, 10), (17, 9), (2048, 64)])
def test_tiny_and_wide_views(renderer, w, h):
    """Width < 16 gives a single pyramid level (src/main.rs:639); 2048 is the widest view the
    reference's 9-image array supports (shaders/compute.glsl:14-15)."""
    check_frame(renderer, R.default_scene(), w, h)


def test_4k_nine_levels(renderer):
    """BASELINE.json configs[4] resolution on path A: 3840x2160 = 9 levels, the cap of
    COMPUTE_IMAGE_COUNT (src/main.rs:359); full-frame comparison against the oracle."""
    rgb, ref = check_frame(renderer, R.cornell_scene(), 3840, 2160, levels=False)
    assert len(renderer.level_info()) == 9 and renderer.level_info()[-1] == (3840, 2160)


def test_resize_and_rescene_cycles(renderer):
    """Re-allocation paths: many resizes and scene swaps must keep giving oracle-identical frames."""
    for w, h, scene in [(64, 64, R.default_scene()), (640, 360, R.cornell_scene()), (96, 64, R.default_scene()), (640, 360, R.cornell_scene())]:
        renderer.set_scene(scene)
        renderer.resize(w, h)
        rgb, depth = renderer.render(want_depth=True)
        ref = O.render_a(oracle_scene(scene), w, h)
        assert np.array_equal(depth, ref["levels"][-1]) and np.abs(rgb - ref["rgb"]).max() <= RGB_TOL


def test_shortened_sqrt_is_correctly_rounded_for_every_input(renderer):
    """rt_device_math.h sqrt_cr (the 9-instruction core of the compiler's 16-instruction IEEE sqrt for
    2^-96 <= x < inf, the compiler's sequence otherwise) against __builtin_sqrtf on the device, all 2^32 inputs."""
    assert renderer.selftest_math() == 0
