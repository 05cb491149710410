"""GPU parity tests at BASELINE.json's FULL sizes on the 1 M-triangle scene: the metric's own workload
(1920x1080, 4 spp), configs[3] (8 spp, tile-split over 8 ranks + gather + de-tile) and configs[4]
(3840x2160, 64 spp, 8 bounces).  Through the C ABI, against oracle B.

Path B has NO reference counterpart: "parity unpinned by the reference".  The oracle
cannot render these frames whole in test time, so each test compares a band of rows of the full-size
frame (the RNG is keyed by the global pixel index, so a band of the oracle's frame is exactly those rows
of the whole frame) bit for bit, and the whole-frame ray counts with the counts the oracle produced
for the whole workload in the authoring container (tests/golden/path_b_tri1m_counts.json, written by
tests/golden/make_golden_counts.py)."""
import json
import os

import numpy as np
import pytest

import oracle as O
from raytracing_engine_amd import scenes

pytestmark = pytest.mark.gpu

SKY = (0.2, 0.2, 0.25)
N_TRIS, EDGE = 1_000_000, 0.08


@pytest.fixture(scope="module")
def tri1m(renderer):
    """The 1 M-triangle soup on the device (BVH built once for the module) and in the oracle."""
    mesh = scenes.soup_scene(N_TRIS, seed=1, edge=EDGE)
    renderer.set_mesh(*mesh)
    renderer.set_partition(0, 1)
    st = renderer.pt_stats()
    assert st["n_tris"] == N_TRIS and st["n_lights"] == 2
    return mesh, O.TriScene(*mesh)


@pytest.fixture(scope="module")
def pinned(golden_dir):
    return json.load(open(os.path.join(golden_dir, "path_b_tri1m_counts.json")))


def counts(st):
    return {k: st[k] for k in ("camera_rays", "bounce_rays", "shadow_rays")}


def test_config2_tri100k_two_level_1080p_4spp(renderer, pinned):
    """configs[2] AS NAMED: 100 k random triangles with the 2-level BVH (top level over 64 bottom-level chunks), 1920x1080,
    4 spp: whole-frame ray counts against the oracle's (pinned), three bands of the frame bit for bit, and the frame equal to
    the single-level tree's.  (Runs before the module's 1 M-triangle fixture puts its mesh on the device.)"""
    mesh = scenes.soup_scene(100_000, seed=1, edge=0.25)
    renderer.set_partition(0, 1)
    renderer.resize(1920, 1080)
    renderer.set_mesh(*mesh, bvh_levels=2, blas_chunks=64)
    st = renderer.pt_stats()
    assert st["bvh_levels"] == 2 and st["blas_chunks"] == 64 and st["tlas_nodes"] >= 9
    rgb = renderer.render_pt(spp=4, bounces=1, seed=1, sky=SKY)
    st = renderer.pt_stats()
    assert st["stack_overflow"] == 0 and np.isfinite(rgb).all()
    assert counts(st) == {k: pinned["tri100k_1080p_4spp"][k] for k in counts(st)}
    sc = O.TriScene(*mesh)
    for rows in [(0, 8), (536, 552), (1072, 1080)]:
        band, _ = sc.render(1920, 1080, spp=4, bounces=1, seed=1, sky=SKY, rows=rows)
        assert np.array_equal(rgb[rows[0]:rows[1]], band), rows
    renderer.set_mesh(*mesh)
    assert np.array_equal(renderer.render_pt(spp=4, bounces=1, seed=1, sky=SKY), rgb)


def test_metric_workload_tri1m_1080p_4spp(renderer, tri1m, pinned):
    """BASELINE.json metric config = bench.py's default workload."""
    _, sc = tri1m
    renderer.resize(1920, 1080)
    rgb = renderer.render_pt(spp=4, bounces=1, seed=1, sky=SKY)
    st = renderer.pt_stats()
    assert st["stack_overflow"] == 0 and np.isfinite(rgb).all()
    assert counts(st) == {k: pinned["tri1m_1080p_4spp"][k] for k in counts(st)}
    for rows in [(0, 8), (536, 552), (1072, 1080)]:
        band, _ = sc.render(1920, 1080, spp=4, bounces=1, seed=1, sky=SKY, rows=rows)
        assert np.array_equal(rgb[rows[0]:rows[1]], band), rows


def test_config3_tri1m_1080p_8spp_split_over_8_ranks(renderer, tri1m, pinned):
    """configs[3]: 8 spp, framebuffer tiles dealt over 8 ranks (emulated one after the other on this GPU),
    tile-major buffers gathered rank-major and de-tiled: equal to the single-context frame, whose band
    equals the oracle's."""
    import torch

    _, sc = tri1m
    w, h, n_ranks = 1920, 1080, 8
    renderer.resize(w, h)
    prm = renderer.pt_params(spp=8, bounces=1, seed=1, sky=SKY)
    single = renderer.render_pt(params=prm)
    assert counts(renderer.pt_stats()) == {k: pinned["tri1m_1080p_8spp"][k] for k in ("camera_rays", "bounce_rays", "shadow_rays")}
    tx, ty, _ = renderer.tile_info()
    per = -(-(tx * ty) // n_ranks)
    gathered = torch.zeros((n_ranks, per, 64, 64, 3), dtype=torch.float32, device="cuda")
    rays = 0
    try:
        for rank in range(n_ranks):
            renderer.set_partition(rank, n_ranks)
            renderer.render_pt_device((0, 0, 0, 1), (0, 0, 0), prm, gathered[rank].data_ptr(), tile_major=True)
            renderer.synchronize()
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        renderer.detile_device(gathered.data_ptr(), n_ranks, per, out.data_ptr())
        renderer.synchronize()
        split = out.cpu().numpy()
        for rank in range(n_ranks):  # ray counts of the ranks add up to the frame's
            renderer.set_partition(rank, n_ranks)
            renderer.render_pt(params=prm)
            st = renderer.pt_stats()
            rays += st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
    finally:
        renderer.set_partition(0, 1)
    assert np.array_equal(split, single)
    assert rays == sum(pinned["tri1m_1080p_8spp"][k] for k in ("camera_rays", "bounce_rays", "shadow_rays"))
    band, _ = sc.render(w, h, spp=8, bounces=1, seed=1, sky=SKY, rows=(300, 308))
    assert np.array_equal(split[300:308], band)


def test_config4_tri1m_4k_64spp_8bounces(renderer, tri1m):
    """configs[4]: 3840x2160, 64 spp, 8 bounces (531 M paths in 16 passes of the 2^25-path wavefront)."""
    _, sc = tri1m
    w, h = 3840, 2160
    renderer.resize(w, h)
    rgb = renderer.render_pt(spp=64, bounces=8, seed=1, sky=SKY)
    st = renderer.pt_stats()
    assert st["stack_overflow"] == 0 and np.isfinite(rgb).all()
    assert st["camera_rays"] == w * h * 64 and st["bounce_rays"] > st["camera_rays"]
    rows = (1200, 1202)
    band, ct = sc.render(w, h, spp=64, bounces=8, seed=1, sky=SKY, rows=rows)
    assert np.array_equal(rgb[rows[0]:rows[1]], band)
    assert ct["camera_rays"] == w * 2 * 64 and ct["bounce_rays"] > ct["camera_rays"]
    renderer.resize(64, 64)  # leave a small view behind for the tests that follow


def test_context_tri16m_scene_beyond_the_infinity_cache(renderer):
    """bench.py's context workload tri16m_1080p_4spp: the headline soup with 16 M triangles (1.6 GB of nodes, triangle records and
    materials = six times the 256 MiB Infinity Cache), the one scene on which the kernels run against HBM.  Two bands of the
    full-size frame bit for bit against oracle B (its own BVH2 over the 16 M triangles), stack within bounds.  (Last in the file:
    it replaces the module's 1 M-triangle mesh.)"""
    mesh = scenes.soup_scene(16_000_000, seed=1, edge=0.032)
    renderer.set_partition(0, 1)
    renderer.resize(1920, 1080)
    renderer.set_mesh(*mesh)
    st = renderer.pt_stats()
    assert st["n_tris"] == 16_000_000 and st["n_nodes"] * 80 + st["n_tris"] * 80 > 4 * (256 << 20)
    rgb = renderer.render_pt(spp=4, bounces=1, seed=1, sky=SKY)
    st = renderer.pt_stats()
    assert st["stack_overflow"] == 0 and np.isfinite(rgb).all() and st["camera_rays"] == 1920 * 1080 * 4
    sc = O.TriScene(*mesh)
    for rows in [(270, 272), (806, 808)]:
        band, _ = sc.render(1920, 1080, spp=4, bounces=1, seed=1, sky=SKY, rows=rows)
        assert np.array_equal(rgb[rows[0]:rows[1]], band), rows
    del sc
    renderer.set_mesh(*scenes.cornell_tri_scene())  # release the 1.6 GB
    renderer.resize(64, 64)
