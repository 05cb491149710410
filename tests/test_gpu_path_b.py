"""GPU parity tests for path B (triangle BVH + wavefront path tracer), through the C ABI.

Path B has NO reference counterpart: parity is HIP kernels vs oracle B on identical
scenes and seeds — "parity unpinned by the reference".  Bar: north_star's 1e-4 max-abs RGB; the
arithmetic contract actually makes the frames bit-identical, which is what the
tests assert, together with exact ray counts."""
import os

import numpy as np
import pytest

import oracle as O
import raytracing_engine_amd as R
from raytracing_engine_amd import scenes

pytestmark = pytest.mark.gpu
RGB_TOL = 1e-4


def check_pt(r, mesh, w, h, rot=(0, 0, 0, 1), pos=(0, 0, 0), exact=True, **kw):
    v, a, e = mesh
    r.set_mesh(v, a, e)
    r.resize(w, h)
    rgb = r.render_pt(rot, pos, **kw)
    okw = {k: kw[k] for k in ("spp", "bounces", "seed", "sky", "ray_eps") if k in kw}
    ref, ct = O.TriScene(v, a, e).render(w, h, rot=rot, pos=pos, **okw)
    err = np.abs(rgb - ref).max()
    assert err <= RGB_TOL, err
    if exact:
        assert np.array_equal(rgb, ref), f"{np.count_nonzero(rgb != ref)} values differ, max {err}"
    st = r.pt_stats()
    assert st["stack_overflow"] == 0
    assert (st["camera_rays"], st["bounce_rays"], st["shadow_rays"]) == (ct["camera_rays"], ct["bounce_rays"], ct["shadow_rays"])
    return rgb, ref, st


def test_trace_rays_matches_bruteforce_oracle(renderer):
    v, a, e = scenes.soup_scene(20000, seed=4, edge=1.0)
    renderer.set_mesh(v, a, e)
    sc = O.TriScene(v, a, e)
    rng = np.random.default_rng(8)
    n = 4000
    o = rng.uniform([-12, 0, -12], [12, 30, 12], size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[:50] = [0, 1, 0]  # axis-aligned rays (zero direction components)
    d[50:100] = [1, 0, 0]
    d[100:150] = [-0.0, 1, -0.0]  # negative zeros
    d[150:200] = [-0.0, -0.0, -1]
    t, tri = renderer.trace_rays(o, d)
    hits = 0
    for i in range(n):
        rt_, rtt = sc.closest_hit(o[i], d[i], use_bvh=False)
        assert tri[i] == rt_, i
        if rt_ >= 0:
            hits += 1
            assert t[i] == np.float32(rtt)
        else:
            assert np.isinf(t[i])
    assert hits > 500
    seg = (d * rng.uniform(1, 25, size=(n, 1))).astype(np.float32)
    _, occ = renderer.trace_rays(o, seg, any_hit=True)
    ref = np.array([sc.occluded(o[i], seg[i], use_bvh=False) for i in range(n)])
    assert np.array_equal(occ.astype(bool), ref) and 0.05 < ref.mean() < 0.95


def test_cornell_parity(renderer):
    rgb, ref, st = check_pt(renderer, scenes.cornell_tri_scene(), 128, 128, pos=(0, 1, 0), spp=4, bounces=2, seed=7)
    assert rgb.mean() > 0.05 and st["camera_rays"] == 128 * 128 * 4


@pytest.mark.parametrize("bounces,spp", [(0, 1), (1, 4), (3, 2), (8, 1)])
def test_bounce_and_spp_grid(renderer, bounces, spp):
    check_pt(renderer, scenes.cornell_tri_scene(), 96, 64, pos=(0, 1, 0), rot=R.camera_quat(0.2, -0.1), spp=spp, bounces=bounces, seed=3)


def test_soup_parity_with_sky(renderer):
    check_pt(renderer, scenes.soup_scene(20000, seed=1, edge=0.8), 160, 90, spp=2, bounces=1, seed=5, sky=(0.3, 0.3, 0.4))


def test_soup_100k_small_view(renderer):
    """BASELINE.json configs[2] scene (100 k random triangles) at a view the oracle finishes in seconds."""
    check_pt(renderer, scenes.soup_scene(100000, seed=1), 192, 108, spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25))


@pytest.mark.parametrize("name,args", [("path_b_cornell_64.npz", dict(kind="cornell", w=64, h=64, spp=4, bounces=2, seed=7, pos=(0, 1, 0))),
                                       ("path_b_soup2k_96x54.npz", dict(kind="soup", w=96, h=54, spp=2, bounces=1, seed=5, sky=(0.3, 0.3, 0.4)))])
def test_against_committed_fixture(renderer, golden_dir, name, args):
    g = np.load(os.path.join(golden_dir, name))
    kind, w, h = args.pop("kind"), args.pop("w"), args.pop("h")
    v, a, e = scenes.cornell_tri_scene() if kind == "cornell" else scenes.soup_scene(2000, seed=3, edge=1.5)
    renderer.set_mesh(v, a, e)
    renderer.resize(w, h)
    pos = args.pop("pos", (0, 0, 0))
    rgb = renderer.render_pt(pos=pos, **args)
    assert np.array_equal(rgb, g["rgb"])
    st = renderer.pt_stats()
    assert [st["camera_rays"], st["bounce_rays"], st["shadow_rays"]] == g["counters"].tolist()


def test_sample_passes_do_not_change_the_image(renderer):
    """spp split into several passes (max_paths) accumulates in the same order as one pass."""
    v, a, e = scenes.cornell_tri_scene()
    renderer.set_mesh(v, a, e)
    renderer.resize(64, 64)
    one = renderer.render_pt(pos=(0, 1, 0), spp=6, bounces=2, seed=2)
    many = renderer.render_pt(pos=(0, 1, 0), spp=6, bounces=2, seed=2, max_paths=2 * 4096)  # 2 samples per pass
    assert np.array_equal(one, many)
    ref, _ = O.TriScene(v, a, e).render(64, 64, spp=6, bounces=2, seed=2, pos=(0, 1, 0))
    assert np.array_equal(one, ref)


def test_rays_through_vertices_edges_and_duplicate_triangles(renderer):
    """The boundary cases of the triangle test (u, v or u + v exactly on an edge; signed zeros) and the (t, id) tie-break: rays
    straight through the vertices, edge midpoints and diagonals of a regular grid of quads, every triangle present TWICE (the
    copy with the smaller original index must win a tie), from both sides (both signs of the determinant).  Closest hits and
    occlusion must be the oracle's brute-force results bit for bit; the straight-line form of the test (tri_test_flat) is what runs."""
    f = np.float32
    tris = []
    for i in range(-4, 4):
        for j in range(-4, 4):
            a_, b_, c_, d_ = (i, 5, j), (i + 1, 5, j), (i + 1, 5, j + 1), (i, 5, j + 1)
            tris += [a_ + b_ + c_, a_ + c_ + d_]
    v = np.array(tris + tris, f)  # coplanar duplicates: ties on t for every hit
    a = np.full((len(v), 3), 0.5, f)
    e = np.zeros((len(v), 3), f)
    e[-1] = 1.0
    renderer.set_mesh(v, a, e)
    sc = O.TriScene(v, a, e)
    xs = np.arange(-4.5, 4.75, 0.25, dtype=f)
    gx, gz = np.meshgrid(xs, xs)
    n = gx.size
    for y0, dy in ((0.0, 1.0), (9.0, -1.0)):  # from below and from above
        o = np.stack([gx.ravel(), np.full(n, y0, f), gz.ravel()], 1).astype(f)
        d = np.tile(np.array([0.0, dy, 0.0], f), (n, 1))
        d[::3, 0] = -0.0  # signed zeros in the direction
        t, tri = renderer.trace_rays(o, d)
        hits = 0
        for k in range(n):
            rt_, rtt = sc.closest_hit(o[k], d[k], use_bvh=False)
            assert tri[k] == rt_, (k, o[k], tri[k], rt_)
            if rt_ >= 0:
                hits += 1
                assert t[k] == np.float32(rtt) and rt_ < len(tris)  # the lower index of a duplicate pair
        assert hits > n // 2
        seg = (d * f(7.0)).astype(f)
        _, occ = renderer.trace_rays(o, seg, any_hit=True)
        ref = np.array([sc.occluded(o[k], seg[k], use_bvh=False) for k in range(n)])
        assert np.array_equal(occ.astype(bool), ref)
    # and through the render kernels (packet kernel for the camera rays): a camera below the grid, looking straight up
    check_pt(renderer, (v, a, e), 65, 65, pos=(0.0, 0.0, 0.0), spp=2, bounces=1, sky=(0.4, 0.5, 0.6))


def test_tiny_and_degenerate_meshes(renderer):
    f = np.float32
    one = (np.array([[-1, 5, -1, 1, 5, -1, 0, 5, 1]], f), np.array([[0.5, 0.6, 0.7]], f), np.zeros((1, 3), f))
    check_pt(renderer, one, 32, 32, spp=2, bounces=1, sky=(1, 1, 1))
    # zero-area triangles are never hit; lights only (every path ends at depth 0)
    v, a, e = scenes.cornell_tri_scene()
    v = np.concatenate([v, np.array([[0, 5, 0, 0, 5, 0, 0, 5, 0]], f)])
    a = np.concatenate([a, np.array([[1, 1, 1]], f)])
    e = np.concatenate([e, np.zeros((1, 3), f)])
    check_pt(renderer, (v, a, e), 48, 48, pos=(0, 1, 0), spp=2, bounces=2)


@pytest.mark.parametrize("n_ranks", [2, 8])
def test_partition_union_equals_single(renderer, n_ranks):
    import torch

    v, a, e = scenes.cornell_tri_scene()
    w, h = 200, 136
    renderer.set_mesh(v, a, e)
    renderer.resize(w, h)
    renderer.set_partition(0, 1)
    prm = renderer.pt_params(spp=2, bounces=2, seed=9)
    full = renderer.render_pt(pos=(0, 1, 0), params=prm)
    tx, ty, _ = renderer.tile_info()
    per = -(-(tx * ty) // n_ranks)
    gathered = torch.zeros((n_ranks, per, 64, 64, 3), dtype=torch.float32, device="cuda")
    try:
        for rank in range(n_ranks):
            renderer.set_partition(rank, n_ranks)
            renderer.render_pt_device((0, 0, 0, 1), (0, 1, 0), prm, gathered[rank].data_ptr(), tile_major=True)
            renderer.synchronize()
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        renderer.detile_device(gathered.data_ptr(), n_ranks, per, out.data_ptr())
        renderer.synchronize()
        assert np.array_equal(out.cpu().numpy(), full)
    finally:
        renderer.set_partition(0, 1)


def test_traversal_counters_and_errors(renderer):
    v, a, e = scenes.soup_scene(5000, seed=2, edge=1.0)
    renderer.set_mesh(v, a, e)
    renderer.resize(64, 64)
    renderer.render_pt(spp=1, bounces=1, count_traversal=True)
    st = renderer.pt_stats()
    rays = st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
    assert st["nodes_visited"] >= rays and st["tris_tested"] > 0 and st["n_tris"] == 5000 and st["bvh_depth"] <= 30
    with pytest.raises(R.RtError):
        renderer.render_pt(spp=0)
    with pytest.raises(R.RtError):
        renderer.render_pt(bounces=99)
    bad = v.copy()
    bad[3, 4] = np.nan
    with pytest.raises(R.RtError):
        renderer.set_mesh(bad, a, e)
    fresh = R.Renderer(0)
    fresh.resize(32, 32)
    with pytest.raises(R.RtError) as ei:
        fresh.render_pt()
    assert ei.value.code == -4
    fresh.close()


def test_full_hd_properties(renderer):
    """BASELINE.json metric size (1920x1080, 4 spp) on the 100 k-triangle scene: too big for the
    oracle in test time, so check size-independent properties: determinism, ray-count identities,
    tile-split identity on a band, finiteness."""
    v, a, e = scenes.soup_scene(100000, seed=1)
    renderer.set_mesh(v, a, e)
    renderer.resize(1920, 1080)
    prm = renderer.pt_params(spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25))
    a1 = renderer.render_pt(params=prm)
    st = renderer.pt_stats()
    a2 = renderer.render_pt(params=prm)
    assert np.array_equal(a1, a2) and np.isfinite(a1).all() and (a1 >= 0).all()
    assert st["camera_rays"] == 1920 * 1080 * 4 and st["bounce_rays"] <= st["camera_rays"] and st["stack_overflow"] == 0
    # band parity at full size: the RNG is keyed by the global pixel index, so the oracle can render
    # just rows 520..536 of the 1920x1080 frame and must reproduce those rows of the GPU frame exactly
    band, _ = O.TriScene(v, a, e).render(1920, 1080, spp=4, bounces=1, seed=1, sky=(0.2, 0.2, 0.25), rows=(520, 536))
    assert np.array_equal(a1[520:536], band)


def test_deep_paths_and_many_samples(renderer):
    """Maximum bounce count (15) and a large spp on a small view."""
    check_pt(renderer, scenes.cornell_tri_scene(), 24, 24, pos=(0, 1, 0), spp=1, bounces=15, seed=4)
    check_pt(renderer, scenes.cornell_tri_scene(), 16, 16, pos=(0, 1, 0), spp=64, bounces=1, seed=6)


def test_mesh_swaps(renderer):
    """rt_set_mesh replaces mesh, BVH and lights; frames follow."""
    for mesh, pos in [(scenes.cornell_tri_scene(), (0, 1, 0)), (scenes.soup_scene(3000, seed=5, edge=1.0), (0, 0, 0)), (scenes.cornell_tri_scene(), (0, 1, 0))]:
        check_pt(renderer, mesh, 64, 48, pos=pos, spp=2, bounces=1, seed=2, sky=(0.1, 0.1, 0.1))


def test_shadow_overlap_does_not_change_results(renderer):
    """The shadow kernel runs beside the next closest-hit kernel by default; serialised it must give
    the same frame (per-path sums keep their order)."""
    v, a, e = scenes.cornell_tri_scene()
    renderer.set_mesh(v, a, e)
    renderer.resize(96, 96)
    one = renderer.render_pt(pos=(0, 1, 0), spp=3, bounces=3, seed=8)  # default: shadow(d) and closest(d + 1) in one persistent launch
    st1 = renderer.pt_stats()
    ref, ct = O.TriScene(v, a, e).render(96, 96, spp=3, bounces=3, seed=8, pos=(0, 1, 0))
    assert np.array_equal(one, ref)
    for mode in (1, 2):  # one launch after the other; two launches on two streams
        two = renderer.render_pt(pos=(0, 1, 0), spp=3, bounces=3, seed=8, tune_no_overlap=mode)
        st2 = renderer.pt_stats()
        assert np.array_equal(one, two)
        assert (st1["camera_rays"], st1["bounce_rays"], st1["shadow_rays"]) == (st2["camera_rays"], st2["bounce_rays"], st2["shadow_rays"]) == (ct["camera_rays"], ct["bounce_rays"], ct["shadow_rays"])
    assert st1["launches_trace_closest"] == 4 and st1["launches_trace_shadow"] == 4


_KNOB_SCENE = {}


def _knob_scene():
    """The knob tests' scene and the ORACLE's frame + ray counts for it (computed once per session)."""
    if not _KNOB_SCENE:
        v, a, e = scenes.soup_scene(30000, seed=9, edge=0.6)
        ref, ct = O.TriScene(v, a, e).render(160, 96, spp=2, bounces=2, seed=5, sky=(0.2, 0.2, 0.25))
        _KNOB_SCENE.update(mesh=(v, a, e), ref=ref, ct=ct)
    return _KNOB_SCENE


@pytest.mark.parametrize("knobs", [dict(tune_refill_min=1), dict(tune_refill_min=64), dict(tune_refill_min=8 | (3 << 8)), dict(tune_blocks_per_cu=1),
                                   dict(tune_blocks_per_cu=3, tune_lds_stack=2), dict(tune_lds_stack=1), dict(tune_lds_stack=40),
                                   dict(tune_no_packet=1), dict(tune_sort_rays=1), dict(tune_sort_rays=1, tune_no_packet=1, tune_no_overlap=1),
                                   dict(tune_no_overlap=1), dict(tune_no_overlap=2),
                                   dict(tune_tri_mode=1), dict(tune_tri_mode=1 | (1 << 8)), dict(tune_tri_mode=1 | (1 << 8), tune_refill_min=1), dict(tune_tri_mode=2), dict(tune_tri_mode=2 | (1 << 8) | (1 << 16)),
                                   dict(tune_tri_mode=2 | (64 << 8) | (255 << 16)), dict(tune_tri_mode=2 | (7 << 8) | (3 << 16), tune_refill_min=1),
                                   dict(tune_tri_mode=2, tune_no_packet=1, tune_no_overlap=1, tune_lds_stack=1),
                                   dict(tune_tri_mode=2, tune_no_overlap=2, tune_refill_min=64), dict(tune_tri_mode=2, tune_sort_rays=1, tune_blocks_per_cu=1),
                                   dict(tune_tri_mode=3), dict(tune_tri_mode=3 | (1 << 8) | (1 << 16)), dict(tune_tri_mode=3 | (64 << 8) | (64 << 16)),
                                   dict(tune_tri_mode=3 | (64 << 8) | (255 << 16), tune_refill_min=1), dict(tune_tri_mode=3 | (16 << 8) | (2 << 16), tune_no_packet=1, tune_no_overlap=1, tune_lds_stack=1),
                                   dict(tune_tri_mode=3, tune_no_overlap=2, tune_refill_min=64),
                                   dict(tune_tri_mode=4), dict(tune_tri_mode=4, tune_refill_min=1), dict(tune_tri_mode=4, tune_refill_min=64), dict(tune_tri_mode=4, tune_refill_min=24 | (3 << 8)),
                                   dict(tune_tri_mode=4, tune_no_packet=1, tune_no_overlap=1, tune_lds_stack=1), dict(tune_tri_mode=4, tune_no_overlap=2, tune_blocks_per_cu=1),
                                   dict(tune_tri_mode=4, tune_sort_rays=1), dict(tune_sort_rays=2), dict(tune_sort_rays=2, tune_no_packet=1, tune_refill_min=8),
                                   dict(tune_no_packet=2), dict(tune_no_packet=3), dict(tune_no_packet=4), dict(tune_no_packet=5), dict(tune_no_packet=3, tune_no_overlap=1)])
def test_scheduling_knobs_do_not_change_the_frame(renderer, knobs):
    """Refill threshold, triangle tests per round, inline / wave-pooled triangle tests and the pool's flush rule, resident
    workgroups, LDS / spill split of the traversal stack, rays sorted in LDS, launch overlap, the packet kernel's node test
    (per-ray slab tests / interval test per pass, with and without the per-ray second step): pure scheduling, so the frame and
    the ray counts must equal the ORACLE's (not just the default configuration's)."""
    k = _knob_scene()
    renderer.set_mesh(*k["mesh"])
    renderer.resize(160, 96)
    got = renderer.render_pt(spp=2, bounces=2, seed=5, sky=(0.2, 0.2, 0.25), **knobs)
    st = renderer.pt_stats()
    assert np.array_equal(got, k["ref"]), f"{np.count_nonzero(got != k['ref'])} values differ from the oracle's frame"
    assert st["stack_overflow"] == 0
    for c in ("camera_rays", "bounce_rays", "shadow_rays"):
        assert st[c] == k["ct"][c], c


def test_queue_streams_cover_every_entry_exactly_once(renderer):
    """The ray queue is consumed through 16 interleaved stream heads (64-entry blocks); queue lengths
    around the block / stream boundaries must neither drop nor repeat an entry: the frame equals the
    oracle's for views of 1 .. a few thousand paths."""
    mesh = scenes.cornell_tri_scene()
    for w, h in [(1, 1), (7, 9), (8, 8), (63, 1), (64, 1), (65, 1), (32, 32), (1023, 1), (1025, 1), (129, 17)]:
        check_pt(renderer, mesh, w, h, pos=(0, 1, 0), spp=1, bounces=2, seed=w * 31 + h)


@pytest.mark.parametrize("tri_mode", [2, 3, 4])
def test_experimental_schedules_on_short_and_ragged_queues(renderer, tri_mode):
    """The wave-pooled (2), postponed (3) and prefetch-ring (4) schedules of the per-lane kernels on queue lengths around the
    block / stream / ring boundaries (1 ray, fewer rays than a prefetch batch, one more than a wave, ...), where their drain
    and exit rules are exercised with nearly empty rings: frames and ray counts equal to the oracle's."""
    mesh = scenes.cornell_tri_scene()
    for w, h in [(1, 1), (3, 2), (7, 1), (9, 1), (8, 8), (65, 1), (33, 31), (129, 17)]:
        check_pt(renderer, mesh, w, h, pos=(0, 1, 0), spp=1, bounces=3, seed=w * 31 + h, tune_tri_mode=tri_mode, tune_no_packet=(w * h) % 2)
    check_pt(renderer, scenes.soup_scene(3000, seed=5, edge=1.0), 40, 30, spp=3, bounces=2, seed=6, sky=(0.1, 0.1, 0.1), tune_tri_mode=tri_mode, tune_refill_min=2)



def test_far_camera_within_the_padded_range_and_rejection_beyond(renderer):
    """The ray/box test's rounding error grows with the distance of the ray origin; the box padding covers
    camera coordinates up to 32 x the mesh's largest |coordinate| (include/rt_abi.h, rt_render_pt).  A
    telephoto view from just inside that range must still equal the oracle (whose own BVH and box test are
    different: any missed box would show); beyond it the call is refused."""
    mesh = scenes.soup_scene(20000, seed=6, edge=0.8)
    m = float(np.abs(mesh[0]).max())
    renderer.set_mesh(*mesh)
    w, h = 96, 64
    renderer.resize(w, h, ratio=(0.016, 0.016 * h / w))  # the scene fills the view from ~800 units away
    pos = (0.0, -31.0 * m, 0.0)
    rgb = renderer.render_pt(pos=pos, spp=2, bounces=1, seed=3, sky=(0.3, 0.3, 0.4))
    ref, ct = O.TriScene(*mesh).render(w, h, pos=pos, ratio=(0.016, 0.016 * h / w), spp=2, bounces=1, seed=3, sky=(0.3, 0.3, 0.4))
    st = renderer.pt_stats()
    assert ct["bounce_rays"] > 0.5 * ct["camera_rays"], "the view must look at the mesh"
    assert np.array_equal(rgb, ref)
    assert (st["camera_rays"], st["bounce_rays"], st["shadow_rays"]) == (ct["camera_rays"], ct["bounce_rays"], ct["shadow_rays"])
    with pytest.raises(R.RtError) as ei:
        renderer.render_pt(pos=(0.0, -33.0 * m, 0.0), spp=1)
    assert ei.value.code == -1 and "padding" in str(ei.value)
    renderer.resize(64, 64)


@pytest.fixture(scope="module")
def walker(tmp_path_factory):
    """tests/native/bvh8_walk.cpp: an independent host-side walk of the same compressed 8-wide BVH."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path_factory.mktemp("walk") / "bvh8_walk"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-ffp-contract=off", "-fno-fast-math", os.path.join(root, "tests", "native", "bvh8_walk.cpp"),
                    os.path.join(root, "raytracing_engine_amd", "csrc", "bvh_build.cpp"), os.path.join(root, "raytracing_engine_amd", "csrc", "bvh_two_level.cpp"),
                    "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.parametrize("n_tris,edge", [(30000, 0.6), (38, 0.0)])
def test_traversal_counts_match_the_host_walk_of_the_same_bvh(renderer, walker, tmp_path, n_tris, edge):
    """bench.py's roofline quotes rt_pt_stats.nodes_visited / tris_tested.  They are produced by the traversal
    step functions themselves (COUNT instantiation); here an independent host program (scalar per-child code
    written from the node layout) walks the same tree for the same rays and must report the same number of
    node fetches and triangle tests for EVERY ray, closest-hit and any-hit, plus the same hits."""
    import subprocess

    mesh = scenes.cornell_tri_scene() if n_tris == 38 else scenes.soup_scene(n_tris, seed=12, edge=edge)
    v = mesh[0]
    renderer.set_mesh(*mesh)
    rng = np.random.default_rng(3)
    n = 6000
    o = rng.uniform([-12, 0, -12], [12, 30, 12], size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # a coherent camera-like fan from one origin, and axis-parallel rays
    o[:2000] = (0, 1, 0)
    fan = np.stack([np.linspace(-0.6, 0.6, 2000), np.ones(2000), 0.4 * np.sin(np.linspace(0, 40, 2000))], 1)
    d[:2000] = (fan / np.linalg.norm(fan, axis=1, keepdims=True)).astype(np.float32)
    d[2000:2050] = (0, 1, 0)
    d[2050:2100] = (-1, 0, 0)
    d[2100:2150] = (-0.0, 1, -0.0)  # negative zeros: the reciprocal is negative, the octant must follow the sign bit
    d[2150:2200] = (-0.0, -0.0, 1)
    for any_hit in (0, 1):
        dd = (d * rng.uniform(1, 25, size=(n, 1))).astype(np.float32) if any_hit else d
        t, tri, counts = renderer.trace_rays(o, dd, any_hit=bool(any_hit), counted=True)
        fin, fout = tmp_path / f"in{any_hit}.bin", tmp_path / f"out{any_hit}.bin"
        with open(fin, "wb") as f:
            f.write(np.array([len(v), n, any_hit], np.uint32).tobytes())
            f.write(np.ascontiguousarray(v, np.float32).tobytes())
            f.write(o.tobytes())
            f.write(dd.tobytes())
        subprocess.run([walker, str(fin), str(fout)], check=True)
        rec = np.fromfile(fout, dtype=np.dtype([("nodes", "<u4"), ("tris", "<u4"), ("t", "<f4"), ("tri", "<i4")]))
        assert len(rec) == n
        assert np.array_equal(rec["tri"], tri) and np.array_equal(rec["t"], t)
        assert np.array_equal(rec["nodes"], counts[:, 0]), f"node fetches differ for {np.count_nonzero(rec['nodes'] != counts[:, 0])} rays"
        assert np.array_equal(rec["tris"], counts[:, 1]), f"triangle tests differ for {np.count_nonzero(rec['tris'] != counts[:, 1])} rays"
        assert counts[:, 0].min() >= 1 and (n_tris == 38 or counts[:, 0].sum() > 3 * n)


def test_two_level_bvh_frames_counts_and_chunk_rebuild(renderer, walker, tmp_path):
    """BASELINE.json configs[2] names a "2-level BVH": rt_set_mesh_ex(bvh_levels = 2) builds a top level over 64
    bottom-level chunks and flattens both into the node array the kernels walk.  The frame must be the oracle's (and
    therefore the single-level frame), the host walker must reproduce the per-ray traversal counts on ITS two-level
    build, and after rt_update_mesh_chunk moved one chunk's triangles the frame must be the oracle's for the moved mesh."""
    import subprocess

    v, a, e = scenes.soup_scene(30000, seed=14, edge=0.6)
    kw = dict(spp=2, bounces=2, seed=11, sky=(0.2, 0.2, 0.25))
    renderer.resize(128, 80)
    renderer.set_mesh(v, a, e)
    one = renderer.render_pt(**kw)
    renderer.set_mesh(v, a, e, bvh_levels=2, blas_chunks=64)
    st = renderer.pt_stats()
    assert st["bvh_levels"] == 2 and st["blas_chunks"] == 64 and st["tlas_nodes"] >= 9 and st["n_nodes"] > st["tlas_nodes"]
    two = renderer.render_pt(**kw)
    ref, ct = O.TriScene(v, a, e).render(128, 80, **kw)
    st = renderer.pt_stats()
    assert np.array_equal(two, ref) and np.array_equal(one, two) and st["stack_overflow"] == 0
    assert (st["camera_rays"], st["bounce_rays"], st["shadow_rays"]) == (ct["camera_rays"], ct["bounce_rays"], ct["shadow_rays"])
    # traversal counts of the flattened two-level tree against the host walk of the same build
    rng = np.random.default_rng(5)
    n = 3000
    o = rng.uniform([-12, 0, -12], [12, 30, 12], size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    t, tri, counts = renderer.trace_rays(o, d, counted=True)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(np.array([len(v), n, 0 | (64 << 8)], np.uint32).tobytes())
        f.write(np.ascontiguousarray(v, np.float32).tobytes())
        f.write(o.tobytes())
        f.write(d.tobytes())
    subprocess.run([walker, str(fin), str(fout)], check=True)
    rec = np.fromfile(fout, dtype=np.dtype([("nodes", "<u4"), ("tris", "<u4"), ("t", "<f4"), ("tri", "<i4")]))
    assert np.array_equal(rec["tri"], tri) and np.array_equal(rec["t"], t)
    assert np.array_equal(rec["nodes"], counts[:, 0]) and np.array_equal(rec["tris"], counts[:, 1])
    # move the triangles of one chunk (the light's chunk too, so the light list has to follow) and rebuild only it
    light_chunk = next(c for c in range(64) if (len(v) - 1) in renderer.mesh_chunk(c))
    sizes = [len(renderer.mesh_chunk(c)) for c in range(64)]  # leaves of the SAH cut: not equal, but the fullest leaf is always split next
    assert sum(sizes) == len(v) and min(sizes) >= 1 and max(sizes) <= 4 * (len(v) // 64)
    assert len(np.unique(np.concatenate([renderer.mesh_chunk(c) for c in range(64)]))) == len(v)
    v2 = v.copy()
    for chunk in (5, light_chunk):
        ids = renderer.mesh_chunk(chunk)
        assert len(np.unique(ids)) == len(ids)
        v2[ids] += np.tile(np.array([0.4, -0.3, 0.2], np.float32), 3)
        renderer.update_mesh_chunk(chunk, v2[ids])
    st = renderer.pt_stats()
    assert st["ms_build_blas"] > 0 and st["bvh_build_ms"] >= st["ms_build_blas"]
    moved = renderer.render_pt(**kw)
    ref2, ct2 = O.TriScene(v2, a, e).render(128, 80, **kw)
    st = renderer.pt_stats()
    assert np.array_equal(moved, ref2) and not np.array_equal(moved, two)
    assert (st["camera_rays"], st["bounce_rays"], st["shadow_rays"]) == (ct2["camera_rays"], ct2["bounce_rays"], ct2["shadow_rays"])
    renderer.set_mesh(v2, a, e, bvh_levels=2)
    assert np.array_equal(renderer.render_pt(**kw), ref2)
    # error behaviour: no such chunk; vertices leaving the range the padding was chosen for (the mesh stays as it was); single-level mesh
    with pytest.raises(R.RtError) as ei:
        renderer.update_mesh_chunk(64, v2[:10])
    assert ei.value.code == -1
    ids = renderer.mesh_chunk(0)
    with pytest.raises(R.RtError) as ei:
        renderer.update_mesh_chunk(0, v2[ids] * 100.0)
    assert ei.value.code == -1
    for wrong in (v2[ids][:-1], np.concatenate([v2[ids], v2[ids][:1]])):  # a vertex array that is not the chunk's size is refused, not read
        with pytest.raises(R.RtError) as ei:
            renderer.update_mesh_chunk(0, wrong)
        assert ei.value.code == -1 and "holds" in str(ei.value)
    assert np.array_equal(renderer.render_pt(**kw), ref2)
    renderer.set_mesh(v, a, e)
    with pytest.raises(R.RtError) as ei:
        renderer.update_mesh_chunk(0, v[:10])
    assert ei.value.code == -4
    with pytest.raises(R.RtError):
        renderer.set_mesh(v, a, e, bvh_levels=3)
    renderer.resize(64, 64)


@pytest.mark.parametrize("yaw,pitch,pos,spp,w,h", [
    (0.0, 0.0, (0, 0, 0), 3, 131, 77),            # spp that does not divide a wave: packets straddle pixels irregularly
    (np.pi, 0.0, (0, 30, 0), 2, 96, 64),          # looking along -Y from behind the mesh: every direction sign flipped
    (np.pi / 2, 0.3, (-14, 15, 0), 1, 100, 60),   # along +X, pitched: the view centre crosses octant boundaries
    (-np.pi / 2, -0.7, (14, 15, 6), 4, 64, 64),   # along -X, looking down
    (0.4, 1.2, (0, 14, -11), 2, 80, 48),          # steep upward pitch: rays towards +Z
    (2.5, -0.2, (1, 15, 2), 5, 70, 40),           # camera INSIDE the mesh volume
])
def test_packet_kernel_camera_poses(renderer, yaw, pitch, pos, spp, w, h):
    """The camera rays go through the wave-uniform packet kernel (one tree walk per 64 paths, lanes of a deviating direction
    octant in a further pass): frames and ray counts must be the oracle's for views whose packets mix octants, for sample
    counts that do not tile a wave, for a camera inside the mesh, and must equal the per-lane kernel's and every node-test
    variant's of the packet kernel (tune_no_packet)."""
    mesh = scenes.soup_scene(20000, seed=17, edge=0.7)
    rot = R.camera_quat(yaw, pitch)
    rgb, ref, st = check_pt(renderer, mesh, w, h, rot=rot, pos=pos, spp=spp, bounces=1, seed=9, sky=(0.3, 0.3, 0.4))
    assert st["camera_rays"] == w * h * spp
    for mode in (1, 2, 3, 4, 5):  # per-lane kernel; packet kernel with per-ray slab tests; interval test + slab tests; interval test only; ... without the cap
        other = renderer.render_pt(rot, pos, spp=spp, bounces=1, seed=9, sky=(0.3, 0.3, 0.4), tune_no_packet=mode)
        assert np.array_equal(rgb, other), mode
    fetched = {}
    for mode in (2, 3, 4):
        renderer.render_pt(rot, pos, spp=spp, bounces=1, seed=9, sky=(0.3, 0.3, 0.4), count_traversal=True, tune_no_packet=mode)
        ct = renderer.pt_stats()
        assert ct["packets"] == -(-(-(-w // 64) * -(-h // 64) * 4096 * spp) // 64) and ct["packet_nodes_fetched"] >= ct["packets"]
        assert ct["stack_overflow"] == 0
        fetched[mode] = (ct["packet_nodes_fetched"], ct["packet_tris_fetched"])
    # the interval test lets through a superset of the children some ray hits, and with the per-ray second step exactly those
    assert fetched[3] == fetched[2] and fetched[4][0] >= fetched[2][0] and fetched[4][1] >= fetched[2][1]


def test_packet_kernel_random_views(renderer):
    """The interval test of the packet kernel is a conservative filter: whatever the view, the frame must be the per-lane kernel's
    bit for bit.  Forty random views of three meshes - a sparse soup, a dense one whose triangles are large against the packets,
    the closed terrain - from outside, inside and far away, axis-parallel directions (a zero direction component makes one
    axis of the interval unbounded) included, with sample counts and sizes that leave lanes of a wave without a ray."""
    rng = np.random.default_rng(20251005)
    meshes = [scenes.soup_scene(6000, seed=3, edge=0.5), scenes.soup_scene(1500, seed=4, edge=4.0), scenes.terrain_scene(48, seed=2)]
    sizes = [(96, 64), (70, 40), (131, 33), (64, 64)]
    for k in range(40):
        mesh = meshes[k % 3]
        w, h = sizes[k % 4]
        renderer.set_mesh(*mesh)
        renderer.resize(w, h)
        if k % 8 == 7:  # axis-parallel view directions
            yaw, pitch = float(rng.integers(0, 4)) * np.pi / 2, float(rng.integers(-1, 2)) * np.pi / 2
        else:
            yaw, pitch = float(rng.uniform(-np.pi, np.pi)), float(rng.uniform(-1.5, 1.5))
        scale = (3.0, 12.0, 60.0)[k % 3]
        pos = tuple(float(x) for x in (rng.uniform(-1, 1, 3) * scale + np.array([0.0, 15.0, 0.0]) * (k % 3 != 2)))
        spp = int(rng.integers(1, 6))
        rot = R.camera_quat(yaw, pitch)
        kw = dict(spp=spp, bounces=1, seed=k, sky=(0.3, 0.3, 0.4))
        ref = renderer.render_pt(rot, pos, tune_no_packet=1, **kw)
        assert renderer.pt_stats()["stack_overflow"] == 0
        for mode in (0, 3, 2):
            got = renderer.render_pt(rot, pos, tune_no_packet=mode, **kw)
            assert np.array_equal(got, ref), (k, mode, yaw, pitch, pos, spp)
            assert renderer.pt_stats()["stack_overflow"] == 0


def test_chunk_update_accepts_the_mesh_it_was_built_from(renderer):
    """rt_update_mesh_chunk checks the new vertices against the coordinate range the box padding was chosen for.  That range is
    stored at build time: reconstructing it as pad / 2e-5 loses an ulp for about 8 % of the ranges (100.0 among them) and then
    refuses the very chunk that holds the extreme vertex, even unchanged."""
    v, a, e = scenes.soup_scene(4000, seed=21, edge=0.5)
    v = v.copy()
    v[123, 0] = 100.0  # largest |coordinate| of the mesh: exactly 100.0
    assert np.abs(v).max() == np.float32(100.0)
    assert np.float32(np.float32(2e-5) * np.float32(100.0)) / np.float32(2e-5) < np.float32(100.0)  # the round trip this test is about
    kw = dict(spp=1, bounces=1, seed=4, sky=(0.2, 0.2, 0.25))
    renderer.resize(96, 64)
    renderer.set_mesh(v, a, e, bvh_levels=2, blas_chunks=16)
    before = renderer.render_pt(**kw)
    for chunk in range(16):
        ids = renderer.mesh_chunk(chunk)
        renderer.update_mesh_chunk(chunk, v[ids])  # unchanged vertices: every chunk, the extreme vertex's too, must be accepted
    assert np.array_equal(renderer.render_pt(**kw), before)
    ref, _ = O.TriScene(v, a, e).render(96, 64, **kw)
    assert np.array_equal(before, ref)
    # the camera reach (32 x the stored range) is exact as well
    assert np.array_equal(renderer.render_pt(pos=(0, -3200.0, 0), **kw), O.TriScene(v, a, e).render(96, 64, pos=(0, -3200.0, 0), **kw)[0])
    with pytest.raises(R.RtError):
        renderer.render_pt(pos=(0, -3200.5, 0), **kw)
    renderer.resize(64, 64)


@pytest.mark.parametrize("n_tris", [1, 3, 38, 700])
def test_two_level_bvh_small_meshes(renderer, n_tris):
    """Two-level builds of meshes too small for 64 chunks (at least four triangles per chunk; one chunk at the limit)."""
    if n_tris == 38:
        mesh, pos = scenes.cornell_tri_scene(), (0, 1, 0)
    elif n_tris == 1:
        f = np.float32
        mesh, pos = (np.array([[-1, 5, -1, 1, 5, -1, 0, 5, 1]], f), np.array([[0.5, 0.6, 0.7]], f), np.zeros((1, 3), f)), (0, 0, 0)
    else:
        mesh, pos = scenes.soup_scene(n_tris, seed=21, edge=2.0), (0, 0, 0)
    v, a, e = mesh
    renderer.set_mesh(v, a, e, bvh_levels=2)
    st = renderer.pt_stats()
    assert st["bvh_levels"] == 2 and 1 <= st["blas_chunks"] <= max(1, n_tris // 4)
    renderer.resize(64, 48)
    rgb = renderer.render_pt(pos=pos, spp=2, bounces=2, seed=5, sky=(0.4, 0.4, 0.5))
    ref, ct = O.TriScene(v, a, e).render(64, 48, pos=pos, spp=2, bounces=2, seed=5, sky=(0.4, 0.4, 0.5))
    st = renderer.pt_stats()
    assert np.array_equal(rgb, ref) and st["stack_overflow"] == 0
    assert (st["camera_rays"], st["bounce_rays"], st["shadow_rays"]) == (ct["camera_rays"], ct["bounce_rays"], ct["shadow_rays"])
    ids = np.concatenate([renderer.mesh_chunk(c) for c in range(st["blas_chunks"])])
    assert sorted(ids.tolist()) == list(range(len(v)))  # the chunks partition the mesh
