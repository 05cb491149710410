"""The native host harness (host/rt_host.cpp, stand-in for the reference's Rust main) drives the C
ABI end to end and writes images; its output must equal the oracle's frame."""
import os
import subprocess

import numpy as np
import pytest

import oracle as O
import raytracing_engine_amd as R
from raytracing_engine_amd import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "rt_host")


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(x) for x in f.readline().split())
        assert float(f.readline()) < 0  # little endian
        return np.frombuffer(f.read(), "<f4").reshape(h, w, 3)


def read_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = (int(x) for x in f.readline().split())
        assert f.readline().strip() == b"255"
        return np.frombuffer(f.read(), np.uint8).reshape(h, w, 3)


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "host"), "-s"], check=True)
    return EXE


def test_default_scene_frame_and_camera_semantics(exe, tmp_path):
    out = tmp_path / "a.pfm"
    # yaw/pitch/move go through Data::rotation / Data::position semantics (src/main.rs:402-414)
    subprocess.run([exe, "--size", "200x120", "--yaw", "0.4", "--pitch", "-0.1", "--move", "1,2,0.5", "--out", str(out)], check=True)
    q = O.camera_quat(0.4, -0.1)
    pos = 1 * O.rotate(q, (1, 0, 0)) + 2 * O.rotate(q, (0, 1, 0)) + 0.5 * O.rotate(q, (0, 0, 1))
    ref = O.render_a(O.default_scene(), 200, 120, rot=q, pos=pos)["rgb"]
    got = read_pfm(out)
    assert np.abs(got - ref).max() <= 2e-4  # sinf/cosf/rotate on the host side differ by an ulp from numpy's
    ppm = tmp_path / "a.ppm"
    subprocess.run([exe, "--size", "64x64", "--out", str(ppm)], check=True)
    ref8 = O.to_unorm8(O.render_a(O.default_scene(), 64, 64)["rgb"])[::-1, :, :3]  # PPM is top-down
    assert np.array_equal(read_ppm(ppm), ref8)


def test_narrow_window_is_squared_up(exe, tmp_path):
    out = tmp_path / "s.pfm"
    subprocess.run([exe, "--size", "96x160", "--out", str(out)], check=True)  # src/main.rs:702-706
    assert read_pfm(out).shape == (96, 96, 3)


def test_soup_scene_path_traced(exe, tmp_path):
    out = tmp_path / "b.pfm"
    subprocess.run([exe, "--size", "96x54", "--scene", "soup:5000", "--spp", "2", "--bounces", "1", "--seed", "3", "--out", str(out)], check=True)
    v, a, e = scenes.soup_scene(5000, seed=1, edge=0.25)
    ref, _ = O.TriScene(v, a, e).render(96, 54, spp=2, bounces=1, seed=3, sky=(0.2, 0.2, 0.25))
    assert np.array_equal(read_pfm(out), ref)
    out2 = tmp_path / "b2.pfm"  # the same frame over the two-level BVH (rt_set_mesh_ex)
    subprocess.run([exe, "--size", "96x54", "--scene", "soup:5000", "--two-level", "--spp", "2", "--bounces", "1", "--seed", "3", "--out", str(out2)], check=True)
    assert np.array_equal(read_pfm(out2), ref)


def test_sketched_variants_and_frames_in_flight(exe, tmp_path):
    """--march / --repeat reach rt_config; --inflight runs the frame loop through the
    fence-per-image slots (f.3) and must write the same picture as the synchronous loop."""
    out = tmp_path / "v.pfm"
    subprocess.run([exe, "--size", "160x96", "--march", "2", "--repeat", "40,0,40", "--out", str(out)], check=True)
    cfg = O.default_config()
    cfg.march_algorithm = 2
    cfg.repeat[:] = (40.0, 0.0, 40.0)
    ref = O.render_a(O.default_scene(), 160, 96, cfg=cfg)["rgb"]
    assert np.abs(read_pfm(out) - ref).max() <= 1e-4
    subprocess.run([exe, "--size", "160x96", "--mirror", "2,0.75", "--out", str(out)], check=True)  # fragment.glsl:125 "TODO: reflection"
    cfg = O.default_config()
    cfg.reflections, cfg.reflectivity = 2, 0.75
    assert np.abs(read_pfm(out) - O.render_a(O.default_scene(), 160, 96, cfg=cfg)["rgb"]).max() <= 1e-4
    for flag, (n, transparency, index) in (("2,0.75", (2, 0.75, 1.0)), ("2,0.75,1.5", (2, 0.75, 1.5))):  # fragment.glsl:124 "TODO: transparency", :126 "TODO: refraction"
        subprocess.run([exe, "--size", "160x96", "--transmit", flag, "--out", str(out)], check=True)
        cfg = O.default_config()
        cfg.transmissions, cfg.transparency, cfg.refraction_index = n, transparency, index
        assert np.abs(read_pfm(out) - O.render_a(O.default_scene(), 160, 96, cfg=cfg)["rgb"]).max() <= 1e-4
    a, b = tmp_path / "sync.ppm", tmp_path / "slots.ppm"
    subprocess.run([exe, "--size", "160x96", "--frames", "5", "--out", str(a)], check=True)
    res = subprocess.run([exe, "--size", "160x96", "--frames", "5", "--inflight", "3", "--out", str(b)], check=True, capture_output=True, text=True)
    assert "through 3 slots" in res.stdout
    assert np.array_equal(read_ppm(a), read_ppm(b))
    # and against the oracle: UNORM8 of oracle A's frame (PPM is top-down); one 8-bit step is allowed where
    # the 1e-4 RGB tolerance straddles a rounding boundary
    ref8 = O.to_unorm8(O.render_a(O.default_scene(), 160, 96)["rgb"])[::-1, :, :3]
    diff = np.abs(read_ppm(b).astype(np.int16) - ref8.astype(np.int16))
    assert diff.max() <= 1 and np.count_nonzero(diff) <= 0.001 * diff.size


def test_roctx_ranges_are_pushed_and_change_nothing(tmp_path):
    """RT_ROCTX=1 makes the library bracket every stage launch with a roctx range (csrc/rt_roctx.h; the marker library is opened
    with dlopen on first use): the frames of both paths must be what they are without it, and the marker library must really be
    in the process (otherwise the ranges were silently off)."""
    import subprocess
    import sys

    code = (
        "import sys, numpy as np\n"
        "import raytracing_engine_amd as R\n"
        "r = R.Renderer(0)\n"
        "r.set_scene(R.default_scene()); r.resize(160, 96)\n"
        "a = r.render(spp=4)\n"
        "r.set_mesh(*R.scenes.cornell_tri_scene())\n"
        "b = r.render_pt(pos=(0, 1, 0), spp=2, bounces=2, seed=3)\n"
        "np.savez(sys.argv[1], a=a, b=b, roctx=np.array('roctx' in open('/proc/self/maps').read()))\n"
    )
    out = {}
    for flag in ("0", "1"):
        path = tmp_path / f"frames_{flag}.npz"
        subprocess.run([sys.executable, "-c", code, str(path)], check=True, env=dict(os.environ, RT_ROCTX=flag), cwd=ROOT)
        out[flag] = np.load(path)
    assert bool(out["1"]["roctx"]) and not bool(out["0"]["roctx"])
    assert np.array_equal(out["0"]["a"], out["1"]["a"]) and np.array_equal(out["0"]["b"], out["1"]["b"])
