"""CPU tests of the boundary and the host logic: the C-ABI library loads and exports every symbol
include/rt_abi.h declares (no compute calls: there is no GPU here), struct layouts match the
header as gcc sees it, the host mirror of src/main.rs behaves like the reference, and the
multi-rank gather/de-tile plumbing works over gloo with world_size 2."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
import raytracing_engine_amd as R
from raytracing_engine_amd import _lib, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_abi.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = R.load()
    names = declared_functions()
    assert len(names) >= 32 and "rt_render" in names and "rt_render_pt" in names
    for n in names:
        assert hasattr(lib, n), f"librt_amd.so does not export {n}"
        assert n in _lib.PROTOTYPES, f"python binding lacks a prototype for {n}"
    assert sorted(_lib.PROTOTYPES) == names
    assert lib.rt_abi_version() == 4


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof/offsetof as gcc computes them from include/rt_abi.h vs the ctypes mirrors."""
    prog = tmp_path / "layout.c"
    prog.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "rt_abi.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(rt_mutable_data), sizeof(rt_material), sizeof(rt_object), sizeof(rt_light),
           sizeof(rt_config), sizeof(rt_stats), sizeof(rt_pt_params), sizeof(rt_pt_stats));
    printf("%zu %zu %zu %zu %zu\n", offsetof(rt_mutable_data, mats), offsetof(rt_mutable_data, objs), offsetof(rt_mutable_data, lights),
           offsetof(rt_stats, ms_total), offsetof(rt_pt_stats, camera_rays));
    return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    sizes = [int(x) for x in out]
    assert sizes[:8] == [656, 32, 16, 32, C.sizeof(R.Config), C.sizeof(R.Stats), C.sizeof(R.PtParams), C.sizeof(R.PtStats)]
    assert sizes[8:] == [16, 272, 400, R.Stats.ms_total.offset, R.PtStats.camera_rays.offset]
    assert (R.MutableData.mats.offset, R.MutableData.objs.offset, R.MutableData.lights.offset) == (16, 272, 400)


def test_no_gpu_means_a_loud_failure_not_a_fallback():
    lib = R.load()
    n = C.c_int(-1)
    rc = lib.rt_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(R.RtError) as e:
        R.Renderer(0)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_defaults_match_reference_constants():
    lib = R.load()
    cfg = R.Config()
    assert lib.rt_default_config(C.byref(cfg)) == 0
    assert (cfg.render_dist, round(cfg.cam_fall_off, 6), round(cfg.light_fall_off, 6), round(cfg.ray_radius, 6)) == (1000.0, 0.01, 0.01, 0.01)
    s = R.MutableData()
    assert lib.rt_default_scene(C.byref(s)) == 0
    assert bytes(s) == bytes(host.default_scene()) == bytes(O.default_scene())  # src/main.rs:524-591, three statements
    assert lib.rt_default_config(None) == -1 and lib.rt_default_scene(None) == -1


@pytest.mark.parametrize("w,h", [(8, 8), (64, 64), (256, 256), (1000, 700), (1920, 1080), (2048, 2048), (3840, 2160)])
def test_pyramid_host_logic_matches_oracle(w, h):
    count = host.level_count(w)
    assert count == O.level_count(w)
    assert [host.level_dims(w, h, count, i) for i in range(count)] == [O.level_dims(w, h, count, i) for i in range(count)]
    np.testing.assert_array_equal(host.default_ratio(w, h), np.array([1.0, np.float32(h) / np.float32(w)], np.float32))


def test_camera_controller_follows_main_rs():
    cam = host.CameraController()
    cam.rotate(0.0, 5.0)  # pitch clamps to +-pi/2 (src/main.rs:770)
    assert abs(cam.rotation[1] - np.pi / 2) < 1e-6
    cam = host.CameraController()
    cam.move_local(0, 2, 0)  # W: +forward = +Y at identity rotation (rotation::FORWARD, :353)
    np.testing.assert_allclose(cam.pos, [0, 2, 0], atol=1e-6)
    cam.rotate(np.pi / 2, 0)  # yaw right by 90 degrees: forward becomes +X (from_rotation_z(-yaw), :403)
    cam.move_local(0, 1, 0)
    np.testing.assert_allclose(cam.pos, [1, 2, 0], atol=1e-6)
    cam.move_local(1, 0, 1)   # right is now -Y, up stays +Z
    np.testing.assert_allclose(cam.pos, [1, 1, 1], atol=1e-6)
    np.testing.assert_allclose(cam.quat(), O.camera_quat(np.pi / 2, 0), atol=1e-6)


def test_tile_partition_round_trip():
    rng = np.random.default_rng(0)
    frame = rng.random((200, 300, 3), dtype=np.float32)
    for n_ranks in (1, 2, 3, 8):
        tx, ty = -(-300 // 64), -(-200 // 64)
        per = -(-(tx * ty) // n_ranks)
        tiles = np.concatenate([host.frame_to_tiles(frame, r, n_ranks, per) for r in range(n_ranks)])
        np.testing.assert_array_equal(host.tiles_to_frame(tiles, n_ranks, per, 300, 200), frame)


_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from raytracing_engine_amd import host
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
w, h = 300, 200
frame = np.random.default_rng(7).random((h, w, 3), dtype=np.float32)   # what a single GPU would render
tx, ty = -(-w // 64), -(-h // 64)
per = -(-(tx * ty) // world)
mine = torch.from_numpy(host.frame_to_tiles(frame, rank, world, per))  # this rank's tiles, tile-major
gathered = torch.empty((world, per, 64, 64, 3)) if rank == 0 else None
host.gather_tiles(mine, gathered, rank, dist)
t = torch.tensor([1.0 + rank])
dist.all_reduce(t, op=dist.ReduceOp.MAX)   # bench.py's max-over-ranks timing reduction
assert float(t) == float(world)
if rank == 0:
    out = host.tiles_to_frame(gathered.numpy().reshape(-1, 64, 64, 3), world, per, w, h)
    assert np.array_equal(out, frame), "tile-split frame differs from the single-rank frame"
# the exchange as rt_gather_tiles does it: every rank sends only the tiles it owns (20 tiles on 3 ranks: 7, 7, 6);
# the tile buffer holds exactly `owned` tiles, as include/rt_abi.h documents, and the pad tile of the short rank
# stays as the root left it
owned = host.owned_tiles(tx * ty, rank, world)
exact = mine[:owned].clone()
gathered2 = torch.full((world, per, 64, 64, 3), -1.0) if rank == 0 else None
host.gather_owned_tiles(exact, gathered2, rank, world, tx * ty, dist)
if rank == 0:
    out = host.tiles_to_frame(gathered2.numpy().reshape(-1, 64, 64, 3), world, per, w, h)
    assert np.array_equal(out, frame), "owned-count exchange differs from the single-rank frame"
    for peer in range(world):
        k = host.owned_tiles(tx * ty, peer, world)
        assert (gathered2[peer, k:] == -1.0).all(), "a pad tile was written"
    print("OK")
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,port", [(2, 29517), (3, 29519)])
def test_gather_over_gloo(tmp_path, world, port):
    """world_size 2 and 3 on CPU: the N>1 path of bench.py (tile ownership, gather to rank 0, de-tile), and the
    owned-count exchange of rt_gather_tiles on an uneven split (300x200 = 20 tiles on 3 ranks: 7, 7, 6)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    assert "OK" in outs[0][0]


def test_graft_entry_build_runs():
    """The driver's "does it build" check: __graft_entry__.build() compiles the library, the oracle and the host
    CLI (incremental here) and verifies the ABI version the loaded library reports."""
    import __graft_entry__ as g

    g.build()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` as a plain command (no torch.distributed.run around it): the parent spawns N rank
    processes before anything touches a GPU, relays rank 0's JSON line and propagates a rank's failure.  Driven
    here through the stub step (gloo, no GPU): launcher, RANK / WORLD_SIZE / MASTER_* plumbing, the timed bracket,
    the max-over-ranks reduction and the tile gather are bench.py's own code."""
    import json

    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "4", "--warmup", "1", "--stub-step"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [json.loads(ln) for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["stub"] is True
    assert out["parity"]["ok"] and out["config"]["rays_per_step"] == 300 * 200
    assert out["ms_per_step"] >= 4.0  # the slower rank (2 x 2 ms per step) sets the time: max over ranks
    bad = subprocess.run([sys.executable, bench, "--gpus", "3", "--steps", "2", "--stub-step", "--stub-fail-rank", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode == 3 and "rank 2 exited" in bad.stderr
    assert not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]
    # under an external launcher the environment decides; a mismatch is refused
    mism = subprocess.run([sys.executable, bench, "--gpus", "2", "--stub-step"], env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert mism.returncode != 0 and "WORLD_SIZE=1" in mism.stderr
