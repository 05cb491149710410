"""The native RCCL exchange behind the C ABI (rt_comm_*, rt_gather_tiles).  A 1-GPU box can only
run the one-rank communicator (RCCL refuses two ranks on one device), so this covers the entry
points, the root's own-tile path and the de-tile; the N-rank control flow is covered by
tests/test_host.py (gloo) and bench.py --rehearse-one-gpu."""
import numpy as np
import pytest

import raytracing_engine_amd as R

pytestmark = pytest.mark.gpu


def test_single_rank_gather_and_detile():
    import torch

    r = R.Renderer(0)
    try:
        r.set_scene(R.default_scene())
        r.resize(300, 200)
        full = r.render()
        uid = R.Renderer.comm_unique_id()
        assert len(uid) == 128
        r.comm_init(uid, 0, 1)
        tx, ty, owned = r.tile_info()
        assert owned == tx * ty
        mine = torch.zeros((owned, 64, 64, 3), dtype=torch.float32, device="cuda")
        gathered = torch.zeros((1, owned, 64, 64, 3), dtype=torch.float32, device="cuda")
        out = torch.empty((200, 300, 3), dtype=torch.float32, device="cuda")
        r.render_device((0, 0, 0, 1), (0, 0, 0), 1, mine.data_ptr(), tile_major=True)
        r.gather_tiles(mine.data_ptr(), gathered.data_ptr(), owned)
        r.detile_device(gathered.data_ptr(), 1, owned, out.data_ptr())
        r.synchronize()
        assert np.array_equal(out.cpu().numpy(), full)
        r.comm_destroy()
        with pytest.raises(R.RtError):
            r.gather_tiles(mine.data_ptr(), gathered.data_ptr(), owned)  # RT_ERR_STATE after destroy
    finally:
        r.close()


@pytest.mark.parametrize("lanes", [3, 1])
def test_bench_exchange_path_with_one_rank(lanes):
    """bench.py's N > 1 path (one stream per frame lane, events, torch.distributed gather over RCCL on the
    exchange stream, de-tile; consecutive frames on different lanes overlap) driven with a 1-rank
    communicator: every lane's de-tiled frame must be the single-context frame."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29540 + lanes))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--exercise-exchange", "--frames-in-flight", str(lanes), "--steps", "4", "--warmup", "2",
                          "--workload", "spheres8_1080p_4spp", "--no-traffic", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["rehearsal_split_equals_single"] is True
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["frames_in_flight"] == lanes
