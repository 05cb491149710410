#!/usr/bin/env python3
"""Exploratory: two processes, native rt_comm_* exchange (unique id through a file).
    python tools/comm_two_ranks.py            # spawns both ranks; each uses device RANK % device_count"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) == 1:
    idf = "/tmp/rt_comm_id.bin"
    if os.path.exists(idf):
        os.remove(idf)
    ps = [subprocess.Popen([sys.executable, __file__, str(r), "2", idf]) for r in range(2)]
    sys.exit(max(p.wait() for p in ps))

rank, n, idf = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
import torch  # noqa: E402

import raytracing_engine_amd as R  # noqa: E402

dev = rank % torch.cuda.device_count()
torch.cuda.set_device(dev)
r = R.Renderer(dev)
if rank == 0:
    uid = R.Renderer.comm_unique_id()
    open(idf + ".tmp", "wb").write(uid)
    os.rename(idf + ".tmp", idf)
else:
    while not os.path.exists(idf):
        time.sleep(0.05)
    uid = open(idf, "rb").read()
r.set_scene(R.default_scene())
r.resize(300, 200)
r.comm_init(uid, rank, n)
tx, ty, owned = r.tile_info()
per = -(-(tx * ty) // n)
mine = torch.zeros((per, 64, 64, 3), dtype=torch.float32, device="cuda")
gathered = torch.zeros((n, per, 64, 64, 3), dtype=torch.float32, device="cuda") if rank == 0 else None
r.render_device((0, 0, 0, 1), (0, 0, 0), 1, mine.data_ptr(), tile_major=True)
r.gather_tiles(mine.data_ptr(), gathered.data_ptr() if rank == 0 else 0, per)
if rank == 0:
    out = torch.empty((200, 300, 3), dtype=torch.float32, device="cuda")
    r.detile_device(gathered.data_ptr(), n, per, out.data_ptr())
    r.synchronize()
    r.comm_destroy()
    r.set_partition(0, 1)
    full = r.render()
    print("two-rank native gather == single frame:", bool(np.array_equal(out.cpu().numpy(), full)))
else:
    r.synchronize()
    r.comm_destroy()
r.close()
