#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json metric: Mrays/s).

    python bench.py --gpus N --steps K --warmup W [--workload NAME]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one synthetic frame: every rank renders its framebuffer
tiles (64x64, dealt round-robin) with inputs resident in HBM, the tiles are gathered to rank 0
over RCCL and de-tiled there.  value = rays traced by all ranks / max-over-ranks wall time.
Rank 0 prints ONE JSON line (with `roofline` for the dominant kernel and `cpu_baseline` = the CPU
oracle timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="tri1m_1080p_4spp", help="tri1m_1080p_4spp (metric config) | spheres8_1080p_4spp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)  # run under rocprofv3 by measure_traffic()
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC child runs (roofline.traffic = null)")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="frame lanes: consecutive steps go round-robin to this many independent contexts (own stream, own "
                         "pyramid / wavefront state, own output and exchange buffers), so the latency-bound parts of a frame - "
                         "coarse pyramid levels, the tails of the persistent traversal kernels, the RCCL gather - overlap with "
                         "the next frames; 1 = strictly one frame after the other")
    ap.add_argument("--exercise-exchange", action="store_true",
                    help="run the N > 1 code path (streams, events, gather, de-tile) with a 1-rank communicator: a test of "
                         "the plumbing on one GPU, not a measurement")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="debug only: all ranks share cuda:0 and gather through host memory over gloo (exercises the N>1 "
                         "control flow on a 1-GPU box; the line it prints is marked rehearsal and is not a measurement)")
    return ap.parse_args()


def host_threads():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 CPUs but grants 16 per GPU)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def measure_traffic(workload, kernel_substr):
    """HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC counters, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
    (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2), with --kernel-trace only; on gfx950 FETCH_SIZE
    tallies 128-byte requests as 64 bytes, so the read side is doubled (calibrated for wide streaming
    reads; for these 16-byte gathers it is an upper bound of the correction).  Each pass profiles a
    child process that runs two steps of the same workload.  Returns None when rocprofv3 is unusable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None
    kb = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="rt_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--traffic-child", "--workload", workload]
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=600, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                        vals.append(float(row["Counter_Value"]))
            shutil.rmtree(d, ignore_errors=True)
            if not vals:
                return None
            kb[counter] = sum(vals) / len(vals)
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    return {"bytes_per_launch": kb["FETCH_SIZE"] * 1024.0 * 2.0 + kb["WRITE_SIZE"] * 1024.0,
            "fetch_size_kb_raw": round(kb["FETCH_SIZE"], 1), "write_size_kb_raw": round(kb["WRITE_SIZE"], 1),
            "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as read"}


# ---- workloads -----------------------------------------------------------------------------
class SpheresWorkload:
    """BASELINE.json configs[1]: the 8-sphere Cornell-style scene, 1920x1080, 4 spp, path A
    (reference-faithful cone marcher + shading; 4 spp = 2x2 stratified full frames, averaged)."""

    name = "spheres8_1080p_4spp"
    width, height, spp = 1920, 1080, 4
    dtype = "f32"
    dominant_kernel = "shade_kernel"

    def __init__(self, R, renderer):
        self.R, self.r = R, renderer
        self.scene = R.cornell_scene()
        self.rot = np.array([0, 0, 0, 1], np.float32)
        self.pos = np.zeros(3, np.float32)
        renderer.set_scene(self.scene)
        renderer.resize(self.width, self.height)

    def describe(self):
        return {"workload": f"{self.name}: Cornell-style room of 8 sphere SDFs + 1 soft-shadowed point light, "
                            f"{self.width}x{self.height}, {self.spp} spp (2x2 stratified), path A cone-march + shade",
                "width": self.width, "height": self.height, "spp": self.spp, "tile": 64}

    def step(self, out_ptr, tile_major):
        self.r.render_device(self.rot, self.pos, self.spp, out_ptr, tile_major)

    def rays_per_step(self):
        """Rays of THIS rank for one step (after at least one synchronous render)."""
        self.r.render(self.rot, self.pos, spp=self.spp)
        st = self.r.stats()
        return st["primary_rays"] + st["shadow_rays"]

    def roofline(self):
        """HIP-event duration of each kernel (profile_stages) -> dominant kernel vs the HBM roofline.
        Algorithmic bytes: cone level = 8 B/thread (4 B parent read + 4 B store),
        shade = 4 B depth read per sample + 12 B rgb store per pixel; the per-level schedule puts the
        spp samples (up to 16) of a pixel into ONE launch per level and one shade launch."""
        cfg = self.r.default_config()
        cfg.profile_stages = 1
        self.r.set_config(cfg)
        reps, lv, sh = 10, None, 0.0
        for _ in range(reps):
            self.r.render(self.rot, self.pos, spp=self.spp)
            st = self.r.stats()
            lv = np.array(st["ms_level"]) if lv is None else lv + np.array(st["ms_level"])
            sh += st["ms_shade"]
        cfg.profile_stages = 0
        self.r.set_config(cfg)
        lv, sh = lv / reps, sh / reps
        dims = self.r.level_info()
        last = len(dims) - 1
        fused = st["ms_fused"]
        if fused > 0:  # one-launch pyramid: every level texel stored once (4 B); shade: 4 B depth + 12 B rgb
            kernels = {"pyramid_tile_kernel": (fused, sum(w * h for w, h in dims) * 4.0),
                       "shade_kernel": (sh, self.width * self.height * 16.0)}
        else:
            nb = min(self.spp, 16)
            kernels = {f"cone_level_kernel(level {last})": (lv[last], dims[last][0] * dims[last][1] * 8.0 * nb),
                       "shade_kernel": (sh, self.width * self.height * (4.0 * nb + 12.0))}
        name = max(kernels, key=lambda k: kernels[k][0])
        ms, nbytes = kernels[name]
        achieved = nbytes / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "avg_kernel_ms": round(float(ms), 4),
                "algorithmic_bytes_per_launch": nbytes,
                "note": "path A is VALU/sqrt-bound by construction (about 16 B of HBM traffic per pixel against "
                        "thousands of flops); the HBM fraction is reported as the contract asks, not as the limiter",
                "all_kernels_ms": {k: round(float(v[0]), 4) for k, v in kernels.items()}}

    def cpu_baseline(self):
        import oracle as O

        threads = host_threads()
        sc = O.scene_from_bytes(bytes(self.scene))
        n = 2
        O.render_a(sc, 64, 64)  # warm the OpenMP pool
        t0 = time.perf_counter()
        rays, reps = 0, 0
        while time.perf_counter() - t0 < 2.0:  # >= 2 s wall on every granted core
            for s in range(self.spp):
                i, j = s % n, s // n
                jit = (((np.float32(2 * i + 1) / np.float32(n)) - np.float32(1)) / np.float32(self.width),
                       ((np.float32(2 * j + 1) / np.float32(n)) - np.float32(1)) / np.float32(self.height))
                ct = O.render_a(sc, self.width, self.height, rot=self.rot, pos=self.pos, jitter=jit, want_levels=False,
                                threads=threads)["counters"]
                rays += self.width * self.height + ct["shadow_rays"]
            reps += 1
        dt = time.perf_counter() - t0
        # fp32 operations of one step as the oracle's own counters give them: an SDF
        # evaluation is 3 sub + 5 (dot) + sqrt + sub = 10, a march step 3 (position fma) + 3 per object
        # (decrement, compare, min); used by main() for the VALU roofline of this compute-bound path
        ops = 0.0
        for s in range(self.spp):
            i, j = s % n, s // n
            jit = (((np.float32(2 * i + 1) / np.float32(n)) - np.float32(1)) / np.float32(self.width),
                   ((np.float32(2 * j + 1) / np.float32(n)) - np.float32(1)) / np.float32(self.height))
            c = O.render_a(sc, self.width, self.height, rot=self.rot, pos=self.pos, jitter=jit, want_levels=False, threads=threads)["counters"]
            ops += 10.0 * (c["cone_sdf"] + c["shadow_sdf"]) + (3.0 + 3.0 * sc.objCount) * (c["cone_steps"] + c["shadow_steps"])
        self.fp32_ops_per_step = ops
        return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                "sample": f"the full workload {reps}x ({self.spp} spp x {self.width}x{self.height}) with oracle A "
                          f"(OpenMP, {threads} threads), {dt:.2f} s"}


class TriWorkload:
    """BASELINE.json metric config: 1 M-triangle BVH scene, 1920x1080, 4 spp, path B (wavefront path
    tracer: camera ray + next-event shadow ray + 1 diffuse bounce with its own shadow ray).  Build-
    defined extension: the reference has no triangles/BVH, parity is against oracle B only."""

    name = "tri1m_1080p_4spp"
    n_tris, edge = 1_000_000, 0.08
    width, height, spp, bounces, seed = 1920, 1080, 4, 1, 1
    sky = (0.2, 0.2, 0.25)
    dtype = "f32"
    dominant_kernel = "pt_trace<false, false>"

    def __init__(self, R, renderer):
        self.R, self.r = R, renderer
        self.mesh = R.scenes.soup_scene(self.n_tris, seed=1, edge=self.edge)
        self.rot = np.array([0, 0, 0, 1], np.float32)
        self.pos = np.zeros(3, np.float32)
        renderer.set_mesh(*self.mesh)
        renderer.resize(self.width, self.height)
        self.params = renderer.pt_params(spp=self.spp, bounces=self.bounces, seed=self.seed, sky=self.sky)

    def describe(self):
        st = self.r.pt_stats()
        return {"workload": f"{self.name}: {self.n_tris} random triangles (edge +-{self.edge}) + 1 emissive quad, compressed BVH8 "
                            f"({st['n_nodes']} nodes, depth {st['bvh_depth']}), {self.width}x{self.height}, {self.spp} spp, "
                            f"{self.bounces} bounce + NEE, path B wavefront path tracer",
                "width": self.width, "height": self.height, "spp": self.spp, "bounces": self.bounces, "tile": 64,
                "bvh_build_ms_host": round(st["bvh_build_ms"], 1)}

    def step(self, out_ptr, tile_major):
        self.r.render_pt_device(self.rot, self.pos, self.params, out_ptr, tile_major)

    def rays_per_step(self):
        self.r.render_pt(self.rot, self.pos, params=self.params)
        st = self.r.pt_stats()
        return st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]

    def roofline(self):
        """Dominant kernel = pt_trace<closest>.  Algorithmic bytes per launch: every BVH
        compressed 8-wide node fetched = 80 B, every triangle tested = 48 B, per ray 32 B ray read + 8 B hit write + 4 B
        queue entry; counts come from the kernel's own instrumented twin (count_traversal)."""
        r = self.r
        prm = r.pt_params(spp=self.spp, bounces=self.bounces, seed=self.seed, sky=self.sky, count_traversal=True)
        r.render_pt(self.rot, self.pos, params=prm)
        ct = r.pt_stats()
        cfg = r.default_config()
        cfg.profile_stages = 1
        r.set_config(cfg)
        reps = 5
        acc = {k: 0.0 for k in ("ms_generate", "ms_trace_closest", "ms_shade", "ms_trace_shadow", "ms_resolve", "ms_total")}
        for _ in range(reps):
            r.render_pt(self.rot, self.pos, params=self.params)
            st = r.pt_stats()
            for k in acc:
                acc[k] += st[k] / reps
        cfg.profile_stages = 0
        r.set_config(cfg)
        closest_rays = ct["camera_rays"] + ct["bounce_rays"]
        bytes_closest = ct["nodes_visited"] * 80.0 + ct["tris_tested"] * 48.0 + closest_rays * 44.0
        bytes_shadow = ct["shadow_nodes_visited"] * 80.0 + ct["shadow_tris_tested"] * 48.0 + ct["shadow_rays"] * 48.0
        n_launch = st["launches_trace_closest"]
        ms = acc["ms_trace_closest"] / n_launch
        achieved = bytes_closest / (acc["ms_trace_closest"] * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "pt_trace<closest>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "avg_kernel_ms": round(ms, 4), "launches_per_step": n_launch,
                "algorithmic_bytes_per_launch": bytes_closest / n_launch,
                "per_ray": {"nodes": round(ct["nodes_visited"] / closest_rays, 2), "tris": round(ct["tris_tested"] / closest_rays, 2),
                            "bytes": round(bytes_closest / closest_rays, 1),
                            "shadow_nodes": round(ct["shadow_nodes_visited"] / max(ct["shadow_rays"], 1), 2),
                            "shadow_bytes": round(bytes_shadow / max(ct["shadow_rays"], 1), 1)},
                "stage_ms": {k: round(v, 4) for k, v in acc.items()},
                "shadow_kernel_GBs": round(bytes_shadow / max(acc["ms_trace_shadow"], 1e-9) / 1e6, 1),
                "note": "algorithmic bytes charge every node / triangle fetch as if it came from HBM; the 66 MB scene is cache "
                        "resident (L1 hit rate 84 %, L2 79 %, `traffic` = the HBM bytes the PMC counters saw), so frac > 1 means "
                        "node visits per second, not HBM saturation: the binding limits are VALU issue and the vector L1's "
                        "access rate (DESIGN.md section 8; gather ceilings measured by tools/l1_gather_bench.hip: 13.9 TB/s "
                        "L2-resident, 4.4 TB/s from the Infinity Cache)"}

    def cpu_baseline(self):
        import oracle as O

        threads = host_threads()
        sc = O.TriScene(*self.mesh)
        w, h = self.width, self.height  # the full workload, once
        sc.render(32, 18, spp=1, bounces=self.bounces, seed=self.seed, sky=self.sky, threads=threads)
        t0 = time.perf_counter()
        _, ct = sc.render(w, h, spp=self.spp, bounces=self.bounces, seed=self.seed, sky=self.sky, rot=self.rot, pos=self.pos, threads=threads)
        dt = time.perf_counter() - t0
        rays = ct["camera_rays"] + ct["bounce_rays"] + ct["shadow_rays"]
        return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                "sample": f"the full workload once ({w}x{h}, {self.spp} spp, same scene/camera/seed), oracle B with its own "
                          f"median-split BVH (OpenMP, {threads} threads), {rays} rays in {dt:.2f} s"}


class TerrainWorkload(TriWorkload):
    """Context workload, not a BASELINE config: 1 M triangles forming a closed height-field surface (rays
    end at their first hit) with the same camera model, resolution, spp and bounce count."""

    name = "terrain1m_1080p_4spp"

    def __init__(self, R, renderer):
        self.R, self.r = R, renderer
        self.mesh = R.scenes.terrain_scene(708, seed=1)
        self.n_tris = len(self.mesh[0])
        self.rot = R.camera_quat(0.0, -0.25)
        self.pos = np.array([0, 0, 4], np.float32)
        self.sky = (0.4, 0.5, 0.7)
        renderer.set_mesh(*self.mesh)
        renderer.resize(self.width, self.height)
        self.params = renderer.pt_params(spp=self.spp, bounces=self.bounces, seed=self.seed, sky=self.sky)

    def describe(self):
        d = super().describe()
        d["workload"] = d["workload"].replace(f"random triangles (edge +-{self.edge})", "height-field triangles (708x708 cells)")
        return d


WORKLOADS = {SpheresWorkload.name: SpheresWorkload, TriWorkload.name: TriWorkload, TerrainWorkload.name: TerrainWorkload}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch
    import torch.distributed as dist

    import raytracing_engine_amd as R

    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.exercise_exchange
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    r = R.Renderer(local_rank)  # raises when librt_amd.so / the GPU is missing: no fallback
    wl = WORKLOADS[args.workload](R, r)
    if args.traffic_child:  # profiled by measure_traffic(): two plain steps, nothing printed
        buf = torch.empty((wl.height, wl.width, 3), dtype=torch.float32, device=dev)
        for _ in range(2):
            wl.step(buf.data_ptr(), False)
        r.synchronize()
        r.close()
        return
    r.set_partition(rank, world)
    tx, ty, owned = r.tile_info()
    tiles_per_rank = -(-(tx * ty) // world)
    T = 64

    # Frame lanes: lane 0 is (r, wl); the others are further contexts with the same scene.  Step i runs on
    # lane i % L.  Inside a lane everything is stream-ordered: render -> [RCCL gather -> de-tile]; lanes only
    # share the GPU (and the communicator, whose gathers are issued in step order on every rank).
    n_lanes = 1 if args.rehearse_one_gpu else max(1, min(args.frames_in_flight, 8))

    class Lane:
        pass

    lanes = []
    for li in range(n_lanes):
        ln = Lane()
        ln.r = r if li == 0 else R.Renderer(local_rank)
        ln.wl = wl if li == 0 else WORKLOADS[args.workload](R, ln.r)
        ln.r.set_partition(rank, world)
        ln.frame = torch.empty((wl.height, wl.width, 3), dtype=torch.float32, device=dev)
        if multi:
            ln.mine = torch.zeros((tiles_per_rank, T, T, 3), dtype=torch.float32, device=dev)
            ln.gathered = torch.empty((world, tiles_per_rank, T, T, 3), dtype=torch.float32, device=dev) if rank == 0 else None
            # An explicit (non-null) stream per lane carries render and de-tile in order.  The null stream must
            # not be used here: rt_set_stream(NULL) selects the context's own non-blocking stream, which does
            # not synchronise with torch's default stream.
            ln.stream = torch.cuda.Stream(device=dev)
            ln.r.set_stream(ln.stream.cuda_stream)
            ln.ev_rendered = torch.cuda.Event()
            ln.ev_gathered = torch.cuda.Event()
        lanes.append(ln)
    frame = lanes[0].frame
    comm = torch.cuda.Stream(device=dev) if multi else None  # every gather, in step order
    state = {"i": 0}

    def step():
        ln = lanes[state["i"] % n_lanes]
        state["i"] += 1
        if not multi:
            ln.wl.step(ln.frame.data_ptr(), False)
            return
        with torch.cuda.stream(ln.stream):
            ln.wl.step(ln.mine.data_ptr(), True)  # enqueued on the lane's stream by the context
            ln.ev_rendered.record(ln.stream)
        if args.rehearse_one_gpu:  # gloo gathers host tensors
            ln.stream.synchronize()
            host_all = torch.empty(ln.gathered.shape) if rank == 0 else None
            R.host.gather_tiles(ln.mine.cpu(), host_all, rank, dist)
            with torch.cuda.stream(ln.stream):
                if rank == 0:
                    ln.gathered.copy_(host_all)
                    ln.r.detile_device(ln.gathered.data_ptr(), world, tiles_per_rank, ln.frame.data_ptr())
            return
        with torch.cuda.stream(comm):
            comm.wait_event(ln.ev_rendered)
            R.host.gather_tiles(ln.mine, ln.gathered, rank, dist)  # RCCL waits for / is waited on by `comm`
            ln.ev_gathered.record(comm)
        with torch.cuda.stream(ln.stream):
            # the lane's next render reuses `mine` / `gathered`: it is ordered behind this wait
            ln.stream.wait_event(ln.ev_gathered)
            if rank == 0:
                ln.r.detile_device(ln.gathered.data_ptr(), world, tiles_per_rank, ln.frame.data_ptr())

    def drain():
        for ln in lanes:
            ln.r.synchronize()

    def fence():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # set-up, not warm-up: every lane renders one frame so that its buffers exist (the first frame of a context
    # allocates pyramid / wavefront state) whatever --warmup is; the lane rotation then starts at lane 0 again
    for _ in range(n_lanes):
        step()
    fence()
    state["i"] = 0
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0

    rays = wl.rays_per_step()
    tot = torch.tensor([dt, float(rays)], dtype=torch.float64, device="cpu" if args.rehearse_one_gpu else dev)
    if world > 1:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt, rays = float(tmax[0]), float(tot[1])
    else:
        dt, rays = float(tot[0]), float(tot[1])

    if rank == 0:
        out = {"metric": "Mrays/s", "value": round(rays * args.steps / dt / 1e6, 3), "unit": "Mrays/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
               "config": dict(wl.describe(), parallelism=f"tile-split x{world}" if world > 1 else "single GPU",
                              rays_per_step=int(rays), frames_in_flight=n_lanes)}
        check_split = (args.rehearse_one_gpu and world > 1) or args.exercise_exchange
        if args.rehearse_one_gpu:
            out["rehearsal"] = "all ranks on one GPU, gloo gather through host memory: NOT a measurement"
        if args.exercise_exchange:
            out["rehearsal"] = "the N > 1 code path with a 1-rank communicator: NOT a measurement"
        if check_split:
            # the de-tiled frame of the split render must equal a single-context render of the same frame
            import numpy as np
            split = [ln.frame.cpu().numpy().copy() for ln in lanes[:max(1, min(n_lanes, args.steps + args.warmup))]]
        if multi:
            torch.cuda.synchronize()
            r.set_stream(None)
        for ln in lanes[1:]:  # lane 0 goes on to the roofline / CPU-baseline legs
            ln.r.close()
        r.set_partition(0, 1)
        if check_split:
            wl.step(frame.data_ptr(), False)
            r.synchronize()
            single = frame.cpu().numpy()
            out["rehearsal_split_equals_single"] = bool(all((single == f).all() for f in split))
        if not multi:
            out["roofline"] = wl.roofline()
            if not args.no_traffic:
                tr = measure_traffic(args.workload, wl.dominant_kernel)
                if tr:
                    out["roofline"]["traffic"] = tr["bytes_per_launch"]
                    out["roofline"]["traffic_detail"] = tr
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = wl.cpu_baseline()
                if getattr(wl, "fp32_ops_per_step", None):  # path A: the roofline that means something for it
                    tflops = wl.fp32_ops_per_step / (out["ms_per_step"] * 1e-3) / 1e12
                    out["roofline"]["valu"] = {"fp32_ops_per_step": wl.fp32_ops_per_step, "achieved": round(tflops, 2), "peak": 157.3,
                                               "unit": "TFLOP/s", "frac": round(tflops / 157.3, 4),
                                               "note": "useful fp32 operations counted by the oracle (10 per SDF evaluation, 3 + 3 x objects per "
                                                       "march step) over the step time, against the vector-fp32 peak (fma = 2); the correctly "
                                                       "rounded sqrt and division sequences, address and control instructions are not counted"}
        print(json.dumps(out), flush=True)
    if rank != 0:
        for ln in lanes[1:]:
            ln.r.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
