#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json metric: Mrays/s).

    python bench.py --gpus N --steps K --warmup W [--workload NAME]
    N > 1 either under a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (the driver:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...) or as a plain
    command: without WORLD_SIZE in the environment this process only spawns the N rank processes (before
    anything touches a GPU), relays rank 0's JSON line and exits non-zero if any rank failed.

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
    ap.add_argument("--workload", default="tri1m_1080p_4spp",
                    help="tri1m_1080p_4spp (metric config) | spheres8_1080p_4spp (configs[1]) | terrain1m_1080p_4spp, tri16m_1080p_4spp (context)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity gate (the timed frame checked against the oracle after the timed region)")
    ap.add_argument("--stub-step", action="store_true", help=argparse.SUPPRESS)  # CPU test of the launcher + N>1 control flow (gloo, no GPU, no kernels)
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)  # run under rocprofv3 by measure_traffic()
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC child runs (roofline.traffic = null)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frame lanes (0 = the workload's default: 3 for path B, 4 for path A - one per hardware queue, measured "
                         "0.406 against 0.432 ms per step with 3): consecutive steps go round-robin to this many independent contexts (own stream, own "
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


def pmc_pass(workload, kernel_substr, counters):
    """One rocprofv3 PMC pass (--kernel-trace + --pmc only, the program directly after "--") over a child process that runs
    two steps of the workload: {counter: average value per launch of the kernel}.  None when rocprofv3 is unusable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None
    d = tempfile.mkdtemp(prefix="rt_pmc_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.abspath(__file__), "--traffic-child", "--workload", workload]
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=900, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        vals = {c: [] for c in counters}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] in vals and kernel_substr in row["Kernel_Name"]:
                    vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
        if not all(vals.values()):
            return None
        return {c: sum(v) / len(v) for c, v in vals.items()}
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def measure_traffic(workload, kernel_substr):
    """Fabric-side bytes per launch of the dominant kernel from rocprofv3 PMC counters, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
    (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2), with --kernel-trace only; on gfx950 FETCH_SIZE
    tallies 128-byte requests as 64 bytes, so the read side is doubled (calibrated for wide streaming
    reads; for these 16-byte gathers it is an upper bound of the correction).  These are L2 MISSES: requests
    the L2 sent to the fabric, whether the Infinity Cache (256 MiB) or HBM answered them - an upper bound of
    the HBM traffic, close to it only when the working set exceeds the Infinity Cache.  Returns None when
    rocprofv3 is unusable."""
    kb = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        got = pmc_pass(workload, kernel_substr, [counter])
        if got is None:
            return None
        kb[counter] = got[counter]
    return {"bytes_per_launch": kb["FETCH_SIZE"] * 1024.0 * 2.0 + kb["WRITE_SIZE"] * 1024.0,
            "fetch_size_kb_raw": round(kb["FETCH_SIZE"], 1), "write_size_kb_raw": round(kb["WRITE_SIZE"], 1),
            "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as read"}


# Issue cost of one wave64 vector instruction on one SIMD, measured by tools/valu_rate_bench.hip at 8 waves per SIMD
# (profiles/r02_valu_issue_rates.txt): v_fma / v_mul / v_add / v_mov_b32 2.65-2.87 cycles, every other class the traversal
# kernels use (v_cvt_f32_ubyteN, v_max3 / v_min3, v_cmp, v_cndmask, shifts, bit-field and logic ops, integer multiplies) 4.3-4.8.
VALU_FAST_CYCLES, VALU_OTHER_CYCLES = 2.75, 4.7
# Share of the fast class among the vector instructions of the per-lane traversal kernels' code (static count over the ISA of
# pt_trace_fused: 0.32; the node step alone: 54 fast-class instructions in 183): tools/valu_mix.py prints it from the built library.
VALU_FAST_SHARE = 0.32
N_SIMDS = 256 * 4


def measure_valu_issue(workload, kernel_substr):
    """The vector-ALU side of the dominant kernel from one PMC pass: wave-level vector instructions per launch, lanes active per
    instruction, the SQ's own busy figure, and the issue-slot fraction = instructions x measured issue cycles / (SIMDs x kernel cycles)."""
    got = pmc_pass(workload, kernel_substr, ["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"])
    if got is None:
        return None
    insts = got["SQ_INSTS_VALU"]
    cycles = got["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
    avg = VALU_FAST_SHARE * VALU_FAST_CYCLES + (1.0 - VALU_FAST_SHARE) * VALU_OTHER_CYCLES
    return {"bound": "valu_issue", "insts_per_launch": insts, "lanes_per_instruction": round(got["SQ_THREAD_CYCLES_VALU"] / max(insts, 1.0), 1),
            "kernel_cycles": round(cycles), "issue_cycles_per_instruction": round(avg, 2),
            "frac": round(insts * avg / (N_SIMDS * max(cycles, 1.0)), 3),
            "sq_active_inst_valu_frac": round(got["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMDS * max(cycles, 1.0)), 3),
            "definition": "SQ_INSTS_VALU x issue cycles per instruction (0.32 x 2.75 + 0.68 x 4.7: tools/valu_mix.py, tools/valu_rate_bench.hip at 8 waves per SIMD) / "
                          "(1024 SIMDs x GRBM_GUI_ACTIVE / 8); about 1 = every issue slot taken (the isolated-stream costs overstate a mixed "
                          "stream by a few per cent); lanes_per_instruction = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU of the same pass; "
                          "sq_active_inst_valu_frac = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / the same denominator"}


# Gather ceilings of the cache hierarchy for the BVH node access pattern (every lane reads one random 80-byte
# record per iteration), measured on MI355X by tools/l1_gather_bench.hip; raw output: profiles/r02_l1_gather_bench.txt
GATHER_CEILING_L2_TBS = 13.9   # table fits one XCD's L2
GATHER_CEILING_IC_TBS = 4.4    # 16-20 MiB table, served from the Infinity Cache over the fabric


def finish_roofline(rl, traffic):
    """`achieved` / `frac` of the HBM roofline from the L2-miss (fabric-side) bytes the PMC counters saw per launch of the
    dominant kernel - Infinity-Cache hits included, so an UPPER bound of the HBM traffic and of the fraction - when they
    could be collected; the algorithmic figure
    (every fetch charged as if it came from HBM) stays beside it under `algorithmic`, where
    a cache-resident scene makes it exceed the HBM peak - that ratio is node visits per second, not a
    roofline fraction, and is not called one."""
    ms = rl["avg_kernel_ms"]
    alg = rl.pop("algorithmic_bytes_per_launch")
    alg_gbs = alg / (ms * 1e-3) / 1e9
    rl["algorithmic"] = {"bytes_per_launch": alg, "GBps": round(alg_gbs, 1), "over_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4),
                         "definition": rl.pop("algorithmic_definition")}
    if traffic:
        ach = traffic["bytes_per_launch"] / (ms * 1e-3) / 1e9
        rl.update(achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4), hbm_counter_frac=round(ach / HBM_PEAK_GBS, 4),
                  traffic=traffic["bytes_per_launch"], traffic_detail=traffic,
                  traffic_over_algorithmic=round(traffic["bytes_per_launch"] / alg, 3),
                  achieved_definition="L2-miss (fabric-side) bytes per launch incl. Infinity-Cache hits, from rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE, "
                                      "separate passes) / avg_kernel_ms: an upper bound of the HBM traffic; equal to it only when the working set "
                                      "exceeds the 256 MiB Infinity Cache (workload tri16m_1080p_4spp)")
    elif alg_gbs <= HBM_PEAK_GBS:  # counters unavailable: the algorithmic figure is still a valid lower bound on the fraction
        rl.update(achieved=round(alg_gbs, 1), frac=round(alg_gbs / HBM_PEAK_GBS, 4), hbm_counter_frac=None, traffic=None,
                  achieved_definition="algorithmic bytes per launch / avg_kernel_ms (PMC traffic not collected)")
    else:
        rl.update(achieved=None, frac=None, hbm_counter_frac=None, traffic=None,
                  achieved_definition="PMC traffic not collected and the algorithmic rate exceeds the HBM peak (cache-resident data): no HBM fraction reported")
    rl.update(bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s")
    return rl


def launch_ranks(n):
    """`python bench.py --gpus N` as a plain command: spawn one rank process per GPU with the environment
    torch.distributed.run would give them (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free
    MASTER_PORT) and wait.  This parent never imports torch or touches a GPU, and nothing is re-exec'd.
    Rank 0 inherits stdout (its JSON line is the output); the other ranks' stdout goes to stderr.  The exit
    code is non-zero if any rank failed; the remaining ranks are terminated then."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n))  # HSA_ENABLE_IPC_MODE_LEGACY: set by main()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=None if r == 0 else sys.stderr) for r in range(n)]
    rc, alive = 0, set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


class stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block: RCCL prints a version banner ("RCCL version : ...", four lines) to
    stdout when its first communicator comes up, and rank 0's stdout must carry ONE JSON line and nothing else."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def timed_region(step, fence, n_setup, warmup, steps, reset):
    """The contract's bracket: set-up + W untimed warm-up steps, then EXACTLY K steps between two fences
    (device synchronisation + barrier over the ranks)."""
    for _ in range(n_setup):
        step()
    fence()
    reset()
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    return time.perf_counter() - t0


def reduce_over_ranks(dist, world, dt, rays, device):
    """value = rays of ALL ranks / MAX over ranks of the bracket time."""
    if world == 1:
        return dt, rays
    import torch

    tot = torch.tensor([dt, float(rays)], dtype=torch.float64, device=device)
    tmax = tot.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return float(tmax[0]), float(tot[1])


def stub_main(args, rank, world):
    """CPU rehearsal of the launcher and of the N > 1 control flow (tests/test_host.py): ranks meet over gloo, a
    step is a sleep plus the tile gather of a fixed random frame, rank 0 prints the JSON line.  No GPU, no
    kernels, no oracle: the line carries "stub": true and is not a measurement."""
    import torch
    import torch.distributed as dist

    from raytracing_engine_amd import host

    if rank == args.stub_fail_rank:
        raise SystemExit(3)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h = 300, 200
    frame = np.random.default_rng(7).random((h, w, 3), dtype=np.float32)
    tx, ty = -(-w // 64), -(-h // 64)
    per = -(-(tx * ty) // world)
    mine = torch.from_numpy(host.frame_to_tiles(frame, rank, world, per))
    gathered = torch.empty((world, per, 64, 64, 3)) if rank == 0 else None

    def step():
        time.sleep(0.002 * (1 + rank))  # the slowest rank sets the time: max over ranks
        if world > 1:
            host.gather_tiles(mine, gathered, rank, dist)
        else:
            gathered[0] = mine

    def fence():
        if world > 1:
            dist.barrier()

    dt = timed_region(step, fence, 1, args.warmup, args.steps, lambda: None)
    rays = float(sum(min(64, w - (t % tx) * 64) * min(64, h - (t // tx) * 64) for t in range(rank, tx * ty, world)))
    dt, rays = reduce_over_ranks(dist, world, dt, rays, "cpu")
    if rank == 0:
        out = host.tiles_to_frame(gathered.numpy().reshape(-1, 64, 64, 3), world, per, w, h)
        print(json.dumps({"metric": "Mrays/s", "value": round(rays * args.steps / dt / 1e6, 3), "unit": "Mrays/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "stub": True,
                          "config": {"workload": "stub: tile gather of a fixed 300x200 frame over gloo", "rays_per_step": int(rays)},
                          "parity": {"ok": bool(np.array_equal(out, frame)), "against": "the frame the tiles were cut from"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def tune_from_env():
    """RT_BENCH_TUNE="tune_no_packet=1,tune_sort_rays=1": tuning knobs of rt_pt_params for A/B profiles (tools/refresh_profiles.sh);
    unset for every measurement that is reported."""
    out = {}
    for item in os.environ.get("RT_BENCH_TUNE", "").split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            out[k.strip()] = int(v)
    return out


# ---- workloads -----------------------------------------------------------------------------
class SpheresWorkload:
    """BASELINE.json configs[1]: the 8-sphere Cornell-style scene, 1920x1080, 4 spp, path A
    (reference-faithful cone marcher + shading; 4 spp = 2x2 stratified full frames, averaged)."""

    name = "spheres8_1080p_4spp"
    default_lanes = 4  # HIP streams of one priority are dealt onto four hardware queues: one lane per queue (a fifth shares one and loses)
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
        return {"kernel": name, "avg_kernel_ms": round(float(ms), 4), "algorithmic_bytes_per_launch": nbytes,
                "algorithmic_definition": "cone level: 8 B per thread (4 B parent read + 4 B store); shade: 4 B depth read per sample + 12 B rgb store per pixel",
                "note": "path A is VALU/sqrt-bound by construction (about 16 B of HBM traffic per pixel against "
                        "thousands of flops); the HBM fraction is reported as the contract asks, not as the limiter: see roofline.valu",
                "all_kernels_ms": {k: round(float(v[0]), 4) for k, v in kernels.items()}}

    def _oracle_frame(self, threads):
        import oracle as O

        sc = O.scene_from_bytes(bytes(self.scene))
        n, acc = 2, None
        for s in range(self.spp):
            i, j = s % n, s // n
            jit = (((np.float32(2 * i + 1) / np.float32(n)) - np.float32(1)) / np.float32(self.width),
                   ((np.float32(2 * j + 1) / np.float32(n)) - np.float32(1)) / np.float32(self.height))
            f = O.render_a(sc, self.width, self.height, rot=self.rot, pos=self.pos, jitter=jit, want_levels=False, threads=threads)
            acc = f["rgb"] if acc is None else acc + f["rgb"]
            self._oracle_shadow_rays = getattr(self, "_oracle_shadow_rays", 0) + f["counters"]["shadow_rays"]
        return acc / np.float32(self.spp)

    def parity(self, frame, rays_per_step):
        """The timed frame against oracle A (whole frame, 4 stratified samples averaged in index order).  Bar: RGB
        max-abs <= 1e-4 (powf in the specular term is the one libm call of the path), ray count equal."""
        self._oracle_shadow_rays = 0
        ref = self._oracle_frame(host_threads())
        err = float(np.abs(frame - ref).max())
        oracle_rays = self.width * self.height * self.spp + self._oracle_shadow_rays
        return {"against": "oracle A, whole frame (parity unpinned by the reference: it ships no fixtures and cannot be built here)",
                "max_abs_err": err, "tolerance": 1e-4, "rows": [0, self.height], "rays_equal_oracle": bool(oracle_rays == rays_per_step),
                "ok": bool(err <= 1e-4 and oracle_rays == rays_per_step)}

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
    dominant_kernel = "pt_trace_packet"

    def __init__(self, R, renderer):
        self.R, self.r = R, renderer
        self.mesh = R.scenes.soup_scene(self.n_tris, seed=1, edge=self.edge)
        self.rot = np.array([0, 0, 0, 1], np.float32)
        self.pos = np.zeros(3, np.float32)
        renderer.set_mesh(*self.mesh)
        renderer.resize(self.width, self.height)
        self.params = renderer.pt_params(spp=self.spp, bounces=self.bounces, seed=self.seed, sky=self.sky, **tune_from_env())

    def describe(self):
        st = self.r.pt_stats()
        tune = tune_from_env()
        return {"workload": f"{self.name}: {self.n_tris} random triangles (edge +-{self.edge}) + 1 emissive quad, single-level compressed BVH8 "
                            f"({st['n_nodes']} nodes, depth {st['bvh_depth']}, one-triangle leaves), {self.width}x{self.height}, {self.spp} spp, "
                            f"{self.bounces} bounce + NEE, path B wavefront path tracer (camera rays: packet kernel)"
                            + (f" [RT_BENCH_TUNE: {tune}]" if tune else ""),
                "width": self.width, "height": self.height, "spp": self.spp, "bounces": self.bounces, "tile": 64,
                "bvh_build_ms_host": round(st["bvh_build_ms"], 1),
                "persistent_workgroups_per_cu": int(self.params.tune_blocks_per_cu) or "context default (7 for a whole frame)"}

    def set_share(self, world, n_lanes):
        """A rank's share of the frame and the frame lanes it is rendered on: with a 1/8 share on several lanes the persistent
        traversal launches of the lanes fill the machine TOGETHER, and two workgroups per CU per launch instead of the four
        the context would pick for a lone frame of that size measured best (tools/partition_scaling.py --tune
        tune_blocks_per_cu=2: 1.43 against 1.48 ms per frame with three lanes; profiles/r03_partition_scaling.txt).  For the
        whole frame the same idea (4 instead of 7 workgroups per CU with three lanes) measured 8.91-8.93 against 9.05-9.10 ms in
        some runs and 9.36 in others (same file): not applied."""
        if world >= 8 and n_lanes >= 2 and "tune_blocks_per_cu" not in tune_from_env():
            self.params.tune_blocks_per_cu = 2

    def step(self, out_ptr, tile_major):
        self.r.render_pt_device(self.rot, self.pos, self.params, out_ptr, tile_major)

    def rays_per_step(self):
        self.r.render_pt(self.rot, self.pos, params=self.params)
        st = self.r.pt_stats()
        return st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]

    golden_counts = "path_b_tri1m_counts.json"  # oracle B's whole-workload counts (tests/golden/make_golden_counts.py)

    def _oracle(self):
        if getattr(self, "_osc", None) is None:
            import oracle as O

            self._osc = O.TriScene(*self.mesh)
        return self._osc

    def _pinned(self):
        path = os.path.join(ROOT, "tests", "golden", self.golden_counts)
        if self.name == "tri1m_1080p_4spp" and os.path.exists(path):
            return json.load(open(path)).get(self.name)
        return None

    def roofline(self):
        """The step's three traversal kernels in the default schedule, timed with HIP events around every launch
        (profile_stages): pt_trace_packet (camera rays, wave-uniform), pt_trace_fused (closest-hit rays of depth 1 + the
        shadow rays of depth 0 in one persistent launch) and the last pt_trace<any>.  The dominant one - the fused launch -
        carries the roofline.  Algorithmic bytes: every BVH8 node record fetched = 80 B, every
        triangle record = 48 B, per ray its state (44 B closest: 32 B ray, 8 B hit, 4 B queue entry; 48 B shadow); the
        per-lane kernels fetch a record once per ray, the packet kernel once per wave of 64 camera rays.  The counts come
        from the kernels' COUNT instantiations; that they are what a walk of the same tree produces is checked per ray
        by tests/test_gpu_path_b.py (host walker tests/native/bvh8_walk.cpp).  Beside them: oracle B's counts on its own
        BVH2 with SURVEY.md section 8d's per-ray formula."""
        r = self.r
        tune = tune_from_env()
        prm = r.pt_params(spp=self.spp, bounces=self.bounces, seed=self.seed, sky=self.sky, count_traversal=True, **tune)
        r.render_pt(self.rot, self.pos, params=prm)
        ct = r.pt_stats()
        cfg = r.default_config()
        cfg.profile_stages = 1
        r.set_config(cfg)
        reps = 5
        keys = ("ms_generate", "ms_trace_packet", "ms_trace_closest", "ms_trace_fused", "ms_shade", "ms_trace_shadow", "ms_resolve", "ms_total")
        acc = {k: 0.0 for k in keys}
        for _ in range(reps):
            r.render_pt(self.rot, self.pos, params=self.params)
            st = r.pt_stats()
            for k in acc:
                acc[k] += st[k] / reps
        cfg.profile_stages = 0
        r.set_config(cfg)
        all_rays = ct["camera_rays"] + ct["bounce_rays"] + ct["shadow_rays"]
        fused_on = st["launches_trace_fused"] > 0
        # bytes each kernel moves through the cache hierarchy
        b_packet = ct["packet_nodes_fetched"] * 80.0 + ct["packet_tris_fetched"] * 48.0 + ct["camera_rays"] * 24.0  # 16 B direction written + 8 B hit
        b_closest = ct["nodes_visited"] * 80.0 + ct["tris_tested"] * 48.0 + (ct["bounce_rays"] if ct["packets"] else ct["bounce_rays"] + ct["camera_rays"]) * 44.0
        b_shadow_fused = ct["fused_shadow_nodes"] * 80.0 + ct["fused_shadow_tris"] * 48.0 + ct["fused_shadow_rays"] * 48.0
        b_shadow_alone = (ct["shadow_nodes_visited"] - ct["fused_shadow_nodes"]) * 80.0 + (ct["shadow_tris_tested"] - ct["fused_shadow_tris"]) * 48.0 \
            + (ct["shadow_rays"] - ct["fused_shadow_rays"]) * 48.0
        kernels = {}
        if ct["packets"]:
            kernels["pt_trace_packet"] = {"ms": acc["ms_trace_packet"], "launches": 1, "bytes": b_packet, "rays": ct["camera_rays"]}
        if fused_on:
            kernels["pt_trace_fused"] = {"ms": acc["ms_trace_fused"], "launches": st["launches_trace_fused"], "bytes": b_closest + b_shadow_fused,
                                         "rays": ct["bounce_rays"] + ct["fused_shadow_rays"]}
        else:
            kernels["pt_trace<closest>"] = {"ms": acc["ms_trace_closest"], "launches": st["launches_trace_closest"] - (1 if ct["packets"] else 0), "bytes": b_closest,
                                            "rays": ct["bounce_rays"] if ct["packets"] else ct["bounce_rays"] + ct["camera_rays"]}
        kernels["pt_trace<any>"] = {"ms": acc["ms_trace_shadow"], "launches": st["launches_trace_shadow"] - st["launches_trace_fused"], "bytes": b_shadow_alone,
                                    "rays": ct["shadow_rays"] - ct["fused_shadow_rays"]}
        name = max(kernels, key=lambda k: kernels[k]["ms"])
        dom = kernels[name]
        self.dominant_kernel = {"pt_trace_fused": "pt_trace_fused", "pt_trace_packet": "pt_trace_packet", "pt_trace<closest>": "pt_trace<false, false>",
                                "pt_trace<any>": "pt_trace<true, false>"}[name]
        n_launch = max(dom["launches"], 1)
        ms = dom["ms"] / n_launch
        l1_tbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e12
        lane_nodes = ct["nodes_visited"] + ct["shadow_nodes_visited"]
        lane_rays = ct["bounce_rays"] + ct["shadow_rays"] + (0 if ct["packets"] else ct["camera_rays"])
        out = {"kernel": name, "avg_kernel_ms": round(ms, 4), "launches_per_step": n_launch,
               "algorithmic_bytes_per_launch": dom["bytes"] / n_launch,
               "algorithmic_definition": "80 B per BVH8 node record fetched + 48 B per triangle record fetched + per-ray state (44 B closest-hit: 32 B ray, 8 B hit, "
                                         "4 B queue entry; 48 B shadow), every fetch charged as if it came from HBM",
               "kernels": {k: {"ms_per_step": round(v["ms"], 4), "launches_per_step": v["launches"], "rays_per_step": int(v["rays"]),
                               "algorithmic_GB_per_step": round(v["bytes"] / 1e9, 3), "algorithmic_GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1),
                               "Mrays_per_s": round(v["rays"] / max(v["ms"], 1e-9) / 1e3, 1)} for k, v in kernels.items()},
               "gather_ceiling": {"achieved_TBps": round(l1_tbs, 2), "l2_resident_TBps": GATHER_CEILING_L2_TBS, "infinity_cache_TBps": GATHER_CEILING_IC_TBS,
                                  "over_l2_resident": round(l1_tbs / GATHER_CEILING_L2_TBS, 3), "over_infinity_cache": round(l1_tbs / GATHER_CEILING_IC_TBS, 3),
                                  "definition": "bytes the dominant kernel moves through the vector L1s (its algorithmic bytes) per second, against the gather "
                                                "throughput tools/l1_gather_bench.hip measures for 80-byte records when the table fits one XCD's L2 and "
                                                "when it is served from the Infinity Cache (profiles/r02_l1_gather_bench.txt); the 20 MB node array + 48 MB "
                                                "of triangles sit between the two"},
               "per_ray": {"per_lane_kernels": {"nodes": round(lane_nodes / max(lane_rays, 1), 2),
                                                "tris": round((ct["tris_tested"] + ct["shadow_tris_tested"]) / max(lane_rays, 1), 2)},
                           "packet_kernel": {"nodes_per_wave": round(ct["packet_nodes_fetched"] / max(ct["packets"], 1), 1),
                                             "tris_per_wave": round(ct["packet_tris_fetched"] / max(ct["packets"], 1), 1), "waves": ct["packets"]},
                           "counted_by": "the kernels' COUNT instantiations on their BVH8; the per-lane step functions are cross-checked per ray against a host "
                                         "walk of the same tree (tests/test_gpu_path_b.py::test_traversal_counts_match_the_host_walk_of_the_same_bvh)"},
               "stage_ms": {k: round(v, 4) for k, v in acc.items()},
               "note": ("the scene (%.0f MB of nodes, triangles and materials) exceeds the 256 MiB Infinity Cache: the fabric-side bytes of `traffic` are HBM bytes"
                        % (self.n_tris * 80e-6 + st["n_nodes"] * 80e-6)) if self.n_tris * 80 + st["n_nodes"] * 80 > (256 << 20) else
                       "the 70 MB scene is cache resident (L1 hit rate 84 %, L2 80 %), so HBM is not what binds these kernels: vector-instruction issue "
                       "does (roofline.valu_issue), with the vector L1's access rate close behind (DESIGN.md section 8)"}
        pin = self._pinned()
        if pin:  # N_node / N_tri counted by oracle B's instrumented traversal (its own BVH2, <= 4-triangle leaves), all rays
            n_rays = pin["camera_rays"] + pin["bounce_rays"] + pin["shadow_rays"]
            nn, nt = pin["nodes_visited"] / n_rays, pin["tris_tested"] / n_rays
            stages = 2.0  # a ray passes a trace stage and a shade stage
            out["oracle_bvh2"] = {"nodes_per_ray": round(nn, 2), "tris_per_ray": round(nt, 2),
                                  "bytes_per_ray": round(32 * nn + 48 * nt + 48 + 16 * stages, 1),
                                  "rays_equal_kernel_counters": bool(n_rays == all_rays),
                                  "definition": "32 N_node + 48 N_tri + 48 + 16 S with oracle B's counts (tests/golden/path_b_tri1m_counts.json); the kernels "
                                                "walk a different, 8-wide tree, so this is context, not the kernel's traffic"}
        return out

    def parity(self, frame, rays_per_step):
        """Bands of the timed frame against oracle B (bit for bit; the RNG is keyed by the global pixel index, so a
        band of the oracle's frame is those rows of the whole frame), and the step's ray count against the oracle's
        count for the whole workload where it has been pinned (tests/golden)."""
        sc = self._oracle()
        threads = host_threads()
        bands = [(0, 4), (self.height // 2 - 4, self.height // 2 + 4), (self.height - 4, self.height)]
        err, exact = 0.0, True
        for a, b in bands:
            ref, _ = sc.render(self.width, self.height, spp=self.spp, bounces=self.bounces, seed=self.seed, sky=self.sky, rot=self.rot, pos=self.pos,
                               rows=(a, b), threads=threads)
            err = max(err, float(np.abs(frame[a:b] - ref).max()))
            exact = exact and bool(np.array_equal(frame[a:b], ref))
        pin = self._pinned()
        rays_ok = None if pin is None else bool(pin["camera_rays"] + pin["bounce_rays"] + pin["shadow_rays"] == rays_per_step)
        return {"against": "oracle B, row bands of the full-size frame (parity unpinned by the reference: it has no triangle path)",
                "max_abs_err": err, "tolerance": 1e-4, "bit_exact": exact, "rows": [list(b) for b in bands], "rays_equal_oracle": rays_ok,
                "ok": bool(err <= 1e-4 and rays_ok is not False)}

    def cpu_baseline(self):
        threads = host_threads()
        sc = self._oracle()
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
        self.params = renderer.pt_params(spp=self.spp, bounces=self.bounces, seed=self.seed, sky=self.sky, **tune_from_env())

    def describe(self):
        d = super().describe()
        d["workload"] = d["workload"].replace(f"random triangles (edge +-{self.edge})", "height-field triangles (708x708 cells)")
        return d


class Tri16mWorkload(TriWorkload):
    """Context workload, not a BASELINE config: the headline soup with 16 M triangles - 320 MB of BVH nodes, 768 MB of triangle
    records, 512 MB of materials - so the scene is six times the 256 MiB Infinity Cache and the L2-miss bytes the counters
    see ARE HBM traffic: the one workload on which "fraction of the HBM roofline" can be judged from counters.  Same
    generator, camera, resolution, spp and bounce count; edges scaled by 16^(-1/3) so the rays meet as much surface."""

    name = "tri16m_1080p_4spp"
    n_tris, edge = 16_000_000, 0.032

    def cpu_baseline(self):  # bounded sample: the oracle's BVH2 over 16 M triangles alone takes a while
        threads = host_threads()
        sc = self._oracle()
        rows = (self.height // 2 - 32, self.height // 2 + 32)
        t0 = time.perf_counter()
        _, ct = sc.render(self.width, self.height, spp=self.spp, bounces=self.bounces, seed=self.seed, sky=self.sky, rot=self.rot, pos=self.pos, rows=rows,
                          threads=threads)
        dt = time.perf_counter() - t0
        rays = ct["camera_rays"] + ct["bounce_rays"] + ct["shadow_rays"]
        return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                "sample": f"rows {rows[0]}..{rows[1]} of the frame at {self.spp} spp, oracle B (OpenMP, {threads} threads), {rays} rays in {dt:.2f} s"}


WORKLOADS = {SpheresWorkload.name: SpheresWorkload, TriWorkload.name: TriWorkload, TerrainWorkload.name: TerrainWorkload, Tri16mWorkload.name: Tri16mWorkload}


def main():
    # dmabuf IPC: the host driver of this pool supports nothing else, and without it RCCL / device-memory sharing across
    # processes fails with "hipIpcGetMemHandle: invalid argument" (task environment notes; exported on the boxes already).
    # Set before anything imports torch or opens the GPU, so both launch modes - this file's own launcher and
    # torch.distributed.run - run their ranks in the same environment.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # plain `python bench.py --gpus N`: this process only launches the ranks
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.stub_step:
        return stub_main(args, rank, world)

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
        with stdout_to_stderr():
            if args.rehearse_one_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()  # brings the communicator up (and its banner out) here, not inside the first timed gather

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
    n_lanes = 1 if args.rehearse_one_gpu else max(1, min(args.frames_in_flight or getattr(wl, "default_lanes", 3), 8))

    class Lane:
        pass

    lanes = []
    for li in range(n_lanes):
        ln = Lane()
        ln.r = r if li == 0 else R.Renderer(local_rank)
        ln.wl = wl if li == 0 else WORKLOADS[args.workload](R, ln.r)
        ln.r.set_partition(rank, world)
        if hasattr(ln.wl, "set_share"):
            ln.wl.set_share(world, n_lanes)
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
    dt = timed_region(step, fence, n_lanes, args.warmup, args.steps, lambda: state.update(i=0))

    rays_rank = wl.rays_per_step()
    dt, rays = reduce_over_ranks(dist, world, dt, rays_rank, "cpu" if args.rehearse_one_gpu else dev)

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
        timed_frame = None if args.no_parity or args.rehearse_one_gpu else lanes[0].frame.cpu().numpy()
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
        if not args.no_parity and not args.rehearse_one_gpu:
            # parity gate, run with every measurement: the last frame lane 0 rendered inside the
            # timed region (for N > 1 the gathered, de-tiled frame on rank 0) against the oracle, same process
            out["parity"] = wl.parity(timed_frame, int(rays))
        if not multi:
            # the regime the roofline block is measured in, next to the timed one: the same K steps on lane 0 alone, one frame after the
            # other (per-kernel HIP-event durations in `roofline` are those of a frame that has the GPU to itself; with several lanes the
            # kernels of different frames run side by side and stretch each other while the frame rate goes up)
            if n_lanes > 1:
                r.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    wl.step(frame.data_ptr(), False)
                r.synchronize()
                one = (time.perf_counter() - t0) / args.steps
                out["one_frame_at_a_time"] = {"ms_per_step": round(one * 1e3, 4), "value": round(rays / one / 1e6, 3),
                                              "note": "lane 0 alone after the timed region; `roofline.avg_kernel_ms` belongs to this regime"}
            rl = wl.roofline()
            tr = None if args.no_traffic else measure_traffic(args.workload, wl.dominant_kernel)
            out["roofline"] = finish_roofline(rl, tr)
            if not args.no_traffic and isinstance(wl, TriWorkload):  # the bound the per-lane traversal kernels actually run into
                out["roofline"]["valu_issue"] = measure_valu_issue(args.workload, wl.dominant_kernel)
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
