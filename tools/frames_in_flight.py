#!/usr/bin/env python3
"""Host-visible frame rate with the pixels in host memory every frame:
synchronous rt_render (render, wait, copy) against the frames-in-flight slots (read-back of frame k
beside the render of frame k+1).   python tools/frames_in_flight.py [--frames 300] [--slots 3]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_engine_amd as R  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--slots", type=int, default=3)
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--spp", type=int, default=1)
a = ap.parse_args()
w, h = (int(v) for v in a.size.split("x"))
r = R.Renderer(0)
r.set_scene(R.cornell_scene())
r.resize(w, h)
cams = [(R.camera_quat(0.002 * k, 0.0), (0.0, 0.01 * k, 0.0)) for k in range(a.frames)]


def fps(fn):
    fn(cams[:10])
    t0 = time.perf_counter()
    fn(cams)
    return len(cams) / (time.perf_counter() - t0)


def sync_loop(cs):
    for rot, pos in cs:
        r.render(rot, pos, spp=a.spp)


def pipelined(fmt):
    r.frames_configure(a.slots, fmt)  # slots (and their render lanes) are set up once, like a swapchain

    def run(cs):
        n = a.slots
        for k, (rot, pos) in enumerate(cs):
            r.frame_submit(k % n, rot, pos, spp=a.spp)
            if k >= n - 1:
                r.frame_wait((k - n + 1) % n, copy=False)
        for k in range(max(0, len(cs) - n + 1), len(cs)):
            r.frame_wait(k % n, copy=False)
    return run


def device_only(cs):  # no read-back at all: the render rate itself
    import torch
    buf = torch.empty(h * w * 3, dtype=torch.float32, device="cuda")
    for rot, pos in cs:
        r.render_device(rot, pos, a.spp, buf.data_ptr(), False)
    r.synchronize()


print(f"path A, {w}x{h}, {a.spp} spp, {a.frames} frames, pixels in host memory every frame")
print(f"  rt_render (synchronous, pageable destination)      {fps(sync_loop):8.1f} frames/s")
print(f"  frames in flight, {a.slots} slots, f32 RGB ({w * h * 12 / 1e6:.1f} MB/frame)   {fps(pipelined(r.FRAME_F32)):8.1f} frames/s")
print(f"  frames in flight, {a.slots} slots, RGBA8 ({w * h * 4 / 1e6:.1f} MB/frame)      {fps(pipelined(r.FRAME_RGBA8)):8.1f} frames/s")
try:
    print(f"  render only (no read-back)                          {fps(device_only):8.1f} frames/s")
except Exception as e:  # torch missing: the other rows stand on their own
    print("  render only: skipped (", e, ")")
