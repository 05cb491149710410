"""Host-side mirror of the reference's Rust host logic for the hot path (src/main.rs), on top of
the C ABI.  Python is used only for tests/bench plumbing; the native stand-in for the Rust
`main` is host/rt_host.cpp.  Citations are file:line in the reference repository.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import Config, Light, Material, MutableData, Object, PtParams, PtStats, RtError, Stats

# src/main.rs:343-364
SPEED_MOVEMENT = 25.0
SPEED_ROTATION = 1.0
SPEED_MOUSE = 1.0
COMPUTE_IMAGE_COUNT = 9
RENDER_DIST = 1000.0
FOV = 1.0


def level_count(width):
    """src/main.rs:639  (view.x / 8.0).log2() as usize + 1, capped at COMPUTE_IMAGE_COUNT."""
    q = int(width) // 8
    return min((q.bit_length() - 1 if q >= 1 else 0) + 1, COMPUTE_IMAGE_COUNT)


def level_dims(width, height, count, level):
    """src/main.rs:209-213  ceil((1 << i) * res / (4 << count)) * 8."""
    den = 4 << count
    return (-((-(width << level)) // den)) * 8, (-((-(height << level)) // den)) * 8


def default_ratio(width, height):
    """src/main.rs:610  ratio = [FOV, FOV * h / w] in f32."""
    return np.array([FOV, np.float32(FOV) * np.float32(height) / np.float32(width)], np.float32)


def camera_quat(yaw, pitch):
    """Data::rotation, src/main.rs:402-404: Quat::from_rotation_z(-yaw) * Quat::from_rotation_x(pitch),
    glam 0.21.3 semantics, returned as to_array() = [x, y, z, w]."""
    hz, hx = np.float32(-yaw) * np.float32(0.5), np.float32(pitch) * np.float32(0.5)
    zs, zc, xs, xc = np.sin(hz), np.cos(hz), np.sin(hx), np.cos(hx)
    return np.array([zc * xs, zs * xs, zs * xc, zc * xc], np.float32)


def quat_mul_vec3(q, v):
    """glam Quat::mul_vec3 (used by Data::position, src/main.rs:409-411)."""
    q = np.asarray(q, np.float64)
    v = np.asarray(v, np.float64)
    b = q[:3]
    return (2.0 * np.dot(b, v) * b + (q[3] * q[3] - np.dot(b, b)) * v + 2.0 * q[3] * np.cross(b, v)).astype(np.float32)


class CameraController:
    """Data<W> camera state + the per-frame update of src/main.rs:732-775 (without the window)."""

    def __init__(self):
        self.rotation = np.zeros(2, np.float32)  # absolute yaw (x) / pitch (y), :383
        self.pos = np.zeros(3, np.float32)       # push_constants.pos, :626

    def rotate(self, d_yaw, d_pitch):
        self.rotation += np.array([d_yaw, d_pitch], np.float32)
        self.rotation[1] = np.clip(self.rotation[1], -0.5 * math.pi, 0.5 * math.pi)  # :770

    def move_local(self, right, forward, up):
        """Data::position (:406-414): displacement along the rotated RIGHT(+X)/FORWARD(+Y)/UP(+Z) axes."""
        q = self.quat()
        d = (right * quat_mul_vec3(q, (1, 0, 0)) + forward * quat_mul_vec3(q, (0, 1, 0)) + up * quat_mul_vec3(q, (0, 0, 1)))
        self.pos = (self.pos + d).astype(np.float32)  # :773

    def quat(self):
        return camera_quat(self.rotation[0], self.rotation[1])


def make_scene(spheres, materials, lights):
    """spheres: [(x,y,z,r)], materials: [(r,g,b,shine,ambient)] (material i belongs to sphere i,
    fragment.glsl:154), lights: [((x,y,z),(r,g,b))]."""
    s = MutableData()
    s.matCount, s.objCount, s.lightCount = len(materials), len(spheres), len(lights)
    for i, (x, y, z, r) in enumerate(spheres):
        s.objs[i].pos[:] = (x, y, z)
        s.objs[i].size = r
    for i, (r, g, b, shine, ambient) in enumerate(materials):
        s.mats[i].color[:] = (r, g, b)
        s.mats[i].diffuse = s.mats[i].specular = 1.0
        s.mats[i].shine, s.mats[i].ambient = shine, ambient
    for i, (p, c) in enumerate(lights):
        s.lights[i].pos[:] = p
        s.lights[i].color[:] = c
    return s


def default_scene():
    """The reference's start-up scene, src/main.rs:524-591."""
    return make_scene(
        [(5, 5, -1, 3), (5, 4, 10, 6), (-3, 3, -3, 1), (4, -1, 0, 2)],
        [(0.2, 0.2, 1.0, 1, 0.05), (0.1, 1.0, 0.1, 10, 0.05), (1.0, 1.0, 0.1, 1, 0.05), (1.0, 0.1, 0.1, 1, 0.05)],
        [((-1, 0, -3), (0.1, 0.5, 0.6)), ((8, -5, 10), (1.2, 0.2, 0.3))])


def cornell_scene():
    """BASELINE.json configs[0]/[1]: Cornell-box-style room made of the reference's only primitive
    (8 spheres = MAX_OBJECTS) + 1 soft-shadowed point light standing in for the area light
    (SURVEY.md §8d).  5 wall spheres of r=50 enclose x,z in [-6,6], back wall at y=22."""
    spheres = [(-56, 10, 0, 50), (56, 10, 0, 50), (0, 10, -56, 50), (0, 10, 56, 50), (0, 72, 0, 50),
               (-2.5, 14, -4, 2), (2.5, 11, -3, 3), (0, 8, -5, 1)]
    mats = [(0.75, 0.15, 0.15, 1, 0.05), (0.15, 0.75, 0.15, 1, 0.05), (0.75, 0.75, 0.75, 1, 0.05),
            (0.75, 0.75, 0.75, 1, 0.05), (0.75, 0.75, 0.75, 1, 0.05), (0.9, 0.9, 0.2, 10, 0.05),
            (0.2, 0.4, 0.9, 30, 0.05), (0.9, 0.5, 0.2, 4, 0.05)]
    lights = [((0, 10, 5), (1.0, 1.0, 1.0))]
    return make_scene(spheres, mats, lights)


def scene_bytes(scene):
    return bytes(scene)


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Renderer:
    """One rt_ctx: one GPU, one stream, single-threaded (like the reference's one queue)."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        rc = self._lib.rt_create(C.byref(self._ctx), int(device))
        if rc != 0:
            raise RtError(rc, self._lib.rt_last_error(None).decode())
        self.width = self.height = 0

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, self._lib.rt_last_error(self._ctx).decode())

    def close(self):
        if self._ctx:
            self._lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def default_config(self):
        cfg = Config()
        self._lib.rt_default_config(C.byref(cfg))
        return cfg

    def set_config(self, cfg):
        self._check(self._lib.rt_set_config(self._ctx, C.byref(cfg)))

    def set_scene(self, scene):
        raw = bytes(scene) if not isinstance(scene, (bytes, bytearray)) else bytes(scene)
        buf = C.create_string_buffer(raw, len(raw))
        self._check(self._lib.rt_set_scene(self._ctx, C.cast(buf, C.c_void_p), len(raw)))

    def resize(self, width, height, ratio=None):
        r = None
        if ratio is not None:
            ratio = np.ascontiguousarray(ratio, np.float32)
            r = _fptr(ratio)
        self._check(self._lib.rt_resize(self._ctx, width, height, r))
        self.width, self.height = width, height

    def level_info(self):
        count = C.c_uint32()
        dims = ((C.c_uint32 * 2) * _lib.RT_MAX_LEVELS)()
        self._check(self._lib.rt_level_info(self._ctx, C.byref(count), C.byref(dims)))
        return [(int(dims[i][0]), int(dims[i][1])) for i in range(count.value)]

    def set_partition(self, rank, n_ranks):
        self._check(self._lib.rt_set_partition(self._ctx, rank, n_ranks))

    def tile_info(self):
        tx, ty, owned = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self._lib.rt_tile_info(self._ctx, C.byref(tx), C.byref(ty), C.byref(owned)))
        return int(tx.value), int(ty.value), int(owned.value)

    def set_stream(self, stream_ptr):
        self._check(self._lib.rt_set_stream(self._ctx, C.c_void_p(stream_ptr)))

    def render(self, rot=(0, 0, 0, 1), pos=(0, 0, 0), spp=1, want_depth=False):
        """Synchronous frame -> (H,W,3) f32 [, last pyramid level]."""
        rot = np.ascontiguousarray(rot, np.float32)
        pos = np.ascontiguousarray(pos, np.float32)
        rgb = np.empty((self.height, self.width, 3), np.float32)
        if spp == 1:
            depth = None
            if want_depth:
                w, h = self.level_info()[-1]
                depth = np.empty((h, w), np.float32)
            self._check(self._lib.rt_render(self._ctx, _fptr(rot), _fptr(pos), _fptr(rgb), _fptr(depth) if want_depth else None))
            return (rgb, depth) if want_depth else rgb
        self._check(self._lib.rt_render_spp(self._ctx, _fptr(rot), _fptr(pos), spp, _fptr(rgb)))
        return rgb

    def render_device(self, rot, pos, spp, dev_ptr, tile_major=False):
        """Asynchronous: enqueue a frame into a device buffer (e.g. torch tensor .data_ptr())."""
        rot = np.ascontiguousarray(rot, np.float32)
        pos = np.ascontiguousarray(pos, np.float32)
        self._check(self._lib.rt_render_device(self._ctx, _fptr(rot), _fptr(pos), spp, C.c_void_p(dev_ptr), int(tile_major)))

    def detile_device(self, tiles_ptr, n_ranks, tiles_per_rank, rgb_ptr):
        self._check(self._lib.rt_detile_device(self._ctx, C.c_void_p(tiles_ptr), n_ranks, tiles_per_rank, C.c_void_p(rgb_ptr)))

    def synchronize(self):
        self._check(self._lib.rt_synchronize(self._ctx))

    # ---- native RCCL exchange (for hosts without torch.distributed) ---------------------------------
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * 128)()
        rc = _lib.load().rt_comm_unique_id(buf)
        if rc != 0:
            raise RtError(rc, "rt_comm_unique_id failed (librccl.so not loadable?)")
        return bytes(buf)

    def comm_init(self, unique_id, rank, n_ranks):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._lib.rt_comm_init(self._ctx, buf, rank, n_ranks))

    def gather_tiles(self, tiles_ptr, gathered_ptr, tiles_per_rank):
        self._check(self._lib.rt_gather_tiles(self._ctx, C.c_void_p(tiles_ptr), C.c_void_p(gathered_ptr), tiles_per_rank))

    def comm_destroy(self):
        self._check(self._lib.rt_comm_destroy(self._ctx))

    def read_level(self, level):
        w, h = C.c_uint32(), C.c_uint32()
        self._check(self._lib.rt_read_level(self._ctx, level, None, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value), np.float32)
        self._check(self._lib.rt_read_level(self._ctx, level, _fptr(out), C.byref(w), C.byref(h)))
        return out

    def read_rgba8(self):
        out = np.empty((self.height, self.width, 4), np.uint8)
        self._check(self._lib.rt_read_rgba8(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    # ---- frames in flight (src/main.rs:664-667, 882-927: one fence per swapchain image) -----------
    FRAME_F32, FRAME_RGBA8 = 0, 1

    def frames_configure(self, n_slots=3, fmt=0):
        """n_slots swapchain-image-like slots; fmt FRAME_F32 (H,W,3 f32) or FRAME_RGBA8 (H,W,4 u8)."""
        self._check(self._lib.rt_frames_configure(self._ctx, int(n_slots), int(fmt)))
        self._frame_fmt = int(fmt)

    def frame_submit(self, slot, rot=(0, 0, 0, 1), pos=(0, 0, 0), spp=1, pt_params=None):
        """Enqueue a frame into `slot` (waits for the slot's previous frame first) and return at once;
        path A by default, path B when pt_params is given."""
        rot = np.ascontiguousarray(rot, np.float32)
        pos = np.ascontiguousarray(pos, np.float32)
        if pt_params is None:
            self._check(self._lib.rt_frame_submit(self._ctx, int(slot), _fptr(rot), _fptr(pos), int(spp)))
        else:
            self._check(self._lib.rt_frame_submit_pt(self._ctx, int(slot), _fptr(rot), _fptr(pos), C.byref(pt_params)))

    def frame_wait(self, slot, copy=True):
        """Block until the slot's pixels are in host memory -> array (a view of the pinned frame when
        copy=False: valid until the slot is submitted again)."""
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        self._check(self._lib.rt_frame_wait(self._ctx, int(slot), C.byref(ptr), C.byref(nbytes)))
        if self._frame_fmt == self.FRAME_RGBA8:
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(self.height, self.width, 4))
        else:
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(self.height, self.width, 3))
        return a.copy() if copy else a

    def frame_ready(self, slot):
        r = C.c_int()
        self._check(self._lib.rt_frame_poll(self._ctx, int(slot), C.byref(r)))
        return bool(r.value)

    def selftest_math(self):
        """Mismatches of the kernels' shortened sqrt against IEEE sqrt over all 2^32 fp32 inputs (0)."""
        n = C.c_uint64()
        self._check(self._lib.rt_selftest_math(self._ctx, C.byref(n)))
        return int(n.value)

    def stats(self):
        s = Stats()
        self._check(self._lib.rt_get_stats(self._ctx, C.byref(s)))
        return s.as_dict()

    # ---- path B: triangle mesh + BVH + wavefront path tracer (no reference counterpart) ----------
    def set_mesh(self, verts, albedo, emission, bvh_levels=1, blas_chunks=0):
        """Upload a triangle mesh and build its BVH (rt_set_mesh / rt_set_mesh_ex).  bvh_levels=2: a top level over
        blas_chunks (0 = 64) bottom-level chunks, flattened into the same node array; frames are identical."""
        verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 9)
        albedo = np.ascontiguousarray(albedo, np.float32).reshape(-1, 3)
        emission = np.ascontiguousarray(emission, np.float32).reshape(-1, 3)
        if not (len(verts) == len(albedo) == len(emission)):
            raise ValueError("verts/albedo/emission disagree on the triangle count")
        opt = _lib.MeshOptions(bvh_levels, blas_chunks)
        self._check(self._lib.rt_set_mesh_ex(self._ctx, _fptr(verts), _fptr(albedo), _fptr(emission), len(verts), C.byref(opt)))

    def mesh_chunk(self, chunk):
        """Original triangle indices of bottom-level chunk `chunk` of a two-level mesh, in rt_update_mesh_chunk's order."""
        n = C.c_uint32()
        self._check(self._lib.rt_mesh_chunk_info(self._ctx, chunk, C.byref(n), None, 0))
        ids = np.empty(n.value, np.uint32)
        self._check(self._lib.rt_mesh_chunk_info(self._ctx, chunk, None, ids.ctypes.data_as(C.POINTER(C.c_uint32)), n.value))
        return ids

    def update_mesh_chunk(self, chunk, verts):
        """New vertices (count x 9, mesh_chunk order) for one chunk of a two-level mesh: only that chunk is rebuilt."""
        verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 9)
        self._check(self._lib.rt_update_mesh_chunk(self._ctx, chunk, _fptr(verts), len(verts)))  # the C side checks the count against the chunk

    def pt_params(self, spp=4, bounces=1, seed=1, sky=(0.0, 0.0, 0.0), ray_eps=1e-3, count_traversal=False, max_paths=0, tune_refill_min=0,
                  tune_blocks_per_cu=0, tune_lds_stack=0, tune_no_overlap=0, tune_no_packet=0, tune_sort_rays=0, tune_tri_mode=0):
        p = PtParams()
        self._lib.rt_default_pt_params(C.byref(p))
        p.spp, p.bounces, p.seed, p.ray_eps, p.count_traversal, p.max_paths = spp, bounces, seed, ray_eps, int(count_traversal), max_paths
        p.tune_refill_min, p.tune_blocks_per_cu, p.tune_lds_stack = tune_refill_min, tune_blocks_per_cu, tune_lds_stack
        p.tune_no_overlap = tune_no_overlap
        p.tune_no_packet = tune_no_packet
        p.tune_sort_rays = tune_sort_rays
        p.tune_tri_mode = tune_tri_mode
        p.sky[:] = [float(np.float32(x)) for x in sky]
        return p

    def render_pt(self, rot=(0, 0, 0, 1), pos=(0, 0, 0), params=None, **kw):
        """Synchronous path-traced frame -> (H,W,3) f32."""
        params = params or self.pt_params(**kw)
        rot = np.ascontiguousarray(rot, np.float32)
        pos = np.ascontiguousarray(pos, np.float32)
        rgb = np.empty((self.height, self.width, 3), np.float32)
        self._check(self._lib.rt_render_pt(self._ctx, _fptr(rot), _fptr(pos), C.byref(params), _fptr(rgb)))
        return rgb

    def render_pt_device(self, rot, pos, params, dev_ptr, tile_major=False):
        rot = np.ascontiguousarray(rot, np.float32)
        pos = np.ascontiguousarray(pos, np.float32)
        self._check(self._lib.rt_render_pt_device(self._ctx, _fptr(rot), _fptr(pos), C.byref(params), C.c_void_p(dev_ptr), int(tile_major)))

    def pt_stats(self):
        s = PtStats()
        self._check(self._lib.rt_get_pt_stats(self._ctx, C.byref(s)))
        return s.as_dict()

    def trace_rays(self, origins, dirs, any_hit=False, counted=False):
        """Test hook: closest hit (t, original triangle index) or occlusion flags for a ray batch;
        counted=True also returns an (n, 2) array of BVH nodes fetched / triangles tested per ray."""
        origins = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = len(origins)
        t = np.empty(n, np.float32)
        tri = np.empty(n, np.int32)
        counts = np.zeros((n, 2), np.uint32) if counted else None
        self._check(self._lib.rt_trace_rays_counted(self._ctx, _fptr(origins), _fptr(dirs), n, int(any_hit), _fptr(t),
                                                    tri.ctypes.data_as(C.POINTER(C.c_int32)),
                                                    counts.ctypes.data_as(C.POINTER(C.c_uint32)) if counted else None))
        return (t, tri, counts) if counted else (t, tri)


def tile_owner(tile_index, n_ranks):
    """Framebuffer partition rule of the C ABI: tile t (row-major 64x64 tiles) -> rank t % n_ranks."""
    return tile_index % n_ranks


def tiles_to_frame(tiles, n_ranks, tiles_per_rank, width, height):
    """numpy model of rt_detile_device: (n_ranks*tiles_per_rank, 64, 64, 3) rank-major -> (H,W,3)."""
    T = _lib.RT_TILE
    tx, ty = -(-width // T), -(-height // T)
    out = np.zeros((ty * T, tx * T, 3), tiles.dtype)
    for t in range(tx * ty):
        r, k = t % n_ranks, t // n_ranks
        y, x = divmod(t, tx)
        out[y * T:(y + 1) * T, x * T:(x + 1) * T] = tiles[r * tiles_per_rank + k]
    return out[:height, :width]


def frame_to_tiles(frame, rank, n_ranks, tiles_per_rank):
    """numpy model of the tile-major output of rt_render*_device(tile_major=1): the tiles owned by
    `rank` (tile t -> rank t % n_ranks), zero-padded to tiles_per_rank x 64 x 64 x 3."""
    T = _lib.RT_TILE
    h, w = frame.shape[:2]
    tx, ty = -(-w // T), -(-h // T)
    out = np.zeros((tiles_per_rank, T, T, 3), frame.dtype)
    for k, t in enumerate(range(rank, tx * ty, n_ranks)):
        y, x = divmod(t, tx)
        blk = frame[y * T:(y + 1) * T, x * T:(x + 1) * T]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out


def gather_tiles(mine, gathered, rank, dist):
    """The one exchange step of a multi-GPU frame: every rank's tile-major buffer goes
    to rank 0 (torch.distributed gather; backend nccl = RCCL over xGMI on GPUs, gloo in the CPU tests).
    `gathered` is a (world, tiles_per_rank, 64, 64, 3) tensor on rank 0, None elsewhere."""
    dist.gather(mine, list(gathered.unbind(0)) if rank == 0 else None, dst=0)


def owned_tiles(total_tiles, rank, n_ranks):
    """Tiles of a frame of `total_tiles` that belong to `rank` (tile t -> rank t % n_ranks); rt_tile_info's `owned`."""
    return (total_tiles - rank + n_ranks - 1) // n_ranks if total_tiles > rank else 0


def gather_owned_tiles(mine, gathered, rank, world, total_tiles, dist):
    """The exchange as rt_gather_tiles (csrc/rt_abi_comm.hip) does it, with torch.distributed point-to-point
    calls: every rank sends exactly the tiles it owns, the root receives each peer's own count into that
    peer's block of `gathered` and copies its own; pad tiles of an uneven split are neither read nor written."""
    if rank == 0:
        k0 = owned_tiles(total_tiles, 0, world)
        gathered[0, :k0] = mine[:k0]
        for peer in range(1, world):
            k = owned_tiles(total_tiles, peer, world)
            if k:
                dist.recv(gathered[peer, :k], src=peer)
    else:
        k = owned_tiles(total_tiles, rank, world)
        if k:
            dist.send(mine[:k].contiguous(), dst=0)
