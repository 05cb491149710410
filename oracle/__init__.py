"""ctypes view of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (raytracing_engine_amd) never imports this module.
"parity unpinned by the reference": see oracle/oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
# ORACLE_SO: another build of the same sources, e.g. `make -C oracle asan` -> _build/liboracle_asan.so (tests/test_oracle_sanitized.py)
_SO_OVERRIDE = os.environ.get("ORACLE_SO")


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile). Building the checker is not using it."""
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in os.listdir(_HERE)
        if f.endswith((".c", ".h"))
    ):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _SO


class Material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("diffuse", C.c_float), ("specular", C.c_float),
                ("shine", C.c_float), ("ambient", C.c_float), ("pad", C.c_uint32)]


class Object(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("size", C.c_float)]


class Light(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("pad0", C.c_uint32), ("color", C.c_float * 3), ("pad1", C.c_uint32)]


class Scene(C.Structure):
    _fields_ = [("matCount", C.c_uint32), ("objCount", C.c_uint32), ("lightCount", C.c_uint32), ("pad", C.c_uint32),
                ("mats", Material * 8), ("objs", Object * 8), ("lights", Light * 8)]


class Config(C.Structure):
    _fields_ = [("render_dist", C.c_float), ("cam_fall_off", C.c_float), ("light_fall_off", C.c_float),
                ("ray_radius", C.c_float), ("max_steps", C.c_uint32), ("march_algorithm", C.c_uint32), ("repeat", C.c_float * 3),
                ("reflections", C.c_uint32), ("reflectivity", C.c_float),
                ("transmissions", C.c_uint32), ("transparency", C.c_float), ("refraction_index", C.c_float)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("cone_threads", "cone_steps", "cone_sdf", "hit_pixels",
                                          "shadow_rays", "shadow_steps", "shadow_sdf", "reflection_rays", "transmission_rays")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _SO_OVERRIDE:
            build()
        L = C.CDLL(_SO_OVERRIDE or _SO)
        fp = C.POINTER(C.c_float)
        L.ora_default_config.argtypes = [C.POINTER(Config)]
        L.ora_default_scene.argtypes = [C.POINTER(Scene)]
        L.ora_level_count.argtypes = [C.c_uint32]
        L.ora_level_count.restype = C.c_uint32
        L.ora_level_dims.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)] * 2
        L.ora_camera_quat.argtypes = [C.c_float, C.c_float, fp]
        L.ora_render_a.argtypes = [C.POINTER(Scene), C.POINTER(Config), C.c_uint32, C.c_uint32, fp, fp, fp, fp,
                                   C.POINTER(fp), fp, C.POINTER(Counters), C.c_int]
        L.ora_render_a.restype = C.c_int
        for name in ("ora_trace_bruteforce", "ora_trace_cone", "ora_shadow_ray"):
            f = getattr(L, name)
            f.argtypes = [C.POINTER(Scene), C.POINTER(Config), fp, fp, C.c_float]
            f.restype = C.c_float
        L.ora_rotate.argtypes = [fp, fp, fp]
        L.ora_to_unorm8.argtypes = [fp, C.c_uint64, C.POINTER(C.c_uint8)]
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def default_config():
    c = Config()
    lib().ora_default_config(C.byref(c))
    return c


def default_scene():
    s = Scene()
    lib().ora_default_scene(C.byref(s))
    return s


def scene_from_bytes(raw):
    assert len(raw) == C.sizeof(Scene) == 656
    return Scene.from_buffer_copy(raw)


def level_count(width):
    return int(lib().ora_level_count(width))


def level_dims(width, height, count, level):
    w, h = C.c_uint32(), C.c_uint32()
    lib().ora_level_dims(width, height, count, level, C.byref(w), C.byref(h))
    return int(w.value), int(h.value)


def camera_quat(yaw, pitch):
    out = np.zeros(4, np.float32)
    lib().ora_camera_quat(yaw, pitch, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def rotate(q, v):
    q_, qp = _f(q)
    v_, vp = _f(v)
    out = np.zeros(3, np.float32)
    lib().ora_rotate(qp, vp, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def render_a(scene, width, height, rot=(0, 0, 0, 1), pos=(0, 0, 0), ratio=None, cfg=None, jitter=(0, 0),
             want_levels=True, want_rgb=True, threads=0):
    """One path-A frame. Returns dict(levels=[...], rgb=HxWx3, counters={...})."""
    L = lib()
    cfg = cfg or default_config()
    if ratio is None:
        ratio = (1.0, np.float32(1.0) * np.float32(height) / np.float32(width))  # src/main.rs:610
    ratio_, ratio_p = _f(ratio)
    rot_, rot_p = _f(rot)
    pos_, pos_p = _f(pos)
    jit_, jit_p = _f(jitter)
    count = level_count(width)
    fp = C.POINTER(C.c_float)
    levels = []
    lv_ptrs = (fp * 9)()
    if want_levels:
        for i in range(count):
            w, h = level_dims(width, height, count, i)
            a = np.zeros((h, w), np.float32)
            levels.append(a)
            lv_ptrs[i] = a.ctypes.data_as(fp)
    rgb = np.zeros((height, width, 3), np.float32) if want_rgb else None
    ct = Counters()
    rc = L.ora_render_a(C.byref(scene), C.byref(cfg), width, height, ratio_p, rot_p, pos_p, jit_p,
                        lv_ptrs if want_levels else None, rgb.ctypes.data_as(fp) if want_rgb else None,
                        C.byref(ct), threads)
    if rc != 0:
        raise RuntimeError(f"ora_render_a failed: {rc}")
    return {"levels": levels, "rgb": rgb, "counters": ct.as_dict()}


def trace_cone(scene, origin, direction, threshold, cfg=None, brute=False):
    cfg = cfg or default_config()
    o_, op = _f(origin)
    d_, dp = _f(direction)
    fn = lib().ora_trace_bruteforce if brute else lib().ora_trace_cone
    return float(fn(C.byref(scene), C.byref(cfg), op, dp, threshold))


def shadow_ray(scene, origin, direction, end, cfg=None):
    cfg = cfg or default_config()
    o_, op = _f(origin)
    d_, dp = _f(direction)
    return float(lib().ora_shadow_ray(C.byref(scene), C.byref(cfg), op, dp, end))


def to_unorm8(rgb):
    rgb_, p = _f(rgb.reshape(-1, 3))
    out = np.zeros((rgb_.shape[0], 4), np.uint8)
    lib().ora_to_unorm8(p, rgb_.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.reshape(rgb.shape[:-1] + (4,))


# ---- oracle B (triangles + BVH + path tracing; no reference counterpart) ---------------------
class _Mesh(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("verts", C.POINTER(C.c_float)), ("albedo", C.POINTER(C.c_float)),
                ("emission", C.POINTER(C.c_float))]


class PtParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("bounces", C.c_uint32),
                ("seed", C.c_uint32), ("ratio", C.c_float * 2), ("rot", C.c_float * 4), ("pos", C.c_float * 3),
                ("sky", C.c_float * 3), ("ray_eps", C.c_float)]


class PtCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("camera_rays", "bounce_rays", "shadow_rays", "nodes_visited", "tris_tested", "n_nodes")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_b_ready = False


def _libb():
    global _b_ready
    L = lib()
    if not _b_ready:
        fp = C.POINTER(C.c_float)
        L.orb_scene_create.argtypes = [C.POINTER(_Mesh)]
        L.orb_scene_create.restype = C.c_void_p
        L.orb_scene_destroy.argtypes = [C.c_void_p]
        L.orb_render.argtypes = [C.c_void_p, C.POINTER(PtParams), fp, C.POINTER(PtCounters), C.c_int, C.c_int]
        L.orb_render.restype = C.c_int
        L.orb_render_rows.argtypes = [C.c_void_p, C.POINTER(PtParams), C.c_uint32, C.c_uint32, fp, C.POINTER(PtCounters), C.c_int, C.c_int]
        L.orb_render_rows.restype = C.c_int
        L.orb_closest_hit.argtypes = [C.c_void_p, fp, fp, fp, C.c_int]
        L.orb_closest_hit.restype = C.c_int32
        L.orb_occluded.argtypes = [C.c_void_p, fp, fp, C.c_int]
        L.orb_occluded.restype = C.c_int
        L.orb_rand.argtypes = [C.c_uint32] * 5
        L.orb_rand.restype = C.c_float
        L.orb_cosine_dir.argtypes = [fp, C.c_float, C.c_float, fp]
        L.orb_sincos_2pi.argtypes = [C.c_float, fp, fp]
        _b_ready = True
    return L


class TriScene:
    """Mesh + the oracle's own BVH (orb_scene)."""

    def __init__(self, verts, albedo, emission):
        self.verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 9)
        self.albedo = np.ascontiguousarray(albedo, np.float32).reshape(-1, 3)
        self.emission = np.ascontiguousarray(emission, np.float32).reshape(-1, 3)
        n = self.verts.shape[0]
        assert self.albedo.shape[0] == n and self.emission.shape[0] == n
        fp = C.POINTER(C.c_float)
        m = _Mesh(n, self.verts.ctypes.data_as(fp), self.albedo.ctypes.data_as(fp), self.emission.ctypes.data_as(fp))
        self._h = _libb().orb_scene_create(C.byref(m))
        if not self._h:
            raise RuntimeError("orb_scene_create failed")

    def __del__(self):
        if getattr(self, "_h", None):
            _libb().orb_scene_destroy(self._h)
            self._h = None

    def render(self, width, height, spp=1, bounces=1, seed=1, rot=(0, 0, 0, 1), pos=(0, 0, 0), ratio=None,
               sky=(0.0, 0.0, 0.0), ray_eps=1e-3, use_bvh=True, threads=0, rows=None):
        p = PtParams()
        p.width, p.height, p.spp, p.bounces, p.seed = width, height, spp, bounces, seed
        if ratio is None:
            ratio = (1.0, np.float32(1.0) * np.float32(height) / np.float32(width))
        p.ratio[:] = [float(np.float32(x)) for x in ratio]
        p.rot[:] = [float(np.float32(x)) for x in rot]
        p.pos[:] = [float(np.float32(x)) for x in pos]
        p.sky[:] = [float(np.float32(x)) for x in sky]
        p.ray_eps = ray_eps
        row0, row1 = rows if rows is not None else (0, height)
        rgb = np.zeros((row1 - row0, width, 3), np.float32)
        ct = PtCounters()
        rc = _libb().orb_render_rows(self._h, C.byref(p), row0, row1, rgb.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ct), int(use_bvh), threads)
        if rc:
            raise RuntimeError(f"orb_render failed: {rc}")
        return rgb, ct.as_dict()

    def closest_hit(self, origin, direction, use_bvh=True):
        o_, op = _f(origin)
        d_, dp = _f(direction)
        t = C.c_float()
        tri = _libb().orb_closest_hit(self._h, op, dp, C.byref(t), int(use_bvh))
        return int(tri), float(t.value)

    def occluded(self, origin, direction, use_bvh=True):
        o_, op = _f(origin)
        d_, dp = _f(direction)
        return bool(_libb().orb_occluded(self._h, op, dp, int(use_bvh)))


def pt_rand(pixel, sample, depth, dim, seed):
    return float(_libb().orb_rand(pixel, sample, depth, dim, seed))


def cosine_dir(n, u1, u2):
    n_, np_ = _f(n)
    out = np.zeros(3, np.float32)
    _libb().orb_cosine_dir(np_, u1, u2, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def sincos_2pi(u):
    s, c = C.c_float(), C.c_float()
    _libb().orb_sincos_2pi(u, C.byref(s), C.byref(c))
    return float(s.value), float(c.value)
