"""ctypes binding of librt_amd.so (include/rt_abi.h).

The library is the product: hand-written HIP kernels for gfx950 behind a C ABI.  There is no
Python/CPU fallback — a missing library or a missing GPU raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB", os.path.join(_HERE, "librt_amd.so"))  # RT_AMD_LIB: A/B builds of the same ABI

RT_OK = 0
RT_MAX_LEVELS = 9
RT_TILE = 64
STATUS_NAMES = {0: "RT_OK", -1: "RT_ERR_INVALID", -2: "RT_ERR_NO_DEVICE", -3: "RT_ERR_HIP", -4: "RT_ERR_STATE",
                -5: "RT_ERR_OOM"}


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {message}")
        self.code = code


class Material(C.Structure):  # shaders/utilities.glsl:8-14, std140, 32 B
    _fields_ = [("color", C.c_float * 3), ("diffuse", C.c_float), ("specular", C.c_float), ("shine", C.c_float),
                ("ambient", C.c_float), ("_pad", C.c_uint32)]


class Object(C.Structure):  # shaders/utilities.glsl:16-19, 16 B
    _fields_ = [("pos", C.c_float * 3), ("size", C.c_float)]


class Light(C.Structure):  # shaders/utilities.glsl:21-24, std140, 32 B
    _fields_ = [("pos", C.c_float * 3), ("_pad0", C.c_uint32), ("color", C.c_float * 3), ("_pad1", C.c_uint32)]


class MutableData(C.Structure):  # shaders/compute.glsl:17-24, 656 B
    _fields_ = [("matCount", C.c_uint32), ("objCount", C.c_uint32), ("lightCount", C.c_uint32), ("_pad", C.c_uint32),
                ("mats", Material * 8), ("objs", Object * 8), ("lights", Light * 8)]


class Config(C.Structure):
    _fields_ = [("render_dist", C.c_float), ("cam_fall_off", C.c_float), ("light_fall_off", C.c_float),
                ("ray_radius", C.c_float), ("max_steps", C.c_uint32), ("profile_stages", C.c_uint32), ("fuse_levels", C.c_uint32),
                ("march_algorithm", C.c_uint32), ("repeat", C.c_float * 3), ("reflections", C.c_uint32), ("reflectivity", C.c_float),
                ("transmissions", C.c_uint32), ("transparency", C.c_float), ("refraction_index", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("level_count", C.c_uint32), ("spp", C.c_uint32),
                ("frames", C.c_uint64), ("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("hit_pixels", C.c_uint64), ("cone_threads", C.c_uint64), ("ms_total", C.c_float),
                ("ms_cone", C.c_float), ("ms_shade", C.c_float), ("ms_level", C.c_float * RT_MAX_LEVELS), ("ms_fused", C.c_float),
                ("reflection_rays", C.c_uint64), ("transmission_rays", C.c_uint64)]

    def as_dict(self):
        d = {}
        for n, _ in self._fields_:
            v = getattr(self, n)
            d[n] = list(v) if hasattr(v, "__len__") else v
        return d


class PtParams(C.Structure):  # rt_pt_params
    _fields_ = [("spp", C.c_uint32), ("bounces", C.c_uint32), ("seed", C.c_uint32), ("sky", C.c_float * 3),
                ("ray_eps", C.c_float), ("count_traversal", C.c_uint32), ("max_paths", C.c_uint32),
                ("tune_refill_min", C.c_uint32), ("tune_blocks_per_cu", C.c_uint32), ("tune_lds_stack", C.c_uint32), ("tune_no_overlap", C.c_uint32),
                ("tune_no_packet", C.c_uint32), ("tune_sort_rays", C.c_uint32), ("tune_tri_mode", C.c_uint32)]


class MeshOptions(C.Structure):  # rt_mesh_options
    _fields_ = [("bvh_levels", C.c_uint32), ("blas_chunks", C.c_uint32)]


class PtStats(C.Structure):  # rt_pt_stats
    _fields_ = [("n_tris", C.c_uint32), ("n_nodes", C.c_uint32), ("bvh_depth", C.c_uint32), ("n_lights", C.c_uint32),
                ("stack_need", C.c_uint32), ("bvh_build_ms", C.c_float), ("stack_overflow", C.c_uint32), ("camera_rays", C.c_uint64),
                ("bounce_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("nodes_visited", C.c_uint64),
                ("tris_tested", C.c_uint64), ("shadow_nodes_visited", C.c_uint64), ("shadow_tris_tested", C.c_uint64),
                ("wave_rounds", C.c_uint64), ("alive_lane_rounds", C.c_uint64),
                ("ms_total", C.c_float), ("ms_generate", C.c_float),
                ("ms_trace_closest", C.c_float), ("ms_shade", C.c_float), ("ms_trace_shadow", C.c_float),
                ("ms_resolve", C.c_float), ("launches_trace_closest", C.c_uint32), ("launches_trace_shadow", C.c_uint32),
                ("packets", C.c_uint64), ("packet_nodes_fetched", C.c_uint64), ("packet_tris_fetched", C.c_uint64), ("fused_shadow_nodes", C.c_uint64),
                ("fused_shadow_tris", C.c_uint64), ("fused_shadow_rays", C.c_uint64), ("ms_trace_packet", C.c_float), ("ms_trace_fused", C.c_float),
                ("launches_trace_fused", C.c_uint32), ("bvh_levels", C.c_uint32), ("blas_chunks", C.c_uint32), ("tlas_nodes", C.c_uint32),
                ("ms_build_blas", C.c_float), ("ms_build_tlas", C.c_float), ("ms_build_flatten", C.c_float),
                ("pool_flushes", C.c_uint64), ("wave_rounds_all", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


assert C.sizeof(MutableData) == 656 and C.sizeof(Material) == 32 and C.sizeof(Object) == 16 and C.sizeof(Light) == 32

# every symbol include/rt_abi.h declares: name -> (restype, argtypes)
_fp = C.POINTER(C.c_float)
_vp = C.c_void_p
_u32p = C.POINTER(C.c_uint32)
PROTOTYPES = {
    "rt_abi_version": (C.c_int, []),
    "rt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rt_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "rt_destroy": (None, [_vp]),
    "rt_last_error": (C.c_char_p, [_vp]),
    "rt_default_config": (C.c_int, [C.POINTER(Config)]),
    "rt_set_config": (C.c_int, [_vp, C.POINTER(Config)]),
    "rt_default_scene": (C.c_int, [C.POINTER(MutableData)]),
    "rt_set_scene": (C.c_int, [_vp, _vp, C.c_size_t]),
    "rt_resize": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _fp]),
    "rt_level_info": (C.c_int, [_vp, _u32p, C.POINTER((C.c_uint32 * 2) * RT_MAX_LEVELS)]),
    "rt_set_partition": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "rt_tile_info": (C.c_int, [_vp, _u32p, _u32p, _u32p]),
    "rt_set_stream": (C.c_int, [_vp, _vp]),
    "rt_render": (C.c_int, [_vp, _fp, _fp, _fp, _fp]),
    "rt_render_spp": (C.c_int, [_vp, _fp, _fp, C.c_uint32, _fp]),
    "rt_render_device": (C.c_int, [_vp, _fp, _fp, C.c_uint32, _vp, C.c_int]),
    "rt_detile_device": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp]),
    "rt_synchronize": (C.c_int, [_vp]),
    "rt_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "rt_comm_init": (C.c_int, [_vp, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]),
    "rt_gather_tiles": (C.c_int, [_vp, _vp, _vp, C.c_uint32]),
    "rt_comm_destroy": (C.c_int, [_vp]),
    "rt_read_level": (C.c_int, [_vp, C.c_uint32, _fp, _u32p, _u32p]),
    "rt_read_rgba8": (C.c_int, [_vp, C.POINTER(C.c_uint8)]),
    "rt_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "rt_default_pt_params": (C.c_int, [C.POINTER(PtParams)]),
    "rt_set_mesh": (C.c_int, [_vp, _fp, _fp, _fp, C.c_uint32]),
    "rt_set_mesh_ex": (C.c_int, [_vp, _fp, _fp, _fp, C.c_uint32, C.POINTER(MeshOptions)]),
    "rt_mesh_chunk_info": (C.c_int, [_vp, C.c_uint32, _u32p, _u32p, C.c_uint32]),
    "rt_update_mesh_chunk": (C.c_int, [_vp, C.c_uint32, _fp, C.c_uint32]),
    "rt_render_pt": (C.c_int, [_vp, _fp, _fp, C.POINTER(PtParams), _fp]),
    "rt_render_pt_device": (C.c_int, [_vp, _fp, _fp, C.POINTER(PtParams), _vp, C.c_int]),
    "rt_get_pt_stats": (C.c_int, [_vp, C.POINTER(PtStats)]),
    "rt_selftest_math": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "rt_frames_configure": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "rt_frame_submit": (C.c_int, [_vp, C.c_uint32, _fp, _fp, C.c_uint32]),
    "rt_frame_submit_pt": (C.c_int, [_vp, C.c_uint32, _fp, _fp, C.POINTER(PtParams)]),
    "rt_frame_wait": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rt_frame_poll": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_int)]),
    "rt_trace_rays": (C.c_int, [_vp, _fp, _fp, C.c_uint32, C.c_int, _fp, C.POINTER(C.c_int32)]),
    "rt_trace_rays_counted": (C.c_int, [_vp, _fp, _fp, C.c_uint32, C.c_int, _fp, C.POINTER(C.c_int32), _u32p]),
}

_lib = None


def _preload_shared_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7).  Two HIP
    runtimes in one process cannot both open the GPU ("No HIP GPUs are available"), so when torch
    is installed its copy is loaded first and librt_amd.so's NEEDED libamdhip64.so.7 resolves to
    it.  A process without torch (host/rt_host.cpp) uses /opt/rocm's runtime via RUNPATH."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def load():
    """Load librt_amd.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C raytracing_engine_amd/csrc). raytracing_engine_amd has no CPU fallback.")
        _preload_shared_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
