"""raytracing_engine_amd — MI355X (gfx950) replacement for the GPU hot path of
IvoteSligte/raytracing_engine: per-pixel ray/scene intersection and shading, as hand-written HIP
kernels behind the C ABI in include/rt_abi.h.  Importing the package does not touch the GPU;
creating a Renderer does, and fails loudly when librt_amd.so or a GPU is missing (no fallback)."""
from . import scenes  # noqa: F401
from ._lib import LIB_PATH, Config, Light, Material, MutableData, Object, PtParams, PtStats, RtError, Stats, load  # noqa: F401
from .host import (CameraController, Renderer, camera_quat, cornell_scene, default_ratio, default_scene,  # noqa: F401
                   level_count, level_dims, make_scene, tiles_to_frame)
