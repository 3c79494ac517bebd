"""raytracer_amd -- MI355X (gfx950) render hot path of AnalogArnold/rayTracer.

Hand-written HIP kernels behind a C ABI (include/rtgpu.h, raytracer_amd/librtgpu.so) plus the
host-side mirror of the reference's Python interface (rtmain / rtscene / rtcamera / rtmaincpp).
Importing the package loads the shared library and fails loudly if it has not been built.
"""
from ._lib import LIB_PATH, RtError, lib  # noqa: F401  (loads librtgpu.so or raises)
from .api import Context, HostScene, tile_cuts_host, mt19937_state_at, mt19937_stream_states, shard_frames, shard_rows, render_frame_banded, write_ppm  # noqa: F401
from .rtcamera import Camera  # noqa: F401
from .rtscene import RenderType, Scene  # noqa: F401
from .rtmain import render_scene  # noqa: F401

__all__ = ["Context", "HostScene", "Camera", "Scene", "RenderType", "render_scene", "RtError", "write_ppm",
           "mt19937_state_at", "mt19937_stream_states", "shard_frames", "shard_rows", "render_frame_banded", "LIB_PATH"]
