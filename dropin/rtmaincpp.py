"""``rtmaincpp`` for the reference's unmodified Python: put ONLY this directory on sys.path.

The reference's rtmain.py:4 does ``from rtmaincpp import cpp_render_scene`` (the nanobind module whose
export is commented out at cpp/rtmain.cpp:157-160).  This module has that name and that function -- the
ten positionals of cpp/rtmain.cpp:29-38 -- and forwards to librtgpu.so (C ABI ``rt_render_scene``).
Nothing else lives in this directory, so ``rtmain`` / ``rtscene`` / ``rtcamera`` keep resolving to the
reference's own files.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)

from raytracer_amd.rtmaincpp import cpp_render_scene  # noqa: E402,F401
