"""``render_scene``: the Python entry point, mirroring the reference's rtmain.py:6-29."""
if not __package__:  # imported as a TOP-LEVEL module (this directory on sys.path, the reference's own layout):
    import importlib as _il  # become an alias of the package module so that there is one set of classes
    import os as _os
    import sys as _sys
    _root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    if _root not in _sys.path:
        _sys.path.append(_root)
    _sys.modules[__name__] = _il.import_module("raytracer_amd." + __name__)
from pathlib import Path

from raytracer_amd.rtmaincpp import cpp_render_scene  # noqa: E402
from raytracer_amd.rtscene import RenderType, Scene, find_max_displacements  # noqa: E402,F401  (rtmain.py:2)


def render_scene(image_height: int, image_width: int, scene: Scene, antialiasing_samples: int,
                 out_directory_path: Path, render_type=RenderType.DYNAMIC, frames_to_render: int = None):
    """Clip / pad the scene like the reference, then hand it to the GPU renderer.

    DYNAMIC renders ``frames_to_render`` timesteps (default: all); STATIC renders one frame.  Asking for
    more frames than there is data for prints the reference's message and returns None."""
    if frames_to_render is None:
        frames_to_render = 1 if render_type == RenderType.STATIC else scene.timestep_count
    if frames_to_render > scene.timestep_count:
        print("Number of requested frames exceeds the number of timesteps with availabile data.")
        return
    scene.clip_scene(frames_to_render, render_type)
    if render_type == RenderType.DYNAMIC:
        scene.fill_empty_timesteps()
    cpp_render_scene(image_height, image_width, antialiasing_samples, out_directory_path, scene.timestep_count,
                     scene.coords_expanded, scene.face_colors, scene.camera_center, scene.pixel_00_center,
                     scene.matrix_pixel_spacing)
