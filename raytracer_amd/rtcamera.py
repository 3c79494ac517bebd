"""Camera -> the three arrays the render hot path consumes.

Host-side mirror of the reference's ``rtcamera.Camera`` (rtcamera.py:11-76): same constructor
arguments, same attribute names, same floating-point evaluation order, so the values handed to
``cpp_render_scene`` are bit-identical (golden values: SURVEY.md section 8(c), tests/test_host_api.py).
"""
if not __package__:  # imported as a TOP-LEVEL module (this directory on sys.path, the reference's own layout):
    import importlib as _il  # become an alias of the package module so that there is one set of classes
    import os as _os
    import sys as _sys
    _root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    if _root not in _sys.path:
        _sys.path.append(_root)
    _sys.modules[__name__] = _il.import_module("raytracer_amd." + __name__)
from math import radians, tan

import numpy as np

_WORLD_UP = (0.0, 1.0, 0.0)


def normalise_vector(vector):
    """vector / sqrt(vector . vector)  (rtcamera.py:7-9)."""
    return vector / np.sqrt(vector.dot(vector))


class Camera:
    """Pinhole camera looking from ``camera_center`` at ``point_camera_target``; vertical field of
    view in degrees.  Defaults: at the origin, looking down -z, viewport one unit away."""

    __slots__ = ("image_width", "image_height", "camera_center", "point_camera_target", "angle_vertical_view",
                 "matrix_camera_to_world", "matrix_world_to_camera", "matrix_pixel_spacing",
                 "viewport_upper_left", "pixel_00_center")

    def __init__(self, image_width, image_height, camera_center=None, point_camera_target=None,
                 angle_vertical_view=90.0):
        self.image_width = image_width
        self.image_height = image_height
        self.camera_center = np.array([0.0, 0.0, 0.0]) if camera_center is None else camera_center
        self.point_camera_target = np.array([0, 0, -1]) if point_camera_target is None else point_camera_target
        self.angle_vertical_view = radians(angle_vertical_view)
        self.create_basis_matrices()

    # -- reference: rtcamera.py:46-55
    def _compute_camera_basis_vectors(self):
        back = self.camera_center - self.point_camera_target  # "forward" points from target to camera
        focal_length = np.sqrt(back.dot(back))
        right = np.cross(np.array(_WORLD_UP), back)
        up = np.cross(back, right)
        return normalise_vector(back), normalise_vector(right), normalise_vector(up), focal_length

    # -- reference: rtcamera.py:33-43
    def create_basis_matrices(self):
        forward, right, up, focal_length = self._compute_camera_basis_vectors()
        m = np.zeros((4, 4))
        m[:, :3] = np.array([right, up, forward, self.camera_center])
        m[3] = np.array([0.0, 0.0, 0.0, 1.0])
        self.matrix_camera_to_world = m
        self.matrix_world_to_camera = np.linalg.inv(m)
        self._create_viewport(forward, right, up, focal_length)

    # -- reference: rtcamera.py:57-73
    def _create_viewport(self, forward, right, up, focal_length):
        viewport_height = 2 * tan(self.angle_vertical_view / 2) * focal_length
        viewport_width = viewport_height * (self.image_width / self.image_height)
        across = viewport_width * right
        down = (-viewport_height) * up
        step_x = across / self.image_width
        step_y = down / self.image_height
        self.matrix_pixel_spacing = np.array([step_x, step_y])
        self.viewport_upper_left = self.camera_center - (focal_length * forward) - across / 2 - down / 2
        self.pixel_00_center = self.viewport_upper_left + 0.5 * (step_x + step_y)

    def add_camera_to_scene(self, scene):
        scene.add_camera(self.camera_center, self.pixel_00_center, self.matrix_pixel_spacing)
