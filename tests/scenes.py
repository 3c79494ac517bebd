"""BASELINE.json's configs as test scenes, at the reference's own sizes (400 x 225: rttests.py:31-33; loop order and
file names: cpp/rtmain.cpp:49-64).  Shared by the CPU and GPU suites; nothing here touches the GPU."""
import os

import numpy as np

from raytracer_amd import meshgen
from raytracer_amd import rtsimdataloader as loader

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H, W = 225, 400


def block():
    """configs[0]: simple_block.e (44 boundary triangles, 9 timesteps), 1 camera, 1 spp ("no anti-aliasing": the
    reference still jitters, rtrender.cpp:53)."""
    m = loader.simdata_to_rtmesh(os.path.join(GOLD, "simple_block.e"), scale=100.0,
                                 world_position=np.array([-0.5, -0.75, -2.0]), color_field="disp_x")
    coords = np.ascontiguousarray(m.node_coords_expanded_over_time)
    colors = np.ascontiguousarray(m.face_colors_over_time)
    assert coords.shape[:2] == (9, 44)
    return [coords], [colors], [meshgen.camera_arrays(W, H)], 1


def sphere():
    """configs[1]: sphere_gmsh_vtk_test_2d_manyelems.vtk (4,992 triangles), 1 camera, 50 spp, 1 frame."""
    m = loader.vtk_mesh_to_rtmesh(os.path.join(GOLD, "sphere_gmsh_vtk_test_2d_manyelems.vtk"), scale=2.0,
                                  world_position=np.array([0.0, 0.0, -2.2]))
    m.set_surface(loader.SurfType.FIELD_COLOR, np.array([0.5, 0.5, 0.5]))
    coords = np.ascontiguousarray(m.node_coords_expanded_over_time)
    assert coords.shape[1] == 4992
    return [coords], [np.ascontiguousarray(m.face_colors_over_time)], [meshgen.camera_arrays(W, H)], 50


def sequence(T=100):
    """configs[2]: a T-timestep displacement sequence (a swaying plate, a drifting sphere), 1 camera, 2 spp."""
    c, k = meshgen.plate(24, T=T, sway=0.01)
    c2, k2 = meshgen.uv_sphere(12, 16, radius=1.0, center=(0.6, -0.3, -3.0), T=T)
    c2 = c2 + np.linspace(0.0, 0.8, T)[:, None, None, None]
    return ([np.ascontiguousarray(c), np.ascontiguousarray(c2)], [np.ascontiguousarray(k), np.ascontiguousarray(k2)],
            [meshgen.camera_arrays(W, H)], 2)


def stereo(T=16, seed=16):
    """configs[4]: 2 cameras x 8 meshes x 50 spp, tree rebuilt per timestep; seed 16 places the plates so that the
    TLAS leaves do not appear in range order at any timestep (quirk 1 active: rtbvh.cpp:552-560 vs
    rtrayintersection.cpp:214-218) while the pair still sees most plates."""
    rng = np.random.default_rng(seed)
    coords, colors = [], []
    for _ in range(8):
        x0, y0 = rng.uniform(-6, 3), rng.uniform(-4, 1)
        c, k = meshgen.plate(12, side=3.0, x0=x0, y0=y0, z0=-6.0 + rng.uniform(-1, 1), T=T, sway=0.02)
        coords.append(np.ascontiguousarray(c))
        colors.append(np.ascontiguousarray(k))
    cams = [meshgen.camera_arrays(W, H, center=(cx, 0.0, 0.0), target=(-1.0, -1.0, -6.0), fov=75) for cx in (-0.4, 0.4)]
    return coords, colors, cams, 50


def as_scene(coords, colors, cams):
    import raytracer_amd as ra
    sc = ra.Scene()
    T = coords[0].shape[0]
    for c, k in zip(coords, colors):
        sc.add_mesh(c, None, None, k, T)
    for cam in cams:
        sc.add_camera(*cam)
    return sc
