"""Host-side mirror of the reference's Python interface: Camera golden values (SURVEY.md section 8(c),
captured from the reference's rtcamera.py), Scene clip / fill semantics (rtscene.py:57-110) and the
render_scene argument plumbing (rtmain.py:6-29).  CPU only: the C++ backend is replaced by a recorder."""
import os
import numpy as np

import raytracer_amd as ra
from raytracer_amd import rtmain
from raytracer_amd.rtscene import RenderType, Scene


def test_camera_golden_rttests_pose():
    cam = ra.Camera(400, 225, np.array([-0.5, 1.1, 1.1]), np.array([0, 0, -1]), 90)
    assert cam.pixel_00_center.tobytes().hex() == "a62b4aa15d680fc098889c1960310140d8cbaceada7b08c0"
    want = np.array([[0.02095042630646232, -0.0, 0.00498819673963388],
                     [-0.00226473404477017, -0.01918847390659814, 0.0095118829880347]])
    assert np.allclose(cam.matrix_pixel_spacing, want, rtol=0, atol=5e-17)
    assert cam.matrix_pixel_spacing.shape == (2, 3) and cam.matrix_pixel_spacing.flags.c_contiguous


def test_camera_golden_default_pose():
    cam = ra.Camera(400, 225)
    assert np.allclose(cam.pixel_00_center, [-1.773333333333333, 0.9955555555555554, -1.0], rtol=0, atol=3e-16)
    assert np.allclose(cam.matrix_pixel_spacing, [[0.00888888888888889, 0, 0], [0, -0.00888888888888889, 0]],
                       rtol=0, atol=2e-17)
    assert np.allclose(cam.matrix_camera_to_world @ cam.matrix_world_to_camera, np.eye(4))


def _scene(T_list, E=3):
    sc = Scene()
    for T in T_list:
        coords = np.arange(T * E * 9, dtype=np.float64).reshape(T, E, 3, 3)
        colors = np.arange(T * E * 3, dtype=np.float64).reshape(T, E, 3) / 100.0
        sc.add_mesh(coords, coords[0].reshape(-1, 3), np.arange(E * 3).reshape(E, 3), colors, T)
    return sc


def test_fill_empty_timesteps_repeats_coords_and_pads_half_grey():
    sc = _scene([3, 1])
    assert sc.timestep_count == 3 and sc.mesh_count == 2
    first = sc.coords_expanded[1][0].copy()
    sc.fill_empty_timesteps()
    assert sc.coords_expanded[1].shape == (3, 3, 3, 3) and sc.coords_expanded[1].flags.c_contiguous
    assert all(np.array_equal(sc.coords_expanded[1][t], first) for t in range(3))
    assert sc.face_colors[1].shape == (3, 3, 3)
    assert np.all(sc.face_colors[1][1:] == 0.5)


def test_clip_dynamic_and_static():
    sc = _scene([4, 4])
    sc.clip_scene(2, RenderType.DYNAMIC)
    assert sc.timestep_count == 2 and sc.coords_expanded[0].shape[0] == 2
    sc = _scene([4, 4])
    sc.clip_scene(2, RenderType.STATIC)
    # the reference keeps [:2] and sets timestep_count = 1, so timestep 0 is what gets rendered
    assert sc.timestep_count == 1 and sc.coords_expanded[0].shape[0] == 2


def test_render_scene_plumbing(monkeypatch, tmp_path, capsys):
    calls = []
    monkeypatch.setattr(rtmain, "cpp_render_scene", lambda *a: calls.append(a))
    sc = _scene([3, 1])
    ra.Camera(8, 4).add_camera_to_scene(sc)
    rtmain.render_scene(4, 8, sc, 2, tmp_path)
    (H, W, spp, out, T, coords, colors, cams, p00s, sps), = calls
    assert (H, W, spp, T) == (4, 8, 2, 3) and out == tmp_path
    assert [c.shape for c in coords] == [(3, 3, 3, 3), (3, 3, 3, 3)]
    assert [c.shape for c in colors] == [(3, 3, 3), (3, 3, 3)]
    assert len(cams) == len(p00s) == len(sps) == 1 and sps[0].shape == (2, 3)
    # too many frames: message + None, nothing rendered
    calls.clear()
    assert rtmain.render_scene(4, 8, _scene([2]), 1, tmp_path, frames_to_render=5) is None
    assert not calls and "exceeds" in capsys.readouterr().out
    # STATIC default renders one frame
    sc = _scene([3])
    ra.Camera(8, 4).add_camera_to_scene(sc)
    rtmain.render_scene(4, 8, sc, 1, tmp_path, render_type=RenderType.STATIC)
    assert calls[-1][4] == 1


def _threads_in_child(env_extra):
    """rt_host_threads() of a fresh process (the pool size is fixed at first use) under the given environment."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RT_HOST_THREADS", "OMP_NUM_THREADS", "LOCAL_WORLD_SIZE")}
    env.update(env_extra)
    code = f"import sys; sys.path.insert(0, {root!r}); from raytracer_amd import _lib as L; print(L.lib.rt_host_threads())"
    return int(subprocess.check_output([sys.executable, "-c", code], env=env, timeout=120).decode().split()[-1])


def test_host_pool_does_not_inherit_torchruns_omp_num_threads():
    """torchrun exports OMP_NUM_THREADS=1 to multi-rank jobs; the per-timestep host BVH build decides the
    multi-GPU frame rate (hard part 3), so a rank takes affinity cores / LOCAL_WORLD_SIZE instead."""
    cores = len(os.sched_getaffinity(0))
    assert _threads_in_child({"OMP_NUM_THREADS": "1"}) == max(1, min(cores, 16))
    assert _threads_in_child({"OMP_NUM_THREADS": "1", "LOCAL_WORLD_SIZE": "2"}) == max(1, min(cores // 2, 16))
    assert _threads_in_child({"OMP_NUM_THREADS": "1", "LOCAL_WORLD_SIZE": "8"}) == max(1, min(cores // 8, 16))
    if cores >= 4:
        assert _threads_in_child({"OMP_NUM_THREADS": "1", "LOCAL_WORLD_SIZE": "2"}) > 1
    if cores >= 16:  # a many-core box: 8 ranks under torchrun still build on more than one thread each
        assert _threads_in_child({"OMP_NUM_THREADS": "1", "LOCAL_WORLD_SIZE": "8"}) > 1
    assert _threads_in_child({"RT_HOST_THREADS": "3", "OMP_NUM_THREADS": "1", "LOCAL_WORLD_SIZE": "8"}) == 3
