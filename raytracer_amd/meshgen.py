"""Deterministic synthetic scenes for parity tests and the benchmark.

All generators return arrays in the renderer's input layout: coordinates (T, E, 3, 3) f64 and face
colours (T, E, 3) f64, wound counter-clockwise as seen from +z (the reference culls back faces,
cpp/rtrayintersection.cpp:84).  No wall-clock randomness: integer seeds only.
"""
import numpy as np


def plate(n, side=10.0, x0=None, y0=None, z0=None, amp=0.05, T=1, sway=0.0):
    """n x n quads on [x0,x0+side] x [y0,y0+side], z = z0 + amp*side*sin(7x/side)*cos(5y/side); each quad
    -> triangles (0,1,2),(0,2,3).  Face colour g = ((31 i + 17 j) mod 97)/96 for quad (i,j).
    T > 1 adds a per-timestep x-displacement ``sway * t * sin(3 y / side)``; colours for t >= 1 are the
    per-timestep min-max normalised face-mean x-displacement (t = 0 -> 0.5), like the reference's loader
    (rtsimdataloader.py:29-35,64)."""
    x0 = -side / 2 if x0 is None else x0
    y0 = -side / 2 if y0 is None else y0
    z0 = -1.05 * (side / 2) if z0 is None else z0
    xs = x0 + side * np.arange(n + 1) / n
    ys = y0 + side * np.arange(n + 1) / n
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    Z = z0 + amp * side * np.sin(7 * X / side) * np.cos(5 * Y / side)
    P = np.stack([X, Y, Z], axis=-1)  # (n+1, n+1, 3)
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    v = [(ii, jj), (ii + 1, jj), (ii + 1, jj + 1), (ii, jj + 1)]  # CCW seen from +z
    quad_idx = [a * (n + 1) + b for a, b in v]
    tri_nodes = np.stack([np.stack([quad_idx[0], quad_idx[1], quad_idx[2]], -1),
                          np.stack([quad_idx[0], quad_idx[2], quad_idx[3]], -1)], axis=2)  # (n,n,2,3)
    conn = tri_nodes.reshape(-1, 3)
    g = ((31 * ii + 17 * jj) % 97) / 96.0
    base_col = np.repeat(g.reshape(-1), 2)
    nodes0 = P.reshape(-1, 3)
    coords = np.empty((T, conn.shape[0], 3, 3))
    colors = np.empty((T, conn.shape[0], 3))
    for t in range(T):
        nodes = nodes0.copy()
        if t > 0 and sway != 0.0:
            disp = sway * t * np.sin(3 * nodes0[:, 1] / side)
            nodes[:, 0] += disp
            norm = (disp - disp.min()) / (disp.max() - disp.min())
            col = norm[conn].mean(axis=1)
        elif T > 1:
            col = np.full(conn.shape[0], 0.5)
        else:
            col = base_col
        coords[t] = nodes[conn]
        colors[t] = col[:, None]
    return coords, colors


def plate_grid(n_per_side, grid=(2, 2), side=10.0, gap=0.0, z0=None, T=1):
    """Several plates (separate BLASes) tiled under one TLAS; together they span
    [-(gx*side)/2, +(gx*side)/2] x [-(gy*side)/2, ...]."""
    gx, gy = grid
    z0 = -1.05 * (max(gx, gy) * side / 2) if z0 is None else z0
    meshes = []
    for a in range(gx):
        for b in range(gy):
            x0 = -gx * side / 2 + a * (side + gap)
            y0 = -gy * side / 2 + b * (side + gap)
            meshes.append(plate(n_per_side, side=side, x0=x0, y0=y0, z0=z0, T=T))
    return [m[0] for m in meshes], [m[1] for m in meshes]


def uv_sphere(n_lat, n_lon, radius=1.0, center=(0.0, 0.0, 0.0), color=0.5, T=1):
    """Closed triangulated sphere, outward-facing CCW triangles, uniform colour (the reference's VTK /
    .vol loaders pre-fill 0.5: rtsimdataloader_new.py:70,86)."""
    cx, cy, cz = center
    tris = []

    def p(i, j):
        th = np.pi * i / n_lat
        ph = 2 * np.pi * (j % n_lon) / n_lon
        return (cx + radius * np.sin(th) * np.cos(ph), cy + radius * np.cos(th), cz + radius * np.sin(th) * np.sin(ph))

    for i in range(n_lat):
        for j in range(n_lon):
            a, b, c, d = p(i, j), p(i + 1, j), p(i + 1, j + 1), p(i, j + 1)
            if i != 0:
                tris.append((a, d, b) if False else (a, b, d))
            if i != n_lat - 1:
                tris.append((b, c, d))
    arr = np.array(tris, dtype=np.float64)  # (E,3,3)
    # make every triangle face outward
    n = np.cross(arr[:, 1] - arr[:, 0], arr[:, 2] - arr[:, 0])
    mid = arr.mean(axis=1) - np.array(center)
    flip = (n * mid).sum(axis=1) < 0
    arr[flip] = arr[flip][:, [0, 2, 1]]
    coords = np.repeat(arr[None], T, axis=0)
    colors = np.full((T, arr.shape[0], 3), float(color))
    return np.ascontiguousarray(coords), colors


def random_soup(n_tris, seed, extent=4.0, size=0.6, z_range=(-6.0, -3.0), T=1):
    """Random triangles facing +z-ish (mixed winding: about half are back faces), random grey."""
    rng = np.random.default_rng(seed)
    c = np.stack([rng.uniform(-extent, extent, n_tris), rng.uniform(-extent, extent, n_tris),
                  rng.uniform(z_range[0], z_range[1], n_tris)], axis=-1)
    off = rng.uniform(-size, size, (n_tris, 3, 3))
    off[:, :, 2] *= 0.3
    arr = c[:, None, :] + off
    coords = np.repeat(arr[None], T, axis=0)
    g = rng.uniform(0, 1, n_tris)
    colors = np.repeat(np.repeat(g[None, :, None], 3, axis=2), T, axis=0)
    return np.ascontiguousarray(coords), np.ascontiguousarray(colors)


def camera_arrays(width, height, center=(0.0, 0.0, 0.0), target=(0.0, 0.0, -1.0), fov=90.0):
    """(camera_center, pixel_00_center, matrix_pixel_spacing) via the Camera mirror."""
    from .rtcamera import Camera
    cam = Camera(width, height, np.array(center, dtype=np.float64), np.array(target, dtype=np.float64), fov)
    return cam.camera_center, cam.pixel_00_center, cam.matrix_pixel_spacing
