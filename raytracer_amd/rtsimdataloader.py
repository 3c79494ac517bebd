"""Mesh input for the two sample files the reference ships, without pyvale / pyvista.

The reference's loaders (rtsimdataloader.py:43-83, rtsimdataloader_new.py:117-180,315-374) go through
pyvale's Exodus reader / RenderMesh and pyvista.  Neither is installable here, so these are independent
readers that produce the same *shapes and conventions* the renderer consumes -- coordinates
(T, E, 3, 3) f64, face colours (T, E, 3) f64:
  * Exodus (NetCDF classic, via scipy.io.netcdf_file): TET4 blocks -> boundary faces (faces owned by exactly
    one element), wound outward; `scale` multiplies coordinates and displacements; deformed coordinates =
    coords + disp(t); face colour = mean over the face's nodes of the min-max normalised nodal x-displacement
    of that timestep, 0.5 at t = 0 (rtsimdataloader.py:29-35,64,74-75).
  * legacy ASCII VTK unstructured grid: the type-5 (triangle) cells, points * scale, uniform colour 0.5
    (rtsimdataloader_new.py:70,86,335-372).
PARITY: unpinned against pyvale/pyvista (triangle ORDER and winding choices inside those libraries are not
observable here); what the reference's own files pin is checked in tests/test_loaders.py -- 44 surface
triangles / (132, 3) expanded nodes for simple_block.e (rttests.py:154,170), 4992 triangles for the sphere.
"""
if not __package__:  # imported as a TOP-LEVEL module (this directory on sys.path, the reference's own layout):
    import importlib as _il  # become an alias of the package module so that there is one set of classes
    import os as _os
    import sys as _sys
    _root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    if _root not in _sys.path:
        _sys.path.append(_root)
    _sys.modules[__name__] = _il.import_module("raytracer_amd." + __name__)
from dataclasses import dataclass
from enum import Enum, IntEnum
from pathlib import Path

import numpy as np

COORDS_PER_NODE = 3
RGB_VALS = 3


class SurfType(str, Enum):
    FIELD_COLOR = "field_color"
    TEXTURE = "texture"

    def __str__(self):
        return self.value


class ElementNodeCount(IntEnum):
    TRI3 = 3
    QUAD4 = 4
    TET10 = 10
    TET14 = 14
    HEX8 = 8
    HEX20 = 20
    HEX27 = 27


@dataclass
class RTMesh:
    """Common mesh interface (reference: rtsimdataloader_new.py:41-56)."""
    node_coords: np.ndarray = None
    connectivity: np.ndarray = None
    node_coords_over_time: np.ndarray = None
    node_coords_expanded_over_time: np.ndarray = None
    face_colors_over_time: np.ndarray = None
    uvs_over_time: np.ndarray = None
    uvs: np.ndarray = None
    texture: np.ndarray = None
    surface_type: SurfType = None
    timestep_count: int = 1
    element_count: int = 0
    node_count: int = 0
    nodes_per_element: int = 3

    def set_surface(self, surface_type=SurfType.FIELD_COLOR, surface_fill=None, uv_coords=None):
        """Field colours: a single RGB (3,) or per timestep (T,E,3); anything else (or nothing) pre-fills mid
        grey 0.5 with the reference's messages (rtsimdataloader_new.py:58-86) -- including, as in the reference,
        a per-element (E,3) fill, whose branch falls through to the grey fill."""
        if self.surface_type is not None and surface_type != self.surface_type:
            self.face_colors_over_time = self.uvs_over_time = self.texture = self.uvs = None
        self.surface_type = surface_type
        shape_te = (self.timestep_count, self.element_count, RGB_VALS)
        if surface_type == SurfType.FIELD_COLOR:
            if surface_fill is None:
                print("No colour data passed. Pre-filling automatically with grey.")
            elif surface_fill.shape == (RGB_VALS,):
                self.face_colors_over_time = np.ones(shape_te) * surface_fill
                return
            elif surface_fill.shape == (self.element_count, RGB_VALS):
                # bug-compatible: the reference's branch has no `return` (rtsimdataloader_new.py:76-78), so the
                # broadcast colours are overwritten by the grey fill below
                self.face_colors_over_time = np.ascontiguousarray(np.broadcast_to(surface_fill[np.newaxis], shape_te))
            elif surface_fill.shape == shape_te:
                self.face_colors_over_time = surface_fill
                return
            else:
                print("Surface fill must be of shape (3,) or (element_count, 3) or (timestep_count, element_count, 3)."
                      "\nPre-filling automatically with grey.")
            self.face_colors_over_time = np.ones(shape_te, dtype=np.float64) * 0.5
        elif surface_type == SurfType.TEXTURE:
            if uv_coords is None:
                print("UV coordinates are required to append texture.")
                return
            if surface_fill is None or surface_fill.ndim != 2:
                print("Wrong number of dimensions. The array containing the texture should be two-dimensional.")
                return
            self.uvs = uv_coords
            self.uvs_over_time = np.broadcast_to(uv_coords[np.newaxis], (self.timestep_count, self.element_count, 3, 2))
            self.texture = surface_fill


# --------------------------------------------------------------------------------------------------
def orient_mesh_in_world(node_coords, world_position=None, world_rotation=None):
    """node_coords @ R^T + position (rtsimdataloader_new.py:236-251); rotation is a 3x3 matrix or an object with
    ``as_matrix()`` (scipy Rotation)."""
    out = np.asarray(node_coords, dtype=np.float64)
    if world_rotation is not None:
        R = world_rotation.as_matrix() if hasattr(world_rotation, "as_matrix") else np.asarray(world_rotation, float)
        out = out @ R.T
    if world_position is not None:
        out = out + np.asarray(world_position, dtype=np.float64)
    return np.ascontiguousarray(out)


def compute_face_colors_averages(field_nodal_values, connectivity):
    """Min-max normalise a nodal field to [0,1], grey RGB per node, mean over each face's nodes
    (rtsimdataloader.py:29-35)."""
    v = np.asarray(field_nodal_values, dtype=np.float64)
    norm = (v - v.min()) / (v.max() - v.min())
    node_colors = np.column_stack((norm, norm, norm))
    return np.mean(node_colors[connectivity], axis=1)


# ---- Exodus --------------------------------------------------------------------------------------
_TET_FACES = ((0, 1, 3), (1, 2, 3), (0, 3, 2), (0, 2, 1))  # Exodus TET4 side numbering; outward for a positive tet


def _names(var):
    return [b"".join(row).decode("ascii", "ignore").strip().strip("\x00") for row in var.data]


def read_exodus(path):
    """-> dict(coords (N,3), tets (M,4) zero-based, times (T,), nodal {name: (T,N)})."""
    from scipy.io import netcdf_file
    with netcdf_file(str(path), "r", mmap=False) as f:
        v = f.variables
        coords = np.stack([np.array(v["coordx"].data, np.float64), np.array(v["coordy"].data, np.float64),
                           np.array(v["coordz"].data, np.float64) if "coordz" in v else 0.0 * v["coordx"].data], 1)
        tets = []
        blk = 1
        while f"connect{blk}" in v:
            var = v[f"connect{blk}"]
            et = getattr(var, "elem_type", b"").decode().upper()
            if not et.startswith("TET"):
                raise ValueError(f"{path}: element block {blk} is {et or '?'}; only TET4 blocks are supported")
            tets.append(np.array(var.data, np.int64)[:, :4] - 1)
            blk += 1
        if not tets:
            raise ValueError(f"{path}: no element blocks")
        times = np.array(v["time_whole"].data, np.float64) if "time_whole" in v else np.zeros(1)
        nodal = {}
        if "name_nod_var" in v:
            for i, name in enumerate(_names(v["name_nod_var"])):
                nodal[name] = np.array(v[f"vals_nod_var{i + 1}"].data, np.float64)
    return dict(coords=coords, tets=np.concatenate(tets), times=times, nodal=nodal)


def tet_boundary_faces(coords, tets):
    """Triangles that belong to exactly one tetrahedron, wound so the normal points away from the owning
    tetrahedron's fourth node.  Order: by element, then by Exodus side number."""
    faces = tets[:, np.array(_TET_FACES)]                # (M,4,3)
    opposite = tets[:, [2, 0, 1, 3]]                     # node of the tet not on each face
    flat = faces.reshape(-1, 3)
    key = np.sort(flat, axis=1)
    _, inverse, counts = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    keep = counts[inverse.reshape(-1)] == 1
    tri = flat[keep]
    opp = opposite.reshape(-1)[keep]
    a, b, c = coords[tri[:, 0]], coords[tri[:, 1]], coords[tri[:, 2]]
    n = np.cross(b - a, c - a)
    inward = np.einsum("ij,ij->i", n, coords[opp] - a) > 0
    tri[inward] = tri[inward][:, [0, 2, 1]]
    return np.ascontiguousarray(tri)


def simdata_to_rtmesh(pypath: Path, field_components=("disp_x", "disp_y", "disp_z"), fields_to_render=("disp_y", "disp_x"),
                      scale: float = 100.0, world_position=None, world_rotation=None, color_field=None):
    """Exodus file -> RTMesh with the surface triangles over all timesteps.  Like the reference
    (rtsimdataloader_new.py:117-180) it leaves ``surface_type`` unset and fills no colours -- call
    ``set_surface`` next.  ``color_field="disp_x"`` (no counterpart in rtsimdataloader_new) additionally fills
    the min-max normalised nodal-field greys of the reference's older loader (rtsimdataloader.py:29-35,64-75)
    and sets FIELD_COLOR."""
    ex = read_exodus(pypath)
    coords = ex["coords"] * scale
    T = len(ex["times"])
    disp = np.zeros((T,) + coords.shape)
    for axis, name in enumerate(field_components[:3]):
        if name in ex["nodal"]:
            disp[:, :, axis] = ex["nodal"][name] * scale
    conn = tet_boundary_faces(coords, ex["tets"])
    mesh = RTMesh()
    mesh.nodes_per_element = 3
    mesh.connectivity = conn
    mesh.timestep_count = T
    mesh.element_count = conn.shape[0]
    mesh.node_count = coords.shape[0]
    over_time = np.empty((T,) + coords.shape)
    expanded = np.empty((T, conn.shape[0], 3, 3))
    colors = np.empty((T, conn.shape[0], 3))
    field = ex["nodal"].get(color_field) if color_field else None
    for t in range(T):
        nodes = orient_mesh_in_world(coords + (disp[t] if t > 0 else 0.0), world_position, world_rotation)
        over_time[t] = nodes
        expanded[t] = nodes[conn]
        if t == 0 or field is None or field[t].max() == field[t].min():
            colors[t] = 0.5
        else:
            colors[t] = compute_face_colors_averages(field[t], conn)
    mesh.node_coords = over_time[0]
    mesh.node_coords_over_time = over_time
    mesh.node_coords_expanded_over_time = expanded
    if color_field:
        mesh.face_colors_over_time = colors
        mesh.surface_type = SurfType.FIELD_COLOR
    return mesh


def add_mesh_to_scene(scene, pypath: Path, field_components=("disp_x", "disp_y", "disp_z"),
                      fields_to_render=("disp_y", "disp_x"), world_position=None, scale: float = 100.0):
    """Exodus file -> Scene.add_mesh(...) with data for all timesteps (reference: rtsimdataloader.py:43-83)."""
    m = simdata_to_rtmesh(pypath, field_components, fields_to_render, scale, world_position, color_field=field_components[0])
    scene.add_mesh(m.node_coords_expanded_over_time, m.node_coords, m.connectivity, m.face_colors_over_time,
                   m.timestep_count)


# ---- legacy VTK ----------------------------------------------------------------------------------
def read_vtk_unstructured_ascii(path):
    """-> (points (N,3) f64, list of (cell_type, node index array))."""
    tokens = Path(path).read_text().split("\n")
    if len(tokens) < 4 or "ASCII" not in tokens[2].upper():
        raise ValueError(f"{path}: only legacy ASCII VTK files are supported")
    words = " ".join(tokens[3:]).split()
    i = 0
    points = cells = types = None
    while i < len(words):
        w = words[i].upper()
        if w == "DATASET":
            if words[i + 1].upper() != "UNSTRUCTURED_GRID":
                raise ValueError(f"{path}: DATASET {words[i + 1]} not supported")
            i += 2
        elif w == "POINTS":
            n = int(words[i + 1])
            points = np.array(words[i + 3:i + 3 + 3 * n], dtype=np.float64).reshape(n, 3)
            i += 3 + 3 * n
        elif w == "CELLS":
            n, size = int(words[i + 1]), int(words[i + 2])
            raw = np.array(words[i + 3:i + 3 + size], dtype=np.int64)
            cells = []
            j = 0
            for _ in range(n):
                k = raw[j]
                cells.append(raw[j + 1:j + 1 + k])
                j += 1 + k
            i += 3 + size
        elif w == "CELL_TYPES":
            n = int(words[i + 1])
            types = np.array(words[i + 2:i + 2 + n], dtype=np.int64)
            i += 2 + n
        else:
            i += 1
    if points is None or cells is None or types is None:
        raise ValueError(f"{path}: POINTS / CELLS / CELL_TYPES missing")
    return points, list(zip(types.tolist(), cells))


def vtk_mesh_to_rtmesh(pypath: Path, scale: float = 100.0, world_position=None, world_rotation=None):
    """Legacy ASCII VTK triangle mesh -> static RTMesh (reference: rtsimdataloader_new.py:315-374)."""
    points, cells = read_vtk_unstructured_ascii(pypath)
    tri = np.array([c for t, c in cells if t == 5], dtype=np.int64)
    if tri.size == 0:
        raise ValueError(f"{pypath}: no triangle (type 5) cells")
    nodes = orient_mesh_in_world(points * scale, world_position, world_rotation)
    mesh = RTMesh()
    mesh.node_coords = nodes
    mesh.connectivity = tri
    mesh.timestep_count = 1
    mesh.element_count = tri.shape[0]
    mesh.node_count = nodes.shape[0]
    mesh.nodes_per_element = 3
    mesh.node_coords_over_time = nodes[None].copy()
    mesh.node_coords_expanded_over_time = np.ascontiguousarray(nodes[tri][None])
    return mesh
