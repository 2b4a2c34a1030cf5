"""Collidable points from collision meshes.

Mirrors the behaviour of the reference's mesh wrapping (``src/jaxsim/parsers/rod/meshes.py:7-104`` and
``create_mesh_collision``, ``parsers/rod/utils.py:228-280``): a mesh collision is replaced by a set of points
taken from the mesh -- by default ALL vertices (``extract_points_vertices``) -- scaled by the ``<mesh scale>``,
moved by the collision pose into the link frame, every point enabled.  Like the reference, meshes are only
processed when ``JAXSIM_COLLISION_MESH_ENABLED`` is set (``parsers/rod/parser.py:333-348``), otherwise the
shape is skipped.

The reference loads meshes with ``trimesh`` (third-party, absent from /root/reference and from this image,
version unpinned).  This module reads Wavefront OBJ and STL (ASCII and binary) itself and, like trimesh's
default ``process=True``, merges duplicate vertices (first occurrence kept, file order preserved -- trimesh's
own ordering of merged vertices is unpinned, so the SET of points is the contract, not their order).

Point selection methods (same names and arguments as ``meshes.py``):

* ``extract_points_vertices(mesh)``
* ``extract_points_select_points_over_axis(mesh, axis, direction, n)`` -- reproduces the reference as written:
  ``arr.sort(axis=0)`` sorts every COLUMN independently (the rows are no longer mesh vertices) before the
  first / last ``n`` rows are taken, and the ``axis`` argument is not used (``meshes.py:45-66``);
* ``extract_points_aap(mesh, axis, upper, lower)`` -- vertices inside an axis-aligned slab;
* ``extract_points_random_surface_sampling(mesh, n)`` / ``extract_points_uniform_surface_sampling(mesh, n)`` --
  area-weighted surface samples; trimesh's random stream cannot be reproduced, a seeded NumPy generator is used
  (``seed`` argument), and the "even" variant rejects samples closer than the radius trimesh uses
  (``sqrt(area / (3 n))``).
"""

from __future__ import annotations

import dataclasses
import os
import pathlib
import struct

import numpy as np

VALID_AXIS = {"x": 0, "y": 1, "z": 2}


@dataclasses.dataclass
class Mesh:
    vertices: np.ndarray  # [V, 3]
    faces: np.ndarray  # [F, 3] int

    @property
    def is_empty(self) -> bool:
        return self.vertices.shape[0] == 0

    def apply_scale(self, scale) -> "Mesh":
        self.vertices = self.vertices * np.asarray(scale, dtype=float).reshape(-1)[:3] if np.size(scale) == 3 else self.vertices * float(np.asarray(scale).reshape(-1)[0])
        return self

    def triangles(self) -> np.ndarray:
        return self.vertices[self.faces]  # [F, 3, 3]

    def area_faces(self) -> np.ndarray:
        t = self.triangles()
        return 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1)


def _merge_vertices(v: np.ndarray, f: np.ndarray) -> Mesh:
    """Merge duplicate vertices (trimesh ``process=True``), keeping the first occurrence and the file order."""
    if v.shape[0] == 0:
        return Mesh(v.reshape(0, 3), f.reshape(0, 3))
    key = np.round(v, 12)
    _, first, inverse = np.unique(key, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)  # unique rows in order of first appearance
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    new_index = rank[inverse.reshape(-1)]
    return Mesh(v[first[order]].astype(float), new_index[f] if f.size else f.reshape(0, 3))


def _load_obj(text: str) -> Mesh:
    verts, faces = [], []
    for line in text.splitlines():
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            verts.append([float(p[1]), float(p[2]), float(p[3])])
        elif p[0] == "f":
            idx = [int(tok.split("/")[0]) for tok in p[1:]]
            idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
            for k in range(1, len(idx) - 1):  # fan triangulation of polygons
                faces.append([idx[0], idx[k], idx[k + 1]])
    return _merge_vertices(np.array(verts, dtype=float).reshape(-1, 3), np.array(faces, dtype=int).reshape(-1, 3))


def _load_stl(data: bytes) -> Mesh:
    head = data[:512].lstrip()
    if head.startswith(b"solid") and b"facet" in data[:4096]:
        verts = []
        for line in data.decode("ascii", errors="ignore").splitlines():
            p = line.split()
            if len(p) == 4 and p[0] == "vertex":
                verts.append([float(p[1]), float(p[2]), float(p[3])])
        v = np.array(verts, dtype=float).reshape(-1, 3)
    else:
        (n,) = struct.unpack_from("<I", data, 80)
        rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
        v = rec["v"].reshape(-1, 3).astype(float)
    return _merge_vertices(v, np.arange(v.shape[0]).reshape(-1, 3))


def load_mesh(path: str | os.PathLike) -> Mesh:
    """OBJ / STL reader (the reference calls ``trimesh.load_mesh(file, file_type=suffix)``)."""
    path = pathlib.Path(path)
    kind = path.suffix.lower().lstrip(".")
    if kind == "obj":
        mesh = _load_obj(path.read_text())
    elif kind == "stl":
        mesh = _load_stl(path.read_bytes())
    else:
        raise RuntimeError(f"unsupported mesh format '{kind}' ({path}); OBJ and STL are read")
    if mesh.is_empty:
        raise RuntimeError(f"Failed to process '{path}'")  # like utils.py:251-252
    return mesh


def resolve_local_uri(uri: str, base_dir: str | os.PathLike | None = None) -> pathlib.Path:
    """``file://``, ``package://`` / ``model://`` (searched under the colon-separated directories of
    ``GZ_SIM_RESOURCE_PATH``, ``IGN_GAZEBO_RESOURCE_PATH``, ``ROS_PACKAGE_PATH``, ``AMENT_PREFIX_PATH``) and plain
    paths (relative ones against the directory of the description file)."""
    for scheme in ("file://",):
        if uri.startswith(scheme):
            return pathlib.Path(uri[len(scheme) :])
    for scheme in ("package://", "model://"):
        if uri.startswith(scheme):
            rel = uri[len(scheme) :]
            roots = []
            for var in ("GZ_SIM_RESOURCE_PATH", "IGN_GAZEBO_RESOURCE_PATH", "ROS_PACKAGE_PATH", "AMENT_PREFIX_PATH", "SDF_PATH"):
                roots += [r for r in os.environ.get(var, "").split(":") if r]
            for r in roots:
                for cand in (pathlib.Path(r) / rel, pathlib.Path(r) / "share" / rel):
                    if cand.exists():
                        return cand
            raise FileNotFoundError(f"cannot resolve '{uri}' (searched {roots})")
    p = pathlib.Path(uri)
    if not p.is_absolute() and base_dir is not None:
        p = pathlib.Path(base_dir) / p
    return p


# ---- point selection (parsers/rod/meshes.py:7-104) ------------------------------------------------------
def extract_points_vertices(mesh: Mesh) -> np.ndarray:
    return mesh.vertices


def _sample_surface(mesh: Mesh, n: int, rng) -> np.ndarray:
    area = mesh.area_faces()
    if area.sum() <= 0:
        raise RuntimeError("mesh without surface: cannot sample it")
    f = rng.choice(area.size, size=n, p=area / area.sum())
    t = mesh.triangles()[f]
    u, v = rng.random(n), rng.random(n)
    flip = u + v > 1.0
    u[flip], v[flip] = 1.0 - u[flip], 1.0 - v[flip]
    return t[:, 0] + u[:, None] * (t[:, 1] - t[:, 0]) + v[:, None] * (t[:, 2] - t[:, 0])


def extract_points_random_surface_sampling(mesh: Mesh, n: int, seed: int = 0) -> np.ndarray:
    return _sample_surface(mesh, int(n), np.random.default_rng(seed))


def extract_points_uniform_surface_sampling(mesh: Mesh, n: int, seed: int = 0) -> np.ndarray:
    n = int(n)
    radius = np.sqrt(mesh.area_faces().sum() / (3.0 * n))
    cand = _sample_surface(mesh, 8 * n, np.random.default_rng(seed))
    keep: list[np.ndarray] = []
    for p in cand:  # greedy rejection of samples closer than `radius` to an accepted one
        if all(np.linalg.norm(p - q) >= radius for q in keep):
            keep.append(p)
            if len(keep) == n:
                break
    return np.array(keep).reshape(-1, 3)


def extract_points_select_points_over_axis(mesh: Mesh, axis: str, direction: str, n: int) -> np.ndarray:
    if axis not in VALID_AXIS:
        raise KeyError(axis)
    dirs = {"higher": np.s_[-n:], "lower": np.s_[:n]}
    arr = np.array(mesh.vertices, dtype=float)
    arr.sort(axis=0)  # as written in the reference: every column is sorted on its own
    return arr[dirs[direction]]


def extract_points_aap(mesh: Mesh, axis: str, upper: float | None = None, lower: float | None = None) -> np.ndarray:
    upper = upper if upper is not None else np.inf
    lower = lower if lower is not None else -np.inf
    assert lower < upper, "Invalid bounds for axis-aligned plane"
    v = mesh.vertices
    return v[(v[:, VALID_AXIS[axis]] >= lower) & (v[:, VALID_AXIS[axis]] <= upper)]


def mesh_collision_points(uri: str, scale, H: np.ndarray, *, base_dir=None, method=None) -> np.ndarray:
    """``create_mesh_collision`` (``parsers/rod/utils.py:228-280``): mesh -> points in the link frame."""
    mesh = load_mesh(resolve_local_uri(uri, base_dir)).apply_scale(scale)
    pts = np.asarray((method or extract_points_vertices)(mesh), dtype=float).reshape(-1, 3)
    return pts @ H[:3, :3].T + H[:3, 3]


def mesh_collisions_enabled() -> bool:
    return bool(int(os.environ.get("JAXSIM_COLLISION_MESH_ENABLED", "0")))
