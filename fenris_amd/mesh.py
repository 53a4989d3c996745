"""Mesh container and procedural generators: host mirror of src/mesh.rs and src/mesh/procedural.rs."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi


@dataclass
class Mesh:
    """Mesh<f64, D, C> (src/mesh.rs:23-40): ``vertices`` N x D float64, ``connectivity`` E x n uint64."""
    vertices: np.ndarray
    connectivity: np.ndarray
    elem_kind: int

    def __post_init__(self):
        d, n = _ffi.ELEM_DIM[self.elem_kind], _ffi.ELEM_NODES[self.elem_kind]
        self.vertices = _ffi.as_f64(self.vertices).reshape(-1, d)
        self.connectivity = _ffi.as_u64(self.connectivity).reshape(-1, n)

    @classmethod
    def from_vertices_and_connectivity(cls, vertices, connectivity, elem_kind):
        return cls(vertices, connectivity, elem_kind)

    def num_elements(self):
        return len(self.connectivity)

    def num_nodes(self):
        return len(self.vertices)


def _gen(fn, d, n, kind, *args):
    nv, nc = C.c_uint64(), C.c_uint64()
    rc = fn(*args, None, None, C.byref(nv), C.byref(nc))
    if rc:
        raise _ffi.FenrisError(rc, "mesh generator")
    v = np.zeros((nv.value, d))
    c = np.zeros((nc.value, n), dtype=np.uint64)
    if nv.value:
        rc = fn(*args, _ffi.fp(v), _ffi.up(c), C.byref(nv), C.byref(nc))
        if rc:
            raise _ffi.FenrisError(rc, "mesh generator")
    return Mesh(v, c, kind)


class procedural:
    @staticmethod
    def create_rectangular_uniform_quad_mesh_2d(unit_length, units_x, units_y, cells_per_unit, top_left):
        """src/mesh/procedural.rs:46-93"""
        tl = _ffi.as_f64(top_left)
        return _gen(_ffi.lib().fh_quad_mesh_2d, 2, 4, _ffi.QUAD4, float(unit_length), units_x, units_y, cells_per_unit,
                    _ffi.fp(tl))

    @staticmethod
    def create_unit_square_uniform_quad_mesh_2d(cells_per_dim):
        """procedural.rs:15-20"""
        return procedural.create_rectangular_uniform_quad_mesh_2d(1.0, 1, 1, cells_per_dim, (0.0, 1.0))

    @staticmethod
    def create_unit_square_uniform_tri_mesh_2d(cells_per_dim):
        """procedural.rs:22-28: the quad mesh with every (convex) cell split into the triangles [0, 1, 2] and [0, 2, 3]
        (split_into_triangles, src/mesh.rs:276-293; fenris-geometry/src/primitives/quad.rs:76-88)"""
        q = procedural.create_unit_square_uniform_quad_mesh_2d(cells_per_dim)
        c = q.connectivity
        tri = np.stack([c[:, [0, 1, 2]], c[:, [0, 2, 3]]], axis=1).reshape(-1, 3)
        return Mesh(q.vertices, np.ascontiguousarray(tri, dtype=np.uint64), _ffi.TRI3)

    @staticmethod
    def create_rectangular_uniform_hex_mesh(unit_length, units_x, units_y, units_z, cells_per_unit):
        """procedural.rs:216-277"""
        return _gen(_ffi.lib().fh_hex_mesh, 3, 8, _ffi.HEX8, float(unit_length), units_x, units_y, units_z,
                    cells_per_unit)

    @staticmethod
    def create_unit_box_uniform_hex_mesh_3d(cells_per_dim):
        """procedural.rs:30-35"""
        return procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, cells_per_dim)

    @staticmethod
    def create_rectangular_uniform_tet_mesh(unit_length, units_x, units_y, units_z, cells_per_unit):
        """procedural.rs:286-403 (BCC lattice)"""
        return _gen(_ffi.lib().fh_tet_mesh, 3, 4, _ffi.TET4, float(unit_length), units_x, units_y, units_z,
                    cells_per_unit)

    @staticmethod
    def create_unit_box_uniform_tet_mesh_3d(cells_per_dim):
        """procedural.rs:37-42"""
        return procedural.create_rectangular_uniform_tet_mesh(1.0, 1, 1, 1, cells_per_dim)


def hex27_mesh_from_hex8(mesh: Mesh) -> Mesh:
    """Hex27Mesh::from(&hex8_mesh) (src/mesh_convert.rs:85-166, 227-330)."""
    assert mesh.elem_kind == _ffi.HEX8
    E = mesh.num_elements()
    out_v = np.zeros((max(27 * E, 1), 3))
    out_c = np.zeros((E, 27), dtype=np.uint64)
    nv = C.c_uint64()
    rc = _ffi.lib().fh_hex8_to_hex27(_ffi.fp(mesh.vertices), mesh.num_nodes(), _ffi.up(mesh.connectivity), E,
                                     _ffi.fp(out_v), C.byref(nv), _ffi.up(out_c))
    if rc:
        raise _ffi.FenrisError(rc, "hex8_to_hex27")
    return Mesh(out_v[: nv.value].copy(), out_c, _ffi.HEX27)


def _refine(mesh: Mesh, from_kind, to_kind) -> Mesh:
    assert mesh.elem_kind == from_kind
    E, d, n1 = mesh.num_elements(), _ffi.ELEM_DIM[to_kind], _ffi.ELEM_NODES[to_kind]
    out_v = np.zeros((max(mesh.num_nodes() + n1 * E, 1), d))
    out_c = np.zeros((E, n1), dtype=np.uint64)
    nv = C.c_uint64()
    rc = _ffi.lib().fh_refine_to_quadratic(from_kind, _ffi.fp(mesh.vertices), mesh.num_nodes(), _ffi.up(mesh.connectivity), E,
                                           _ffi.fp(out_v), C.byref(nv), _ffi.up(out_c))
    if rc:
        raise _ffi.FenrisError(rc, "refine_to_quadratic")
    return Mesh(out_v[: nv.value].copy(), out_c, to_kind)


def tet10_mesh_from_tet4(mesh: Mesh) -> Mesh:
    """Tet10Mesh::from(&tet4_mesh) (src/mesh_convert.rs:42-83, 227-330, 444-452)"""
    return _refine(mesh, _ffi.TET4, _ffi.TET10)


def tri6_mesh_from_tri3(mesh: Mesh) -> Mesh:
    """Mesh2d<Tri6d2Connectivity>::from(tri3_mesh) (src/mesh_convert.rs:332-383)"""
    return _refine(mesh, _ffi.TRI3, _ffi.TRI6)


def quad9_mesh_from_quad4(mesh: Mesh) -> Mesh:
    """Mesh2d<Quad9d2Connectivity>::from(quad4_mesh) (src/mesh_convert.rs:385-442)"""
    return _refine(mesh, _ffi.QUAD4, _ffi.QUAD9)


def hex20_mesh_from_hex8(mesh: Mesh) -> Mesh:
    """Hex20Mesh::from(&hex8_mesh) (src/mesh_convert.rs:168-217, 227-330, 481-490)"""
    return _refine(mesh, _ffi.HEX8, _ffi.HEX20)


def tet20_mesh_from_tet4(mesh: Mesh) -> Mesh:
    """Tet20Mesh::from(&tet4_mesh) (src/mesh_convert.rs:658-775)"""
    assert mesh.elem_kind == _ffi.TET4
    E = mesh.num_elements()
    out_v = np.zeros((max(20 * E, 1), 3))
    out_c = np.zeros((E, 20), dtype=np.uint64)
    nv = C.c_uint64()
    rc = _ffi.lib().fh_tet4_to_tet20(_ffi.fp(mesh.vertices), mesh.num_nodes(), _ffi.up(mesh.connectivity), E, _ffi.fp(out_v),
                                     C.byref(nv), _ffi.up(out_c))
    if rc:
        raise _ffi.FenrisError(rc, "tet4_to_tet20")
    return Mesh(out_v[: nv.value].copy(), out_c, _ffi.TET20)
