"""Vertex / element reordering: host mirror of src/mesh/reorder.rs (reverse Cuthill-McKee).

A locality-preserving numbering is what the owner-computes assembly kernel feeds on: its node blocks are
contiguous index ranges, and consecutive blocks of a sweep chain reuse the staged elements they share."""
from __future__ import annotations

import numpy as np

from . import _ffi
from .mesh import Mesh


class InvalidPermutation(ValueError):
    """src/mesh/reorder.rs:103-114"""


class Permutation:
    """src/mesh/reorder.rs:97-169: ``perm[target_index] = source_index``"""

    def __init__(self, perm):
        self._perm = _ffi.as_u64(perm)

    @classmethod
    def from_vec(cls, perm):
        p = np.asarray(perm, dtype=np.int64)
        if len(p) and (p.min() < 0 or p.max() >= len(p) or len(np.unique(p)) != len(p)):
            raise InvalidPermutation("Invalid permutation")
        return cls(p)

    def __len__(self):
        return len(self._perm)

    def __eq__(self, other):
        return isinstance(other, Permutation) and np.array_equal(self._perm, other._perm)

    def perm(self):
        return self._perm

    def reverse(self):
        self._perm = np.ascontiguousarray(self._perm[::-1])

    def source_index(self, target_index):
        return int(self._perm[target_index])

    def inverse(self) -> "Permutation":
        inv = np.empty(len(self._perm), dtype=np.uint64)
        inv[self._perm.astype(np.int64)] = np.arange(len(self._perm), dtype=np.uint64)
        return Permutation(inv)

    def apply_to_slice(self, array):
        array = np.asarray(array)
        if len(array) != len(self._perm):
            raise ValueError("Slice and permutation must have the same size.")
        return array[self._perm.astype(np.int64)]


class MeshPermutation:
    """src/mesh/reorder.rs:13-52"""

    def __init__(self, vertex_perm: Permutation, connectivity_perm: Permutation):
        self._v, self._c = vertex_perm, connectivity_perm

    def vertex_permutation(self):
        return self._v

    def connectivity_permutation(self):
        return self._c

    def apply(self, mesh: Mesh) -> Mesh:
        new_vertices = self._v.apply_to_slice(mesh.vertices)
        inv = self._v.inverse().perm()
        new_conn = inv[self._c.apply_to_slice(mesh.connectivity).astype(np.int64)]
        return Mesh(np.ascontiguousarray(new_vertices), np.ascontiguousarray(new_conn, dtype=np.uint64), mesh.elem_kind)


def cuthill_mckee(row_offsets, col_indices) -> Permutation:
    """src/mesh/reorder.rs:171-233 on a square sparsity pattern"""
    ro, ci = _ffi.as_u64(row_offsets), _ffi.as_u64(col_indices)
    n = len(ro) - 1
    perm = np.zeros(max(n, 1), dtype=np.uint64)
    cip = ci if len(ci) else np.zeros(1, dtype=np.uint64)
    rc = _ffi.lib().fh_cuthill_mckee(n, _ffi.up(ro), _ffi.up(cip), _ffi.up(perm))
    if rc:
        raise _ffi.FenrisError(rc, "fh_cuthill_mckee failed")
    return Permutation(perm[:n])


def reverse_cuthill_mckee(row_offsets, col_indices) -> Permutation:
    """src/mesh/reorder.rs:235-239"""
    p = cuthill_mckee(row_offsets, col_indices)
    p.reverse()
    return p


def reorder_mesh_par(mesh: Mesh) -> MeshPermutation:
    """src/mesh/reorder.rs:54-95"""
    conn = _ffi.as_u64(mesh.connectivity)
    N, E = mesh.num_nodes(), mesh.num_elements()
    vp, cp = np.zeros(max(N, 1), dtype=np.uint64), np.zeros(max(E, 1), dtype=np.uint64)
    rc = _ffi.lib().fh_reorder_mesh(N, conn.shape[1], _ffi.up(conn), E, _ffi.up(vp), _ffi.up(cp))
    if rc:
        raise _ffi.FenrisError(rc, "fh_reorder_mesh failed")
    return MeshPermutation(Permutation(vp[:N]), Permutation(cp[:E]))
