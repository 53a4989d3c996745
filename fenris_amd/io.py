"""Mesh input: host mirror of src/io/msh.rs (Gmsh MSH 4.1)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from .mesh import Mesh


class MshError(ValueError):
    """eyre errors of load_msh_from_bytes (src/io/msh.rs:47-111)"""


def load_msh_from_bytes(data: bytes, elem_kind: int) -> Mesh:
    """src/io/msh.rs:47-111: ``elem_kind`` plays the role of the connectivity type parameter (Tri3d2, Quad4d2, Tet4,
    Hex8, Hex27): only element blocks of that Gmsh type and entity dimension are read."""
    lib = _ffi.lib()
    nv, ne = C.c_uint64(), C.c_uint64()
    rc = lib.fh_load_msh(data, len(data), elem_kind, None, C.byref(nv), None, C.byref(ne))
    if rc:
        raise MshError((lib.fh_msh_last_error() or b"").decode())
    d, n = _ffi.ELEM_DIM[elem_kind], _ffi.ELEM_NODES[elem_kind]
    v = np.zeros((max(nv.value, 1), d))
    c = np.zeros((max(ne.value, 1), n), dtype=np.uint64)
    rc = lib.fh_load_msh(data, len(data), elem_kind, _ffi.fp(v), C.byref(nv), _ffi.up(c), C.byref(ne))
    if rc:
        raise MshError((lib.fh_msh_last_error() or b"").decode())
    return Mesh(v[: nv.value].copy(), c[: ne.value].copy(), elem_kind)


def load_msh_from_file(path, elem_kind: int) -> Mesh:
    """src/io/msh.rs:35-45"""
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError as exc:
        raise MshError(f"failed to read file: {exc}") from exc
    return load_msh_from_bytes(data, elem_kind)
