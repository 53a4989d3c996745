"""Quadrature rules: host mirror of fenris-quadrature (univariate.rs, tensor.rs, polyquad tables).

Rules are returned like the reference's ``Rule<D> = (weights, points)`` (src/quadrature.rs:23).
Computed by the host C++ of libfenris_hip (fenris_amd/csrc/host_inputs.cpp).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi


def _check(rc, what):
    if rc != 0:
        raise _ffi.FenrisError(rc, what)


class univariate:
    @staticmethod
    def gauss(num_points):
        """fenris-quadrature/src/univariate.rs:66-118"""
        w, x = np.empty(num_points), np.empty(num_points)
        _check(_ffi.lib().fh_gauss(num_points, _ffi.fp(w), _ffi.fp(x)), "gauss: number of points must be positive")
        return w, x.reshape(-1, 1)


class tensor:
    @staticmethod
    def quadrilateral_gauss(num_points_per_dim):
        """tensor.rs:13-31"""
        n = num_points_per_dim
        w, p = np.empty(n * n), np.empty((n * n, 2))
        _check(_ffi.lib().fh_quadrilateral_gauss(n, _ffi.fp(w), _ffi.fp(p)), "quadrilateral_gauss")
        return w, p

    @staticmethod
    def hexahedron_gauss(num_points_per_dim):
        """tensor.rs:36-55"""
        n = num_points_per_dim
        w, p = np.empty(n ** 3), np.empty((n ** 3, 3))
        _check(_ffi.lib().fh_hexahedron_gauss(n, _ffi.fp(w), _ffi.fp(p)), "hexahedron_gauss")
        return w, p


class total_order:
    """src/quadrature/total_order.rs -> polyquad tables; smallest tabulated strength >= requested."""

    @staticmethod
    def tetrahedron(strength):
        w, p, n = np.empty(128), np.empty((128, 3)), C.c_uint32()
        _check(_ffi.lib().fh_tetrahedron_rule(max(strength, 1), _ffi.fp(w), _ffi.fp(p), C.byref(n)),
               f"tetrahedron rule of strength {strength} is not tabulated")
        return w[: n.value].copy(), p[: n.value].copy()

    @staticmethod
    def triangle(strength):
        w, p, n = np.empty(128), np.empty((128, 2)), C.c_uint32()
        _check(_ffi.lib().fh_triangle_rule(max(strength, 1), _ffi.fp(w), _ffi.fp(p), C.byref(n)),
               f"triangle rule of strength {strength} is not tabulated")
        return w[: n.value].copy(), p[: n.value].copy()
