"""Operators and materials: host mirror of src/assembly/operators/laplace.rs and fenris-solid."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

from . import _ffi


class LaplaceOperator:
    """src/assembly/operators/laplace.rs:13-73 (SolutionDim = 1, Parameters = ())"""
    op_kind = _ffi.LAPLACE


@dataclass(frozen=True)
class LameParameters:
    """fenris-solid/src/materials.rs:8-12"""
    mu: float = 0.0
    lambda_: float = 0.0

    @classmethod
    def from_young_poisson(cls, yp: "YoungPoisson"):
        """impl From<YoungPoisson> for LameParameters (materials.rs:31-43)"""
        mu, lam = C.c_double(), C.c_double()
        _ffi.lib().fh_lame_from_young_poisson(yp.young, yp.poisson, C.byref(mu), C.byref(lam))
        return cls(mu.value, lam.value)

    def as_pair(self):
        return (self.mu, self.lambda_)


@dataclass(frozen=True)
class YoungPoisson:
    """materials.rs:25-29"""
    young: float
    poisson: float


class LinearElasticMaterial:
    """materials.rs:66-123"""
    op_kind = _ffi.LINEAR_ELASTIC


class NeoHookeanMaterial:
    """materials.rs:225-353"""
    op_kind = _ffi.NEO_HOOKEAN


class StVKMaterial:
    """materials.rs:370-469"""
    op_kind = _ffi.STVK


class TensorEllipticOperator:
    """An elliptic operator given as data (FH_TENSOR, include/fenris_hip.h): the contraction
    C(a, b)[i][k] = sum_jl a[j] A[i][j][k][l] b[l] of an `EllipticContraction` (src/assembly/operators.rs:146-189) whose coefficients do not
    depend on grad u, SolutionDim = GeometryDim.  ``tensor``: (d, d, d, d) for every quadrature point, or (nq, d, d, d, d).
    ``symmetric=True`` is `Symmetry::Symmetric` (the caller asserts A[i][j][k][l] == A[k][l][i][j]; the upper block triangle is formed and
    mirrored, operators.rs:176-181), ``False`` is `Symmetry::NonSymmetric` (every block formed)."""
    op_kind = _ffi.TENSOR

    def __init__(self, tensor, symmetric=False):
        import numpy as np

        self.tensor = np.asarray(tensor, dtype=np.float64)
        self.symmetric = bool(symmetric)

    def tensors_for(self, nq, d):
        import numpy as np

        t = self.tensor
        if t.shape == (d, d, d, d):
            t = np.broadcast_to(t, (nq, d, d, d, d))
        if t.shape != (nq, d, d, d, d):
            raise ValueError(f"TensorEllipticOperator: tensor of shape {self.tensor.shape} for {nq} points in {d} dimensions")
        return np.ascontiguousarray(t).reshape(nq, d ** 4)

    @staticmethod
    def linear_elastic(mu, lambda_, d=3):
        """the tensor of LinearElasticMaterial (materials.rs:108-118): mu (delta_ik delta_jl + delta_il delta_jk) + lambda delta_ij delta_kl"""
        import numpy as np

        I = np.eye(d)
        return mu * (np.einsum("ik,jl->ijkl", I, I) + np.einsum("il,jk->ijkl", I, I)) + lambda_ * np.einsum("ij,kl->ijkl", I, I)


class MaterialEllipticOperator:
    """fenris-solid/src/lib.rs:412-508: turns a hyperelastic material into an elliptic operator
    (SolutionDim = GeometryDim, Parameters = LameParameters)."""

    def __init__(self, material):
        self.material = material
        self.op_kind = material.op_kind

    @classmethod
    def new(cls, material):
        return cls(material)


@dataclass(frozen=True)
class Density:
    """src/assembly/local/mass.rs:15-17"""
    value: float = 0.0

    def as_pair(self):
        return (self.value, 0.0)


class GravitySource:
    """fenris-solid/src/gravity_source.rs:21-65: force density rho * g (SolutionDim = GeometryDim,
    Parameters = Density per quadrature point)."""

    def __init__(self, gravitational_acceleration):
        self.gravitational_acceleration = [float(x) for x in gravitational_acceleration]

    @classmethod
    def from_acceleration(cls, gravitational_acceleration):
        return cls(gravitational_acceleration)

    @property
    def solution_dim(self):
        return len(self.gravitational_acceleration)


class SourceFunction:
    """src/assembly/local/source.rs:14-22.  The reference evaluates arbitrary Rust code per quadrature point; here
    ``evaluate(x, data)`` is a vectorised Python callable: ``x`` is the (E, nq, d) array of physical points, ``data``
    the table's per-point data (nq, 2) or None, and it returns (E, nq, solution_dim) values that the device kernel
    integrates against the basis."""

    def __init__(self, solution_dim, evaluate):
        self.solution_dim = int(solution_dim)
        self.evaluate = evaluate
