"""ctypes wrapper around oracle/libfenris_oracle.so -- the CPU restatement of the fenris assembly path.

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by the product package ``fenris_amd``.
See ``fenris_oracle.h`` for the parity-pin status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfenris_oracle.so")

QUAD4, HEX8, TET4, HEX27, TRI3, TET10, QUAD9, TRI6, HEX20, TET20 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
LAPLACE, LINEAR_ELASTIC, NEO_HOOKEAN, STVK, MASS_SCALAR, MASS_VECTOR, TENSOR = 0, 1, 2, 3, 4, 5, 6
OK, SINGULAR_JACOBIAN, BAD_ARGUMENT, COLUMN_NOT_FOUND = 0, 1, 2, 4

_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)


class _Assembler(C.Structure):
    _fields_ = [
        ("elem_kind", C.c_int),
        ("op_kind", C.c_int),
        ("vertices", _f64p),
        ("num_nodes", C.c_uint64),
        ("connectivity", _u64p),
        ("num_elements", C.c_uint64),
        ("u", _f64p),
        ("q_weights", _f64p),
        ("q_points", _f64p),
        ("nq", C.c_uint32),
        ("q_params", _f64p),
        ("elem_to_rule", _u64p),
        ("rule_params", _f64p),
        ("num_rules", C.c_uint64),
        ("q_tensor", _f64p),
        ("tensor_symmetric", C.c_int),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with its Makefile (gcc)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("fenris_oracle.c", "fenris_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.fo_material_energy_density.restype = C.c_double
        _lib.fo_material_energy_density.argtypes = [C.c_int, C.c_int, _f64p, C.c_double, C.c_double]
        _lib.fo_lame_from_young_poisson.argtypes = [C.c_double, C.c_double, _f64p, _f64p]
        _lib.fo_lame_from_young_poisson.restype = None
        _lib.fo_material_stress_tensor.argtypes = [C.c_int, C.c_int, _f64p, C.c_double, C.c_double, _f64p]
        _lib.fo_material_stress_tensor.restype = None
        _lib.fo_material_stress_contraction.argtypes = [C.c_int, C.c_int, _f64p, _f64p, _f64p, C.c_double,
                                                        C.c_double, _f64p]
        _lib.fo_material_stress_contraction.restype = None
        _lib.fo_free.argtypes = [C.c_void_p]
        _lib.fo_free.restype = None
        _lib.fo_create_rectangular_uniform_quad_mesh_2d.argtypes = [
            C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, _f64p, C.POINTER(_f64p), _u64p, C.POINTER(_u64p), _u64p]
        for f in (_lib.fo_create_rectangular_uniform_hex_mesh, _lib.fo_create_rectangular_uniform_tet_mesh):
            f.argtypes = [C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(_f64p), _u64p,
                          C.POINTER(_u64p), _u64p]
        _lib.fo_tet4_to_tet20.argtypes = [_f64p, C.c_uint64, _u64p, C.c_uint64, C.POINTER(_f64p), _u64p, C.POINTER(_u64p)]
        _lib.fo_refine_to_quadratic.argtypes = [C.c_int, _f64p, C.c_uint64, _u64p, C.c_uint64, C.POINTER(_f64p), _u64p,
                                                C.POINTER(_u64p)]
        _lib.fo_hex8_to_hex27.argtypes = [_f64p, C.c_uint64, _u64p, C.c_uint64, C.POINTER(_f64p), _u64p,
                                          C.POINTER(_u64p)]
        _lib.fo_assemble_pattern.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _u64p, _u64p, _u64p, _u64p, _u64p]
        _lib.fo_color_elements.argtypes = [C.c_uint64, _u64p, _u64p, _u64p, _u64p, _u64p]
        ap = C.POINTER(_Assembler)
        _lib.fo_assemble_element_matrix.argtypes = [ap, C.c_uint64, _f64p]
        _lib.fo_assemble_element_vector.argtypes = [ap, C.c_uint64, _f64p]
        _lib.fo_assemble_element_scalar.argtypes = [ap, C.c_uint64, _f64p]
        _lib.fo_assemble_into_csr.argtypes = [ap, _u64p, _u64p, _f64p, _u64p]
        _lib.fo_par_assemble_into_csr.argtypes = [ap, C.c_uint64, _u64p, _u64p, _u64p, _u64p, _f64p, C.c_int, _u64p]
        _lib.fo_assemble_vector_into.argtypes = [ap, _f64p, _u64p]
        _lib.fo_par_assemble_vector_into.argtypes = [ap, C.c_uint64, _u64p, _u64p, _f64p, C.c_int, _u64p]
        _lib.fo_assemble_scalar.argtypes = [ap, _f64p, _u64p]
        _lib.fo_assemble_source_vector_into.argtypes = [ap, C.c_int, _f64p, _f64p, _f64p]
        _lib.fo_assemble_element_source_vector.argtypes = [ap, C.c_uint64, C.c_int, _f64p, _f64p, _f64p]
        _lib.fo_physical_quadrature_points.argtypes = [ap, _f64p]
        _lib.fo_cuthill_mckee.argtypes = [C.c_uint64, _u64p, _u64p, _u64p]
        _lib.fo_reorder_mesh.argtypes = [C.c_uint64, C.c_uint64, _u64p, C.c_uint64, _u64p, _u64p]
        _lib.fo_cg_solve.argtypes = [C.c_uint64, _u64p, _u64p, _f64p, _f64p, _f64p, C.c_int, C.c_double, C.c_uint64, _u64p]
        _lib.fo_estimate_error_squared.argtypes = [ap, C.c_int, C.c_int, _f64p, _f64p, _f64p]
        _lib.fo_apply_homogeneous_dirichlet_bc_csr.argtypes = [C.c_uint64, _u64p, _u64p, _f64p, _u64p, C.c_uint64,
                                                               C.c_uint64]
        _lib.fo_element_gradients.argtypes = [C.c_int, _f64p, _f64p]
        _lib.fo_element_basis.argtypes = [C.c_int, _f64p, _f64p]
        _lib.fo_element_reference_jacobian.argtypes = [C.c_int, _f64p, _f64p, _f64p]
    return _lib


def _f(a):
    return a.ctypes.data_as(_f64p)


def _u(a):
    return a.ctypes.data_as(_u64p)


def element_num_nodes(kind):
    return lib().fo_element_num_nodes(kind)


def element_dim(kind):
    return lib().fo_element_dim(kind)


def solution_dim(op, d):
    return lib().fo_operator_solution_dim(op, d)


# ---------------------------------------------------------------- quadrature
def gauss(n):
    w, x = np.empty(n), np.empty(n)
    assert lib().fo_gauss(n, _f(w), _f(x)) == 0
    return w, x


def quadrilateral_gauss(n):
    w, p = np.empty(n * n), np.empty((n * n, 2))
    assert lib().fo_quadrilateral_gauss(n, _f(w), _f(p)) == 0
    return w, p


def hexahedron_gauss(n):
    w, p = np.empty(n ** 3), np.empty((n ** 3, 3))
    assert lib().fo_hexahedron_gauss(n, _f(w), _f(p)) == 0
    return w, p


def tetrahedron_rule(strength):
    w, p = np.empty(64), np.empty((64, 3))
    n = lib().fo_tetrahedron_rule(strength, _f(w), _f(p))
    assert n > 0
    return w[:n].copy(), p[:n].copy()


def triangle_rule(strength):
    w, p = np.empty(64), np.empty((64, 2))
    n = lib().fo_triangle_rule(strength, _f(w), _f(p))
    assert n > 0
    return w[:n].copy(), p[:n].copy()


# ---------------------------------------------------------------- meshes
def _take(vp, nv, cp, nc, d, n):
    nv, nc = int(nv.value), int(nc.value)
    if nv == 0:
        v = np.zeros((0, d))
    else:
        v = np.ctypeslib.as_array(vp, shape=(nv, d)).copy()
    if nc == 0:
        c = np.zeros((0, n), dtype=np.uint64)
    else:
        c = np.ctypeslib.as_array(cp, shape=(nc, n)).copy()
    lib().fo_free(vp)
    lib().fo_free(cp)
    return v, c


def quad_mesh(unit_length, ux, uy, cells_per_unit, top_left=(0.0, 1.0)):
    vp, cp, nv, nc = _f64p(), _u64p(), C.c_uint64(), C.c_uint64()
    tl = np.array(top_left, dtype=np.float64)
    st = lib().fo_create_rectangular_uniform_quad_mesh_2d(unit_length, ux, uy, cells_per_unit, _f(tl), C.byref(vp),
                                                          C.byref(nv), C.byref(cp), C.byref(nc))
    assert st == 0
    return _take(vp, nv, cp, nc, 2, 4)


def unit_square_quad_mesh(cells):
    return quad_mesh(1.0, 1, 1, cells)


def hex_mesh(unit_length, ux, uy, uz, cells_per_unit):
    vp, cp, nv, nc = _f64p(), _u64p(), C.c_uint64(), C.c_uint64()
    st = lib().fo_create_rectangular_uniform_hex_mesh(unit_length, ux, uy, uz, cells_per_unit, C.byref(vp),
                                                      C.byref(nv), C.byref(cp), C.byref(nc))
    assert st == 0
    return _take(vp, nv, cp, nc, 3, 8)


def unit_box_hex_mesh(cells):
    return hex_mesh(1.0, 1, 1, 1, cells)


def tet_mesh(unit_length, ux, uy, uz, cells_per_unit):
    vp, cp, nv, nc = _f64p(), _u64p(), C.c_uint64(), C.c_uint64()
    st = lib().fo_create_rectangular_uniform_tet_mesh(unit_length, ux, uy, uz, cells_per_unit, C.byref(vp),
                                                      C.byref(nv), C.byref(cp), C.byref(nc))
    assert st == 0
    return _take(vp, nv, cp, nc, 3, 4)


def unit_box_tet_mesh(cells):
    return tet_mesh(1.0, 1, 1, 1, cells)


def hex8_to_hex27(vertices, conn):
    vertices = np.ascontiguousarray(vertices, dtype=np.float64)
    conn = np.ascontiguousarray(conn, dtype=np.uint64)
    vp, cp, nv = _f64p(), _u64p(), C.c_uint64()
    st = lib().fo_hex8_to_hex27(_f(vertices), len(vertices), _u(conn), len(conn), C.byref(vp), C.byref(nv),
                                C.byref(cp))
    assert st == 0
    return _take(vp, nv, cp, C.c_uint64(len(conn)), 3, 27)


def tet4_to_tet20(vertices, conn):
    """Tet20Mesh::from(&tet4_mesh) (src/mesh_convert.rs:658-775)"""
    vertices = np.ascontiguousarray(vertices, dtype=np.float64)
    conn = np.ascontiguousarray(conn, dtype=np.uint64)
    vp, cp, nv = _f64p(), _u64p(), C.c_uint64()
    st = lib().fo_tet4_to_tet20(_f(vertices), len(vertices), _u(conn), len(conn), C.byref(vp), C.byref(nv), C.byref(cp))
    assert st == 0
    return _take(vp, nv, cp, C.c_uint64(len(conn)), 3, 20)


def refine_to_quadratic(from_kind, vertices, conn):
    """Tet4 -> Tet10, Tri3 -> Tri6, Quad4 -> Quad9 (src/mesh_convert.rs:42-83, 332-452)"""
    d, n1 = {TET4: (3, 10), TRI3: (2, 6), QUAD4: (2, 9), HEX8: (3, 20)}[from_kind]
    vertices = np.ascontiguousarray(vertices, dtype=np.float64)
    conn = np.ascontiguousarray(conn, dtype=np.uint64)
    vp, cp, nv = _f64p(), _u64p(), C.c_uint64()
    st = lib().fo_refine_to_quadratic(from_kind, _f(vertices), len(vertices), _u(conn), len(conn), C.byref(vp), C.byref(nv),
                                      C.byref(cp))
    assert st == 0
    return _take(vp, nv, cp, C.c_uint64(len(conn)), d, n1)


# ---------------------------------------------------------------- materials
def lame_from_young_poisson(young, poisson):
    mu, lam = C.c_double(), C.c_double()
    lib().fo_lame_from_young_poisson(young, poisson, C.byref(mu), C.byref(lam))
    return mu.value, lam.value


def _colmajor(F):
    return np.asfortranarray(np.asarray(F, dtype=np.float64))


def material_energy_density(op, F, mu, lam):
    F = _colmajor(F)
    return lib().fo_material_energy_density(op, F.shape[0], _f(F), mu, lam)


def material_stress_tensor(op, F, mu, lam):
    F = _colmajor(F)
    P = np.zeros_like(F, order="F")
    lib().fo_material_stress_tensor(op, F.shape[0], _f(F), mu, lam, _f(P))
    return np.array(P)


def material_stress_contraction(op, F, a, b, mu, lam):
    F = _colmajor(F)
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    Cm = np.zeros_like(F, order="F")
    lib().fo_material_stress_contraction(op, F.shape[0], _f(F), _f(a), _f(b), mu, lam, _f(Cm))
    return np.array(Cm)


def element_gradients(kind, xi):
    n, d = element_num_nodes(kind), element_dim(kind)
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    g = np.zeros((d, n), order="F")
    assert lib().fo_element_gradients(kind, _f(xi), _f(g)) == 0
    return np.array(g)


def element_basis(kind, xi):
    n = element_num_nodes(kind)
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    p = np.zeros(n)
    assert lib().fo_element_basis(kind, _f(xi), _f(p)) == 0
    return p


# ---------------------------------------------------------------- assembler descriptor
class ElementAssembler:
    """Mirror of ElementEllipticAssembler<Mesh, Op, UniformQuadratureTable> for the oracle."""

    def __init__(self, elem_kind, op_kind, vertices, connectivity, weights, points, params=None, u=None, elem_to_rule=None,
                 rule_params=None, tensor=None, tensor_symmetric=False):
        self.elem_kind, self.op_kind = elem_kind, op_kind
        self.n, self.d = element_num_nodes(elem_kind), element_dim(elem_kind)
        self.s = solution_dim(op_kind, self.d)
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, self.d)
        self.connectivity = np.ascontiguousarray(connectivity, dtype=np.uint64).reshape(-1, self.n)
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        self.points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, self.d)
        nq = len(self.weights)
        if params is None:
            self.params = None
        else:
            p = np.asarray(params, dtype=np.float64)
            if p.ndim == 1:  # uniform data (with_uniform_data quadrature_table.rs:264-266)
                p = np.tile(p, (nq, 1))
            self.params = np.ascontiguousarray(p)
        self.u = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        self.N, self.E = len(self.vertices), len(self.connectivity)
        # CompactQuadratureTable with shared points / weights: rule_params (R, nq, 2), elem_to_rule (E,)
        self.elem_to_rule = None if elem_to_rule is None else np.ascontiguousarray(elem_to_rule, dtype=np.uint64)
        self.rule_params = None if rule_params is None else np.ascontiguousarray(rule_params, dtype=np.float64).reshape(-1, nq, 2)
        self._st = _Assembler(elem_kind, op_kind, _f(self.vertices), self.N, _u(self.connectivity), self.E,
                              _f(self.u) if self.u is not None else None, _f(self.weights), _f(self.points), nq,
                              _f(self.params) if self.params is not None else None,
                              _u(self.elem_to_rule) if self.elem_to_rule is not None else None,
                              _f(self.rule_params) if self.rule_params is not None else None,
                              0 if self.rule_params is None else len(self.rule_params),
                              None, 0)
        # TENSOR: one d x d x d x d coefficient tensor per quadrature point (or one for all: broadcast)
        self.tensor = None
        if tensor is not None:
            t = np.asarray(tensor, dtype=np.float64)
            if t.ndim == 4:
                t = np.tile(t, (nq, 1, 1, 1, 1))
            self.tensor = np.ascontiguousarray(t.reshape(nq, self.d ** 4))
            self._st.q_tensor = _f(self.tensor)
            self._st.tensor_symmetric = 1 if tensor_symmetric else 0

    # ElementConnectivityAssembler (src/assembly/local.rs:18-47)
    def solution_dim(self):
        return self.s

    def num_elements(self):
        return self.E

    def num_nodes(self):
        return self.N

    def element_matrix(self, e):
        ld = self.s * self.n
        ke = np.zeros((ld, ld), order="F")
        st = lib().fo_assemble_element_matrix(C.byref(self._st), e, _f(ke))
        return st, np.array(ke)

    def element_vector(self, e):
        fe = np.zeros(self.s * self.n)
        st = lib().fo_assemble_element_vector(C.byref(self._st), e, _f(fe))
        return st, fe

    def element_scalar(self, e):
        out = C.c_double()
        st = lib().fo_assemble_element_scalar(C.byref(self._st), e, C.byref(out))
        return st, out.value

    def elem_offsets_nodes(self):
        offs = np.arange(0, (self.E + 1) * self.n, self.n, dtype=np.uint64)
        return offs, self.connectivity.reshape(-1)


def assemble_pattern(sdim, num_nodes, elem_offsets, elem_nodes):
    """CsrAssembler::assemble_pattern on a ragged connectivity."""
    elem_offsets = np.ascontiguousarray(elem_offsets, dtype=np.uint64)
    elem_nodes = np.ascontiguousarray(elem_nodes, dtype=np.uint64)
    E = len(elem_offsets) - 1
    ro = np.zeros(sdim * num_nodes + 1, dtype=np.uint64)
    nnz = C.c_uint64()
    en = elem_nodes if len(elem_nodes) else np.zeros(1, dtype=np.uint64)
    st = lib().fo_assemble_pattern(sdim, num_nodes, E, _u(elem_offsets), _u(en), _u(ro), None, C.byref(nnz))
    assert st == 0, st
    ci = np.zeros(max(int(nnz.value), 1), dtype=np.uint64)
    st = lib().fo_assemble_pattern(sdim, num_nodes, E, _u(elem_offsets), _u(en), _u(ro), _u(ci), C.byref(nnz))
    assert st == 0, st
    return ro, ci[: int(nnz.value)]


def pattern_for(asm: ElementAssembler):
    offs, nodes = asm.elem_offsets_nodes()
    return assemble_pattern(asm.s, asm.N, offs, nodes)


def color_elements(elem_offsets, elem_nodes):
    elem_offsets = np.ascontiguousarray(elem_offsets, dtype=np.uint64)
    elem_nodes = np.ascontiguousarray(elem_nodes, dtype=np.uint64)
    E = len(elem_offsets) - 1
    nc = C.c_uint64()
    co = np.zeros(E + 2, dtype=np.uint64)
    labels = np.zeros(max(E, 1), dtype=np.uint64)
    en = elem_nodes if len(elem_nodes) else np.zeros(1, dtype=np.uint64)
    st = lib().fo_color_elements(E, _u(elem_offsets), _u(en), C.byref(nc), _u(co), _u(labels))
    assert st == 0
    k = int(nc.value)
    return co[: k + 1].copy(), labels[:E].copy()


def color_nodes(asm: ElementAssembler):
    """color_nodes (src/assembly/global.rs:540-551)."""
    return color_elements(*asm.elem_offsets_nodes())


def assemble_into_csr(asm, ro, ci, values):
    failed = C.c_uint64(0)
    st = lib().fo_assemble_into_csr(C.byref(asm._st), _u(ro), _u(ci), _f(values), C.byref(failed))
    return st, int(failed.value)


def assemble(asm):
    """CsrAssembler::assemble (global.rs:124-131): pattern + zeros + assemble_into_csr."""
    ro, ci = pattern_for(asm)
    values = np.zeros(len(ci))
    st, failed = assemble_into_csr(asm, ro, ci, values)
    return st, failed, ro, ci, values


def par_assemble_into_csr(asm, colors, ro, ci, values, num_threads=0):
    co, labels = colors
    failed = C.c_uint64(0)
    st = lib().fo_par_assemble_into_csr(C.byref(asm._st), len(co) - 1, _u(co), _u(labels), _u(ro), _u(ci),
                                        _f(values), num_threads, C.byref(failed))
    return st, int(failed.value)


def assemble_vector(asm, out=None):
    if out is None:
        out = np.zeros(asm.s * asm.N)
    failed = C.c_uint64(0)
    st = lib().fo_assemble_vector_into(C.byref(asm._st), _f(out), C.byref(failed))
    return st, int(failed.value), out


def par_assemble_vector(asm, colors, out=None, num_threads=0):
    co, labels = colors
    if out is None:
        out = np.zeros(asm.s * asm.N)
    failed = C.c_uint64(0)
    st = lib().fo_par_assemble_vector_into(C.byref(asm._st), len(co) - 1, _u(co), _u(labels), _f(out), num_threads,
                                           C.byref(failed))
    return st, int(failed.value), out


def assemble_source_vector(asm, s, g=None, values=None, out=None):
    """VectorAssembler + ElementSourceAssembler (source.rs:219-278); asm.op_kind is ignored"""
    if out is None:
        out = np.zeros(s * asm.N)
    g = None if g is None else np.ascontiguousarray(g, dtype=np.float64)
    values = None if values is None else np.ascontiguousarray(values, dtype=np.float64)
    st = lib().fo_assemble_source_vector_into(C.byref(asm._st), s, _f(g) if g is not None else None,
                                              _f(values) if values is not None else None, _f(out))
    return st, out


def cuthill_mckee(ro, ci):
    """src/mesh/reorder.rs:171-233; perm[target] = source"""
    ro = np.ascontiguousarray(ro, dtype=np.uint64)
    ci = np.ascontiguousarray(ci, dtype=np.uint64)
    perm = np.zeros(len(ro) - 1, dtype=np.uint64)
    st = lib().fo_cuthill_mckee(len(ro) - 1, _u(ro), _u(ci if len(ci) else np.zeros(1, dtype=np.uint64)), _u(perm))
    assert st == 0
    return perm


def reorder_mesh(num_vertices, connectivity):
    """reorder_mesh_par (reorder.rs:54-95) -> (vertex_perm, connectivity_perm)"""
    c = np.ascontiguousarray(connectivity, dtype=np.uint64)
    vp, cp = np.zeros(num_vertices, dtype=np.uint64), np.zeros(len(c), dtype=np.uint64)
    st = lib().fo_reorder_mesh(num_vertices, c.shape[1], _u(c), len(c), _u(vp), _u(cp))
    assert st == 0
    return vp, cp


def cg_solve(ro, ci, values, b, x0=None, jacobi=True, tol=1e-9, max_iter=10000):
    """ConjugateGradient as driven by solve_linear_system (poisson_mms_common.rs:142-163); returns (status, x, iterations)"""
    ro = np.ascontiguousarray(ro, dtype=np.uint64)
    ci = np.ascontiguousarray(ci, dtype=np.uint64)
    values = np.ascontiguousarray(values, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(len(b)) if x0 is None else np.array(x0, dtype=np.float64)
    it = C.c_uint64(0)
    st = lib().fo_cg_solve(len(b), _u(ro), _u(ci), _f(values), _f(b), _f(x), 1 if jacobi else 0, tol, max_iter, C.byref(it))
    return st, x, int(it.value)


def estimate_error_squared(asm, which, s, u_h, exact):
    """src/error.rs:287-372: which = 0 L2, 1 H1 seminorm; exact sampled at the physical quadrature points"""
    out = C.c_double()
    u_h = np.ascontiguousarray(u_h, dtype=np.float64)
    exact = np.ascontiguousarray(exact, dtype=np.float64)
    st = lib().fo_estimate_error_squared(C.byref(asm._st), which, s, _f(u_h), _f(exact), C.byref(out))
    return st, out.value


def physical_quadrature_points(asm):
    x = np.zeros((asm.E, len(asm.weights), asm.d))
    st = lib().fo_physical_quadrature_points(C.byref(asm._st), _f(x))
    assert st == 0
    return x


def assemble_scalar(asm):
    out, failed = C.c_double(), C.c_uint64(0)
    st = lib().fo_assemble_scalar(C.byref(asm._st), C.byref(out), C.byref(failed))
    return st, int(failed.value), out.value


def apply_homogeneous_dirichlet_bc_csr(ro, ci, values, nodes, solution_dim):
    nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
    nn = nodes if len(nodes) else np.zeros(1, dtype=np.uint64)
    st = lib().fo_apply_homogeneous_dirichlet_bc_csr(len(ro) - 1, _u(ro), _u(ci), _f(values), _u(nn), len(nodes),
                                                     solution_dim)
    assert st == 0


def max_threads():
    return lib().fo_max_threads()
