"""Global / local assembly: host mirror of src/assembly/{local,global}.rs over the C ABI.

Names and call shapes follow the reference so that tests read like the reference's own tests:

    quadrature = UniformQuadratureTable.from_points_and_weights(points, weights)
    assembler = (ElementEllipticAssemblerBuilder()
                 .with_finite_element_space(mesh).with_operator(LaplaceOperator())
                 .with_quadrature_table(quadrature).with_u(u).build())
    a_global = CsrAssembler().assemble(assembler)                  # examples/poisson2d.rs:33-60
    colors = color_nodes(mesh); CsrParAssembler().assemble(colors, assembler)

All numerics run in libfenris_hip.so on the GPU; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _ffi
from .compose import _Composable, is_composed
from . import compose as _compose
from ._ffi import (ASSEMBLE_OVERWRITE, SCATTER_ATOMIC, SCATTER_COLORED, SCATTER_GATHER, FenrisError,
                   SingularJacobianError)
from .mesh import Mesh


# ------------------------------------------------------------------------------------------ engine
class Engine:
    """Owns one fh_ctx (one device, one stream)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._lib = _ffi.lib()
        self._h = self._lib.fh_create(device)
        if not self._h:
            raise FenrisError(_ffi.FH_HIP_ERROR, f"cannot create a context on HIP device {device} (no GPU?)")
        self.device = device
        if stream is not None:
            self._check(self._lib.fh_set_stream(self._h, C.c_void_p(stream)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        """fh_last_error: the message of the last failed call on this context"""
        return (self._lib.fh_last_error(self._h) or b"").decode()

    def _check(self, rc, failed=None):
        if rc == _ffi.FH_OK:
            return
        msg = (self._lib.fh_last_error(self._h) or b"").decode()
        if rc == _ffi.FH_SINGULAR_JACOBIAN:
            raise SingularJacobianError(msg, int(failed.value) if failed is not None else -1)
        raise FenrisError(rc, msg)

    # inputs
    def set_mesh(self, mesh: Mesh):
        self._mesh = mesh  # keep the host arrays alive
        self._check(self._lib.fh_set_mesh(self._h, mesh.elem_kind, _ffi.fp(mesh.vertices), mesh.num_nodes(),
                                          _ffi.up(mesh.connectivity), mesh.num_elements()))

    def set_connectivity_ragged(self, sdim, num_nodes, elem_offsets, elem_nodes):
        eo, en = _ffi.as_u64(elem_offsets), _ffi.as_u64(elem_nodes)
        en_p = en if len(en) else np.zeros(1, dtype=np.uint64)
        self._check(self._lib.fh_set_connectivity_ragged(self._h, sdim, num_nodes, _ffi.up(eo), _ffi.up(en_p), len(eo) - 1))

    def set_active_elements(self, mask):
        """Numerics only over elements with mask != 0; the pattern still comes from all elements."""
        if mask is None:
            self._check(self._lib.fh_set_active_elements(self._h, None))
        else:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert len(m) == self.num_elements()
            self._check(self._lib.fh_set_active_elements(self._h, m.ctypes.data_as(C.c_char_p)))

    def set_row_range(self, node_begin, node_end):
        """owner-computes assembly of the rows of nodes [node_begin, node_end) only (fh_set_row_range)"""
        self._check(self._lib.fh_set_row_range(self._h, int(node_begin), int(node_end)))

    def set_operator(self, op_kind):
        self._check(self._lib.fh_set_operator(self._h, op_kind))

    def set_operator_tensor(self, tensors, symmetric):
        """fh_set_operator_tensor: (nq, d^4) coefficient tensors of FH_TENSOR for the quadrature table in use"""
        t = _ffi.as_f64(tensors)
        self._check(self._lib.fh_set_operator_tensor(self._h, _ffi.fp(t), t.shape[0], 1 if symmetric else 0))

    def set_quadrature_uniform(self, weights, points, params=None):
        w, p = _ffi.as_f64(weights), _ffi.as_f64(points)
        q = None if params is None else _ffi.as_f64(params)
        self._check(self._lib.fh_set_quadrature_uniform(self._h, _ffi.fp(w), _ffi.fp(p), len(w), _ffi.fp(q)))

    def set_quadrature_table(self, qtable):
        """UniformQuadratureTable or CompactQuadratureTable"""
        if hasattr(qtable, "rules"):  # rule-set table: grouped and walked inside the library (fh_set_quadrature_rules)
            rules = qtable.rules
            offs = np.zeros(len(rules) + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(r[0]) for r in rules])
            w = np.ascontiguousarray(np.concatenate([r[0] for r in rules]), dtype=np.float64)
            p = np.ascontiguousarray(np.concatenate([np.asarray(r[1], dtype=np.float64).reshape(len(r[0]), -1) for r in rules]))
            has_d = [r[2] is not None for r in rules]
            if any(has_d) and not all(has_d):
                raise ValueError("either every rule carries data or none does")
            d = np.ascontiguousarray(np.concatenate([r[2] for r in rules]), dtype=np.float64) if all(has_d) else None
            emap = _ffi.as_u64(qtable.element_to_rule_map)
            self._keep_q = (offs, w, p, d, emap)
            self._check(self._lib.fh_set_quadrature_rules(self._h, len(rules), _ffi.up(offs), _ffi.fp(w), _ffi.fp(p), _ffi.fp(d),
                                                          _ffi.up(emap)))
        elif hasattr(qtable, "rule_params"):
            w, p = _ffi.as_f64(qtable.weights), _ffi.as_f64(qtable.points)
            self._keep_q = (w, p, qtable.rule_params, qtable.element_to_rule_map)
            self._check(self._lib.fh_set_quadrature_compact(self._h, _ffi.fp(w), _ffi.fp(p), len(w), len(qtable.rule_params),
                                                            _ffi.fp(qtable.rule_params), _ffi.up(qtable.element_to_rule_map)))
        else:
            self.set_quadrature_uniform(qtable.weights, qtable.points, qtable.data)

    def update_vertices(self, vertices):
        """new coordinates for the mesh that is set (same connectivity: pattern and topology tables stay)"""
        self._check(self._lib.fh_update_vertices(self._h, _ffi.fp(_ffi.as_f64(np.ascontiguousarray(vertices)))))

    def set_u(self, u):
        if u is None:
            self._check(self._lib.fh_set_u(self._h, None))
        elif _is_torch(u):
            self._check(self._lib.fh_set_u_dev(self._h, C.c_void_p(u.data_ptr())))
        else:
            self._check(self._lib.fh_set_u(self._h, _ffi.fp(_ffi.as_f64(u))))

    # queries
    def solution_dim(self):
        return int(self._lib.fh_solution_dim(self._h))

    def num_elements(self):
        return int(self._lib.fh_num_elements(self._h))

    def num_nodes(self):
        return int(self._lib.fh_num_nodes(self._h))

    def num_rows(self):
        return int(self._lib.fh_num_rows(self._h))

    def nnz(self):
        return int(self._lib.fh_nnz(self._h))

    def last_kernel_name(self):
        return (self._lib.fh_last_kernel_name(self._h) or b"").decode()

    def synchronize(self):
        self._check(self._lib.fh_synchronize(self._h))

    def set_affine_tolerance(self, rel_tol: float):
        """fh_set_affine_tolerance: 0 switches the affine-element fast path off."""
        self._check(self._lib.fh_set_affine_tolerance(self._h, float(rel_tol)))

    def affine_stats(self):
        """(elements found affine, node blocks on the affine kernel, node blocks on the general kernels)"""
        a, b, g = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.fh_affine_stats(self._h, C.byref(a), C.byref(b), C.byref(g)))
        return int(a.value), int(b.value), int(g.value)

    # pattern
    def pattern(self, want_cols=True):
        R = self.num_rows()
        ro = np.zeros(R + 1, dtype=np.uint64)
        nnz = C.c_uint64()
        self._check(self._lib.fh_pattern(self._h, _ffi.up(ro), C.byref(nnz)))
        if not want_cols:
            return ro, None
        ci = np.zeros(max(int(nnz.value), 1), dtype=np.uint64)
        self._check(self._lib.fh_pattern_cols(self._h, _ffi.up(ci)))
        return ro, ci[: int(nnz.value)]

    def build_pattern(self):
        nnz = C.c_uint64()
        self._check(self._lib.fh_pattern(self._h, None, C.byref(nnz)))
        return int(nnz.value)

    def pattern_dev(self, row_offsets_t=None, col_indices_t=None):
        self._check(self._lib.fh_pattern_dev(self._h, _ptr(row_offsets_t), _ptr(col_indices_t)))

    # colouring
    def color(self):
        E = self.num_elements()
        nc = C.c_uint64()
        co = np.zeros(E + 2, dtype=np.uint64)
        lab = np.zeros(max(E, 1), dtype=np.uint64)
        self._check(self._lib.fh_color(self._h, C.byref(nc), _ffi.up(co), _ffi.up(lab)))
        return DisjointSubsetsColors(co[: nc.value + 1].copy(), lab[:E].copy())

    def color_parallel(self):
        """fh_color_parallel: a valid element colouring computed on the device (not the reference's sequential greedy one)"""
        E = self.num_elements()
        nc = C.c_uint64()
        co = np.zeros(E + 2, dtype=np.uint64)
        lab = np.zeros(max(E, 1), dtype=np.uint64)
        self._check(self._lib.fh_color_parallel(self._h, C.byref(nc), _ffi.up(co), _ffi.up(lab)))
        return DisjointSubsetsColors(co[: nc.value + 1].copy(), lab[:E].copy())

    def set_colors(self, colors: "DisjointSubsetsColors"):
        co, lab = _ffi.as_u64(colors.color_offsets), _ffi.as_u64(colors.labels)
        lab_p = lab if len(lab) else np.zeros(1, dtype=np.uint64)
        self._check(self._lib.fh_set_colors(self._h, len(co) - 1, _ffi.up(co), _ffi.up(lab_p)))

    # numeric
    def assemble_matrix(self, values, flags=SCATTER_ATOMIC):
        failed = C.c_uint64(0)
        if _is_torch(values):
            rc = self._lib.fh_assemble_matrix_dev(self._h, C.c_void_p(values.data_ptr()), flags, C.byref(failed))
        else:
            assert values.dtype == np.float64 and values.flags.c_contiguous
            rc = self._lib.fh_assemble_matrix(self._h, _ffi.fp(values), flags, C.byref(failed))
        self._check(rc, failed)

    def assemble_matrix_async(self, values_t, flags):
        self._check(self._lib.fh_assemble_matrix_async_dev(self._h, C.c_void_p(values_t.data_ptr()), flags))

    def set_option(self, name, value):
        """fh_set_option: a FENRIS_HIP_* switch of this context (value None removes it)"""
        self._check(self._lib.fh_set_option(self._h, name.encode(), None if value is None else str(value).encode()))

    def time_assembly(self, values_t, flags, reps=3):
        """fh_time_assembly_dev: milliseconds per assembly (events on the context's stream), after one untimed assembly"""
        ms = C.c_double(0.0)
        self._check(self._lib.fh_time_assembly_dev(self._h, C.c_void_p(values_t.data_ptr()), flags, int(reps), C.byref(ms)))
        return float(ms.value)

    def tune_placement(self, values_t, flags, tries=3):
        """fh_tune_placement_dev: the better of several allocations of the library's own streamed buffer; (ms before, ms after)"""
        a, b = C.c_double(0.0), C.c_double(0.0)
        self._check(self._lib.fh_tune_placement_dev(self._h, C.c_void_p(values_t.data_ptr()), flags, int(tries), C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def assemble_matrix_rows_async(self, values_t, flags, node_begin, node_end):
        """rows of the nodes [node_begin, node_end) only, with the context's second set of tables (fh_assemble_matrix_rows_async_dev)"""
        self._check(self._lib.fh_assemble_matrix_rows_async_dev(self._h, C.c_void_p(values_t.data_ptr()), flags,
                                                                C.c_uint64(int(node_begin)), C.c_uint64(int(node_end))))

    def assemble_matrix_rows(self, values_t, flags, node_begin, node_end):
        failed = C.c_uint64(0)
        self._check(self._lib.fh_assemble_matrix_rows_dev(self._h, C.c_void_p(values_t.data_ptr()), flags,
                                                          C.c_uint64(int(node_begin)), C.c_uint64(int(node_end)), C.byref(failed)), failed)

    def poll_status(self):
        failed = C.c_uint64(0)
        self._check(self._lib.fh_poll_status(self._h, C.byref(failed)), failed)

    def assemble_vector(self, out):
        failed = C.c_uint64(0)
        if _is_torch(out):
            rc = self._lib.fh_assemble_vector_dev(self._h, C.c_void_p(out.data_ptr()), C.byref(failed))
        else:
            rc = self._lib.fh_assemble_vector(self._h, _ffi.fp(out), C.byref(failed))
        self._check(rc, failed)

    def assemble_vector_async(self, out):
        """fh_assemble_vector_async_dev: enqueue only (device tensor); poll_status() reports a singular element"""
        self._check(self._lib.fh_assemble_vector_async_dev(self._h, C.c_void_p(out.data_ptr())))

    def assemble_source_vector(self, out, solution_dim, g=None, values=None):
        """fh_assemble_source_vector(_dev): out += sum_q w |det J| f phi (source.rs:219-278)"""
        gp = _ffi.fp(np.ascontiguousarray(g, dtype=np.float64)) if g is not None else None
        if _is_torch(out):
            vp = C.c_void_p(values.data_ptr()) if values is not None else None
            rc = self._lib.fh_assemble_source_vector_dev(self._h, solution_dim, gp, vp, C.c_void_p(out.data_ptr()))
        else:
            vp = _ffi.fp(np.ascontiguousarray(values, dtype=np.float64)) if values is not None else None
            rc = self._lib.fh_assemble_source_vector(self._h, solution_dim, gp, vp, _ffi.fp(out))
        self._check(rc)

    def physical_quadrature_points(self, nq):
        d = _ffi.ELEM_DIM[self._mesh.elem_kind]
        x = np.zeros((self.num_elements(), nq, d))
        self._check(self._lib.fh_physical_quadrature_points(self._h, _ffi.fp(x)))
        return x

    def spmv(self, values_t, x_t, y_t):
        self._check(self._lib.fh_spmv_dev(self._h, C.c_void_p(values_t.data_ptr()), C.c_void_p(x_t.data_ptr()),
                                          C.c_void_p(y_t.data_ptr())))

    def cg_solve(self, values, b, x, preconditioner=1, rel_tol=1e-9, max_iter=0):
        """fh_cg_solve(_dev): x is the initial guess on entry and the solution on return; returns the iteration count.
        Raises CgSolveError (carrying the iteration count) like ConjugateGradient::solve_with_guess."""
        it = C.c_uint64(0)
        if _is_torch(values):
            rc = self._lib.fh_cg_solve_dev(self._h, C.c_void_p(values.data_ptr()), C.c_void_p(b.data_ptr()),
                                           C.c_void_p(x.data_ptr()), preconditioner, rel_tol, max_iter, C.byref(it))
        else:
            rc = self._lib.fh_cg_solve(self._h, _ffi.fp(values), _ffi.fp(b), _ffi.fp(x), preconditioner, rel_tol, max_iter,
                                       C.byref(it))
        if rc in (7, 8, 9):
            raise CgSolveError(rc, (self._lib.fh_last_error(self._h) or b"").decode(), int(it.value))
        self._check(rc)
        return int(it.value)

    def estimate_error_squared(self, which, solution_dim, u_h, exact):
        out = C.c_double()
        names = ("fh_estimate_L2_error_squared", "fh_estimate_H1_seminorm_error_squared")
        if _is_torch(u_h):
            fn = getattr(self._lib, names[which] + "_dev")
            rc = fn(self._h, solution_dim, C.c_void_p(u_h.data_ptr()), C.c_void_p(exact.data_ptr()), C.byref(out))
        else:
            fn = getattr(self._lib, names[which])
            rc = fn(self._h, solution_dim, _ffi.fp(np.ascontiguousarray(u_h, dtype=np.float64)),
                    _ffi.fp(np.ascontiguousarray(exact, dtype=np.float64)), C.byref(out))
        failed = C.c_uint64(0)
        self._check(rc, failed)
        return out.value

    def assemble_scalar(self):
        out, failed = C.c_double(), C.c_uint64(0)
        self._check(self._lib.fh_assemble_scalar(self._h, C.byref(out), C.byref(failed)), failed)
        return out.value

    def element_matrices(self, first, count):
        s, n = self.solution_dim(), _ffi.ELEM_NODES[self._mesh.elem_kind]
        ld = s * n
        out = np.zeros((count, ld, ld))
        self._check(self._lib.fh_assemble_element_matrices(self._h, first, count, _ffi.fp(out)))
        return out.transpose(0, 2, 1).copy()  # column-major blocks -> [e][row][col]

    def apply_dirichlet_csr_dev(self, values_t, nodes):
        nodes = _ffi.as_u64(nodes)
        self._check(self._lib.fh_apply_dirichlet_csr_dev(self._h, C.c_void_p(values_t.data_ptr()), _ffi.up(nodes), len(nodes)))

    def apply_dirichlet_rhs_dev(self, rhs_t, nodes):
        nodes = _ffi.as_u64(nodes)
        self._check(self._lib.fh_apply_dirichlet_rhs_dev(self._h, C.c_void_p(rhs_t.data_ptr()), _ffi.up(nodes), len(nodes)))


class CgSolveError(FenrisError):
    """SolveError of fenris-sparse/src/cg.rs:277-318 (kind + iterations so far)"""

    KINDS = {7: "MaxIterationsReached", 8: "IndefiniteOperator", 9: "IndefinitePreconditioner"}

    def __init__(self, code, message, num_iterations):
        super().__init__(code, message)
        self.kind = self.KINDS.get(code, "?")
        self.num_iterations = num_iterations


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "device")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


# ------------------------------------------------------------------------------------------ tables
class UniformQuadratureTable:
    """src/assembly/local/quadrature_table.rs:213-298"""

    def __init__(self, points, weights, data=None):
        self.points = _ffi.as_f64(points)
        self.weights = _ffi.as_f64(weights)
        self.data = data  # nq x 2 (mu, lambda) or None

    @classmethod
    def from_points_and_weights(cls, points, weights):
        """NOTE the argument order (points, weights) -- a rule is (weights, points); quadrature_table.rs:232"""
        return cls(points, weights)

    @classmethod
    def from_quadrature(cls, rule):
        weights, points = rule
        return cls(points, weights)

    def with_uniform_data(self, data):
        """quadrature_table.rs:264-266"""
        pair = data.as_pair() if hasattr(data, "as_pair") else tuple(data)
        return UniformQuadratureTable(self.points, self.weights, np.tile(np.asarray(pair, dtype=np.float64), (len(self.weights), 1)))

    def with_data(self, data):
        """quadrature_table.rs:252-262: one Parameters value per quadrature point"""
        arr = np.array([d.as_pair() if hasattr(d, "as_pair") else tuple(d) for d in data], dtype=np.float64)
        assert arr.shape == (len(self.weights), 2)
        return UniformQuadratureTable(self.points, self.weights, arr)


class CompactQuadratureTable:
    """src/assembly/local/quadrature_table.rs:300-439 (``from_quadrature_rules_and_map``) for rules that share one
    set of points and weights and differ in their data: ``rule_data[r]`` holds one Parameters value per point (or a
    single value, repeated), ``element_to_rule_map[e]`` picks the rule of element e.  Piecewise material data."""

    def __init__(self, points, weights, rule_data, element_to_rule_map):
        self.points = _ffi.as_f64(points)
        self.weights = _ffi.as_f64(weights)
        nq = len(self.weights)
        rules = []
        for rd in rule_data:
            if hasattr(rd, "as_pair"):
                rd = [rd] * nq
            arr = np.array([d.as_pair() if hasattr(d, "as_pair") else tuple(d) for d in rd], dtype=np.float64)
            if arr.shape != (nq, 2):
                raise ValueError("every rule needs one data value per quadrature point")  # check_rules_consistency
            rules.append(arr)
        self.rule_params = np.ascontiguousarray(np.stack(rules))
        self.element_to_rule_map = _ffi.as_u64(element_to_rule_map)
        if len(self.element_to_rule_map) and int(self.element_to_rule_map.max()) >= len(rules):
            raise ValueError("Each rule index must correspond to a provided quadrature rule.")  # quadrature_table.rs:366-372
        self.data = self.rule_params[0]

    @classmethod
    def from_quadrature_rules_and_map(cls, points, weights, data, element_to_rule_map):
        return cls(points, weights, data, element_to_rule_map)


class _RuleSetTable:
    """Quadrature tables whose rules differ in their POINTS (GeneralQuadratureTable, and CompactQuadratureTable with
    different point sets; quadrature_table.rs:57-210, 300-439).  The device kernels stage one rule per launch, so the
    global assemblers walk the distinct rules: uniform table of rule r + the element mask of the elements that use it
    (fh_set_active_elements), accumulating into the same output.  Identical rules are merged."""

    def __init__(self, rules, element_to_rule_map):
        self.rules = rules  # list of (weights, points, data (nq, 2) or None)
        self.element_to_rule_map = _ffi.as_u64(element_to_rule_map)
        self.weights, self.points, self.data = rules[0]

    @staticmethod
    def _norm(points, weights, data):
        w, p = _ffi.as_f64(weights), _ffi.as_f64(points)
        if data is None:
            d = None
        else:
            d = np.array([x.as_pair() if hasattr(x, "as_pair") else tuple(x) for x in data], dtype=np.float64).reshape(len(w), 2)
        if len(p) != len(w):
            raise ValueError("Rules must have an equal number of points and weights.")  # check_rules_consistency
        return w, p, d

    @classmethod
    def build(cls, per_entry_rules, entry_of_element):
        """dedupe identical (points, weights, data) triples"""
        keys, rules, remap = {}, [], []
        for w, p, d in per_entry_rules:
            key = (w.tobytes(), p.tobytes(), None if d is None else d.tobytes())
            if key not in keys:
                keys[key] = len(rules)
                rules.append((w, p, d))
            remap.append(keys[key])
        remap = np.asarray(remap, dtype=np.uint64)
        return cls(rules, remap[np.asarray(entry_of_element, dtype=np.int64)])


class GeneralQuadratureTable(_RuleSetTable):
    """src/assembly/local/quadrature_table.rs:57-210: one rule (points, weights, data) per element"""

    @classmethod
    def from_points_and_weights(cls, points, weights):
        return cls.from_points_weights_and_data(points, weights, None)

    @classmethod
    def from_points_weights_and_data(cls, points, weights, data):
        n = len(weights)
        if len(points) != n or (data is not None and len(data) != n):
            raise ValueError("Quadrature point arrays must have the same number of rules.")
        rules = [cls._norm(points[e], weights[e], None if data is None else data[e]) for e in range(n)]
        return cls.build(rules, np.arange(n))


def compact_quadrature_table(points, weights, data, element_to_rule_map):
    """CompactQuadratureTable::from_quadrature_rules_and_map (quadrature_table.rs:357-383) for NestedVec inputs: rules
    that share points and weights take the single-launch device table (CompactQuadratureTable), anything else is walked
    rule by rule."""
    rules = [_RuleSetTable._norm(points[r], weights[r], None if data is None else data[r]) for r in range(len(weights))]
    emap = _ffi.as_u64(element_to_rule_map)
    if len(emap) and int(emap.max()) >= len(rules):
        raise ValueError("Each rule index must correspond to a provided quadrature rule.")
    same = all(np.array_equal(r[0], rules[0][0]) and np.array_equal(r[1], rules[0][1]) for r in rules)
    if same and all(r[2] is not None for r in rules):
        return CompactQuadratureTable(rules[0][1], rules[0][0], [r[2] for r in rules], emap)
    return _RuleSetTable.build(rules, emap)


@dataclass
class DisjointSubsetsColors:
    """Vec<DisjointSubsets> (fenris-paradis/src/lib.rs:171-181) flattened: elements of colour c are
    labels[color_offsets[c]:color_offsets[c+1]] in ascending order."""
    color_offsets: np.ndarray
    labels: np.ndarray

    def __len__(self):
        return len(self.color_offsets) - 1

    def color(self, c):
        return self.labels[int(self.color_offsets[c]): int(self.color_offsets[c + 1])]


class CsrMatrix:
    """nalgebra_sparse::CsrMatrix triple (row_offsets, col_indices, values); values may be numpy or a
    torch tensor on the engine's device."""

    def __init__(self, row_offsets, col_indices, values):
        self.row_offsets, self.col_indices, self.values = row_offsets, col_indices, values

    def nnz(self):
        return len(self.values)

    def nrows(self):
        return len(self.row_offsets) - 1

    def to_scipy(self):
        import scipy.sparse as sp

        v = self.values.cpu().numpy() if _is_torch(self.values) else self.values
        n = self.nrows()
        return sp.csr_matrix((v, self.col_indices.astype(np.int64), self.row_offsets.astype(np.int64)), shape=(n, n))


# ------------------------------------------------------------------------------------------ local
class ElementEllipticAssemblerBuilder:
    """src/assembly/local/elliptic.rs:63-150"""

    def __init__(self, engine: Optional[Engine] = None):
        self._engine, self._space, self._op, self._qtable, self._u, self._has_u = engine, None, None, None, None, False

    @classmethod
    def new(cls, engine: Optional[Engine] = None):
        return cls(engine)

    def with_finite_element_space(self, space: Mesh):
        self._space = space
        return self

    def with_operator(self, op):
        self._op = op
        return self

    def with_quadrature_table(self, qtable: UniformQuadratureTable):
        self._qtable = qtable
        return self

    def with_u(self, u):
        self._u, self._has_u = u, True
        return self

    def build(self) -> "ElementEllipticAssembler":
        if self._space is None or self._op is None or self._qtable is None or not self._has_u:
            raise ValueError("builder incomplete: space, operator, quadrature table and u are all required")
        return ElementEllipticAssembler(self._engine or Engine(), self._space, self._op, self._qtable, self._u)


class ElementEllipticAssembler(_Composable):
    """ElementEllipticAssembler<Mesh, Op, UniformQuadratureTable> (elliptic.rs:152-340), device resident."""

    def __init__(self, engine, space, op, qtable, u):
        self.engine, self.space, self.op, self.qtable = engine, space, op, qtable
        engine.set_mesh(space)
        engine.set_operator(op.op_kind)
        s = engine.solution_dim()
        if u is not None and not _is_torch(u):
            u = _ffi.as_f64(u)
            if len(u) != s * space.num_nodes():
                raise ValueError("Local element dofs (u) dimension mismatch")  # elliptic.rs:385-389
        engine.set_quadrature_table(qtable)
        if op.op_kind == _ffi.TENSOR:   # the operator's data: one tensor per point of the table just set
            engine.set_operator_tensor(op.tensors_for(len(qtable.weights), space.vertices.shape[1]), op.symmetric)
        engine.set_u(u)

    # ElementConnectivityAssembler (src/assembly/local.rs:18-47)
    def solution_dim(self):
        return self.engine.solution_dim()

    def num_elements(self):
        return self.engine.num_elements()

    def num_nodes(self):
        return self.engine.num_nodes()

    def element_node_count(self, _element_index):
        return _ffi.ELEM_NODES[self.space.elem_kind]

    def populate_element_nodes(self, output, element_index):
        output[:] = self.space.connectivity[element_index]

    # ElementMatrixAssembler::assemble_element_matrix_into (local.rs:78)
    def assemble_element_matrix(self, element_index):
        return self.engine.element_matrices(element_index, 1)[0]

    def with_u(self, u):
        self.engine.set_u(u)
        return self


class ElementMassAssembler(ElementEllipticAssembler):
    """ElementMassAssembler (src/assembly/local/mass.rs:48-160): ``with_solution_dim(s).with_space(mesh)
    .with_quadrature_table(table)`` where the table's data is ``Density`` per point.  s must be 1 or the geometry
    dimension.  Only the matrix form exists (ElementMatrixAssembler)."""

    def __init__(self, solution_dim, engine=None):
        self._sdim, self._engine0, self._space0, self._qt0 = solution_dim, engine, None, None

    @classmethod
    def with_solution_dim(cls, solution_dim, engine: Optional[Engine] = None):
        return cls(solution_dim, engine)

    def with_space(self, space: Mesh):
        self._space0 = space
        return self._maybe_build()

    def with_quadrature_table(self, qtable: "UniformQuadratureTable"):
        self._qt0 = qtable
        return self._maybe_build()

    def _maybe_build(self):
        if self._space0 is None or self._qt0 is None:
            return self
        d = _ffi.ELEM_DIM[self._space0.elem_kind]
        if self._sdim not in (1, d):
            raise ValueError("solution_dim must be 1 or the geometry dimension")
        if self._qt0.data is None:
            raise ValueError("the mass assembler needs a Density per quadrature point")

        class _Op:
            op_kind = _ffi.MASS_SCALAR if self._sdim == 1 else _ffi.MASS_VECTOR

        ElementEllipticAssembler.__init__(self, self._engine0 or Engine(), self._space0, _Op(), self._qt0, None)
        return self


class ElementSourceAssemblerBuilder:
    """src/assembly/local/source.rs:24-94"""

    def __init__(self, engine: Optional[Engine] = None):
        self._engine, self._space, self._source, self._qtable = engine, None, None, None

    @classmethod
    def new(cls, engine: Optional[Engine] = None):
        return cls(engine)

    def with_finite_element_space(self, space: Mesh):
        self._space = space
        return self

    def with_source(self, source):
        self._source = source
        return self

    def with_quadrature_table(self, qtable: "UniformQuadratureTable"):
        self._qtable = qtable
        return self

    def build(self) -> "ElementSourceAssembler":
        if self._space is None or self._source is None or self._qtable is None:
            raise ValueError("space, source and quadrature table are required")
        return ElementSourceAssembler(self._engine or Engine(), self._space, self._source, self._qtable)


class ElementSourceAssembler(_Composable):
    """src/assembly/local/source.rs:96-216: an ElementVectorAssembler for the (f, v) term.  ``source`` is a
    GravitySource (uniform Density table) or a SourceFunction (sampled on the host at the physical quadrature
    points, integrated on the device)."""

    def __init__(self, engine, space, source, qtable):
        self.engine, self.space, self.source, self.qtable = engine, space, source, qtable
        d = _ffi.ELEM_DIM[space.elem_kind]
        if source.solution_dim not in (1, d):
            raise ValueError("solution_dim must be 1 or the geometry dimension")
        engine.set_mesh(space)
        engine.set_quadrature_table(qtable)

    def solution_dim(self):
        return self.source.solution_dim

    def num_elements(self):
        return self.space.num_elements()

    def num_nodes(self):
        return self.space.num_nodes()

    def element_node_count(self, _element_index):
        return _ffi.ELEM_NODES[self.space.elem_kind]

    def populate_element_nodes(self, output, element_index):
        output[:] = self.space.connectivity[element_index]

    def assemble_vector_into_engine(self, output):
        s = self.source.solution_dim
        if hasattr(self.source, "gravitational_acceleration"):
            if self.qtable.data is None:
                raise ValueError("GravitySource needs a Density per quadrature point")
            self.engine.assemble_source_vector(output, s, g=self.source.gravitational_acceleration)
            return
        x = self.engine.physical_quadrature_points(len(self.qtable.weights))
        vals = np.ascontiguousarray(self.source.evaluate(x, self.qtable.data), dtype=np.float64)
        if vals.shape != (self.num_elements(), len(self.qtable.weights), s):
            raise ValueError("source values must have shape (E, nq, solution_dim)")
        if _is_torch(output):
            import torch

            vals_t = torch.from_numpy(vals).to(output.device)
            self.engine.assemble_source_vector(output, s, values=vals_t)
        else:
            self.engine.assemble_source_vector(output, s, values=vals)


class MockElementAssembler:
    """Generic ElementConnectivityAssembler with ragged node lists
    (tests/unit_tests/assembly/global.rs MockElementAssembler)."""

    def __init__(self, solution_dim, num_nodes, element_connectivities, engine: Optional[Engine] = None):
        self.engine = engine or Engine()
        offs = np.cumsum([0] + [len(c) for c in element_connectivities]).astype(np.uint64)
        nodes = np.array([x for c in element_connectivities for x in c], dtype=np.uint64)
        self.engine.set_connectivity_ragged(solution_dim, num_nodes, offs, nodes)


# ------------------------------------------------------------------------------------------ global
class CsrAssembler:
    """src/assembly/global.rs:24-183.  ``scatter`` picks the device strategy (default: atomic adds)."""

    def __init__(self, scatter=SCATTER_ATOMIC):
        self.scatter = scatter

    def assemble_pattern(self, element_assembler):
        """global.rs:65-120 -> (row_offsets, col_indices) as uint64 arrays"""
        if is_composed(element_assembler):
            return _compose.aggregate_pattern(element_assembler)
        return element_assembler.engine.pattern()

    def assemble(self, element_assembler, device_values=False):
        """global.rs:124-131"""
        if is_composed(element_assembler):  # AggregateElementAssembler / MapElementNodes / TransformElementMatrix
            import torch

            ro, ci = _compose.aggregate_pattern(element_assembler)
            values = torch.zeros(len(ci), dtype=torch.float64, device="cuda")
            _compose.assemble_matrix_into(ro, ci, values, element_assembler, self.scatter)
            return CsrMatrix(ro, ci, values if device_values else values.cpu().numpy())
        eng = element_assembler.engine
        ro, ci = eng.pattern()
        if device_values:
            import torch

            values = torch.zeros(len(ci), dtype=torch.float64, device=f"cuda:{eng.device}")
        else:
            values = np.zeros(len(ci))
        csr = CsrMatrix(ro, ci, values)
        self.assemble_into_csr(csr, element_assembler)
        return csr

    def assemble_into_csr(self, csr: CsrMatrix, element_assembler):
        """global.rs:133-182: accumulates into csr.values"""
        if is_composed(element_assembler):
            import torch

            on_dev = _is_torch(csr.values)
            values = csr.values if on_dev else torch.from_numpy(np.ascontiguousarray(csr.values)).cuda()
            _compose.assemble_matrix_into(csr.row_offsets, csr.col_indices, values, element_assembler, self.scatter)
            if not on_dev:
                csr.values[:] = values.cpu().numpy()
            return
        eng = element_assembler.engine
        if eng.nnz() == 0 and len(csr.values):
            eng.build_pattern()
        if len(csr.values) != eng.nnz():
            raise ValueError("CSR matrix does not have the pattern of this element assembler")
        flags = self.scatter

        if flags == SCATTER_COLORED:
            eng.color()
        eng.assemble_matrix(csr.values, flags)


class CsrParAssembler:
    """src/assembly/global.rs:185-377: colour-by-colour scatter without atomics."""

    def assemble_pattern(self, element_assembler):
        return element_assembler.engine.pattern()

    def assemble(self, colors: DisjointSubsetsColors, element_assembler):
        eng = element_assembler.engine
        ro, ci = eng.pattern()
        csr = CsrMatrix(ro, ci, np.zeros(len(ci)))
        self.assemble_into_csr(csr, colors, element_assembler)
        return csr

    def assemble_into_csr(self, csr: CsrMatrix, colors: DisjointSubsetsColors, element_assembler):
        eng = element_assembler.engine
        if len(csr.values) != eng.nnz():
            raise ValueError("CSR matrix does not have the pattern of this element assembler")
        eng.set_colors(colors)
        eng.assemble_matrix(csr.values, SCATTER_COLORED)


class VectorAssembler:
    """src/assembly/global.rs:569-616"""

    def assemble_vector(self, element_assembler):
        out = np.zeros(element_assembler.solution_dim() * element_assembler.num_nodes())
        self.assemble_vector_into(out, element_assembler)
        return out

    def assemble_vector_into(self, output, element_assembler):
        n = element_assembler.solution_dim() * element_assembler.num_nodes()
        if (output.numel() if _is_torch(output) else len(output)) != n:
            raise ValueError("Output dimensions mismatch")  # global.rs:592
        if is_composed(element_assembler):
            import torch

            on_dev = _is_torch(output)
            out_t = output if on_dev else torch.from_numpy(np.ascontiguousarray(output)).cuda()
            _compose.assemble_vector_into(out_t, element_assembler)
            if not on_dev:
                output[:] = out_t.cpu().numpy()
        elif hasattr(element_assembler, "assemble_vector_into_engine"):  # ElementSourceAssembler
            element_assembler.assemble_vector_into_engine(output)
        else:
            element_assembler.engine.assemble_vector(output)


class VectorParAssembler(VectorAssembler):
    """src/assembly/global.rs:618-686 (colours are not needed on the device: atomic adds)"""

    def assemble_vector(self, colors, element_assembler):  # noqa: D401 - signature of the reference
        return VectorAssembler.assemble_vector(self, element_assembler)


def assemble_scalar(element_assembler):
    """global.rs:697-711"""
    if is_composed(element_assembler):
        return _compose.assemble_scalar(element_assembler)
    return float(element_assembler.engine.assemble_scalar())


def color_nodes(mesh_or_assembler, engine: Optional[Engine] = None) -> DisjointSubsetsColors:
    """global.rs:540-551 + sequential_greedy_coloring (fenris-paradis/src/coloring.rs:6-70)"""
    if hasattr(mesh_or_assembler, "engine"):
        return mesh_or_assembler.engine.color()
    eng = engine or Engine()
    eng.set_mesh(mesh_or_assembler)
    return eng.color()


def apply_homogeneous_dirichlet_bc_csr(csr: CsrMatrix, nodes, solution_dim, element_assembler):
    """global.rs:379-451 on device-resident values"""
    assert solution_dim == element_assembler.solution_dim()
    eng = element_assembler.engine
    if _is_torch(csr.values):
        eng.apply_dirichlet_csr_dev(csr.values, nodes)
    else:
        import torch

        t = torch.from_numpy(csr.values).to(f"cuda:{eng.device}")
        eng.apply_dirichlet_csr_dev(t, nodes)
        csr.values[:] = t.cpu().numpy()


def apply_homogeneous_dirichlet_bc_rhs(rhs, nodes, solution_dim):
    """global.rs:479-495 (host arrays: trivial)"""
    nodes = np.asarray(nodes, dtype=np.int64)
    for i in range(solution_dim):
        rhs[solution_dim * nodes + i] = 0.0


# ------------------------------------------------------------------------------------------ solve + error estimation
class RelativeResidualCriterion:
    """fenris-sparse/src/cg.rs:86-124: ||r|| <= tol ||b|| on CG's own residual (default 1e-8 for f64)"""

    def __init__(self, tol=1e-8):
        self.tol = float(tol)


class JacobiPreconditioner:
    """the inverse diagonal, built on the device (matrix.diagonal_as_csr() + recip, poisson_mms_common.rs:148-151)"""


class IdentityOperator:
    """fenris-sparse/src/cg.rs:54-61"""


class ConjugateGradient:
    """fenris-sparse/src/cg.rs:196-478, builder style.  The operator is the CSR matrix assembled by an element
    assembler of this package (its engine holds the pattern); values, right-hand side and solution may be numpy
    arrays or device tensors (all of one kind)."""

    def __init__(self):
        self._csr, self._asm, self._pre, self._crit, self._max_iter = None, None, IdentityOperator(), None, 0

    @classmethod
    def new(cls):
        return cls()

    def with_operator(self, csr: "CsrMatrix", element_assembler):
        self._csr, self._asm = csr, element_assembler
        return self

    def with_preconditioner(self, preconditioner):
        self._pre = preconditioner
        return self

    def with_max_iter(self, max_iter):
        self._max_iter = int(max_iter)
        return self

    def with_stopping_criterion(self, criterion: RelativeResidualCriterion):
        self._crit = criterion
        return self

    def solve_with_guess(self, b, x):
        """x: initial guess, overwritten by the solution; returns CgOutput.num_iterations"""
        if self._csr is None or self._crit is None:
            raise ValueError("operator and stopping criterion are required")
        pre = 1 if isinstance(self._pre, JacobiPreconditioner) else 0
        return self._asm.engine.cg_solve(self._csr.values, b, x, pre, self._crit.tol, self._max_iter)


def _sample(space_assembler, fn, nq):
    x = space_assembler.engine.physical_quadrature_points(nq)
    return np.ascontiguousarray(fn(x), dtype=np.float64)


def estimate_L2_error_squared(element_assembler, u, u_h, solution_dim=None):
    """src/error.rs:287-311.  ``u(x)`` is vectorised: (E, nq, d) physical points -> (E, nq, s) values."""
    s = solution_dim or element_assembler.solution_dim()
    nq = len(element_assembler.qtable.weights)
    return element_assembler.engine.estimate_error_squared(0, s, u_h, _sample(element_assembler, u, nq))


def estimate_L2_error(element_assembler, u, u_h, solution_dim=None):
    """src/error.rs:313-328"""
    return float(np.sqrt(estimate_L2_error_squared(element_assembler, u, u_h, solution_dim)))


def estimate_H1_seminorm_error_squared(element_assembler, u_grad, u_h, solution_dim=None):
    """src/error.rs:330-354.  ``u_grad(x)``: (E, nq, d) points -> (E, nq, d, s) with [i][k] = d u_k / d x_i."""
    s = solution_dim or element_assembler.solution_dim()
    nq = len(element_assembler.qtable.weights)
    return element_assembler.engine.estimate_error_squared(1, s, u_h, _sample(element_assembler, u_grad, nq))


def estimate_H1_seminorm_error(element_assembler, u_grad, u_h, solution_dim=None):
    """src/error.rs:356-372"""
    return float(np.sqrt(estimate_H1_seminorm_error_squared(element_assembler, u_grad, u_h, solution_dim)))
