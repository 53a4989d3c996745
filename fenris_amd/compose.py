"""Composition of element assemblers (src/assembly/local.rs:152-340): ``AggregateElementAssembler``, ``MapElementNodes`` and
``TransformElement{Scalar,Vector,Matrix}`` for the closed family the device can express (a scale factor instead of an
arbitrary closure).  Every body keeps its own engine -- mesh, operator, quadrature table, element kind, its fastest kernels;
the aggregate's pattern comes from the union of the mapped connectivities (``fh_set_connectivity_ragged``), and what a body
assembled in its own node numbering is added, scaled, into the aggregate's matrix / vector on the device
(``fh_add_mapped_matrix_dev`` / ``fh_add_mapped_vector_dev``).  The global assemblers of ``assembly.py`` accept these
objects wherever they accept an element assembler."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi


class _Composable:
    """mix-in: the adapter methods of the reference's ``ElementAssemblerTransformations`` extension trait"""

    def map_element_nodes(self, num_nodes, function):
        """assembler.map_element_nodes(num_nodes, |node| ...) (local.rs:300-340): `function` is a callable or an index array"""
        return MapElementNodes(self, num_nodes, function)

    def transform_element_scalar(self, scale):
        return TransformElementScalar(self, float(scale))

    def transform_element_vector(self, scale):
        return TransformElementVector(self, float(scale))

    def transform_element_matrix(self, scale):
        return TransformElementMatrix(self, float(scale))


class _Wrapper(_Composable):
    def __init__(self, assembler):
        self.assembler = assembler

    def solution_dim(self):
        return self.assembler.solution_dim()

    def num_elements(self):
        return self.assembler.num_elements()

    def num_nodes(self):
        return self.assembler.num_nodes()

    def element_node_count(self, e):
        return self.assembler.element_node_count(e)

    def populate_element_nodes(self, output, e):
        self.assembler.populate_element_nodes(output, e)

    # a body = (base assembler with an engine, node map or None, scales of scalar / vector / matrix)
    def _bodies(self):
        return _bodies(self.assembler)


class MapElementNodes(_Wrapper):
    """local.rs:300-340: the element nodes pass through `function`; `num_nodes` is the size of the target index space"""

    def __init__(self, assembler, num_nodes, function):
        super().__init__(assembler)
        self._num_nodes = int(num_nodes)
        n = assembler.num_nodes()
        self.node_map = (np.array([function(i) for i in range(n)], dtype=np.uint64) if callable(function)
                         else np.ascontiguousarray(function, dtype=np.uint64))
        if len(self.node_map) != n:
            raise ValueError("node map must have one entry per node of the wrapped assembler")

    def num_nodes(self):
        return self._num_nodes

    def populate_element_nodes(self, output, e):
        self.assembler.populate_element_nodes(output, e)
        output[:] = self.node_map[np.asarray(output, dtype=np.int64)]

    def _bodies(self):
        out = []
        for base, m, ss, sv, sm in _bodies(self.assembler):
            out.append((base, self.node_map if m is None else self.node_map[m.astype(np.int64)], ss, sv, sm))
        return out


class _Scale(_Wrapper):
    WHICH = 0

    def __init__(self, assembler, scale):
        super().__init__(assembler)
        self.scale = scale

    def _bodies(self):
        out = []
        for base, m, ss, sv, sm in _bodies(self.assembler):
            s = [ss, sv, sm]
            s[self.WHICH] *= self.scale
            out.append((base, m, *s))
        return out


class TransformElementScalar(_Scale):
    """transform_element_scalar(|s| Ok(scale * s))"""
    WHICH = 0


class TransformElementVector(_Scale):
    """transform_element_vector(|mut v| { v *= scale; Ok(()) })"""
    WHICH = 1


class TransformElementMatrix(_Scale):
    """transform_element_matrix(|mut m| { m *= scale; Ok(()) })"""
    WHICH = 2


class AggregateElementAssembler(_Composable):
    """local.rs:152-267: the elements of several assemblers over ONE node index space, one after the other"""

    def __init__(self, assemblers):
        assemblers = list(assemblers)
        if not assemblers:
            raise ValueError("Must have at least one assembler in aggregate")
        if any(a.solution_dim() != assemblers[0].solution_dim() for a in assemblers):
            raise ValueError("All assemblers must have the same solution dimension")
        if any(a.num_nodes() != assemblers[0].num_nodes() for a in assemblers):
            raise ValueError("All assemblers must share the same node index space (same num_nodes)")
        self.assemblers = assemblers
        self.element_offsets = np.cumsum([0] + [a.num_elements() for a in assemblers])

    @classmethod
    def from_assemblers(cls, assemblers):
        return cls(assemblers)

    def solution_dim(self):
        return self.assemblers[0].solution_dim()

    def num_nodes(self):
        return self.assemblers[0].num_nodes()

    def num_elements(self):
        return int(self.element_offsets[-1])

    def _find(self, e):
        k = int(np.searchsorted(self.element_offsets, e, side="right")) - 1
        return self.assemblers[k], e - int(self.element_offsets[k])

    def element_node_count(self, e):
        a, le = self._find(e)
        return a.element_node_count(le)

    def populate_element_nodes(self, output, e):
        a, le = self._find(e)
        a.populate_element_nodes(output, le)

    def _bodies(self):
        return [b for a in self.assemblers for b in _bodies(a)]


def _bodies(assembler):
    if hasattr(assembler, "_bodies"):
        return assembler._bodies()
    return [(assembler, None, 1.0, 1.0, 1.0)]


def is_composed(assembler):
    return hasattr(assembler, "_bodies")


def _connectivity(base):
    """(offsets, nodes) of a base assembler's elements in its own numbering"""
    space = getattr(base, "space", None)
    if space is not None:
        c = np.ascontiguousarray(space.connectivity, dtype=np.uint64)
        return np.arange(0, c.size + 1, c.shape[1], dtype=np.uint64), c.reshape(-1)
    offs, nodes = [0], []
    for e in range(base.num_elements()):
        out = np.zeros(base.element_node_count(e), dtype=np.uint64)
        base.populate_element_nodes(out, e)
        nodes.append(out)
        offs.append(offs[-1] + len(out))
    return np.asarray(offs, dtype=np.uint64), np.concatenate(nodes) if nodes else np.zeros(0, dtype=np.uint64)


def aggregate_pattern(assembler):
    """assemble_pattern of a composed assembler: (row_offsets, col_indices) over its node space (global.rs:65-120)"""
    from .assembly import Engine

    offs, nodes = [np.zeros(1, dtype=np.uint64)], []
    for base, m, _, _, _ in _bodies(assembler):
        o, n = _connectivity(base)
        if m is not None:
            n = m[n.astype(np.int64)]
        offs.append(o[1:] + offs[-1][-1])
        nodes.append(n)
    eng = Engine()
    try:
        eng.set_connectivity_ragged(assembler.solution_dim(), assembler.num_nodes(), np.concatenate(offs), np.concatenate(nodes))
        return eng.pattern()
    finally:
        eng.close()


def assemble_matrix_into(csr_ro, csr_ci, values_t, assembler, scatter):
    """values_t (torch, on the device of the bodies' engines) += the composed assembler's matrix"""
    import torch

    from .assembly import ASSEMBLE_OVERWRITE

    dev = values_t.device
    ro_t = torch.from_numpy(np.ascontiguousarray(csr_ro).view(np.int64)).to(dev)
    ci_t = torch.from_numpy(np.ascontiguousarray(csr_ci).view(np.int64)).to(dev)
    lib = _ffi.lib()
    for base, m, _, _, sm in _bodies(assembler):
        eng = base.engine
        nnz = eng.build_pattern()
        if scatter is not None and (scatter & 0xf) == 1:  # coloured
            eng.color()
        part = torch.zeros(nnz, dtype=torch.float64, device=dev)
        eng.assemble_matrix(part, scatter | ASSEMBLE_OVERWRITE)
        m_t = None if m is None else torch.from_numpy(m.view(np.int64)).to(dev)
        eng._check(lib.fh_add_mapped_matrix_dev(eng._h, C.c_void_p(part.data_ptr()), C.c_void_p(m_t.data_ptr()) if m_t is not None else None,
                                                float(sm), assembler.num_nodes(), C.c_void_p(ro_t.data_ptr()), C.c_void_p(ci_t.data_ptr()),
                                                C.c_void_p(values_t.data_ptr())))
        torch.cuda.synchronize()


def assemble_vector_into(out_t, assembler):
    import torch

    dev = out_t.device
    lib = _ffi.lib()
    for base, m, _, sv, _ in _bodies(assembler):
        eng = base.engine
        part = torch.zeros(base.solution_dim() * base.num_nodes(), dtype=torch.float64, device=dev)
        # a body may be an ElementSourceAssembler (local/source.rs:159-278: the multi-body right-hand side): its context has no
        # operator, the vector comes from the source and the solution dimension goes along explicitly
        if hasattr(base, "assemble_vector_into_engine"):
            base.assemble_vector_into_engine(part)
        else:
            eng.assemble_vector(part)
        m_t = None if m is None else torch.from_numpy(m.view(np.int64)).to(dev)
        eng._check(lib.fh_add_mapped_vector_sdim_dev(eng._h, C.c_void_p(part.data_ptr()), C.c_void_p(m_t.data_ptr()) if m_t is not None else None,
                                                     float(sv), int(base.solution_dim()), assembler.num_nodes(), C.c_void_p(out_t.data_ptr())))
        torch.cuda.synchronize()


def assemble_scalar(assembler):
    return float(sum(ss * base.engine.assemble_scalar() for base, _, ss, _, _ in _bodies(assembler)))
