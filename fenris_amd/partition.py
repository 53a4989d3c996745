"""Multi-GPU assembly of an ARBITRARY mesh: element partition ``elem_to_part[]`` + exchange of interface rows.

fenris itself is single-process; what this replaces on every rank is ``CsrParAssembler::assemble_into_csr`` (global.rs:314-376) over the
rank's own elements.  The slab partition of ``distributed.py`` is the special case for structured boxes (one contiguous node range per
interface, which lets a rank send ONE segment of its values); here a rank may have any number of neighbours and its interface rows lie
anywhere, so they travel through packed index lists.

* every element belongs to exactly one part (``elem_to_part[e]``); a node is OWNED by the lowest part that has an element touching it
  (nodes without elements: part 0);
* rank r's *extended* local mesh = its own elements + every element that touches a node its own elements touch (one ring of halo
  elements).  Local node numbers are the global ones in ascending order (``l2g``), so column order is preserved and the rows of every
  node that r's own elements touch carry the complete GLOBAL pattern -- identical layouts on all ranks that contribute to such a row;
* numerics run over the own elements only (``Engine.set_active_elements``; ``mode="halo"``: also over the halo elements that touch an
  owned node, which completes the owned rows locally and needs no exchange);
* exchange (``mode="exchange"``): for each neighbour q, r sends its partial rows of the nodes q owns that r's own elements touch, and
  adds what q sends for r's owned nodes; both sides list these nodes in ascending global order, so the buffers line up without any
  index traffic.  Point to point (RCCL ``ncclSend`` / ``ncclRecv`` under ``torch.distributed`` on the GPU box, gloo in the CPU tests):
  never a collective over the matrix.
Concatenating the owned rows of all ranks (mapped through ``l2g``) gives the single-process CSR: indices bit for bit, values to rounding.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from .mesh import Mesh


@dataclass
class PartProblem:
    mesh: Mesh                      # extended local mesh (own + halo elements), local node numbering
    l2g: np.ndarray                 # global node id of every local node (ascending)
    elem_l2g: np.ndarray            # global element id of every local element
    active: np.ndarray              # uint8 per local element: 1 = its numerics run on this rank
    owned: np.ndarray               # local ids of the nodes this rank owns (ascending)
    send: Dict[int, np.ndarray] = field(default_factory=dict)   # neighbour -> local node ids whose partial rows go there (ascending global id)
    recv: Dict[int, np.ndarray] = field(default_factory=dict)   # neighbour -> local (owned) node ids that receive its partial rows
    rank: int = 0
    world: int = 1
    mode: str = "exchange"
    own_elements: int = 0

    def num_own_elements(self):
        return int(self.own_elements)

    def num_active_elements(self):
        return int(self.active.sum())


def morton_partition(mesh: Mesh, world: int) -> np.ndarray:
    """Default partitioner: elements sorted by the Morton key of their centroids (21 bits per axis), cut into ``world`` runs of
    (almost) equal length.  Returns ``elem_to_part`` (int32 per element).  Any other array works with ``make_part`` just as well."""
    conn = np.asarray(mesh.connectivity).astype(np.int64)
    cent = mesh.vertices[conn].mean(axis=1)
    lo, hi = cent.min(axis=0), cent.max(axis=0)
    scale = np.where(hi > lo, 2097151.0 / np.where(hi > lo, hi - lo, 1.0), 0.0)
    q = ((cent - lo) * scale).astype(np.uint64)

    def spread(v):   # 21 bits -> every third bit
        v = v & np.uint64(0x1fffff)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v

    key = spread(q[:, 0])
    for a in range(1, q.shape[1]):
        key |= spread(q[:, a]) << np.uint64(a)
    order = np.argsort(key, kind="stable")
    part = np.empty(len(conn), dtype=np.int32)
    bounds = (np.arange(world + 1, dtype=np.int64) * len(conn)) // world
    for r in range(world):
        part[order[bounds[r]:bounds[r + 1]]] = r
    return part


def node_owners(connectivity, elem_to_part, num_nodes: int) -> np.ndarray:
    """owner of every node: the lowest part with an element touching it (nodes without elements: part 0)"""
    conn = np.asarray(connectivity).astype(np.int64)
    part = np.asarray(elem_to_part).astype(np.int64)
    owner = np.full(num_nodes, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(owner, conn.reshape(-1), np.repeat(part, conn.shape[1]))
    owner[owner == np.iinfo(np.int64).max] = 0
    return owner


def make_part(mesh: Mesh, elem_to_part, rank: int, world: int, mode: str = "exchange") -> PartProblem:
    """Rank ``rank``'s share of ``mesh`` under the element partition ``elem_to_part`` (see the module docstring)."""
    if mode not in ("exchange", "halo"):
        raise ValueError("mode must be 'exchange' or 'halo'")
    conn = np.asarray(mesh.connectivity).astype(np.int64)
    part = np.asarray(elem_to_part).astype(np.int64)
    if part.shape != (len(conn),) or (len(part) and (part.min() < 0 or part.max() >= world)):
        raise ValueError("elem_to_part: one part in [0, world) per element")
    n = mesh.num_nodes()
    owner = node_owners(conn, part, n)
    own_e = part == rank
    touched = np.zeros(n, dtype=bool)                      # nodes my own elements touch
    touched[conn[own_e].reshape(-1)] = True
    mine = owner == rank
    # extended mesh: every element that touches a node my own elements touch (their rows must carry the global pattern) -- and, for
    # nodes I own without touching them (only nodes without any element: empty rows), nothing
    ext_e = touched[conn].any(axis=1)
    ext_nodes = np.zeros(n, dtype=bool)
    ext_nodes[conn[ext_e].reshape(-1)] = True
    ext_nodes |= mine                                       # (isolated owned nodes keep their empty rows)
    l2g = np.flatnonzero(ext_nodes)
    g2l = np.full(n, -1, dtype=np.int64)
    g2l[l2g] = np.arange(len(l2g))
    elem_l2g = np.flatnonzero(ext_e)
    local = Mesh(mesh.vertices[l2g].copy(), g2l[conn[elem_l2g]].astype(np.uint64), mesh.elem_kind)
    active = own_e[elem_l2g].astype(np.uint8)
    if mode == "halo":
        # also the halo elements that touch a node I own: my rows are then complete without any traffic
        active |= (mine[conn[elem_l2g]].any(axis=1)).astype(np.uint8)
    prob = PartProblem(local, l2g, elem_l2g, active, g2l[np.flatnonzero(mine)], {}, {}, rank, world, mode, int(own_e.sum()))
    if mode == "exchange":
        # what I send: nodes my own elements touch that another part owns; what I receive: my nodes that another part's elements touch
        for q in np.unique(owner[touched & ~mine]):
            prob.send[int(q)] = g2l[np.flatnonzero(touched & (owner == q))]
        for q in range(world):
            if q == rank:
                continue
            tq = np.zeros(n, dtype=bool)
            tq[conn[part == q].reshape(-1)] = True
            nodes = np.flatnonzero(tq & mine)
            if len(nodes):
                prob.recv[q] = g2l[nodes]
    return prob


class PartExchange:
    """Interface rows of a ``PartProblem``: packed index lists, one send and one receive buffer per neighbour, all transfers posted at
    once (``torch.distributed.batch_isend_irecv``).  Works on any torch tensor (CUDA with nccl = RCCL, CPU with gloo)."""

    def __init__(self, prob: PartProblem, group=None):
        self.prob, self.group = prob, group
        self.values = None
        self.send_idx: Dict[int, object] = {}
        self.recv_idx: Dict[int, object] = {}
        self.recv_buf: Dict[int, object] = {}
        self._reqs: List[object] = []
        self._send_bufs: List[object] = []

    @staticmethod
    def _row_indices(row_offsets, s, nodes, device):
        """positions in ``values`` of all s rows of every node in ``nodes`` (in that order), as an int64 tensor"""
        import torch

        ro = np.asarray(row_offsets).astype(np.int64)
        first, last = ro[s * nodes], ro[s * nodes + s]
        n = last - first
        start = np.cumsum(n) - n
        idx = np.repeat(first - start, n) + np.arange(int(n.sum()), dtype=np.int64)
        return torch.as_tensor(idx, device=device)

    def bind_offsets(self, row_offsets, solution_dim: int, values):
        import torch

        self.values = values
        for q, nodes in self.prob.send.items():
            self.send_idx[q] = self._row_indices(row_offsets, solution_dim, np.asarray(nodes), values.device)
        for q, nodes in self.prob.recv.items():
            self.recv_idx[q] = self._row_indices(row_offsets, solution_dim, np.asarray(nodes), values.device)
            self.recv_buf[q] = torch.empty(self.recv_idx[q].numel(), dtype=values.dtype, device=values.device)
        return self

    def bind(self, engine, values):
        ro, _ = engine.pattern(want_cols=False)
        return self.bind_offsets(ro, engine.solution_dim(), values)

    def bytes_sent(self):
        return int(sum(8 * i.numel() for i in self.send_idx.values()))

    def start(self, comm_stream=None):
        import torch
        import torch.distributed as dist

        self._reqs, self._send_bufs = [], []
        if not self.send_idx and not self.recv_idx:
            return

        def post():
            ops = []
            for q in sorted(self.send_idx):
                buf = self.values.index_select(0, self.send_idx[q])
                self._send_bufs.append(buf)
                ops.append(dist.P2POp(dist.isend, buf, q, self.group))
            for q in sorted(self.recv_idx):
                ops.append(dist.P2POp(dist.irecv, self.recv_buf[q], q, self.group))
            return dist.batch_isend_irecv(ops)

        if comm_stream is not None:
            comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(comm_stream):
                self._reqs = post()
        else:
            self._reqs = post()

    def finish(self):
        for req in self._reqs:
            req.wait()
        self._reqs = []
        for q in sorted(self.recv_idx):       # fixed order of the additions: the owned rows come out the same bits every run
            self.values.index_add_(0, self.recv_idx[q], self.recv_buf[q])
        self._send_bufs = []

    def run(self):
        self.start()
        self.finish()

    def run_vector(self, vec, components: int):
        """the same exchange for a node vector (``components`` values per node: the residual of ``PartAssembly.assemble_vector``):
        partial sums at the nodes a neighbour owns go there and are added; entries of nodes this rank does not own are scratch afterwards"""
        import torch
        import torch.distributed as dist

        def idx(nodes):
            n = torch.as_tensor(np.asarray(nodes, dtype=np.int64), device=vec.device)
            return (n[:, None] * components + torch.arange(components, device=vec.device)[None, :]).reshape(-1)

        send = {q: idx(nodes) for q, nodes in self.prob.send.items()}
        recv = {q: idx(nodes) for q, nodes in self.prob.recv.items()}
        if not send and not recv:
            return
        bufs = {q: torch.empty(len(i), dtype=vec.dtype, device=vec.device) for q, i in recv.items()}
        keep, ops = [], []
        for q in sorted(send):
            keep.append(vec.index_select(0, send[q]))
            ops.append(dist.P2POp(dist.isend, keep[-1], q, self.group))
        for q in sorted(recv):
            ops.append(dist.P2POp(dist.irecv, bufs[q], q, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for q in sorted(recv):
            vec.index_add_(0, recv[q], bufs[q])


def _u64p(a):
    import ctypes as C

    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def morton_partition_abi(mesh: Mesh, world: int) -> np.ndarray:
    """``fh_morton_partition`` (fenris_amd/csrc/partition.cpp): what a Rust / C host calls; same result as ``morton_partition``"""
    import ctypes as C

    from . import _ffi

    conn = np.ascontiguousarray(mesh.connectivity, dtype=np.uint64)
    verts = np.ascontiguousarray(mesh.vertices, dtype=np.float64)
    part = np.empty(len(conn), dtype=np.int32)
    rc = _ffi.lib().fh_morton_partition(verts.shape[1], verts.ctypes.data_as(C.POINTER(C.c_double)), len(verts), conn.shape[1], _u64p(conn), len(conn),
                                        world, part.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise ValueError(f"fh_morton_partition: status {rc}")
    return part


def make_part_abi(mesh: Mesh, elem_to_part, rank: int, world: int, mode: str = "exchange") -> PartProblem:
    """``make_part`` through the C ABI (``fh_partition_*``, fenris_amd/csrc/partition.cpp) -- the host logic as a Rust caller gets it."""
    import ctypes as C

    from . import _ffi

    if mode not in ("exchange", "halo"):
        raise ValueError("mode must be 'exchange' or 'halo'")
    lib = _ffi.lib()
    conn = np.ascontiguousarray(mesh.connectivity, dtype=np.uint64)
    part = np.ascontiguousarray(elem_to_part, dtype=np.int32)
    if part.shape != (len(conn),):
        raise ValueError("elem_to_part: one part in [0, world) per element")
    h = lib.fh_partition_create(mesh.num_nodes(), conn.shape[1], _u64p(conn), len(conn), part.ctypes.data_as(C.POINTER(C.c_int32)), rank, world,
                                1 if mode == "halo" else 0)
    if not h:
        raise ValueError("elem_to_part: one part in [0, world) per element")
    h = C.c_void_p(h)
    try:
        sz = np.zeros(8, dtype=np.uint64)
        assert lib.fh_partition_sizes(h, _u64p(sz)) == 0
        nl, el, no, own_e, nsp, nsn, nrp, nrn = (int(x) for x in sz)
        l2g, elem_l2g = np.empty(nl, dtype=np.uint64), np.empty(el, dtype=np.uint64)
        lconn, active, owned = np.empty((el, conn.shape[1]), dtype=np.uint64), np.empty(el, dtype=np.uint8), np.empty(no, dtype=np.uint64)
        assert lib.fh_partition_mesh(h, _u64p(l2g), _u64p(elem_l2g), _u64p(lconn), active.ctypes.data_as(C.POINTER(C.c_uint8)), _u64p(owned)) == 0
        sp, so, sn = np.empty(nsp, dtype=np.int32), np.empty(nsp + 1, dtype=np.uint64), np.empty(nsn, dtype=np.uint64)
        rp, ro, rn = np.empty(nrp, dtype=np.int32), np.empty(nrp + 1, dtype=np.uint64), np.empty(nrn, dtype=np.uint64)
        i32 = C.POINTER(C.c_int32)
        assert lib.fh_partition_exchange(h, sp.ctypes.data_as(i32), _u64p(so), _u64p(sn), rp.ctypes.data_as(i32), _u64p(ro), _u64p(rn)) == 0
    finally:
        lib.fh_partition_destroy(h)
    l2g = l2g.astype(np.int64)
    local = Mesh(mesh.vertices[l2g].copy(), lconn, mesh.elem_kind)
    prob = PartProblem(local, l2g, elem_l2g.astype(np.int64), active, owned.astype(np.int64), {}, {}, rank, world, mode, own_e)
    for k in range(nsp):
        prob.send[int(sp[k])] = sn[int(so[k]):int(so[k + 1])].astype(np.int64)
    for k in range(nrp):
        prob.recv[int(rp[k])] = rn[int(ro[k]):int(ro[k + 1])].astype(np.int64)
    return prob


class AbiPartExchange:
    """The interface-row exchange of a ``PartProblem`` through the C ABI: ``fh_group_set_exchange_nodes`` + ``fh_group_exchange_start`` /
    ``_finish`` (fenris_amd/csrc/group.hip) -- pack kernel, one RCCL group of ncclSend / ncclRecv, unpack-add kernel, all on the library's
    streams.  GPU only.  ``self_loop``: a one-rank communicator in which every peer is this rank (test mode: whatever is packed for
    a peer comes back as that peer's contribution)."""

    def __init__(self, prob: PartProblem, engine, group=None, self_loop: bool = False):
        import ctypes as C

        import torch
        import torch.distributed as dist

        self.prob, self.engine = prob, engine
        lib = _ffi_lib()
        idbuf = (C.c_uint8 * 128)()
        rank, world = (0, 1) if self_loop else (prob.rank, prob.world)
        if world > 1:
            t = torch.zeros(128, dtype=torch.uint8, device=f"cuda:{engine.device}")
            if rank == 0:
                engine._check(lib.fh_group_unique_id(idbuf))
                t.copy_(torch.tensor(list(idbuf), dtype=torch.uint8))
            dist.broadcast(t, 0, group=group)
            for i, b in enumerate(t.cpu().tolist()):
                idbuf[i] = b
        else:
            engine._check(lib.fh_group_unique_id(idbuf))
        h = C.c_void_p()
        engine._check(lib.fh_group_create(engine._h, idbuf, rank, world, C.byref(h)))
        self._g, self._lib, self._self_loop = h, lib, self_loop
        self.values = None

    def bind(self, engine, values):
        import ctypes as C

        def lists(d):
            peers = sorted(d)
            off = np.zeros(len(peers) + 1, dtype=np.uint64)
            off[1:] = np.cumsum([len(d[q]) for q in peers])
            nodes = np.concatenate([np.asarray(d[q], dtype=np.uint64) for q in peers]) if peers else np.zeros(0, dtype=np.uint64)
            ids = np.asarray([0 if self._self_loop else q for q in peers], dtype=np.int32)
            return len(peers), ids, off, np.ascontiguousarray(nodes)

        ns, sp, so, sn = lists(self.prob.send)
        nr, rp, ro, rn = lists(self.prob.recv)
        i32 = C.POINTER(C.c_int32)
        self.values = values
        row_off, s = np.asarray(engine.pattern(want_cols=False)[0]).astype(np.int64), engine.solution_dim()
        self._bytes = int(8 * (row_off[s * sn.astype(np.int64) + s] - row_off[s * sn.astype(np.int64)]).sum()) if len(sn) else 0
        self.engine._check(self._lib.fh_group_set_exchange_nodes(self._g, ns, sp.ctypes.data_as(i32), _u64p(so), _u64p(sn),
                                                                 nr, rp.ctypes.data_as(i32), _u64p(ro), _u64p(rn)))
        return self

    def bytes_sent(self):
        return self._bytes

    def start(self, comm_stream=None):
        import ctypes as C

        self.engine._check(self._lib.fh_group_exchange_start(self._g, C.c_void_p(self.values.data_ptr())))

    def finish(self):
        import ctypes as C

        self.engine._check(self._lib.fh_group_exchange_finish(self._g, C.c_void_p(self.values.data_ptr())))

    def run(self):
        self.start()
        self.finish()

    def run_vector(self, vec, components: int):
        """``fh_group_exchange_vector_start`` / ``_finish``: the node vector through the same lists"""
        import ctypes as C

        self.engine._check(self._lib.fh_group_exchange_vector_start(self._g, C.c_void_p(vec.data_ptr()), components))
        self.engine._check(self._lib.fh_group_exchange_vector_finish(self._g, C.c_void_p(vec.data_ptr()), components))

    def close(self):
        if self._g:
            self._lib.fh_group_destroy(self._g)
            self._g = None


def _ffi_lib():
    from . import _ffi

    return _ffi.lib()


class PartAssembly:
    """One rank of the multi-GPU stiffness assembly of an arbitrary mesh (the general counterpart of ``distributed.SlabAssembly``).
    ``configure(engine, mesh)`` sets operator / quadrature / u on an engine for the given (extended local) mesh."""

    def __init__(self, prob: PartProblem, configure, device: int = 0, group=None, stream=None, exchange: str = "torch"):
        import torch

        from .assembly import Engine

        if exchange not in ("torch", "abi"):
            raise ValueError("exchange must be 'torch' or 'abi'")
        self.prob = prob
        self.main = Engine(device, stream=stream)
        configure(self.main, prob.mesh)
        self.main.set_active_elements(prob.active)
        nnz = self.main.build_pattern()
        self.values = torch.zeros(nnz, dtype=torch.float64, device=f"cuda:{device}")
        self.exchange = (AbiPartExchange(prob, self.main, group) if exchange == "abi" else PartExchange(prob, group)).bind(self.main, self.values)
        self.placement = None     # (same attributes as distributed.SlabAssembly for callers that drive either)
        self.comm = None

    def enqueue(self, flags):
        """the rank's share of ``assemble_into_csr``.  ``flags`` must carry ``ASSEMBLE_OVERWRITE``: the exchange ships the interface rows as
        they stand after the local assembly and the owner ADDS them, so rows that accumulated on top of earlier content would hand that
        content on again -- counted twice or more across the ranks, silently.  (To accumulate into an existing matrix, assemble with
        OVERWRITE into ``values`` and add the owned rows to it afterwards.)"""
        from .assembly import ASSEMBLE_OVERWRITE

        if not (int(flags) & ASSEMBLE_OVERWRITE):
            raise ValueError("PartAssembly.enqueue: flags must include ASSEMBLE_OVERWRITE (the interface exchange adds what the rows hold)")
        self.main.assemble_matrix_async(self.values, flags)
        self.exchange.run()

    def assemble_vector(self, out):
        """the rank's share of ``VectorAssembler::assemble_vector_into`` (global.rs:582-608): the own elements' contributions are summed
        into a ZEROED scratch vector, the partial sums at nodes other ranks own travel to their owners, and the complete entries of the
        OWNED nodes are then ADDED to ``out`` (device, one entry per local dof) -- the reference accumulates into its output.  Entries of
        ``out`` at nodes this rank does not own are left as they were.  (Assembling straight into ``out`` would ship whatever a non-owned
        entry held before to its owner once per rank that touches the node.)"""
        import torch

        s = self.main.solution_dim()
        # one scratch vector per (shape, dtype, device), cleared in place -- not an allocation per residual
        key = (tuple(out.shape), out.dtype, str(out.device))
        if getattr(self, "_vec_scratch_key", None) != key:
            self._vec_scratch = torch.zeros_like(out)
            self._vec_scratch_key = key
        else:
            self._vec_scratch.zero_()
        scratch = self._vec_scratch
        self.main.assemble_vector(scratch)
        self.exchange.run_vector(scratch, s)
        okey = (s, str(out.device), id(self.prob.owned), len(self.prob.owned))
        if getattr(self, "_owned_dofs_key", None) != okey:
            owned = np.asarray(self.prob.owned, dtype=np.int64)
            dofs = (s * owned[:, None] + np.arange(s, dtype=np.int64)[None, :]).reshape(-1)
            self._owned_dofs = torch.from_numpy(dofs).to(out.device)
            self._owned_dofs_key = okey
        out.index_add_(0, self._owned_dofs, scratch.index_select(0, self._owned_dofs))

    def poll_status(self):
        self.main.poll_status()

    def owned_rows(self, row_offsets=None):
        """(global row ids, positions in ``values``) of the scalar rows this rank owns -- what a caller concatenates over the ranks"""
        ro = row_offsets if row_offsets is not None else self.main.pattern(want_cols=False)[0]
        s = self.main.solution_dim()
        rows_l = (s * np.asarray(self.prob.owned)[:, None] + np.arange(s)[None, :]).reshape(-1)
        rows_g = (s * self.prob.l2g[np.asarray(self.prob.owned)][:, None] + np.arange(s)[None, :]).reshape(-1)
        return rows_g, rows_l, np.asarray(ro)

    def close(self):
        if hasattr(self.exchange, "close"):
            self.exchange.close()
        self.main.close()
