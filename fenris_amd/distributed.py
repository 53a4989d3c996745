"""Multi-GPU assembly: slab partition of a structured hex mesh + exchange of interface rows.

fenris itself is single-process; this is the one exchange the assembly path needs when a mesh
is partitioned: rows of nodes on a partition interface receive contributions from the elements of two
partitions.  One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).

Partition (z-slabs, interface plane owned by the LOWER slab):
  * rank r owns element layers [L0, L1) and the node planes (L0, L1]  (rank 0 also owns plane 0);
  * its *extended* local mesh additionally holds one halo element layer below (if r > 0) and above
    (if r < P-1).  The sparsity pattern is built on the extended mesh, so the rows of both interface planes
    carry the complete GLOBAL pattern and have identical layouts on the two ranks that share them;
  * numerics run over the own elements only (``Engine.set_active_elements``);
  * exchange: the partial rows of the bottom ghost plane L0 go to rank r-1, which adds them to its (owned)
    top plane -- a point-to-point transfer per interface (each rides one xGMI link; all interfaces proceed
    concurrently), never an all-reduce over the matrix.
Local node numbering is the global one shifted by a constant, so concatenating the owned row blocks of all
ranks gives the single-GPU CSR (indices after the shift bit-exact, values to rounding).

Alternative without communication (``mode="halo"``, SURVEY.md 8e "halo-element recomputation"): the rank also
runs the numerics of the halo element layer ABOVE its top interface plane, which completes the rows of that
(owned) plane locally; nothing is sent.  Costs one extra element layer per interior interface; the rows of the two
non-owned planes of the extended mesh stay partial and are ignored.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import _ffi
from .mesh import Mesh, procedural


@dataclass
class SlabProblem:
    mesh: Mesh                    # extended local mesh (own + halo layers), global coordinates
    active: np.ndarray            # uint8 per local element: 1 = own element
    node_offset: int              # global index of local node 0
    owned_nodes: Tuple[int, int]  # local node range [lo, hi) owned by this rank
    send_nodes: Optional[Tuple[int, int]]  # local node range of the bottom ghost plane (to rank-1) or None
    recv_nodes: Optional[Tuple[int, int]]  # local node range of the owned top interface plane (from rank+1)
    rank: int
    world: int
    mode: str = "exchange"        # "exchange": interface rows are sent to the owner; "halo": recomputed, no traffic
    own_elements: int = 0         # elements of the partition proper (the unit of the throughput metric)

    def num_own_elements(self):
        return int(self.own_elements)

    def num_active_elements(self):
        return int(self.active.sum())


def slab_layers(cells_z_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Own element layers [L0, L1) of ``rank`` -- as even as possible, lower ranks take the remainder."""
    base, rem = divmod(cells_z_total, world)
    l0 = rank * base + min(rank, rem)
    return l0, l0 + base + (1 if rank < rem else 0)


def make_slab(unit_length: float, units_x: int, units_y: int, units_z: int, cells_per_unit: int, rank: int,
              world: int, mode: str = "exchange") -> SlabProblem:
    """Slab ``rank`` of create_rectangular_uniform_hex_mesh(unit_length, units_x, units_y, units_z, cells_per_unit)."""
    if mode not in ("exchange", "halo"):
        raise ValueError("mode must be 'exchange' or 'halo'")
    cx, cy, cz = units_x * cells_per_unit, units_y * cells_per_unit, units_z * cells_per_unit
    if world > cz:
        raise ValueError("more ranks than element layers")
    l0, l1 = slab_layers(cz, rank, world)
    e0 = l0 - (1 if rank > 0 else 0)            # first element layer of the extended mesh
    e1 = l1 + (1 if rank < world - 1 else 0)    # one past the last
    # extended box: same generator, then the z coordinate is re-evaluated from the GLOBAL plane index exactly
    # like the reference generator does (T::from_usize(k) * cell_size, procedural.rs:247-251)
    h = unit_length / cells_per_unit
    local = _box(cx, cy, e1 - e0, h)
    npl = (cx + 1) * (cy + 1)
    planes = np.arange(e0, e1 + 1, dtype=np.float64)
    local.vertices[:, 2] = np.repeat(planes * h, npl)
    active = np.zeros(local.num_elements(), dtype=np.uint8)
    per_layer = cx * cy
    active[(l0 - e0) * per_layer:(l1 - e0) * per_layer] = 1
    if mode == "halo" and rank < world - 1:
        active[(l1 - e0) * per_layer:(l1 + 1 - e0) * per_layer] = 1  # the layer above completes the owned top plane
    own_lo_plane = (l0 + 1 if rank > 0 else 0) - e0
    own_hi_plane = l1 - e0
    send = ((l0 - e0) * npl, (l0 - e0 + 1) * npl) if (rank > 0 and mode == "exchange") else None
    recv = ((l1 - e0) * npl, (l1 - e0 + 1) * npl) if (rank < world - 1 and mode == "exchange") else None
    return SlabProblem(local, active, e0 * npl, (own_lo_plane * npl, (own_hi_plane + 1) * npl), send, recv, rank, world,
                       mode, (l1 - l0) * per_layer)


def _box(cx, cy, cz, h):
    """cx x cy x cz cells of size h via the engine's host generator (vertex/cell order of procedural.rs:241-271)."""
    import ctypes as C

    nv, nc = (cx + 1) * (cy + 1) * (cz + 1), cx * cy * cz
    v = np.zeros((nv, 3))
    c = np.zeros((nc, 8), dtype=np.uint64)
    a, b = C.c_uint64(), C.c_uint64()
    # unit_length = h, one cell per unit: cell_size = h / 1 exactly
    rc = _ffi.lib().fh_hex_mesh(float(h), cx, cy, cz, 1, _ffi.fp(v), _ffi.up(c), C.byref(a), C.byref(b))
    if rc or a.value != nv or b.value != nc:
        raise _ffi.FenrisError(rc, "slab mesh generation failed")
    # x, y from the global generator formula as well
    return Mesh(v, c, _ffi.HEX8)


class InterfaceExchange:
    """Sends the partial rows of the bottom ghost plane to rank-1 and adds the rows received from rank+1 to the
    owned top plane.  Works on any torch tensor (CUDA with nccl, CPU with gloo).

    ``pack`` (default): only the entries that can be non-zero travel.  The rows of an interface plane carry the complete global
    pattern -- columns in the plane below, the plane itself and the plane above, the same number in each (the extended mesh has a
    halo layer on either side) and in this order (columns are sorted, the numbering is plane-major) -- but the sender's own elements
    lie ABOVE its ghost plane, so the first third of every scalar row is structurally zero: two thirds of the bytes go over the link
    (Hex8 elasticity 216 x 216: 61 MB instead of 91.5 MB per interface and step), gathered into a send buffer on the transfer stream
    and added through the same index list on the receiving side.  A row whose length is not a multiple of three switches packing off."""

    def __init__(self, slab: SlabProblem, group=None, pack: bool = True):
        self.slab, self.group, self.pack = slab, group, pack
        self.values = None
        self.send_seg = self.recv_seg = None
        self.send_idx = self.recv_idx = None
        self.send_buf = self.recv_buf = None

    @staticmethod
    def _upper_two_thirds(row_offsets, r0, r1, device):
        """indices (int64 tensor) of the last two thirds of every scalar row r0 <= r < r1, or None when a row does not divide"""
        import torch

        ro = torch.as_tensor(np.asarray(row_offsets[r0:r1 + 1]).astype(np.int64), device=device)
        length = ro[1:] - ro[:-1]
        if int((length % 3).sum()) != 0 or int(length.sum()) == 0:
            return None
        n = 2 * (length // 3)
        first = ro[:-1] + length // 3
        start = torch.cumsum(n, 0) - n
        return torch.repeat_interleave(first - start, n) + torch.arange(int(n.sum()), dtype=torch.int64, device=device)

    def bind_offsets(self, row_offsets: np.ndarray, solution_dim: int, values, col_indices=None):
        """``col_indices`` (optional, small cases / tests): checks that what packing leaves out are exactly the columns below the plane"""
        import torch

        s = solution_dim
        self.values = values

        def seg(nodes):
            return (int(row_offsets[s * nodes[0]]), int(row_offsets[s * nodes[1]])) if nodes else None

        self.send_seg, self.recv_seg = seg(self.slab.send_nodes), seg(self.slab.recv_nodes)
        self.send_idx = self.recv_idx = None
        if self.pack:
            if self.slab.send_nodes:
                self.send_idx = self._upper_two_thirds(row_offsets, s * self.slab.send_nodes[0], s * self.slab.send_nodes[1], values.device)
            if self.slab.recv_nodes:
                self.recv_idx = self._upper_two_thirds(row_offsets, s * self.slab.recv_nodes[0], s * self.slab.recv_nodes[1], values.device)
            # both sides of an interface must take the same decision: they see the same rows (identical layouts), so they do
            if col_indices is not None and self.send_idx is not None:
                keep = np.zeros(self.send_seg[1] - self.send_seg[0], dtype=bool)
                keep[self.send_idx.cpu().numpy() - self.send_seg[0]] = True
                cols = np.asarray(col_indices[self.send_seg[0]:self.send_seg[1]]).astype(np.int64) // s
                assert np.all(cols[~keep] < self.slab.send_nodes[0]) and np.all(cols[keep] >= self.slab.send_nodes[0]), \
                    "packed exchange: the first third of a ghost row is not the plane below"
        if self.recv_seg:
            n = self.recv_idx.numel() if self.recv_idx is not None else self.recv_seg[1] - self.recv_seg[0]
            self.recv_buf = torch.empty(n, dtype=values.dtype, device=values.device)
        return self

    def _packing_is_valid(self):
        """What packing leaves out must be structurally zero on the sender: for every node of an interface plane, the first third of
        its (sorted) neighbours must be exactly the nodes BELOW the plane, the second third the plane itself, the last third the nodes
        above -- true for ``make_slab`` by construction, checked here from the slab's connectivity (the interface nodes only) so that any
        other ``SlabProblem`` (say, an interface plane on the mesh boundary: rows of 2 x 9 blocks) falls back to whole rows instead of
        dropping contributions silently."""
        conn = np.asarray(self.slab.mesh.connectivity).astype(np.int64)
        for nodes in (self.slab.send_nodes, self.slab.recv_nodes):
            if not nodes:
                continue
            lo, hi = int(nodes[0]), int(nodes[1])
            on_plane = (conn >= lo) & (conn < hi)
            elems = conn[on_plane.any(axis=1)]
            n = elems.shape[1]
            rows = np.repeat(elems, n, axis=1).reshape(-1)          # (i, j) for all node pairs of these elements
            cols = np.tile(elems, (1, n)).reshape(-1)
            keep = (rows >= lo) & (rows < hi)
            pairs = np.unique(np.stack([rows[keep], cols[keep]], axis=1), axis=0)
            below = np.bincount(pairs[pairs[:, 1] < lo, 0] - lo, minlength=hi - lo)
            inside = np.bincount(pairs[(pairs[:, 1] >= lo) & (pairs[:, 1] < hi), 0] - lo, minlength=hi - lo)
            above = np.bincount(pairs[pairs[:, 1] >= hi, 0] - lo, minlength=hi - lo)
            if not (np.array_equal(below, inside) and np.array_equal(inside, above)):
                return False
        return True

    def bind(self, engine, values):
        ro, _ = engine.pattern(want_cols=False)
        if self.pack and not self._packing_is_valid():
            self.pack = False      # (both sides of an interface see the same rows and take the same decision)
        return self.bind_offsets(ro, engine.solution_dim(), values)

    def bytes_sent(self):
        if not self.send_seg:
            return 0
        return 8 * (self.send_idx.numel() if self.send_idx is not None else self.send_seg[1] - self.send_seg[0])

    def start(self, comm_stream=None):
        """Post the send of the bottom ghost plane and the receive for the owned top plane.  With a CUDA
        ``comm_stream`` the transfers are ordered after the work enqueued so far on the current stream and run
        beside whatever is enqueued next (the rows to send must already be queued: launch them first)."""
        import torch
        import torch.distributed as dist

        self._reqs = []
        if not self.send_seg and not self.recv_seg:
            return

        def post():
            ops = []
            if self.send_seg:
                if self.send_idx is not None:
                    self.send_buf = self.values.index_select(0, self.send_idx)   # (on the transfer stream when there is one)
                    ops.append(dist.P2POp(dist.isend, self.send_buf, self.slab.rank - 1, self.group))
                else:
                    ops.append(dist.P2POp(dist.isend, self.values[self.send_seg[0]:self.send_seg[1]], self.slab.rank - 1, self.group))
            if self.recv_seg:
                ops.append(dist.P2POp(dist.irecv, self.recv_buf, self.slab.rank + 1, self.group))
            return dist.batch_isend_irecv(ops)

        if comm_stream is not None:
            comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(comm_stream):
                self._reqs = post()
        else:
            self._reqs = post()

    def finish(self):
        """Wait for the transfers (the current stream waits) and add the received rows to the owned top plane."""
        for req in getattr(self, "_reqs", []):
            req.wait()
        self._reqs = []
        if self.recv_seg:
            if self.recv_idx is not None:
                self.values.index_add_(0, self.recv_idx, self.recv_buf)   # every index once: no two additions meet
            else:
                self.values[self.recv_seg[0]:self.recv_seg[1]] += self.recv_buf

    def run(self):
        self.start()
        self.finish()


class AbiInterfaceExchange:
    """The same exchange through the C ABI (fh_group_*, fenris_amd/csrc/group.hip): RCCL ncclSend / ncclRecv on the
    library's own side stream -- what a Rust host drives.  The 128-byte id is drawn by rank 0 and handed to the other ranks
    over ``torch.distributed`` here (any transport does).  GPU only."""

    def __init__(self, slab: SlabProblem, engine, group=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        self.slab, self.engine = slab, engine
        lib = _ffi.lib()
        idbuf = (C.c_uint8 * 128)()
        world = slab.world
        if world > 1:
            t = torch.zeros(128, dtype=torch.uint8, device=f"cuda:{engine.device}")
            if slab.rank == 0:
                engine._check(lib.fh_group_unique_id(idbuf))
                t.copy_(torch.tensor(list(idbuf), dtype=torch.uint8))
            dist.broadcast(t, 0, group=group)
            for i, b in enumerate(t.cpu().tolist()):
                idbuf[i] = b
        else:
            engine._check(lib.fh_group_unique_id(idbuf))
        h = C.c_void_p()
        engine._check(lib.fh_group_create(engine._h, idbuf, slab.rank, world, C.byref(h)))
        self._g, self._lib = h, lib
        self.values = None

    def bind(self, engine, values):
        ro, _ = engine.pattern(want_cols=False)
        s = engine.solution_dim()

        def seg(nodes):
            return (int(ro[s * nodes[0]]), int(ro[s * nodes[1]])) if nodes else None

        snd, rcv = seg(self.slab.send_nodes), seg(self.slab.recv_nodes)
        self.values = values
        self.engine._check(self._lib.fh_group_set_exchange(
            self._g, self.slab.rank - 1 if snd else -1, snd[0] if snd else 0, snd[1] - snd[0] if snd else 0,
            self.slab.rank + 1 if rcv else -1, rcv[0] if rcv else 0, rcv[1] - rcv[0] if rcv else 0))
        return self

    def start(self, comm_stream=None):
        import ctypes as C

        self.engine._check(self._lib.fh_group_exchange_start(self._g, C.c_void_p(self.values.data_ptr())))

    def finish(self):
        import ctypes as C

        self.engine._check(self._lib.fh_group_exchange_finish(self._g, C.c_void_p(self.values.data_ptr())))

    def run(self):
        self.start()
        self.finish()

    def size(self):
        """ranks of the communicator as RCCL reports it (ncclCommCount)"""
        import ctypes as C

        n = C.c_int(0)
        self.engine._check(self._lib.fh_group_size(self._g, C.byref(n)))
        return int(n.value)

    def close(self):
        if self._g:
            self._lib.fh_group_destroy(self._g)
            self._g = None


class SlabAssembly:
    """One rank of the multi-GPU stiffness assembly.

    ``configure(engine, mesh)`` sets operator / quadrature / u on an engine for the given mesh (typically by building
    an ElementEllipticAssembler).  In "exchange" mode with ``overlap`` the rows of the bottom ghost plane are
    produced by a first, small launch (the same context's second set of tables, ``fh_assemble_matrix_rows_async_dev``); their
    transfer to the owner rides a separate stream while the main launch computes all other rows; the received
    rows are added at the end.  In "halo" mode nothing is exchanged."""

    def __init__(self, slab: SlabProblem, configure, device: int = 0, overlap: bool = True, group=None, stream=None,
                 exchange: str = "torch", placement_tries: int = 0):
        import torch

        from .assembly import Engine

        self.slab = slab
        self.main = Engine(device, stream=stream)
        configure(self.main, slab.mesh)
        self.main.set_active_elements(slab.active)
        nnz = self.main.build_pattern()
        self.values = torch.zeros(nnz, dtype=torch.float64, device=f"cuda:{device}")
        self.split = None
        self.comm = None
        if overlap and slab.send_nodes is not None:
            # the rows of the bottom ghost plane come first, from the same context's second set of tables
            # (fh_assemble_matrix_rows_async_dev); the nodes below the plane carry no active element
            self.split = int(slab.send_nodes[1])
            self.main.set_row_range(self.split, slab.mesh.num_nodes())
        self.placement = None
        if placement_tries > 0:
            # the better of several allocations of this rank's values and of the library's record buffer (fh_time_assembly_dev /
            # fh_tune_placement_dev: the time of the owner-computes kernels follows the physical memory behind them); rank-local,
            # no collective, before anything is bound to the array
            from .assembly import ASSEMBLE_OVERWRITE, SCATTER_GATHER

            from .placement import probe_placement, settle_device

            import time

            t0 = time.perf_counter()
            fl = SCATTER_GATHER | ASSEMBLE_OVERWRITE
            settle = settle_device(self.main, self.values, fl)
            self.values, self.placement = probe_placement(self.main, self.values, fl, placement_tries)
            self.placement["device_settle"] = settle
            self.placement["seconds"] = round(time.perf_counter() - t0, 3)   # (not part of the pattern build a caller may be timing)
        if overlap and (slab.send_nodes is not None or slab.recv_nodes is not None):
            self.comm = torch.cuda.Stream(device=device)
        if exchange == "abi":   # RCCL inside the library (fh_group_*); the torch path stays the test harness
            self.exchange = AbiInterfaceExchange(slab, self.main, group).bind(self.main, self.values)
        else:
            self.exchange = InterfaceExchange(slab, group).bind(self.main, self.values)

    def enqueue(self, flags):
        """one assembly of this rank's rows (values overwritten or accumulated according to ``flags``)"""
        if self.split is not None:
            self.main.assemble_matrix_rows_async(self.values, flags, 0, self.split)
        if self.comm is not None:
            self.exchange.start(self.comm)
            self.main.assemble_matrix_async(self.values, flags)
            self.exchange.finish()
        else:
            self.main.assemble_matrix_async(self.values, flags)
            self.exchange.run()

    def poll_status(self):
        self.main.poll_status()

    def close(self):
        # a group refers to its context (stream, error slot): it goes first
        if hasattr(self.exchange, "close"):
            self.exchange.close()
        self.main.close()


def make_slab_problem(cells: int, rank: int, world: int, mode: str = "exchange"):
    """bench.py weak scaling: cells x cells x (cells * world) box, one z-slab of cells^3 own elements per rank.
    Returns (extended local mesh, exchange); the caller sets the element mask through ``exchange.slab.active``.
    With ``mode="halo"`` the exchange object has nothing to send (its ``run`` is a no-op)."""
    slab = make_slab(1.0, 1, 1, world, cells, rank, world, mode)
    return slab.mesh, InterfaceExchange(slab)
