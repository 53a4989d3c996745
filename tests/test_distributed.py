"""Multi-partition assembly: slab partition + interface-row exchange (fenris_amd/distributed.py).

CPU part (not gpu): two processes over gloo run the product's partition and exchange code; the per-rank
partial values come from the oracle (the checker standing in for the GPU numerics), the result is compared
with the oracle's single-process assembly of the global mesh.
GPU part: the same partition driven through the engine (element mask + owner-computes kernel) in one process.
"""
import os
import socket

import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import distributed as fd
from fenris_amd import quadrature

LAME = (416666.6666666667, 277777.7777777778)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _global_reference(oracle, units_z, cells, op):
    v, c = oracle.hex_mesh(1.0, 1, 1, units_z, cells)
    w, p = oracle.hexahedron_gauss(2)
    ref = oracle.ElementAssembler(oracle.HEX8, op, v, c, w, p, params=None if op == oracle.LAPLACE else LAME)
    st, _, ro, ci, vals = oracle.assemble(ref)
    assert st == 0
    return ro, ci, vals


def _oracle_partial(oracle, slab, op):
    """Pattern on the extended mesh, numerics over the own elements only (what Engine.set_active_elements does)."""
    m = slab.mesh
    w, p = oracle.hexahedron_gauss(2)
    params = None if op == oracle.LAPLACE else LAME
    full = oracle.ElementAssembler(oracle.HEX8, op, m.vertices, m.connectivity, w, p, params=params)
    ro, ci = oracle.pattern_for(full)
    own = oracle.ElementAssembler(oracle.HEX8, op, m.vertices, m.connectivity[slab.active.astype(bool)], w, p, params=params)
    vals = np.zeros(len(ci))
    st, _ = oracle.assemble_into_csr(own, ro, ci, vals)
    assert st == 0
    return ro, ci, vals


def _check_owned_rows(slab, s, ro, ci, vals, gro, gci, gvals):
    lo, hi = slab.owned_nodes
    a, b = int(ro[s * lo]), int(ro[s * hi])
    g0 = s * (slab.node_offset + lo)
    ga, gb = int(gro[g0]), int(gro[g0 + s * (hi - lo)])
    assert b - a == gb - ga
    assert np.array_equal(ro[s * lo:s * hi + 1] - ro[s * lo], gro[g0:g0 + s * (hi - lo) + 1] - gro[g0])
    assert np.array_equal(ci[a:b] + np.uint64(s * slab.node_offset), gci[ga:gb])  # indices bit-exact after the shift
    scale = np.abs(gvals).max()
    assert np.abs(vals[a:b] - gvals[ga:gb]).max() <= 1e-12 * scale


def test_slab_partition_covers_mesh():
    cells, world = 3, 4
    seen_elems, seen_nodes = 0, 0
    for r in range(world):
        slab = fd.make_slab(1.0, 1, 1, world, cells, r, world)
        seen_elems += slab.num_own_elements()
        seen_nodes += slab.owned_nodes[1] - slab.owned_nodes[0]
        assert (slab.send_nodes is None) == (r == 0) and (slab.recv_nodes is None) == (r == world - 1)
        # vertices are the global generator's vertices (bit-exact)
        g = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, world, cells)
        n = slab.mesh.num_nodes()
        assert np.array_equal(slab.mesh.vertices, g.vertices[slab.node_offset:slab.node_offset + n])
    assert seen_elems == cells * cells * cells * world
    assert seen_nodes == (cells + 1) ** 2 * (cells * world + 1)


@pytest.mark.parametrize("op_name", ["LAPLACE", "LINEAR_ELASTIC"])
def test_halo_recompute_needs_no_exchange(oracle, op_name):
    """mode="halo" (SURVEY 8e alternative): own elements + the halo layer above complete every owned row locally"""
    op = getattr(oracle, op_name)
    s = 1 if op_name == "LAPLACE" else 3
    world, cells = 3, 2
    gro, gci, gvals = _global_reference(oracle, world, cells, op)
    own_total = 0
    for r in range(world):
        slab = fd.make_slab(1.0, 1, 1, world, cells, r, world, mode="halo")
        assert slab.send_nodes is None and slab.recv_nodes is None
        assert slab.num_active_elements() == slab.num_own_elements() + (cells * cells if r < world - 1 else 0)
        own_total += slab.num_own_elements()
        ro, ci, vals = _oracle_partial(oracle, slab, op)
        _check_owned_rows(slab, s, ro, ci, vals, gro, gci, gvals)
        ex = fd.InterfaceExchange(slab).bind_offsets(ro, s, vals)
        assert ex.bytes_sent() == 0
        ex.run()  # no process group needed: nothing to send or receive
    assert own_total == cells * cells * cells * world


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gloo_worker(rank, world, port, op_name, q, units_z=None, cells=2):
    import torch
    import torch.distributed as dist

    from oracle import oracle

    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        op = getattr(oracle, op_name)
        s = 1 if op_name == "LAPLACE" else 3
        units_z = units_z or world          # units_z * cells element layers over `world` ranks (uneven when not divisible)
        slab = fd.make_slab(1.0, 1, 1, units_z, cells, rank, world)
        ro, ci, vals = _oracle_partial(oracle, slab, op)
        vals0 = vals.copy()                   # (values shares its memory with vals)
        values = torch.from_numpy(vals)
        ex = fd.InterfaceExchange(slab).bind_offsets(ro, s, values, col_indices=ci)   # packed: the structurally zero third stays home
        full = fd.InterfaceExchange(slab, pack=False).bind_offsets(ro, s, values)
        if slab.send_nodes is not None:
            assert ex.send_idx is not None and 3 * ex.bytes_sent() == 2 * full.bytes_sent() > 0
        ex.run()
        gro, gci, gvals = _global_reference(oracle, units_z, cells, op)
        _check_owned_rows(slab, s, ro, ci, values.numpy(), gro, gci, gvals)
        # the unpacked form gives the same owned rows bit for bit (what it adds on top are zeros)
        values2 = torch.from_numpy(vals0)
        fd.InterfaceExchange(slab, pack=False).bind_offsets(ro, s, values2).run()
        lo, hi = int(ro[s * slab.owned_nodes[0]]), int(ro[s * slab.owned_nodes[1]])
        assert np.array_equal(values.numpy()[lo:hi], values2.numpy()[lo:hi])
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as exc:  # pragma: no cover - reported to the parent
        import traceback

        q.put((rank, traceback.format_exc() + repr(exc)))


@pytest.mark.parametrize("op_name", ["LAPLACE", "LINEAR_ELASTIC"])
def test_two_rank_exchange_over_gloo(op_name):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, op_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_four_rank_exchange_over_gloo_uneven_layers():
    """world_size 4, seven element layers (1 x 1 x 7 units of one cell): slabs of 2, 2, 2 and 1 layers, three interfaces
    exchanged concurrently; every rank's owned rows equal the single-mesh oracle matrix"""
    import torch.multiprocessing as mp

    assert [fd.slab_layers(7, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 7)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 4
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, "LINEAR_ELASTIC", q, 7, 1)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


@pytest.mark.gpu
def test_row_range_split_equals_single_launch():
    """SlabAssembly's two launches from ONE context (rows of the ghost plane first with the context's second set of
    tables, fh_assemble_matrix_rows_async_dev; then the rest, fh_set_row_range) write exactly the rows a single launch
    writes, with the same values up to the summation order inside the row accumulators"""
    import torch

    world, cells = 3, 4
    w, p = quadrature.tensor.hexahedron_gauss(2)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(fa.LameParameters(*LAME))

    def configure(engine, mesh):
        return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh)
                .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(None).build())

    slab = fd.make_slab(1.0, 1, 1, world, cells, 1, world)
    sa = fd.SlabAssembly(slab, configure, device=0, overlap=True)
    assert sa.split is not None and sa.comm is not None
    flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
    sa.values.fill_(7.0)
    sa.main.assemble_matrix_rows_async(sa.values, flags, 0, sa.split)
    torch.cuda.synchronize()
    a, b = sa.exchange.send_seg
    v1 = sa.values.cpu().numpy().copy()
    assert np.all(v1[b:] == 7.0)  # rows beyond the ghost plane untouched by the first launch
    sa.main.assemble_matrix_async(sa.values, flags)
    sa.poll_status()
    torch.cuda.synchronize()
    v2 = sa.values.cpu().numpy()
    assert np.array_equal(v1[:b], v2[:b])  # ... and the ghost-plane rows untouched by the main launch
    ref = fa.Engine(0)
    configure(ref, slab.mesh)
    ref.set_active_elements(slab.active)
    ref.build_pattern()
    full = torch.zeros_like(sa.values)
    ref.assemble_matrix(full, flags)
    fv = full.cpu().numpy()
    assert np.abs(v2 - fv).max() <= 1e-12 * np.abs(fv).max()
    with pytest.raises(fa.FenrisError):  # a row range is an owner-computes notion
        sa.main.assemble_matrix_rows(sa.values, fa.SCATTER_ATOMIC, 0, sa.split)
    with pytest.raises(fa.FenrisError):
        sa.main.assemble_matrix(sa.values, fa.SCATTER_ATOMIC)
    # the second set of tables follows the context: another range, then a changed mask, then the first range again
    n = slab.mesh.num_nodes()
    sa.values.fill_(7.0)
    sa.main.assemble_matrix_rows(sa.values, flags, sa.split, n)       # same rows as the context's own range
    v3 = sa.values.cpu().numpy().copy()
    assert np.all(v3[:b] == 7.0) and np.array_equal(v3[b:], v2[b:])
    mask = slab.active.copy()
    mask[np.flatnonzero(mask)[::2]] = 0
    sa.main.set_active_elements(mask)
    ref.set_active_elements(mask)
    ref.assemble_matrix(full, flags)
    fv = full.cpu().numpy()
    sa.values.fill_(7.0)
    sa.main.assemble_matrix_rows(sa.values, flags, 0, sa.split)
    sa.main.assemble_matrix(sa.values, flags)
    v4 = sa.values.cpu().numpy()
    assert np.abs(v4 - fv).max() <= 1e-12 * np.abs(fv).max()
    ref.close()
    sa.close()


@pytest.mark.gpu
@pytest.mark.parametrize("opname", ["LAPLACE", "LINEAR_ELASTIC"])
def test_slab_with_many_positions_per_workgroup_against_the_atomic_scatter(opname):
    """A middle slab big enough that every workgroup of the owner-computes kernels walks several positions (40 x 40 x 15 layers: 3 800
    positions for 768 workgroups), its mask set, both launches of SlabAssembly into an array of garbage -- against the ATOMIC scatter of the
    same context (another engine with the same kernels, as in the test above, would share a defect: until round 3 the affine kernel let a
    block without an active element inherit stale values, invisible at one position per workgroup)"""
    import torch

    w, p = quadrature.tensor.hexahedron_gauss(2)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if opname != "LAPLACE":
        qt = qt.with_uniform_data(fa.LameParameters(*LAME))
    op = fa.LaplaceOperator() if opname == "LAPLACE" else fa.MaterialEllipticOperator(fa.LinearElasticMaterial())

    def configure(engine, mesh):
        return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(op).with_quadrature_table(qt)
                .with_u(None).build())

    slab = fd.make_slab(1.0, 1, 1, 1, 40, 1, 3)
    sa = fd.SlabAssembly(slab, configure, device=0, overlap=True)
    flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
    sa.values.fill_(-11.5)
    sa.main.assemble_matrix_rows_async(sa.values, flags, 0, sa.split)
    sa.main.assemble_matrix_async(sa.values, flags)
    sa.poll_status()
    torch.cuda.synchronize()
    assert "k_affine_rows" in sa.main.last_kernel_name()
    got = sa.values.cpu().numpy()
    n = slab.mesh.num_nodes()
    sa.main.set_row_range(0, n)
    want = torch.zeros_like(sa.values)
    sa.main.assemble_matrix(want, fa.SCATTER_ATOMIC)
    wv = want.cpu().numpy()
    assert np.abs(got - wv).max() <= 1e-12 * np.abs(wv).max()
    sa.close()


@pytest.mark.gpu
def test_rows_call_reports_a_singular_element_of_its_range():
    """fh_assemble_matrix_rows_*: a degenerate element touching the range is reported by the call / by the next poll, and a
    later clean call clears it (the rows call has its own status slot)"""
    import torch

    mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(3)
    verts = mesh.vertices.copy()
    e0 = mesh.connectivity[0]
    verts[e0] = verts[e0[0]]                                          # element 0 collapsed to a point: det J == 0 exactly
    bad = fa.Mesh(verts, mesh.connectivity.copy(), mesh.elem_kind)
    w, p = quadrature.tensor.hexahedron_gauss(2)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(fa.LameParameters(*LAME))
    eng = fa.Engine(0)
    (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(bad)
     .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(None).build())
    nnz = eng.build_pattern()
    vals = torch.zeros(nnz, dtype=torch.float64, device="cuda:0")
    flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
    n = bad.num_nodes()
    with pytest.raises(fa.FenrisError, match="Singular"):
        eng.assemble_matrix_rows(vals, flags, 0, 8)                   # node 0 belongs to element 0
    eng.assemble_matrix_rows_async(vals, flags, 0, 8)
    eng.set_row_range(8, n)
    with pytest.raises(fa.FenrisError, match="Singular"):
        eng.poll_status()                                             # reported although the context's own slot is clean
    eng.assemble_matrix_rows(vals, flags, n - 4, n)                   # far corner: clean, and the slot is clear again
    eng.poll_status()
    eng.close()


@pytest.mark.gpu
def test_halo_slabs_through_engine_match_global_oracle(oracle):
    """halo-recompute partition driven through the engine: every rank's owned rows are complete without exchange"""
    import torch

    world, cells = 3, 3
    w, p = quadrature.tensor.hexahedron_gauss(2)
    gro, gci, gvals = _global_reference(oracle, world, cells, oracle.LINEAR_ELASTIC)
    for r in range(world):
        slab = fd.make_slab(1.0, 1, 1, world, cells, r, world, mode="halo")
        eng = fa.Engine(0)
        qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(fa.LameParameters(*LAME))
        (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(slab.mesh)
         .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(None).build())
        eng.set_active_elements(slab.active)
        ro, ci = eng.pattern()
        values = torch.zeros(len(ci), dtype=torch.float64, device="cuda")
        eng.assemble_matrix(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
        _check_owned_rows(slab, 3, ro, ci, values.cpu().numpy(), gro, gci, gvals)
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scatter", [fa.SCATTER_GATHER, fa.SCATTER_ATOMIC, fa.SCATTER_COLORED])
def test_slabs_through_engine_match_global_oracle(oracle, scatter):
    """3 slabs assembled one after the other on one GPU; the exchange is replayed by hand (no process group):
    exercises the element mask, halo pattern and row-segment logic of the multi-GPU path on the device."""
    import torch

    world, cells = 3, 3
    w, p = quadrature.tensor.hexahedron_gauss(2)
    lame = fa.LameParameters(*LAME)
    gro, gci, gvals = _global_reference(oracle, world, cells, oracle.LINEAR_ELASTIC)
    slabs, parts = [], []
    for r in range(world):
        slab = fd.make_slab(1.0, 1, 1, world, cells, r, world)
        eng = fa.Engine(0)
        qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
        asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(slab.mesh)
               .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt)
               .with_u(None).build())
        eng.set_active_elements(slab.active)
        ro, ci = eng.pattern()
        values = torch.zeros(len(ci), dtype=torch.float64, device="cuda")
        if scatter == fa.SCATTER_COLORED:
            eng.color()
        eng.assemble_matrix(values, scatter | fa.ASSEMBLE_OVERWRITE)
        ex = fd.InterfaceExchange(slab).bind_offsets(ro, 3, values)
        slabs.append((slab, ro, ci, values, ex))
        # partial values equal the oracle's own-element assembly on the extended pattern
        oro, oci, ovals = _oracle_partial(oracle, slab, oracle.LINEAR_ELASTIC)
        assert np.array_equal(ro, oro) and np.array_equal(ci, oci)
        assert np.abs(values.cpu().numpy() - ovals).max() <= 1e-12 * np.abs(ovals).max()
        parts.append(eng)
    for r in range(world - 1):  # rank r+1 sends its bottom ghost plane to rank r
        lo_ex, up_ex = slabs[r][4], slabs[r + 1][4]
        seg = slabs[r + 1][3][up_ex.send_seg[0]:up_ex.send_seg[1]]
        slabs[r][3][lo_ex.recv_seg[0]:lo_ex.recv_seg[1]] += seg
    for slab, ro, ci, values, _ in slabs:
        _check_owned_rows(slab, 3, ro, ci, values.cpu().numpy(), gro, gci, gvals)
    for eng in parts:
        eng.close()


@pytest.mark.gpu
def test_overlapped_slab_assembly_two_processes_one_device():
    """SlabAssembly with overlap (rows of the ghost plane first, their transfer on a side stream beside the main launch,
    fh_set_row_range) under a REAL process group: two fresh child processes -- started before anything here touches the
    GPU state they use -- share cuda:0 over gloo, 5 element layers cut 3 + 2; every rank checks its owned rows against the
    single-mesh oracle matrix (tests/slab_overlap_worker.py)."""
    import subprocess
    import sys

    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "slab_overlap_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), "4", "5"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            pr.kill()
            out, _ = pr.communicate()
        outs.append((pr.returncode, out))
    for r, (rc, out) in enumerate(outs):
        assert rc == 0 and f"rank {r} ok" in out, f"rank {r} rc={rc}\n{out[-3000:]}"


@pytest.mark.gpu
def test_group_abi_single_rank_plumbing():
    """fh_group_* (include/fenris_hip.h, fenris_amd/csrc/group.hip) with one rank: RCCL loads, a communicator is created, an
    exchange without peers starts and finishes, argument errors are reported.  (Transfers between ranks need one GPU per
    rank: RCCL refuses two ranks on one device, so on the one-GPU test box only the plumbing can run.)"""
    import ctypes as C

    import torch

    from fenris_amd import _ffi

    lib = _ffi.lib()
    eng = fa.Engine(0)
    try:
        idbuf = (C.c_uint8 * 128)()
        assert lib.fh_group_unique_id(idbuf) == 0 and any(idbuf)
        g = C.c_void_p()
        assert lib.fh_group_create(eng._h, idbuf, 0, 1, C.byref(g)) == 0 and g.value
        vals = torch.ones(1000, dtype=torch.float64, device="cuda")
        assert lib.fh_group_set_exchange(g, -1, 0, 0, -1, 0, 0) == 0
        assert lib.fh_group_exchange_start(g, C.c_void_p(vals.data_ptr())) == 0
        assert lib.fh_group_exchange_start(g, C.c_void_p(vals.data_ptr())) == _ffi.FH_INVALID_STATE  # already started
        assert lib.fh_group_exchange_finish(g, C.c_void_p(vals.data_ptr())) == 0
        assert lib.fh_group_exchange_finish(g, C.c_void_p(vals.data_ptr())) == _ffi.FH_INVALID_STATE  # nothing started
        assert lib.fh_group_set_exchange(g, 0, 0, 10, -1, 0, 0) == _ffi.FH_BAD_ARGUMENT  # a rank cannot send to itself
        assert lib.fh_group_set_exchange(g, 3, 0, 10, -1, 0, 0) == _ffi.FH_BAD_ARGUMENT  # peer outside the group
        torch.cuda.synchronize()
        assert float(vals.sum()) == 1000.0
        lib.fh_group_destroy(g)
        bad = C.c_void_p()
        assert lib.fh_group_create(eng._h, idbuf, 2, 1, C.byref(bad)) == _ffi.FH_BAD_ARGUMENT
    finally:
        eng.close()


def test_group_reports_unsupported_when_rccl_cannot_be_loaded():
    """librccl missing: the group calls return FH_UNSUPPORTED -- they used to crash in the error path (dlerror() called twice).
    A fresh process, because a loaded RCCL stays loaded; FENRIS_HIP_RCCL_LIB names a file that does not exist.  No GPU needed."""
    import subprocess
    import sys

    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from fenris_amd import _ffi; lib = _ffi.lib(); "
            "buf = (C.c_uint8 * 128)(); rc = lib.fh_group_unique_id(buf); print('rc', rc, _ffi.FH_UNSUPPORTED); "
            "sys.exit(0 if rc == _ffi.FH_UNSUPPORTED else 1)") % ROOT
    env = dict(os.environ, FENRIS_HIP_RCCL_LIB="/nonexistent/librccl.so.1")
    pr = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stdout + pr.stderr


@pytest.mark.gpu
def test_group_create_reports_unsupported_when_rccl_cannot_be_loaded():
    import subprocess
    import sys

    code = ("import ctypes as C, sys; sys.path.insert(0, %r); import fenris_amd as fa; from fenris_amd import _ffi; lib = _ffi.lib(); "
            "eng = fa.Engine(0); buf = (C.c_uint8 * 128)(); g = C.c_void_p(); rc = lib.fh_group_create(eng._h, buf, 0, 1, C.byref(g)); "
            "print('rc', rc, eng.last_error()); sys.exit(0 if rc == _ffi.FH_UNSUPPORTED and 'dlopen' in eng.last_error() else 1)") % ROOT
    env = dict(os.environ, FENRIS_HIP_RCCL_LIB="/nonexistent/librccl.so.1")
    pr = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stdout + pr.stderr


@pytest.mark.gpu
def test_group_abi_two_ranks_rccl():
    """fh_group_* between two ranks over RCCL, one GPU each (tests/group_rccl_worker.py): the communicator reports two ranks, the
    sent segment arrives and is added.  Needs two devices: skipped on the one-GPU test box."""
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "group_rccl_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            pr.kill()
            out, _ = pr.communicate()
        outs.append((pr.returncode, out))
    for r, (rc, out) in enumerate(outs):
        assert rc == 0 and f"rank {r} ok" in out, f"rank {r} rc={rc}\n{out[-3000:]}"


def test_packing_is_switched_off_when_the_first_third_is_not_the_plane_below():
    """InterfaceExchange.bind (the engine path) verifies what packing assumes from the slab's connectivity: a SlabProblem whose `send`
    plane has nothing below it (rows of 2 x 9 instead of 3 x 9 blocks, every length still divisible by 3 for s = 3) must travel whole"""
    class _Eng:   # pattern accessor only: no device needed
        def __init__(self, ro):
            self.ro = ro

        def pattern(self, want_cols=True):
            return self.ro, None

        def solution_dim(self):
            return 3

    import torch

    from oracle import oracle as o

    good = fd.make_slab(1.0, 1, 1, 3, 2, 1, 3)
    bad = fd.make_slab(1.0, 1, 1, 3, 2, 1, 3)
    npl = (2 + 1) ** 2
    bad.send_nodes = (0, npl)                       # the bottom plane of the extended mesh: no nodes below it
    for slab, want_pack in ((good, True), (bad, False)):
        w, p = o.hexahedron_gauss(2)
        ro, ci = o.pattern_for(o.ElementAssembler(o.HEX8, o.LINEAR_ELASTIC, slab.mesh.vertices, slab.mesh.connectivity, w, p, params=LAME))
        ex = fd.InterfaceExchange(slab)
        ex.bind(_Eng(ro), torch.zeros(len(ci), dtype=torch.float64))
        assert ex.pack == want_pack
        assert (ex.send_idx is not None) == want_pack
