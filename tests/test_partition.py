"""Multi-GPU assembly of arbitrary meshes: element partition `elem_to_part[]` + packed interface-row exchange (fenris_amd/partition.py).

SURVEY.md 8e: "For unstructured meshes: any element partition (e.g. Morton / RCB on element centroids) -- the engine takes elem_to_part[]";
what every rank's launch replaces is CsrParAssembler::assemble_into_csr over its own elements (global.rs:314-376).

CPU part: 2 and 4 processes over gloo run the product's partition and exchange code on the reference's unstructured sphere fixture and on a
BCC tetrahedral mesh with permuted numbering; the per-rank partial values come from the oracle (the checker standing in for the GPU
numerics); every rank's owned rows must equal the oracle's single-process matrix -- indices bit for bit (through the local -> global node
map), values to 1e-12.  GPU part: the same partitions through the engine (element mask + the owner-computes kernels), all ranks in one
process on the one device, the transfers replaced by device copies through the same index lists."""
import os
import socket

import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import partition as fp
from fenris_amd import quadrature
from conftest import load_golden_mesh

LAME = (416666.6666666667, 277777.7777777778)


def _mesh(name):
    if name == "sphere":
        v, c = load_golden_mesh("sphere_tet4_593")
        return fa.Mesh(v, c, fa.TET4)
    if name == "bcc":   # C3's kind of mesh at a size the oracle finishes in a second: BCC tetrahedra, vertices and elements permuted
        m = fa.procedural.create_unit_box_uniform_tet_mesh_3d(4)
        rng = np.random.Generator(np.random.MT19937(12345))
        vp = rng.permutation(m.num_nodes())
        inv = np.empty_like(vp)
        inv[vp] = np.arange(len(vp))
        return fa.Mesh(m.vertices[vp], inv[np.asarray(m.connectivity).astype(np.int64)][rng.permutation(m.num_elements())].astype(np.uint64), fa.TET4)
    if name == "hex":   # distorted hexahedra with holes
        m = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 7, 5, 4, 1)
        rng = np.random.default_rng(5)
        c = np.asarray(m.connectivity)[rng.random(m.num_elements()) > 0.15]
        return fa.Mesh(m.vertices + 0.1 * rng.uniform(-1, 1, m.vertices.shape), c, fa.HEX8)
    raise ValueError(name)


def _rule(mesh):
    return quadrature.total_order.tetrahedron(2) if mesh.elem_kind == fa.TET4 else quadrature.tensor.hexahedron_gauss(2)


def _okind(oracle, mesh):
    return oracle.TET4 if mesh.elem_kind == fa.TET4 else oracle.HEX8


def _global_reference(oracle, mesh, op):
    w, p = _rule(mesh)
    ref = oracle.ElementAssembler(_okind(oracle, mesh), op, mesh.vertices, mesh.connectivity, w, p, params=None if op == oracle.LAPLACE else LAME)
    st, _, ro, ci, vals = oracle.assemble(ref)
    assert st == 0
    return ro, ci, vals


def _oracle_partial(oracle, prob, op):
    """pattern on the extended local mesh, numerics over the active elements only (what Engine.set_active_elements does)"""
    m = prob.mesh
    w, p = _rule(m)
    params = None if op == oracle.LAPLACE else LAME
    full = oracle.ElementAssembler(_okind(oracle, m), op, m.vertices, m.connectivity, w, p, params=params)
    ro, ci = oracle.pattern_for(full)
    own = oracle.ElementAssembler(_okind(oracle, m), op, m.vertices, np.asarray(m.connectivity)[prob.active.astype(bool)], w, p, params=params)
    vals = np.zeros(len(ci))
    st, _ = oracle.assemble_into_csr(own, ro, ci, vals)
    assert st == 0
    return ro, ci, vals


def _check_owned_rows(prob, s, ro, ci, vals, gro, gci, gvals):
    ro, ci, gro, gci = (np.asarray(x).astype(np.int64) for x in (ro, ci, gro, gci))
    scale = np.abs(gvals).max()
    for l in np.asarray(prob.owned):
        g = int(prob.l2g[l])
        for k in range(s):
            a, b = ro[s * l + k], ro[s * l + k + 1]
            ga, gb = gro[s * g + k], gro[s * g + k + 1]
            assert b - a == gb - ga, (l, g)
            if b == a:
                continue           # a node without elements: an empty row
            cols = s * prob.l2g[ci[a:b] // s] + ci[a:b] % s          # local -> global columns: bit-exact
            assert np.array_equal(cols, gci[ga:gb])
            assert np.abs(vals[a:b] - gvals[ga:gb]).max() <= 1e-12 * scale


@pytest.mark.parametrize("name,world", [("sphere", 3), ("bcc", 4), ("hex", 2)])
def test_partition_covers_the_mesh_and_halo_mode_needs_no_exchange(oracle, name, world):
    mesh = _mesh(name)
    part = fp.morton_partition(mesh, world)
    counts = np.bincount(part, minlength=world)
    assert counts.sum() == mesh.num_elements() and counts.max() - counts.min() <= 1      # balanced runs of the Morton order
    owner = fp.node_owners(mesh.connectivity, part, mesh.num_nodes())
    gro, gci, gvals = _global_reference(oracle, mesh, oracle.LINEAR_ELASTIC)
    owned_total, own_total = 0, 0
    for r in range(world):
        prob = fp.make_part(mesh, part, r, world, mode="halo")
        assert not prob.send and not prob.recv
        assert np.array_equal(prob.l2g[prob.owned], np.flatnonzero(owner == r))
        owned_total += len(prob.owned)
        own_total += prob.num_own_elements()
        assert np.array_equal(mesh.vertices[prob.l2g], prob.mesh.vertices)
        ro, ci, vals = _oracle_partial(oracle, prob, oracle.LINEAR_ELASTIC)
        _check_owned_rows(prob, 3, ro, ci, vals, gro, gci, gvals)       # complete without any exchange
        ex = fp.make_part(mesh, part, r, world)                          # exchange mode: who talks to whom is symmetric
        for q, nodes in ex.send.items():
            other = fp.make_part(mesh, part, q, world)
            assert np.array_equal(ex.l2g[nodes], other.l2g[other.recv[r]])
    assert owned_total == mesh.num_nodes() and own_total == mesh.num_elements()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gloo_worker(rank, world, port, name, op_name, q):
    import torch
    import torch.distributed as dist

    from oracle import oracle

    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        op = getattr(oracle, op_name)
        s = 1 if op_name == "LAPLACE" else 3
        mesh = _mesh(name)
        part = fp.morton_partition(mesh, world)
        prob = fp.make_part(mesh, part, rank, world)
        ro, ci, vals = _oracle_partial(oracle, prob, op)
        values = torch.from_numpy(vals)
        ex = fp.PartExchange(prob).bind_offsets(ro, s, values)
        assert world == 1 or ex.bytes_sent() > 0 or len(prob.recv) > 0
        ex.run()
        gro, gci, gvals = _global_reference(oracle, mesh, op)
        _check_owned_rows(prob, s, ro, ci, values.numpy(), gro, gci, gvals)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as exc:  # pragma: no cover - reported to the parent
        import traceback

        q.put((rank, traceback.format_exc() + repr(exc)))


@pytest.mark.parametrize("name,world,op_name", [("sphere", 2, "LINEAR_ELASTIC"), ("sphere", 4, "LAPLACE"), ("bcc", 2, "LAPLACE"),
                                                ("bcc", 4, "LINEAR_ELASTIC")])
def test_exchange_over_gloo(name, world, op_name):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, name, op_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,op_name", [("sphere", 4, "LINEAR_ELASTIC"), ("bcc", 3, "LINEAR_ELASTIC"), ("hex", 4, "LINEAR_ELASTIC"),
                                                ("hex", 2, "LAPLACE")])
def test_parts_through_the_engine_match_the_global_oracle(oracle, name, world, op_name):
    """every rank's share through the engine on the one GPU of the test box (element mask, owner-computes kernels, values overwritten in an
    array of garbage); the transfers are device copies through the exchange's own index lists"""
    import torch

    op = getattr(oracle, op_name)
    s = 1 if op_name == "LAPLACE" else 3
    mesh = _mesh(name)
    w, p = _rule(mesh)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if s == 3:
        qt = qt.with_uniform_data(fa.LameParameters(*LAME))
    fop = fa.LaplaceOperator() if s == 1 else fa.MaterialEllipticOperator(fa.LinearElasticMaterial())

    def configure(engine, m):
        return fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(m).with_operator(fop).with_quadrature_table(qt).with_u(None).build()

    part = fp.morton_partition(mesh, world)
    ranks = [fp.PartAssembly(fp.make_part(mesh, part, r, world), configure, device=0) for r in range(world)]
    try:
        for pa in ranks:
            pa.values.fill_(3.5)
            pa.main.assemble_matrix_async(pa.values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
            pa.poll_status()
        torch.cuda.synchronize()
        partial = [pa.values.clone() for pa in ranks]
        for r, pa in enumerate(ranks):       # what PartExchange.run() does between processes
            for q_ in sorted(pa.exchange.recv_idx):
                pa.values.index_add_(0, pa.exchange.recv_idx[q_], partial[q_].index_select(0, ranks[q_].exchange.send_idx[r]))
        gro, gci, gvals = _global_reference(oracle, mesh, op)
        for pa in ranks:
            ro, ci = pa.main.pattern()
            _check_owned_rows(pa.prob, s, ro, ci, pa.values.cpu().numpy(), gro, gci, gvals)
    finally:
        for pa in ranks:
            pa.close()


# ---- the same logic behind the C ABI (fenris_amd/csrc/partition.cpp, group.hip): what a Rust / C host drives


@pytest.mark.parametrize("name,world", [("sphere", 3), ("bcc", 4), ("hex", 2), ("hex", 5)])
def test_c_abi_partition_equals_the_python_mirror(name, world):
    mesh = _mesh(name)
    part = fp.morton_partition(mesh, world)
    assert np.array_equal(fp.morton_partition_abi(mesh, world), part)
    rng = np.random.default_rng(17)
    for elem_to_part in (part, rng.integers(0, world, mesh.num_elements()).astype(np.int32)):    # Morton runs and a scattered partition
        for mode in ("exchange", "halo"):
            for r in range(world):
                a, b = fp.make_part(mesh, elem_to_part, r, world, mode), fp.make_part_abi(mesh, elem_to_part, r, world, mode)
                assert np.array_equal(a.l2g, b.l2g) and np.array_equal(a.elem_l2g, b.elem_l2g)
                assert np.array_equal(np.asarray(a.mesh.connectivity), np.asarray(b.mesh.connectivity))
                assert np.array_equal(a.active, b.active) and np.array_equal(a.owned, b.owned)
                assert a.num_own_elements() == b.num_own_elements()
                assert sorted(a.send) == sorted(b.send) and sorted(a.recv) == sorted(b.recv)
                for q in a.send:
                    assert np.array_equal(a.send[q], b.send[q])
                for q in a.recv:
                    assert np.array_equal(a.recv[q], b.recv[q])


def test_c_abi_partition_rejects_bad_arguments():
    mesh = _mesh("hex")
    part = np.zeros(mesh.num_elements(), dtype=np.int32)
    part[3] = 2
    with pytest.raises(ValueError):
        fp.make_part_abi(mesh, part, 0, 2)                    # a part outside [0, world)
    with pytest.raises(ValueError):
        fp.make_part(mesh, part, 0, 2)
    part[3] = -1
    with pytest.raises(ValueError):
        fp.make_part_abi(mesh, part, 0, 2)
    bad = fa.Mesh(mesh.vertices[:10], mesh.connectivity, fa.HEX8)       # node indices beyond the vertices
    with pytest.raises(ValueError):
        fp.make_part_abi(bad, np.zeros(mesh.num_elements(), dtype=np.int32), 0, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("op_name", ["LINEAR_ELASTIC", "LAPLACE"])
def test_c_abi_list_exchange_on_a_one_rank_communicator(op_name):
    """fh_group_set_exchange_nodes / _start / _finish with this rank as its own peer (RCCL on one GPU: two ranks cannot share a device):
    pack kernel -> ncclSend / ncclRecv to self in one group -> unpack-add kernel.  Two 'peers', the received lists in another order than the
    sent ones (rows of equal length: interior nodes of a box), against numpy on the same lists."""
    import torch

    s = 1 if op_name == "LAPLACE" else 3
    mesh = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 6, 5, 4, 1)
    w, p = _rule(mesh)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if s == 3:
        qt = qt.with_uniform_data(fa.LameParameters(*LAME))
    fop = fa.LaplaceOperator() if s == 1 else fa.MaterialEllipticOperator(fa.LinearElasticMaterial())
    eng = fa.Engine(0)
    try:
        fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(fop).with_quadrature_table(qt).with_u(None).build()
        nnz = eng.build_pattern()
        values = torch.zeros(nnz, dtype=torch.float64, device="cuda:0")
        eng.assemble_matrix(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
        ro = np.asarray(eng.pattern(want_cols=False)[0]).astype(np.int64)
        nblk = (ro[s::s] - ro[:-1:s]) // (s * s)                   # column blocks per node
        interior = np.flatnonzero(nblk == 27)
        assert len(interior) == 5 * 4 * 3
        rng = np.random.default_rng(3)
        sendA, sendB = rng.permutation(interior)[:40], rng.permutation(interior)[:25]
        recvA, recvB = rng.permutation(interior)[:40], rng.permutation(interior)[:25]
        prob = fp.PartProblem(mesh, np.arange(mesh.num_nodes()), np.arange(mesh.num_elements()), np.ones(mesh.num_elements(), np.uint8),
                              np.arange(mesh.num_nodes()), {1: sendA, 2: sendB}, {1: recvA, 2: recvB}, 0, 1)
        ex = fp.AbiPartExchange(prob, eng, self_loop=True).bind(eng, values)
        try:
            assert ex.bytes_sent() == 8 * 65 * 27 * s * s
            before = values.cpu().numpy().copy()
            want = before.copy()
            for snd, rcv in ((sendA, recvA), (sendB, recvB)):
                for a, b in zip(snd, rcv):
                    want[ro[s * b]: ro[s * b + s]] += before[ro[s * a]: ro[s * a + s]]
            ex.run()
            torch.cuda.synchronize()
            assert np.array_equal(values.cpu().numpy(), want)     # copies and one addition per entry: exact
            # a second exchange on the result (buffers reused), started and finished separately with work in between
            before = want.copy()
            for snd, rcv in ((sendA, recvA), (sendB, recvB)):
                for a, b in zip(snd, rcv):
                    want[ro[s * b]: ro[s * b + s]] += before[ro[s * a]: ro[s * a + s]]
            ex.start()
            torch.cuda.synchronize()
            ex.finish()
            torch.cuda.synchronize()
            assert np.array_equal(values.cpu().numpy(), want)
            # bad lists are refused with the engine's error, the group stays usable
            with pytest.raises(Exception):
                fp.AbiPartExchange.bind(type("X", (), {"prob": fp.PartProblem(mesh, None, None, None, None, {0: np.array([10 ** 7])}, {}, 0, 1),
                                                       "engine": eng, "_lib": ex._lib, "_g": ex._g, "_self_loop": True})(), eng, values)
        finally:
            ex.close()
    finally:
        eng.close()


# ---- residual vectors of a partition: own elements' contributions, then the partial sums at other ranks' nodes travel (run_vector)


def _oracle_partial_vector(oracle, prob, op, u_global):
    m = prob.mesh
    w, p = _rule(m)
    s = 1 if op == oracle.LAPLACE else m.vertices.shape[1]
    u = u_global.reshape(-1, s)[prob.l2g].reshape(-1)
    own = oracle.ElementAssembler(_okind(oracle, m), op, m.vertices, np.asarray(m.connectivity)[prob.active.astype(bool)], w, p,
                                  params=None if op == oracle.LAPLACE else LAME, u=u)
    st, _, f = oracle.assemble_vector(own)
    assert st == 0
    return s, u, f


def _global_vector(oracle, mesh, op, u_global):
    w, p = _rule(mesh)
    ref = oracle.ElementAssembler(_okind(oracle, mesh), op, mesh.vertices, mesh.connectivity, w, p, params=None if op == oracle.LAPLACE else LAME, u=u_global)
    st, _, f = oracle.assemble_vector(ref)
    assert st == 0
    return f


def _u_for(mesh, s):
    return 1e-3 * np.sin(np.arange(s * mesh.num_nodes()) * 0.37)


def _gloo_vector_worker(rank, world, port, name, op_name, q):
    import torch
    import torch.distributed as dist

    from oracle import oracle

    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        op = getattr(oracle, op_name)
        mesh = _mesh(name)
        sdim = 1 if op_name == "LAPLACE" else 3
        ug = _u_for(mesh, sdim)
        prob = fp.make_part(mesh, fp.morton_partition(mesh, world), rank, world)
        s, _, f = _oracle_partial_vector(oracle, prob, op, ug)
        vec = torch.from_numpy(f.copy())
        fp.PartExchange(prob).run_vector(vec, s)
        want = _global_vector(oracle, mesh, op, ug).reshape(-1, s)
        got = vec.numpy().reshape(-1, s)
        owned = np.asarray(prob.owned)
        assert np.abs(got[owned] - want[prob.l2g[owned]]).max() <= 1e-12 * np.abs(want).max()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as exc:  # pragma: no cover - reported to the parent
        import traceback

        q.put((rank, traceback.format_exc() + repr(exc)))


@pytest.mark.parametrize("name,world,op_name", [("sphere", 3, "LINEAR_ELASTIC"), ("bcc", 4, "LAPLACE")])
def test_vector_exchange_over_gloo(name, world, op_name):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_vector_worker, args=(r, world, port, name, op_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,op_name", [("sphere", 3, "LINEAR_ELASTIC"), ("hex", 4, "LINEAR_ELASTIC"), ("bcc", 2, "LAPLACE")])
def test_partition_residuals_through_the_engine(oracle, name, world, op_name):
    """every rank's residual through the engine on the one GPU (element mask -> the tiled element pass zeroes the halo elements), the
    interface sums moved by device copies through the exchange's node lists; owned entries against the single-process oracle vector"""
    import torch

    op = getattr(oracle, op_name)
    s = 1 if op_name == "LAPLACE" else 3
    mesh = _mesh(name)
    ug = _u_for(mesh, s)
    w, p = _rule(mesh)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if s == 3:
        qt = qt.with_uniform_data(fa.LameParameters(*LAME))
    fop = fa.LaplaceOperator() if s == 1 else fa.MaterialEllipticOperator(fa.LinearElasticMaterial())
    part = fp.morton_partition(mesh, world)
    probs = [fp.make_part(mesh, part, r, world) for r in range(world)]

    def configure_for(prob):
        u_local = ug.reshape(-1, s)[prob.l2g].reshape(-1)

        def configure(engine, m):
            return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(m).with_operator(fop).with_quadrature_table(qt)
                    .with_u(u_local).build())
        return configure

    ranks = [fp.PartAssembly(pr, configure_for(pr), device=0) for pr in probs]
    try:
        outs = []
        for pa in ranks:
            out = torch.zeros(s * pa.prob.mesh.num_nodes(), dtype=torch.float64, device="cuda:0")
            pa.main.assemble_vector(out)
            assert "tiled" in pa.main.last_kernel_name()
            outs.append(out)
        partial = [o.clone() for o in outs]
        for r, pa in enumerate(ranks):      # what run_vector does between processes
            for q_ in sorted(pa.prob.recv):
                src = torch.as_tensor(np.asarray(ranks[q_].prob.send[r]), device="cuda:0")
                dst = torch.as_tensor(np.asarray(pa.prob.recv[q_]), device="cuda:0")
                outs[r].view(-1, s).index_add_(0, dst, partial[q_].view(-1, s).index_select(0, src))
        want = _global_vector(oracle, mesh, op, ug).reshape(-1, s)
        for r, pa in enumerate(ranks):
            owned = np.asarray(pa.prob.owned)
            got = outs[r].cpu().numpy().reshape(-1, s)
            assert np.abs(got[owned] - want[pa.prob.l2g[owned]]).max() <= 1e-12 * np.abs(want).max(), (name, r)
    finally:
        for pa in ranks:
            pa.close()


@pytest.mark.gpu
def test_c_abi_vector_exchange_on_a_one_rank_communicator():
    """fh_group_exchange_vector_start / _finish with the rank as its own peer: pack kernel -> ncclSend / ncclRecv -> unpack-add, against numpy"""
    import torch

    mesh = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 6, 5, 4, 1)
    w, p = _rule(mesh)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(fa.LameParameters(*LAME))
    eng = fa.Engine(0)
    try:
        (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial()))
         .with_quadrature_table(qt).with_u(None).build())
        eng.build_pattern()
        n = mesh.num_nodes()
        rng = np.random.default_rng(4)
        sendA, sendB = rng.permutation(n)[:50], rng.permutation(n)[:31]
        recvA, recvB = rng.permutation(n)[:50], rng.permutation(n)[:31]
        prob = fp.PartProblem(mesh, np.arange(n), np.arange(mesh.num_elements()), np.ones(mesh.num_elements(), np.uint8), np.arange(n),
                              {1: sendA, 2: sendB}, {1: recvA, 2: recvB}, 0, 1)
        values = torch.zeros(1, dtype=torch.float64, device="cuda:0")
        ex = fp.AbiPartExchange(prob, eng, self_loop=True).bind(eng, values)
        try:
            for comp in (3, 1):
                vec = torch.as_tensor(rng.standard_normal(comp * n), device="cuda:0")
                before = vec.cpu().numpy().reshape(-1, comp).copy()
                want = before.copy()
                for snd, rcv in ((sendA, recvA), (sendB, recvB)):
                    np.add.at(want, rcv, before[snd])
                ex.run_vector(vec, comp)
                torch.cuda.synchronize()
                assert np.array_equal(vec.cpu().numpy().reshape(-1, comp), want), comp
        finally:
            ex.close()
    finally:
        eng.close()


@pytest.mark.gpu
def test_part_assembly_accumulate_contract(oracle):
    """Advisor finding of round 4: the interface exchange ships rows / vector entries as they stand and the owner adds them, so (a) the matrix of
    a partition must be assembled with ASSEMBLE_OVERWRITE -- anything else is rejected -- and (b) ``assemble_vector`` accumulates into ``out`` like
    the reference (global.rs:582-608) by way of a zeroed scratch vector: a pre-filled ``out`` keeps its content and gains the owned entries."""
    import torch

    mesh = _mesh("sphere")
    op = oracle.LINEAR_ELASTIC
    s = 3
    ug = _u_for(mesh, s)
    w, p = _rule(mesh)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(fa.LameParameters(*LAME))
    fop = fa.MaterialEllipticOperator(fa.LinearElasticMaterial())
    part = fp.morton_partition(mesh, 1)
    prob = fp.make_part(mesh, part, 0, 1)

    def configure(engine, m):
        return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(m).with_operator(fop).with_quadrature_table(qt)
                .with_u(ug.reshape(-1, s)[prob.l2g].reshape(-1)).build())

    pa = fp.PartAssembly(prob, configure, device=0)
    try:
        with pytest.raises(ValueError):
            pa.enqueue(fa.SCATTER_GATHER)
        pa.enqueue(fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
        pa.poll_status()
        want = _global_vector(oracle, mesh, op, ug)
        out = torch.full((s * prob.mesh.num_nodes(),), 2.5, dtype=torch.float64, device="cuda:0")
        pa.assemble_vector(out)
        pa.assemble_vector(out)      # accumulates: twice the vector on top of the earlier content
        got = out.cpu().numpy().reshape(-1, s)
        owned = np.asarray(prob.owned)
        ref = want.reshape(-1, s)[prob.l2g[owned]]
        assert np.abs(got[owned] - (2.5 + 2.0 * ref)).max() <= 1e-12 * max(1.0, np.abs(want).max())
    finally:
        pa.close()
