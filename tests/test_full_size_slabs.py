"""What ONE RANK of the multi-GPU runs assembles, at the FULL size of BASELINE's configurations, and the N > 1 line of bench.py itself.

Round 3 showed that green small-mesh tests can hide a defect that only appears when a persistent workgroup walks thousands of positions
under an element mask -- i.e. in every multi-GPU partition at benchmark size.  These tests put the full-size masked slabs of
`bench.py --gpus 8` (ns: 216 x 216 x 216 own cells per rank, c5: 256 x 256 x 32) and the bench's own N = 2 path (all ranks on the one
device of the test box, gloo) under `pytest -m gpu`:

* rank 1 of 8 (a middle slab: halo layer below and above, ghost plane to send, top plane to receive into), both launches of
  SlabAssembly into an array of garbage, against (i) the atomic scatter of the same context and (ii) the CPU oracle at 1e-12 on one row block
  each of the bottom ghost plane, the owned top interface plane and the interior (translation invariance: the rows of a node of a uniform
  box depend only on the cell size and on WHICH of its eight elements are active);
* `FENRIS_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2` as a fresh child process.

Reference: CsrParAssembler::assemble_into_csr (global.rs:314-376) is what every rank's launch replaces; SURVEY 8e is the partition."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import distributed as fd
from fenris_amd import quadrature

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
TOL = 1e-12


def _oracle_rows_of_centre_node(oracle, h, layers):
    """rows of the centre node of a 2 x 2 x 2 box of cell size h in which only the element layers `layers` (subset of {0, 1}) are assembled --
    the pattern is the one of the whole box (what the halo layers of the extended mesh give an interface plane)"""
    small = fa.procedural.create_rectangular_uniform_hex_mesh(2 * h, 1, 1, 1, 2)
    w, p = quadrature.tensor.hexahedron_gauss(2)
    full = oracle.ElementAssembler(oracle.HEX8, oracle.LINEAR_ELASTIC, small.vertices, small.connectivity, w, p, params=LAME.as_pair())
    ro, ci = oracle.pattern_for(full)
    conn = np.asarray(small.connectivity)
    layer_of = np.arange(len(conn)) // 4                      # cells are generated layer by layer (procedural.rs:253-271)
    sub = oracle.ElementAssembler(oracle.HEX8, oracle.LINEAR_ELASTIC, small.vertices, conn[np.isin(layer_of, layers)], w, p,
                                  params=LAME.as_pair())
    vals = np.zeros(len(ci))
    st, _ = oracle.assemble_into_csr(sub, ro, ci, vals)
    assert st == 0
    centre = 1 + 3 + 9
    assert ro[3 * centre + 1] - ro[3 * centre] == 81          # all 27 neighbours
    return vals[int(ro[3 * centre]): int(ro[3 * centre + 3])]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["ns-slab", "c5-slab"])
def test_rank_1_of_8_at_full_size(oracle, cfg):
    import torch

    w, p = quadrature.tensor.hexahedron_gauss(2)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(LAME)

    def configure(engine, mesh):
        return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh)
                .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(None).build())

    if cfg == "ns-slab":      # bench.py --gpus 8 (weak scaling): 216 x 216 x (216 * 8) cells, 216 own layers per rank
        cells, slab = 216, fd.make_slab(1.0, 1, 1, 8, 216, 1, 8)
        assert slab.num_own_elements() == 216 ** 3
    else:                     # bench.py --gpus 8 --config c5: BASELINE's 256^3 cut into eight slabs of 32 layers
        cells, slab = 256, fd.make_slab(1.0, 1, 1, 1, 256, 1, 8)
        assert slab.num_own_elements() == 256 * 256 * 32
    h = 1.0 / cells
    sa = fd.SlabAssembly(slab, configure, device=0, overlap=True, stream=torch.cuda.current_stream().cuda_stream)
    try:
        flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
        sa.values.fill_(-11.5)
        sa.main.assemble_matrix_rows_async(sa.values, flags, 0, sa.split)
        sa.main.assemble_matrix_async(sa.values, flags)
        sa.poll_status()
        torch.cuda.synchronize()
        assert "k_affine_rows" in sa.main.last_kernel_name()
        ro, _ = sa.main.pattern(want_cols=False)
        # (ii) the oracle on three row blocks: translation invariance of the uniform box
        npl = (cells + 1) ** 2
        inplane = (cells // 2) + (cells + 1) * (cells // 3)                  # away from the box faces
        send_plane, recv_plane = slab.send_nodes[0] // npl, slab.recv_nodes[0] // npl
        for plane, layers in ((send_plane, [1]),          # bottom ghost plane: own elements lie above it only
                              (recv_plane, [0]),          # owned top plane: own elements below it; the rest arrives from rank 2
                              ((send_plane + recv_plane) // 2, [0, 1])):
            node = plane * npl + inplane
            got = sa.values[int(ro[3 * node]): int(ro[3 * node + 3])].cpu().numpy()
            want = _oracle_rows_of_centre_node(oracle, h, layers)
            assert got.shape == want.shape
            scale = np.abs(_oracle_rows_of_centre_node(oracle, h, [0, 1])).max()
            assert np.abs(got - want).max() <= TOL * scale, (cfg, plane, np.abs(got - want).max() / scale)
        # (i) every value against the atomic scatter of the same context (generic element kernel + fp64 atomics, no owner tables)
        sa.main.set_row_range(0, slab.mesh.num_nodes())
        want = torch.zeros_like(sa.values)
        sa.main.assemble_matrix(want, fa.SCATTER_ATOMIC)
        scale = want.abs().max().item()
        diff = (sa.values - want).abs().max().item() / scale
        assert diff <= 1e-12, (cfg, diff)
        del want
    finally:
        sa.close()
        torch.cuda.empty_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,extra", [("ns", []), ("c5", []), ("ns", ["--exchange", "torch", "--no-overlap"]), ("ns", ["--partition", "halo"]),
                                       ("c3", []), ("ns-perturbed", []), ("c3", ["--partition", "halo"])])
def test_bench_two_ranks_sharing_the_device(cfg, extra):
    """the N > 1 path of bench.py end to end (self-launch through torch.distributed.run, slabs or -- c3, ns-perturbed: meshes without planes --
    element partitions with packed interface lists, masks, the launches, the exchange, the max-over-ranks timing, rank 0's line) on the one
    GPU of the test box: validation mode, gloo instead of RCCL"""
    env = dict(os.environ, FENRIS_BENCH_SHARE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cells = 8 if cfg in ("c3", "ns-perturbed") else 24
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", cfg, "--cells", str(cells), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-traffic", "--placement-tries", "0", "--no-settle"] + extra
    pr = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, pr.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["value"] > 0 and line["higher_is_better"] is True
    assert line["scaling"] == ("weak" if cfg == "ns" else "strong")
    own = {"ns": 2 * 24 ** 3, "c5": 24 ** 3, "c3": 12 * 8 ** 3, "ns-perturbed": 8 ** 3}[cfg]
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - own) <= 1e-6 * own      # whole-job units / max-over-ranks time
    if cfg in ("c3", "ns-perturbed"):
        assert sum(line["config"]["elements_per_rank"]) == own
    if "halo" not in extra:
        # rank 0's owned interface rows are complete only once rank 1's rows have arrived and been added: stiffness rows sum to zero
        assert float(line["config"]["interface_row_sum_over_max"]) <= 1e-12, line["config"]


@pytest.mark.gpu
def test_bench_two_ranks_c5_at_full_size_sharing_the_device():
    """C5 (Hex8 elasticity 256^3) cut into two slabs of 128 layers, both ranks on the one GPU of the test box, through
    bench.py's own N > 1 path: the interface plane of 257^2 nodes (3 x 81 values per node row: 128 MB) goes through the exchange the scaling
    run will use, and rank 0's owned interface rows sum to zero once rank 1's rows have been added."""
    env = dict(os.environ, FENRIS_BENCH_SHARE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c5", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-traffic", "--placement-tries", "0", "--no-settle"]
    pr = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, pr.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    own = 256 ** 3
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - own) <= 1e-6 * own
    assert float(line["config"]["interface_row_sum_over_max"]) <= 1e-12, line["config"]
    # per-rank times and the exchange of the line (item 6c): both ranks report, the exchange is a part of the step
    pr_ms = line["config"]["per_rank_ms_per_step"]
    assert len(pr_ms) == 2 and min(pr_ms) > 0 and max(pr_ms) <= line["ms_per_step"] * 1.0001
    assert line["config"]["exchange_ms"] is not None and line["config"]["exchange_ms"] >= 0.0


@pytest.mark.gpu
def test_c_abi_list_exchange_at_c5_interface_size_on_a_one_rank_communicator():
    """the library's own RCCL exchange (fh_group_set_exchange_nodes: pack kernel, ncclSend / ncclRecv in one group,
    unpack-add kernel) with C5's real interface volume -- a plane of 257 x 257 nodes, 3 x 81 doubles per node row (128 MB) -- with the rank as
    its own peer (two ranks cannot share a device under RCCL).  Rows of the plane are added onto themselves: exactly twice the assembled rows."""
    import torch

    from fenris_amd import partition as fp

    mesh = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 256)
    # three layers of the 256^3 mesh around one interface plane: 256 x 256 x 2 cells
    nx = 257
    keep = np.flatnonzero(np.asarray(mesh.vertices)[:, 2] <= 2.0 / 256 + 1e-12)
    assert len(keep) == nx * nx * 3
    conn = np.asarray(mesh.connectivity).astype(np.int64)
    inside = np.all(np.isin(conn, keep), axis=1)
    remap = -np.ones(mesh.num_nodes(), dtype=np.int64)
    remap[keep] = np.arange(len(keep))
    sub = fa.Mesh(np.asarray(mesh.vertices)[keep], remap[conn[inside]].astype(np.uint64), fa.HEX8)
    del mesh
    w, p = quadrature.tensor.hexahedron_gauss(2)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2)))
    eng = fa.Engine(0)
    try:
        (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(sub).with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial()))
         .with_quadrature_table(qt).with_u(None).build())
        nnz = eng.build_pattern()
        values = torch.zeros(nnz, dtype=torch.float64, device="cuda:0")
        flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
        eng.assemble_matrix(values, flags)
        plane = np.flatnonzero(np.abs(np.asarray(sub.vertices)[:, 2] - 1.0 / 256) <= 1e-12)
        assert len(plane) == nx * nx
        prob = fp.PartProblem(sub, np.arange(sub.num_nodes()), np.arange(sub.num_elements()), np.ones(sub.num_elements(), np.uint8),
                              np.arange(sub.num_nodes()), {1: plane}, {1: plane}, 0, 1)
        ex = fp.AbiPartExchange(prob, eng, self_loop=True).bind(eng, values)
        try:
            ro = np.asarray(eng.pattern(want_cols=False)[0]).astype(np.int64)
            lo, hi = ro[3 * plane[0]], ro[3 * plane[-1] + 3]
            before = values.clone()
            sent = ex.bytes_sent()
            # a node row of the plane: 3 x 3 blocks with up to 27 neighbour nodes = 3 rows of 81 doubles (128 MB for the plane)
            assert sent >= 8 * (nx - 2) * (nx - 2) * 27 * 9 and sent <= 8 * nx * nx * 27 * 9
            ex.run()
            torch.cuda.synchronize()
            # the plane's rows are contiguous (consecutive node ids): doubled; everything else untouched
            assert torch.equal(values[lo:hi], 2.0 * before[lo:hi])
            assert torch.equal(values[:lo], before[:lo]) and torch.equal(values[hi:], before[hi:])
        finally:
            ex.close()
    finally:
        eng.close()
        torch.cuda.empty_cache()
