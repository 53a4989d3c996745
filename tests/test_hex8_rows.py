"""k_hex8_rows (fenris_amd/csrc/hex8_rows.hip): the row-owner form of the general Hex8 stiffness kernel (Laplace / uniform LinearElastic,
hexahedron_gauss(2)) -- what every hexahedral mesh that is not a box of parallelepipeds runs, `bench.py --config ns-perturbed`.

It replaces the LDS atomics of k_gather_pipelined by sums in registers with a fixed order, so beyond parity with the oracle
(elliptic.rs:361-439, materials.rs:108-118, laplace.rs:60-68) these tests pin what the old kernel could not promise: K symmetric bit for
bit (util.rs:38-51 clone_upper_to_lower), the same bits from run to run, for every launch grid (a workgroup that walks many positions
keeps the gradients of shared elements staged; one that walks a single position computes them all), with and without the bank-conflict
tuning of the lane tables."""
import os

import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature

pytestmark = pytest.mark.gpu
TOL = 1e-12
LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
OPS = {"LAPLACE": lambda: fa.LaplaceOperator(), "LINEAR_ELASTIC": lambda: fa.MaterialEllipticOperator(fa.LinearElasticMaterial())}


def _box(nx, ny, nz, seed=3, amp=0.15, holes=0.0, mirror=0.0):
    m = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, nx, ny, nz, 1)
    rng = np.random.default_rng(seed)
    v = m.vertices + amp * rng.uniform(-1.0, 1.0, m.vertices.shape)
    c = np.asarray(m.connectivity).copy()
    if mirror:
        flip = rng.random(len(c)) < mirror
        c[flip] = c[flip][:, [4, 5, 6, 7, 0, 1, 2, 3]]     # bottom and top face swapped: det J < 0
    if holes:
        keep = rng.random(len(c)) >= holes
        keep[0] = True
        c = c[keep]
    return fa.Mesh(v, c, fa.HEX8)


def _asm(engine, mesh, op):
    w, p = quadrature.tensor.hexahedron_gauss(2)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if op != "LAPLACE":
        qt = qt.with_uniform_data(LAME)
    return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(OPS[op]()).with_quadrature_table(qt)
            .with_u(None).build()), (w, p)


def _engine_with(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return fa.Engine(0)     # fh_create reads the FENRIS_HIP_* switches
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC"])
def test_against_the_oracle_exact_symmetry_and_every_variant_bit_for_bit(oracle, op):
    mesh = _box(9, 6, 5, mirror=0.3)
    s = 1 if op == "LAPLACE" else 3
    ref_k = None
    variants = [{}, {"FENRIS_HIP_NO_LANE_TUNING": 1}, {"FENRIS_HIP_PIPE_GRID": 1}, {"FENRIS_HIP_PIPE_GRID": 2}, {"FENRIS_HIP_PIPE_GRID": 7}]
    for env in variants:
        eng = _engine_with(env)
        try:
            asm, (w, p) = _asm(eng, mesh, op)
            k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
            assert eng.last_kernel_name() == "k_hex8_rows", env
            if ref_k is None:
                ref = oracle.ElementAssembler(oracle.HEX8, getattr(oracle, op), mesh.vertices, mesh.connectivity, w, p,
                                              params=None if s == 1 else LAME.as_pair())
                st, _, oro, oci, ovals = oracle.assemble(ref)
                assert st == 0
                assert np.array_equal(k.row_offsets, oro) and np.array_equal(k.col_indices, oci)
                assert np.abs(k.values - ovals).max() <= TOL * np.abs(ovals).max()
                a = k.to_scipy()
                assert (a != a.T).nnz == 0                                   # symmetric bit for bit
                again = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
                assert np.array_equal(again.values, k.values)                # and from run to run
                ref_k = k.values
            else:
                # another launch grid / lane arrangement: the same sums in the same order
                assert np.array_equal(k.values, ref_k), env
        finally:
            eng.close()
    # the kernel it replaces (LDS atomics: hardware order) agrees to rounding
    eng = _engine_with({"FENRIS_HIP_NO_HEX8_ROWS": 1})
    try:
        asm, _ = _asm(eng, mesh, op)
        k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert eng.last_kernel_name() == "k_gather_pipelined"
        assert np.abs(k.values - ref_k).max() <= TOL * np.abs(ref_k).max()
    finally:
        eng.close()


@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC"])
def test_holes_many_positions_per_workgroup_accumulate_and_overwrite(op):
    """a box with holes (rows of different lengths, positions that do not continue each other in memory: no carried lines), two
    workgroups for some two hundred positions, into garbage and on top of existing values -- against the atomic scatter"""
    import torch

    mesh = _box(14, 9, 8, seed=11, holes=0.2)
    eng = _engine_with({"FENRIS_HIP_PIPE_GRID": 2})
    try:
        _asm(eng, mesh, op)
        nnz = eng.build_pattern()
        want = torch.zeros(nnz, dtype=torch.float64, device="cuda")
        eng.assemble_matrix(want, fa.SCATTER_ATOMIC)
        scale = want.abs().max().item()
        got = torch.full((nnz,), -7.25, dtype=torch.float64, device="cuda")
        eng.assemble_matrix(got, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
        assert "k_hex8_rows" in eng.last_kernel_name()
        assert (got - want).abs().max().item() <= TOL * scale
        eng.assemble_matrix(got, fa.SCATTER_GATHER)                          # accumulates: 2 K
        assert (got - 2.0 * want).abs().max().item() <= 2 * TOL * scale
    finally:
        eng.close()


def test_singular_element_is_reported_with_its_index():
    mesh = _box(5, 4, 3, amp=0.1)
    v = mesh.vertices.copy()
    conn = np.asarray(mesh.connectivity).astype(np.int64)
    bad = 17
    v[conn[bad]] = 0.0                                                       # J == 0 exactly at every point
    eng = fa.Engine(0)
    try:
        asm, _ = _asm(eng, fa.Mesh(v, mesh.connectivity, fa.HEX8), "LINEAR_ELASTIC")
        with pytest.raises(fa.SingularJacobianError) as ei:
            fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert "k_hex8_rows" in eng.last_kernel_name()
        assert ei.value.element <= bad
    finally:
        eng.close()


def test_other_rules_keep_the_pipelined_kernel_and_masks_run_here():
    """the row-owner kernel is laid out for the eight-point rule and uniform parameters: other rules stay where they were; an element mask
    (every multi-GPU partition) runs here -- blocks without an active element get lanes that store zeros"""
    mesh = _box(6, 5, 4)
    eng = fa.Engine(0)
    try:
        w3, p3 = quadrature.tensor.hexahedron_gauss(3)
        asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(OPS["LINEAR_ELASTIC"]())
               .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p3, w3).with_uniform_data(LAME)).with_u(None).build())
        fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert eng.last_kernel_name() == "k_gather_pipelined"
        asm, _ = _asm(eng, mesh, "LINEAR_ELASTIC")
        mask = np.ones(mesh.num_elements(), dtype=np.uint8)
        mask[::3] = 0
        eng.set_active_elements(mask)
        import torch
        nnz = eng.build_pattern()
        got = torch.full((nnz,), 9.75, dtype=torch.float64, device="cuda")
        eng.assemble_matrix(got, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
        assert eng.last_kernel_name() == "k_hex8_rows"
        want = torch.zeros_like(got)
        eng.assemble_matrix(want, fa.SCATTER_ATOMIC)
        assert (got - want).abs().max().item() <= TOL * want.abs().max().item()
        eng.set_active_elements(None)
        fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert eng.last_kernel_name() == "k_hex8_rows"
    finally:
        eng.close()
