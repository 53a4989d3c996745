"""Assembled-matrix fixtures (tests/golden/csr/*.npz, made by tests/golden/make_csr_fixtures.py): the oracle (CPU) and the HIP
path (GPU, through the C ABI) against STORED offsets / indices / values -- index arrays bit-exact, values to 1e-12 of the
largest entry, NaN positions identical (NeoHookean, det F <= 0: fenris-solid/src/materials.rs:298-300).  SURVEY.md 7 step 2 / 8c.
Also configuration C1 of BASELINE.json (examples/poisson2d.rs:33-60: Quad4 on the 64 x 64 unit square) against the oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import csr_cases  # noqa: E402

import fenris_amd as fa  # noqa: E402

TOL = 1e-12
NAMES = ["quad4_4x4_laplace", "hex8_2_laplace", "hex8_3_laplace", "hex8_2_elastic", "hex8_3_elastic", "tet4_bcc1_laplace",
         "tet4_bcc2_elastic", "tet4_sphere593_elastic", "hex27_2_neohookean", "hex27_2_neohookean_inverted"]


def _check(ro, ci, vals, name):
    gro, gci, gvals = csr_cases.load(name)
    assert np.array_equal(np.asarray(ro, dtype=np.uint64), gro) and np.array_equal(np.asarray(ci, dtype=np.uint64), gci)
    nan = np.isnan(gvals)
    assert np.array_equal(np.isnan(vals), nan)
    assert np.abs(vals[~nan] - gvals[~nan]).max() <= TOL * np.abs(gvals[~nan]).max()
    return int(nan.sum())


def test_fixture_table_is_complete(oracle):
    assert sorted(NAMES) == sorted(csr_cases.cases(oracle))
    for name in NAMES:
        assert os.path.exists(os.path.join(csr_cases.CSR_DIR, name + ".npz")), name


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_stored_csr(oracle, name):
    asm, _ = csr_cases.oracle_assembler(oracle, name)
    st, _, ro, ci, vals = oracle.assemble(asm)
    assert st == 0
    n_nan = _check(ro, ci, vals, name)
    assert (n_nan > 0) == name.endswith("_inverted")


KIND = {"QUAD4": fa.QUAD4, "HEX8": fa.HEX8, "TET4": fa.TET4, "HEX27": fa.HEX27}


def _hip_assembler(engine, oracle, name):
    _, (kind, op, v, c, w, p, params, u) = csr_cases.oracle_assembler(oracle, name)
    mesh = fa.Mesh(v, c, KIND[kind])
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if op == "LAPLACE":
        oper = fa.LaplaceOperator()
    else:
        qt = qt.with_uniform_data(fa.LameParameters(*params))
        oper = fa.MaterialEllipticOperator(fa.LinearElasticMaterial() if op == "LINEAR_ELASTIC" else fa.NeoHookeanMaterial())
    return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(oper)
            .with_quadrature_table(qt).with_u(u).build())


@pytest.mark.gpu
@pytest.mark.parametrize("scatter", ["gather", "atomic", "colored"])
@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_stored_csr(oracle, name, scatter):
    engine = fa.Engine(0)
    try:
        asm = _hip_assembler(engine, oracle, name)
        mode = {"gather": fa.SCATTER_GATHER, "atomic": fa.SCATTER_ATOMIC, "colored": fa.SCATTER_COLORED}[scatter]
        k = fa.CsrAssembler(mode).assemble(asm)
        _check(k.row_offsets, k.col_indices, k.values, name)
    finally:
        engine.close()


@pytest.mark.gpu
def test_c1_quad4_64x64_against_oracle(oracle):
    """BASELINE config C1: examples/poisson2d.rs -- Quad4 on create_unit_square_uniform_quad_mesh_2d(64), quadrilateral_gauss(2),
    CsrAssembler::assemble.  4 096 elements, 4 225 nodes, nnz = 193^2 = 37 249; indices bit-exact, values 1e-12."""
    v, c = oracle.unit_square_quad_mesh(64)
    w, p = oracle.quadrilateral_gauss(2)
    ref = oracle.ElementAssembler(oracle.QUAD4, oracle.LAPLACE, v, c, w, p)
    st, _, ro, ci, vals = oracle.assemble(ref)
    assert st == 0 and len(v) == 4225 and len(c) == 4096 and len(vals) == 193 * 193
    engine = fa.Engine(0)
    try:
        mesh = fa.procedural.create_unit_square_uniform_quad_mesh_2d(64)
        assert np.array_equal(mesh.vertices, v) and np.array_equal(mesh.connectivity, c)
        asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(fa.LaplaceOperator())
               .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w)).with_u(None).build())
        for mode in (fa.SCATTER_GATHER, fa.SCATTER_ATOMIC, fa.SCATTER_COLORED):
            k = fa.CsrAssembler(mode).assemble(asm)
            assert np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
            assert np.abs(k.values - vals).max() <= TOL * np.abs(vals).max()
            # constants are in the null space of the Laplace stiffness matrix
            rowsum = np.add.reduceat(k.values, ro[:-1].astype(np.int64))
            assert np.abs(rowsum).max() <= 1e-12 * np.abs(vals).max() * 9
    finally:
        engine.close()
