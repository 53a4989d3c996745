"""Affine-element fast path of the owner-computes stiffness assembly (fenris_amd/csrc/affine_kernel.hpp): node blocks whose
elements are all parallelepipeds run on k_affine_rows (K_ab = |det J| C(J^-T Ghat_ab J^-1), no quadrature loop), every other
block keeps the general kernels.  Parity against the oracle (elliptic.rs:361-439 restated) on affine, mixed and non-affine
meshes; exact symmetry like clone_upper_to_lower (util.rs:38-51); run-to-run reproducibility; the switch in the ABI."""
import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature

pytestmark = pytest.mark.gpu
TOL = 1e-12
LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))


@pytest.fixture()
def engine():
    eng = fa.Engine(0)
    yield eng
    eng.close()


def _sheared(mesh, A, b=(0.0, 0.0, 0.0)):
    """affine image of a mesh: every element stays a parallelepiped (up to rounding of the coordinates)"""
    return fa.Mesh(mesh.vertices @ np.asarray(A, dtype=np.float64).T + np.asarray(b), mesh.connectivity, fa.HEX8)


def _meshes():
    rng = np.random.default_rng(5)
    out = {}
    out["box9"] = fa.procedural.create_unit_box_uniform_hex_mesh_3d(9)
    out["slab_17x3x2"] = fa.procedural.create_rectangular_uniform_hex_mesh(0.5, 17, 3, 2, 1)
    out["single_element"] = fa.procedural.create_unit_box_uniform_hex_mesh_3d(1)
    # graded box: element sizes differ from element to element, every element is still a box
    g = fa.procedural.create_unit_box_uniform_hex_mesh_3d(6)
    out["graded"] = fa.Mesh(np.stack([g.vertices[:, 0] ** 1.7, g.vertices[:, 1] ** 0.8 * 2.0, np.expm1(g.vertices[:, 2])], axis=1),
                            g.connectivity, fa.HEX8)
    out["sheared"] = _sheared(fa.procedural.create_unit_box_uniform_hex_mesh_3d(5),
                              [[1.0, 0.3, 0.1], [0.0, 0.8, -0.2], [0.25, 0.0, 1.4]], (3.0, -1.0, 0.5))
    # mirrored (det J < 0): x -> -x
    out["mirrored"] = _sheared(fa.procedural.create_unit_box_uniform_hex_mesh_3d(4), np.diag([-1.0, 1.0, 1.0]))
    # mixed: the vertices of one corner region are perturbed, the rest stays affine
    m = fa.procedural.create_unit_box_uniform_hex_mesh_3d(10)
    v = m.vertices.copy()
    sel = (v[:, 0] > 0.55) & (v[:, 1] > 0.35)
    v[sel] += 0.02 * rng.standard_normal((int(sel.sum()), 3))
    out["mixed"] = fa.Mesh(v, m.connectivity, fa.HEX8)
    # scattered non-affine elements: single vertices moved
    v = m.vertices.copy()
    idx = rng.choice(len(v), 25, replace=False)
    v[idx] += 0.015 * rng.standard_normal((25, 3))
    out["pocked"] = fa.Mesh(v, m.connectivity, fa.HEX8)
    # nothing affine
    out["perturbed"] = fa.Mesh(m.vertices + 0.1 / 10 * rng.uniform(-1, 1, m.vertices.shape), m.connectivity, fa.HEX8)
    return out


def _assemblers(engine, oracle, mesh, op, rule=2):
    w, p = quadrature.tensor.hexahedron_gauss(rule)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if op == "LAPLACE":
        oper, oparams, oop = fa.LaplaceOperator(), None, oracle.LAPLACE
    else:
        qt = qt.with_uniform_data(LAME)
        oper, oparams, oop = fa.MaterialEllipticOperator(fa.LinearElasticMaterial()), LAME.as_pair(), oracle.LINEAR_ELASTIC
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(oper)
           .with_quadrature_table(qt).with_u(None).build())
    ref = oracle.ElementAssembler(oracle.HEX8, oop, mesh.vertices, mesh.connectivity, w, p, params=oparams)
    return asm, ref


EXPECT = {  # which kernels must have run
    "box9": "affine", "slab_17x3x2": "affine", "single_element": "affine", "graded": "affine", "sheared": "affine",
    "mirrored": "affine", "mixed": "both", "pocked": "both", "perturbed": "general",
}


def _symmetric_bitwise(k):
    import scipy.sparse as sp

    n = len(k.row_offsets) - 1
    A = sp.csr_matrix((k.values, k.col_indices.astype(np.int64), k.row_offsets.astype(np.int64)), shape=(n, n))
    D = (A - A.T).tocoo()
    return D.nnz == 0 or not np.any(D.data != 0.0)


@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC"])
@pytest.mark.parametrize("name", sorted(EXPECT))
def test_affine_path_matches_oracle(engine, oracle, name, op):
    mesh = _meshes()[name]
    asm, ref = _assemblers(engine, oracle, mesh, op)
    st, _, ro, ci, vals = oracle.assemble(ref)
    assert st == 0
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    kern = engine.last_kernel_name()
    n_aff_el, n_aff_blk, n_gen_blk = engine.affine_stats()
    if EXPECT[name] == "affine":
        assert kern == "k_affine_rows" and n_gen_blk == 0 and n_aff_el == mesh.num_elements()
    elif EXPECT[name] == "both":
        assert kern.startswith("k_affine_rows + ") and n_aff_blk > 0 and n_gen_blk > 0
        assert 0 < n_aff_el < mesh.num_elements()
    else:
        assert "affine" not in kern and n_aff_blk == 0 and n_aff_el == 0
    assert np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
    assert np.abs(k.values - vals).max() <= TOL * np.abs(vals).max()
    if EXPECT[name] == "affine":
        # clone_upper_to_lower makes K_e symmetric scalar by scalar (util.rs:46-50): so is the assembled matrix
        assert _symmetric_bitwise(k)
        # fixed summation order, no atomics: bitwise reproducible
        k2 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert np.array_equal(k.values, k2.values)
    # accumulate on top of existing values (assemble_into_csr does not zero, global.rs:133-182)
    fa.CsrAssembler(fa.SCATTER_GATHER).assemble_into_csr(k, asm)
    assert np.abs(k.values - 2.0 * vals).max() <= 2 * TOL * np.abs(vals).max()
    # the switch: tolerance 0 = general kernels only, same result to the tolerance
    engine.set_affine_tolerance(0.0)
    k0 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert "affine" not in engine.last_kernel_name()
    assert np.abs(k0.values - vals).max() <= TOL * np.abs(vals).max()
    engine.set_affine_tolerance(2.0 ** -46)
    k1 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert engine.last_kernel_name() == kern
    assert np.abs(k1.values - vals).max() <= TOL * np.abs(vals).max()


@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC"])
def test_affine_mask_row_range_and_moving_mesh(engine, oracle, op):
    mesh = _meshes()["mixed"]
    asm, ref = _assemblers(engine, oracle, mesh, op)
    st, _, ro, ci, vals = oracle.assemble(ref)
    assert st == 0
    n = mesh.num_nodes()
    s_dim = 1 if op == "LAPLACE" else 3
    # element mask (multi-GPU partitions): pattern from all elements, numerics from the active ones
    active = (np.arange(mesh.num_elements()) % 5 != 2)
    engine.set_active_elements(active)
    km = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert engine.last_kernel_name().startswith("k_affine_rows")
    ka = fa.CsrAssembler(fa.SCATTER_ATOMIC).assemble(asm)
    assert np.abs(km.values - ka.values).max() <= TOL * np.abs(ka.values).max()
    engine.set_active_elements(None)
    # row range: only these rows are produced, the others stay untouched
    for lo_n, hi_n in ((n // 4, n // 2), (0, 3), (n - 5, n)):
        engine.set_row_range(lo_n, hi_n)
        kr = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        lo, hi = int(ro[s_dim * lo_n]), int(ro[s_dim * hi_n])
        assert np.abs(kr.values[lo:hi] - vals[lo:hi]).max() <= TOL * np.abs(vals).max()
        assert not kr.values[:lo].any() and not kr.values[hi:].any()
    engine.set_row_range(0, n)
    # moving mesh: new coordinates reclassify the elements (here: everything becomes non-affine, then affine again)
    rng = np.random.default_rng(9)
    v2 = mesh.vertices + 0.004 * rng.standard_normal(mesh.vertices.shape)
    engine._check(engine._lib.fh_update_vertices(engine._h, fa._ffi.fp(np.ascontiguousarray(v2))))
    w, p = quadrature.tensor.hexahedron_gauss(2)
    ref2 = oracle.ElementAssembler(oracle.HEX8, ref.op_kind, v2, mesh.connectivity, w, p, params=None if op == "LAPLACE" else LAME.as_pair())
    _, _, _, _, vals2 = oracle.assemble(ref2)
    k2 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert "affine" not in engine.last_kernel_name()
    assert np.abs(k2.values - vals2).max() <= TOL * np.abs(vals2).max()
    box = fa.procedural.create_unit_box_uniform_hex_mesh_3d(10)
    engine._check(engine._lib.fh_update_vertices(engine._h, fa._ffi.fp(np.ascontiguousarray(box.vertices))))
    ref3 = oracle.ElementAssembler(oracle.HEX8, ref.op_kind, box.vertices, box.connectivity, w, p, params=None if op == "LAPLACE" else LAME.as_pair())
    _, _, _, _, vals3 = oracle.assemble(ref3)
    k3 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert engine.last_kernel_name() == "k_affine_rows"
    assert np.abs(k3.values - vals3).max() <= TOL * np.abs(vals3).max()


def test_affine_other_rules_and_negative_weights(engine, oracle):
    """The reference blocks are built from whatever rule the table holds: a 27-point rule, and a rule with a negative weight
    (which the sqrt-scaled general kernels cannot take: fast_ok = false there)."""
    mesh = _meshes()["sheared"]
    for rule in (1, 3):
        asm, ref = _assemblers(engine, oracle, mesh, "LINEAR_ELASTIC", rule=rule)
        _, _, ro, ci, vals = oracle.assemble(ref)
        k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert engine.last_kernel_name() == "k_affine_rows"
        assert np.abs(k.values - vals).max() <= TOL * np.abs(vals).max()
    # a (made-up) 9-point rule with one negative weight: 8 Gauss points scaled by 9/8 and the centre with weight -1
    w, p = quadrature.tensor.hexahedron_gauss(2)
    w9 = np.concatenate([np.asarray(w) * 9.0 / 8.0, [-1.0]])
    p9 = np.concatenate([np.asarray(p), np.zeros((1, 3))])
    qt = fa.UniformQuadratureTable.from_points_and_weights(p9, w9).with_uniform_data(LAME)
    for name in ("sheared", "perturbed"):
        mesh = _meshes()[name]
        asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh)
               .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(None).build())
        ref = oracle.ElementAssembler(oracle.HEX8, oracle.LINEAR_ELASTIC, mesh.vertices, mesh.connectivity, w9, p9, params=LAME.as_pair())
        st, _, ro, ci, vals = oracle.assemble(ref)
        assert st == 0
        k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        # with a negative weight the pre-scaled-gradient kernels are off: the generic one-pass gather takes everything
        assert engine.last_kernel_name() == "k_assemble_matrix<gather>"
        assert np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
        assert np.abs(k.values - vals).max() <= TOL * np.abs(vals).max()


def test_affine_singular_element_reported(engine):
    """det J == 0 on an (affine) element: FH_SINGULAR_JACOBIAN with the lowest failing element, like elliptic.rs:401-404"""
    mesh = fa.procedural.create_unit_box_uniform_hex_mesh_3d(4)
    # x' = x + z, y' = y, z' = 0: flattened, every element is degenerate (det J == 0 exactly) and affine
    flat = _sheared(mesh, [[1.0, 0.0, 1.0], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0]])
    w, p = quadrature.tensor.hexahedron_gauss(2)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(LAME)
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(flat)
           .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(None).build())
    with pytest.raises(fa.SingularJacobianError) as ei:
        fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert ei.value.element == 0
    assert engine.last_kernel_name() == "k_affine_rows"


@pytest.mark.gpu
def test_placement_tuning_keeps_the_matrix():
    """fh_time_assembly_dev / fh_tune_placement_dev (round 3): re-allocating the element records and timing real assemblies must leave
    K exactly as it was (the kernel is reproducible bit for bit), and the tuning refuses to run without FH_ASSEMBLE_OVERWRITE."""
    import torch

    mesh = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 10)
    w, p = quadrature.tensor.hexahedron_gauss(2)
    lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
    eng = fa.Engine(0)
    try:
        (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial()))
         .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)).with_u(None).build())
        nnz = eng.build_pattern()
        flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
        ref = torch.zeros(nnz, dtype=torch.float64, device="cuda")
        eng.assemble_matrix(ref, flags)
        assert eng.last_kernel_name() == "k_affine_rows"
        vals = torch.zeros_like(ref)
        ms = eng.time_assembly(vals, flags, reps=2)
        assert ms > 0.0 and torch.equal(vals, ref)
        before, after = eng.tune_placement(vals, flags, tries=3)
        assert 0.0 < after <= before
        vals.zero_()
        eng.assemble_matrix(vals, flags)
        assert torch.equal(vals, ref)
        with pytest.raises(fa.FenrisError):
            eng.tune_placement(vals, fa.SCATTER_GATHER, tries=1)
        # a launch variant switched inside the context (fh_set_option): one request ahead instead of two gives the same bits
        eng.set_option("FENRIS_HIP_AFFINE_DEPTH", 1)
        vals.zero_()
        eng.assemble_matrix(vals, flags)
        assert torch.equal(vals, ref)
        eng.set_option("FENRIS_HIP_AFFINE_DEPTH", None)
        with pytest.raises(fa.FenrisError):   # fh_time_assembly_dev runs real assemblies into `vals` as well
            eng.time_assembly(vals, fa.SCATTER_GATHER, reps=1)
    finally:
        eng.close()


@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC"])
@pytest.mark.parametrize("grid", ["1", "2", "5"])
def test_affine_masks_and_holes_with_many_positions_per_workgroup(engine, oracle, op, grid):
    """Round 3, found by scripts/fuzz_gather.py: the small meshes of the other tests give every workgroup ONE position, the benchmark sizes give
    it thousands.  With FENRIS_HIP_AFFINE_GRID a small mesh walks the same paths: (a) under an element mask a block without an active element
    inherited the values a COMPLETE position had left in the staging buffer two steps earlier; (b) a node without elements (a hole) at the head of
    a run of consecutive rows handed the zeros below its own start on to the next position, which wrote them over the end of the previous rows."""
    import torch

    base = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 7, 6, 5, 1)
    rng = np.random.default_rng(42)
    keep = rng.random(base.num_elements()) >= 0.4                      # holes: some nodes lose every element
    mesh = fa.Mesh(base.vertices, np.asarray(base.connectivity)[keep], base.elem_kind)
    asm, ref = _assemblers(engine, oracle, mesh, op)
    nnz = engine.build_pattern()
    engine.set_option("FENRIS_HIP_AFFINE_GRID", grid)
    try:
        for masked in (False, True):
            engine.set_active_elements((rng.random(mesh.num_elements()) < 0.7) if masked else None)
            want = torch.zeros(nnz, dtype=torch.float64, device="cuda")
            engine.assemble_matrix(want, fa.SCATTER_ATOMIC)
            for fill in (0.0, -2.5):
                got = torch.full((nnz,), fill, dtype=torch.float64, device="cuda")
                engine.assemble_matrix(got, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
                assert engine.last_kernel_name().startswith("k_affine_rows")
                w, g = want.cpu().numpy(), got.cpu().numpy()
                assert np.abs(g - w).max() <= TOL * np.abs(w).max(), (masked, fill)
        st, _, ro, ci, vals = oracle.assemble(ref)      # and the unmasked matrix is the oracle's
        engine.set_active_elements(None)
        k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert st == 0 and np.abs(k.values - vals).max() <= TOL * np.abs(vals).max()
    finally:
        engine.set_active_elements(None)
        engine.set_option("FENRIS_HIP_AFFINE_GRID", None)


@pytest.mark.parametrize("shape", ["box", "sheared", "half distorted"])
def test_scalar_mass_matrix_on_the_affine_kernel(oracle, shape):
    """ElementMassAssembler (mass.rs:131-286) with s = 1 on parallelepiped hexahedra rides k_affine_rows<Laplace>: records (|det J|, 0 ...),
    reference blocks (sum_q w rho_q phi_a phi_b, 0 ...) -- densities that differ from point to point, an element mask, and a mesh whose
    distorted half stays on the generic kernel"""
    rng = np.random.default_rng(12)
    mesh = fa.procedural.create_rectangular_uniform_hex_mesh(0.5, 7, 6, 5, 1)
    v = mesh.vertices.copy()
    if shape == "sheared":
        v = v @ np.array([[1.0, 0.2, 0.0], [0.1, 0.9, 0.3], [0.0, -0.2, 1.1]])
    if shape == "half distorted":
        far = v[:, 0] > 2.6          # (the node blocks of the lines' first halves keep affine elements only)
        v[far] += 0.03 * rng.uniform(-1, 1, (int(far.sum()), 3))
    mesh = fa.Mesh(v, mesh.connectivity, fa.HEX8)
    w, p = quadrature.tensor.hexahedron_gauss(2)
    rho = 1.0 + rng.random(len(w))
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_data([fa.Density(r) for r in rho])
    eng = fa.Engine(0)
    try:
        asm = fa.ElementMassAssembler.with_solution_dim(1, eng).with_space(mesh).with_quadrature_table(qt)
        ref = oracle.ElementAssembler(oracle.HEX8, oracle.MASS_SCALAR, mesh.vertices, mesh.connectivity, w, p,
                                      params=np.stack([rho, np.zeros_like(rho)], axis=1))
        st, _, ro, ci, vals = oracle.assemble(ref)
        assert st == 0
        k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        if shape == "half distorted":   # a mesh with any non-affine node block stays on the generic gather entirely
            assert eng.last_kernel_name() == "k_assemble_matrix<gather>"
        else:
            assert eng.last_kernel_name() == "k_affine_rows", eng.last_kernel_name()
        assert np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
        assert np.abs(k.values - vals).max() <= TOL * np.abs(vals).max()
        a = k.to_scipy()
        if shape != "half distorted":
            assert (a != a.T).nnz == 0                      # exactly symmetric (the affine kernel's term order)
        assert abs(a.sum() - (w * rho).sum() / 8.0 * 0.125 * mesh.num_elements()) <= 1e-9 * a.sum() or shape != "box"   # total mass of the box
        # element mask, values overwritten in an array of garbage
        import torch

        mask = (rng.random(mesh.num_elements()) < 0.7).astype(np.uint8)
        eng.set_active_elements(mask)
        sub = oracle.ElementAssembler(oracle.HEX8, oracle.MASS_SCALAR, mesh.vertices, np.asarray(mesh.connectivity)[mask == 1], w, p,
                                      params=np.stack([rho, np.zeros_like(rho)], axis=1))
        want = np.zeros(len(ci))
        st, _ = oracle.assemble_into_csr(sub, ro, ci, want)
        assert st == 0
        buf = torch.full((len(ci),), -7.5, dtype=torch.float64, device="cuda:0")
        eng.assemble_matrix(buf, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
        assert np.abs(buf.cpu().numpy() - want).max() <= TOL * np.abs(vals).max()
    finally:
        eng.close()


@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC"])
@pytest.mark.parametrize("grid", [None, "2"])
def test_affine_rows_many_positions_per_workgroup(oracle, op, grid):
    """k_affine_records + k_affine_rows with few workgroups (FENRIS_HIP_AFFINE_GRID: many positions each -- the steady state of the loader's
    pipeline) and with the default grid: the oracle's matrix, symmetric bit for bit, reproducible, accumulating, under an element mask, and
    the singular element is reported.  (Round 5 ran these checks on the fused-records form, which is retired to scripts/attic/.)"""
    names = ("box9", "graded", "sheared", "mirrored", "mixed", "slab_17x3x2", "single_element")
    for name in names:
        eng = fa.Engine(0)
        try:
            if grid:
                eng.set_option("FENRIS_HIP_AFFINE_GRID", grid)
            mesh = _meshes()[name]
            asm, ref = _assemblers(eng, oracle, mesh, op)
            st, _, ro, ci, vals = oracle.assemble(ref)
            assert st == 0
            k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
            assert eng.last_kernel_name().startswith("k_affine_rows")
            assert np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
            assert np.abs(k.values - vals).max() <= TOL * np.abs(vals).max(), name
            if EXPECT[name] == "affine":
                assert _symmetric_bitwise(k)
            k2 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
            assert np.array_equal(k.values, k2.values)
            fa.CsrAssembler(fa.SCATTER_GATHER).assemble_into_csr(k, asm)
            assert np.abs(k.values - 2.0 * vals).max() <= 2 * TOL * np.abs(vals).max()
            active = (np.arange(mesh.num_elements()) % 5 != 2)
            if mesh.num_elements() > 5:
                eng.set_active_elements(active)
                km = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
                ka = fa.CsrAssembler(fa.SCATTER_ATOMIC).assemble(asm)
                assert np.abs(km.values - ka.values).max() <= TOL * np.abs(ka.values).max(), name
        finally:
            eng.close()
    # singular element (det J == 0 exactly): reported with the lowest element, like elliptic.rs:401-404
    eng = fa.Engine(0)
    try:
        if grid:
            eng.set_option("FENRIS_HIP_AFFINE_GRID", grid)
        flat = _sheared(fa.procedural.create_unit_box_uniform_hex_mesh_3d(4), [[1.0, 0.0, 1.0], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0]])
        asm, _ = _assemblers(eng, oracle, flat, op)
        with pytest.raises(fa.SingularJacobianError) as ei:
            fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
        assert ei.value.element == 0
    finally:
        eng.close()


@pytest.mark.parametrize("op", ["LAPLACE", "LINEAR_ELASTIC"])
def test_node_range_cut_on_the_device_equals_the_host_cut(oracle, op):
    """Round 5: numberings made of grid lines are cut into owner blocks on the device (k_cut_runs); FENRIS_HIP_HOST_CUT=1 keeps the host loop.
    Same blocks -> the same kernel, the same sums in the same order: identical matrices bit for bit, on affine, mixed and general meshes, under
    a row range (slabs) and with runs of many lengths (17 x 3 x 2: lines of 18, 4 and 3 nodes -- the short ones send the cut to the host)."""
    for name in ("box9", "graded", "mixed", "perturbed", "slab_17x3x2", "single_element"):
        mesh = _meshes()[name]
        got = {}
        for host in (0, 1):
            eng = fa.Engine(0)
            try:
                eng.set_option("FENRIS_HIP_HOST_CUT", host)
                asm, ref = _assemblers(eng, oracle, mesh, op)
                k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
                got[host] = (k.values.copy(), eng.last_kernel_name())
                if host == 0:
                    st, _, ro, ci, vals = oracle.assemble(ref)
                    assert st == 0
                    assert np.abs(k.values - vals).max() <= TOL * np.abs(vals).max(), name
            finally:
                eng.close()
        assert got[0][1] == got[1][1], name
        assert np.array_equal(got[0][0], got[1][0]), name
