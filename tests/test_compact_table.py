"""CompactQuadratureTable (src/assembly/local/quadrature_table.rs:300-439) with shared points / weights: per-element
material data.  Oracle properties on the CPU, HIP parity on the GPU."""
import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature

RULES = np.array([[416666.67, 277777.78], [8.0e4, 1.2e5], [3.0e5, 2.0e5]])  # (mu, lambda) per rule


def _setup(kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "HEX8":
        m = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 3)
        w, p = quadrature.tensor.hexahedron_gauss(2)
    elif kind == "TET4":
        m = fa.procedural.create_unit_box_uniform_tet_mesh_3d(2)
        w, p = quadrature.total_order.tetrahedron(2)
    else:
        m = fa.procedural.create_unit_square_uniform_quad_mesh_2d(4)
        w, p = quadrature.tensor.quadrilateral_gauss(2)
    m = fa.Mesh(m.vertices + rng.uniform(-0.02, 0.02, m.vertices.shape), m.connectivity, m.elem_kind)
    emap = rng.integers(0, len(RULES), m.num_elements()).astype(np.uint64)
    # per-point variation inside a rule as well
    rp = RULES[:, None, :] * (1.0 + 0.05 * np.arange(len(w))[None, :, None])
    return m, w, p, emap, np.ascontiguousarray(rp)


# ------------------------------------------------------------------------------------------- CPU: oracle properties
@pytest.mark.parametrize("op_name", ["LINEAR_ELASTIC", "NEO_HOOKEAN"])
def test_oracle_compact_table_is_sum_over_rules(oracle, op_name):
    """K(compact) == sum_r K(uniform table with rule r's data, elements of rule r only)"""
    m, w, p, emap, rp = _setup("HEX8")
    op = getattr(oracle, op_name)
    u = 0.01 * np.random.default_rng(3).standard_normal(3 * m.num_nodes())
    full = oracle.ElementAssembler(oracle.HEX8, op, m.vertices, m.connectivity, w, p, params=rp[0], u=u,
                                   elem_to_rule=emap, rule_params=rp)
    st, _, ro, ci, vals = oracle.assemble(full)
    assert st == 0
    acc = np.zeros_like(vals)
    for r in range(len(rp)):
        sub = oracle.ElementAssembler(oracle.HEX8, op, m.vertices, m.connectivity[emap == r], w, p, params=rp[r], u=u)
        st, _ = oracle.assemble_into_csr(sub, ro, ci, acc)
        assert st == 0
    assert np.abs(acc - vals).max() <= 1e-12 * np.abs(vals).max()


def test_oracle_compact_with_one_rule_equals_uniform(oracle):
    m, w, p, emap, rp = _setup("TET4")
    a = oracle.ElementAssembler(oracle.TET4, oracle.LINEAR_ELASTIC, m.vertices, m.connectivity, w, p, params=rp[1])
    b = oracle.ElementAssembler(oracle.TET4, oracle.LINEAR_ELASTIC, m.vertices, m.connectivity, w, p, params=rp[0],
                                elem_to_rule=np.ones(m.num_elements(), dtype=np.uint64), rule_params=rp)
    va, vb = oracle.assemble(a)[4], oracle.assemble(b)[4]
    assert np.array_equal(va, vb)


def test_compact_table_rejects_bad_rule_index():
    m, w, p, emap, rp = _setup("HEX8")
    emap[0] = 7
    with pytest.raises(ValueError):
        fa.CompactQuadratureTable(p, w, [[fa.LameParameters(*x) for x in r] for r in rp], emap)


# ------------------------------------------------------------------------------------------- GPU parity
@pytest.fixture(scope="module")
def engine():
    eng = fa.Engine(0)
    yield eng
    eng.close()


def _table(p, w, rp, emap, cls=fa.LameParameters):
    return fa.CompactQuadratureTable.from_quadrature_rules_and_map(p, w, [[cls(*x) if cls is fa.LameParameters else cls(x[0]) for x in r]
                                                                          for r in rp], emap)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,op_name", [("HEX8", "LINEAR_ELASTIC"), ("HEX8", "NEO_HOOKEAN"), ("TET4", "STVK"),
                                          ("QUAD4", "LINEAR_ELASTIC")])
@pytest.mark.parametrize("scatter", ["gather", "atomic", "colored"])
def test_compact_table_matrix_vector_scalar_match_oracle(engine, oracle, kind, op_name, scatter):
    m, w, p, emap, rp = _setup(kind, seed=2)
    d = m.vertices.shape[1]
    u = 0.01 * np.random.default_rng(5).standard_normal(d * m.num_nodes())
    mat = {"LINEAR_ELASTIC": fa.LinearElasticMaterial, "NEO_HOOKEAN": fa.NeoHookeanMaterial, "STVK": fa.StVKMaterial}[op_name]()
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(m).with_operator(fa.MaterialEllipticOperator(mat))
           .with_quadrature_table(_table(p, w, rp, emap)).with_u(u).build())
    flags = {"gather": fa.SCATTER_GATHER, "atomic": fa.SCATTER_ATOMIC, "colored": fa.SCATTER_COLORED}[scatter]
    if scatter == "colored":
        k = fa.CsrParAssembler().assemble(fa.color_nodes(asm), asm)
    else:
        k = fa.CsrAssembler(flags).assemble(asm)
    oasm = oracle.ElementAssembler(getattr(oracle, kind), getattr(oracle, op_name), m.vertices, m.connectivity, w, p, params=rp[0], u=u,
                                   elem_to_rule=emap, rule_params=rp)
    st, _, ro, ci, vals = oracle.assemble(oasm)
    assert st == 0
    assert np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
    assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()
    if scatter == "gather":
        f = fa.VectorAssembler().assemble_vector(asm)
        st, _, fo = oracle.assemble_vector(oasm)
        assert st == 0 and np.abs(f - fo).max() <= 1e-12 * np.abs(fo).max()
        e = fa.assemble_scalar(asm)
        st, _, eo = oracle.assemble_scalar(oasm)
        assert st == 0 and abs(e - eo) <= 1e-12 * abs(eo)


@pytest.mark.gpu
def test_compact_table_mass_and_gravity_match_oracle(engine, oracle):
    m, w, p, emap, rp = _setup("HEX8", seed=4)
    rho = np.ascontiguousarray(np.stack([1000.0 + 100.0 * rp[:, :, 0] / rp[0, 0, 0], np.zeros(rp.shape[:2])], axis=-1))
    qt = _table(p, w, rho, emap, cls=fa.Density)
    mass = fa.ElementMassAssembler.with_solution_dim(3, engine).with_space(m).with_quadrature_table(qt)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(mass)
    oasm = oracle.ElementAssembler(oracle.HEX8, oracle.MASS_VECTOR, m.vertices, m.connectivity, w, p, params=rho[0],
                                   elem_to_rule=emap, rule_params=rho)
    st, _, ro, ci, vals = oracle.assemble(oasm)
    assert st == 0 and np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()
    src = (fa.ElementSourceAssemblerBuilder.new(engine).with_finite_element_space(m)
           .with_source(fa.GravitySource([0.0, 0.0, -9.81])).with_quadrature_table(qt).build())
    f = fa.VectorAssembler().assemble_vector(src)
    st, fo = oracle.assemble_source_vector(oasm, 3, g=[0.0, 0.0, -9.81])
    assert st == 0 and np.abs(f - fo).max() <= 1e-12 * np.abs(fo).max()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["HEX8", "TET4", "QUAD4"])
def test_piecewise_constant_material_takes_the_pipelined_kernel(engine, oracle, kind):
    """Rules that are constant over their points (one LameParameters pair per element -- the multi-material case) keep the
    owner-computes pipelined kernel, which then reads (mu, lambda) per element; values as the oracle's."""
    m, w, p, emap, rp = _setup(kind, seed=9)
    rp = np.ascontiguousarray(np.broadcast_to(RULES[:, None, :], rp.shape))
    op = fa.MaterialEllipticOperator(fa.LinearElasticMaterial())
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(m).with_operator(op)
           .with_quadrature_table(_table(p, w, rp, emap)).with_u(None).build())
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    fast_kernel = "k_gather_rows" if kind == "TET4" else "k_gather_pipelined"  # Tet4: row-owner form of the kernel
    assert engine.last_kernel_name() == fast_kernel
    oasm = oracle.ElementAssembler(getattr(oracle, kind), oracle.LINEAR_ELASTIC, m.vertices, m.connectivity, w, p, params=rp[0],
                                   elem_to_rule=emap, rule_params=rp)
    st, _, ro, ci, vals = oracle.assemble(oasm)
    assert st == 0 and np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
    assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()
    # the element-centric kernels and the residual see the same per-element data
    ka = fa.CsrAssembler(fa.SCATTER_ATOMIC).assemble(asm)
    assert np.abs(ka.values - vals).max() <= 1e-12 * np.abs(vals).max()
    # accumulate mode, twice: 2 K on top of the first result
    fa.CsrAssembler(fa.SCATTER_GATHER).assemble_into_csr(k, asm)
    assert np.abs(k.values - 2.0 * vals).max() <= 2e-12 * np.abs(vals).max()
    # a different map on the same pattern: the per-slot parameters are rebuilt
    emap2 = ((emap + 1) % len(RULES)).astype(np.uint64)
    asm2 = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(m).with_operator(op)
            .with_quadrature_table(_table(p, w, rp, emap2)).with_u(None).build())
    k3 = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm2)
    assert engine.last_kernel_name() == fast_kernel
    oasm2 = oracle.ElementAssembler(getattr(oracle, kind), oracle.LINEAR_ELASTIC, m.vertices, m.connectivity, w, p, params=rp[0],
                                    elem_to_rule=emap2, rule_params=rp)
    vals2 = oracle.assemble(oasm2)[4]
    assert np.abs(k3.values - vals2).max() <= 1e-12 * np.abs(vals2).max()


@pytest.mark.gpu
def test_rules_varying_over_points_take_the_generic_kernel(engine, oracle):
    m, w, p, emap, rp = _setup("HEX8", seed=10)
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(m)
           .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial()))
           .with_quadrature_table(_table(p, w, rp, emap)).with_u(None).build())
    fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert engine.last_kernel_name() == "k_assemble_matrix<gather>"


@pytest.mark.gpu
def test_uniform_table_after_compact_restores_fast_path(engine, oracle):
    m, w, p, emap, rp = _setup("HEX8", seed=6)
    lame = fa.LameParameters(*RULES[0])
    op = fa.MaterialEllipticOperator(fa.LinearElasticMaterial())
    (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(m).with_operator(op)
     .with_quadrature_table(_table(p, w, rp, emap)).with_u(None).build())
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(m).with_operator(op)
           .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)).with_u(None).build())
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    assert engine.last_kernel_name() == "k_hex8_rows"   # (Hex8, eight-point rule, uniform parameters: the row-owner kernel, hex8_rows.hip)
    oasm = oracle.ElementAssembler(oracle.HEX8, oracle.LINEAR_ELASTIC, m.vertices, m.connectivity, w, p, params=RULES[0])
    vals = oracle.assemble(oasm)[4]
    assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()


# ------------------------------------------------------------------------------------------- rules with different points
def _mixed_rules(kind="HEX8"):
    """two rules with different point sets (Gauss 2 and Gauss 3) and different data"""
    w2, p2 = quadrature.tensor.hexahedron_gauss(2)
    w3, p3 = quadrature.tensor.hexahedron_gauss(3)
    d2 = [fa.LameParameters(*RULES[0])] * len(w2)
    d3 = [fa.LameParameters(RULES[1][0] * (1 + 0.01 * q), RULES[1][1]) for q in range(len(w3))]
    return [(w2, p2, d2), (w3, p3, d3)]


def _oracle_sum_over_rules(oracle, mesh, op, rules, emap, u):
    w, p, _ = rules[0]
    full = oracle.ElementAssembler(oracle.HEX8, op, mesh.vertices, mesh.connectivity, w, p, params=[1.0, 1.0])
    ro, ci = oracle.pattern_for(full)
    vals, f, e = np.zeros(len(ci)), np.zeros(3 * mesh.num_nodes()), 0.0
    for r, (w, p, d) in enumerate(rules):
        params = np.array([x.as_pair() for x in d])
        sub = oracle.ElementAssembler(oracle.HEX8, op, mesh.vertices, mesh.connectivity[emap == r], w, p, params=params, u=u)
        st, _ = oracle.assemble_into_csr(sub, ro, ci, vals)
        assert st == 0
        st, _, f = oracle.assemble_vector(sub, out=f)
        assert st == 0
        st, _, er = oracle.assemble_scalar(sub)
        e += er
    return ro, ci, vals, f, e


@pytest.mark.gpu
@pytest.mark.parametrize("table", ["compact", "general"])
def test_rules_with_different_points_are_walked_rule_by_rule(engine, oracle, table):
    """CompactQuadratureTable / GeneralQuadratureTable in full generality (quadrature_table.rs:57-210, 300-439): elements
    with a 2-point and a 3-point Gauss rule in one mesh"""
    m, _, _, _, _ = _setup("HEX8", seed=8)
    rules = _mixed_rules()
    emap = (np.arange(m.num_elements()) % 3 == 0).astype(np.uint64)
    u = 0.01 * np.random.default_rng(9).standard_normal(3 * m.num_nodes())
    if table == "compact":
        qt = fa.compact_quadrature_table([r[1] for r in rules], [r[0] for r in rules], [r[2] for r in rules], emap)
    else:
        qt = fa.GeneralQuadratureTable.from_points_weights_and_data([rules[int(r)][1] for r in emap], [rules[int(r)][0] for r in emap],
                                                                     [rules[int(r)][2] for r in emap])
        assert len(qt.rules) == 2  # identical per-element rules are merged
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(m)
           .with_operator(fa.MaterialEllipticOperator(fa.NeoHookeanMaterial())).with_quadrature_table(qt).with_u(u).build())
    ro, ci, vals, fo, eo = _oracle_sum_over_rules(oracle, m, oracle.NEO_HOOKEAN, rules, emap, u)
    for scatter in (fa.SCATTER_GATHER, fa.SCATTER_ATOMIC):
        k = fa.CsrAssembler(scatter).assemble(asm)
        assert np.array_equal(k.row_offsets, ro) and np.array_equal(k.col_indices, ci)
        assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()
    k = fa.CsrParAssembler().assemble(fa.color_nodes(asm), asm)
    assert np.abs(k.values - vals).max() <= 1e-12 * np.abs(vals).max()
    f = fa.VectorAssembler().assemble_vector(asm)
    assert np.abs(f - fo).max() <= 1e-12 * np.abs(fo).max()
    assert abs(fa.assemble_scalar(asm) - eo) <= 1e-12 * abs(eo)


@pytest.mark.gpu
@pytest.mark.parametrize("collapsed", [0, 1, 5])
def test_async_vector_over_a_rule_set_reports_a_singular_element_of_any_group(engine, collapsed):
    """fh_assemble_vector_async_dev over a rule-set table: one launch per rule group, nothing read back in between -- a singular
    element of ANY group (not only of the last one walked) must come out of the next fh_poll_status.  (Advisor, round 3: the per-group
    reset of the status slot erased what an earlier group had reported.)"""
    import torch

    m, _, _, _, _ = _setup("HEX8", seed=8)
    rules = _mixed_rules()
    emap = (np.arange(m.num_elements()) % 3 == 0).astype(np.uint64)      # elements 0, 3, ... rule 1; the others rule 0
    v = m.vertices.copy()
    conn = np.asarray(m.connectivity).astype(np.int64)
    v[conn[collapsed]] = 0.0                                             # all eight vertices at the origin: J == 0 exactly at every point
    qt = fa.compact_quadrature_table([r[1] for r in rules], [r[0] for r in rules], [r[2] for r in rules], emap)
    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(fa.Mesh(v, m.connectivity, fa.HEX8))
           .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt)
           .with_u(np.zeros(3 * m.num_nodes())).build())
    assert asm.engine is engine
    out = torch.zeros(3 * m.num_nodes(), dtype=torch.float64, device="cuda")
    engine.assemble_vector_async(out)                                    # the enqueue itself succeeds
    with pytest.raises(fa.SingularJacobianError):
        engine.poll_status()
    # the blocking call reports the lowest failing element over all groups (global.rs:154: the first error aborts the serial loop)
    with pytest.raises(fa.SingularJacobianError) as exc:
        engine.assemble_vector(out)
    assert exc.value.element <= collapsed


def test_compact_table_helper_picks_the_single_launch_form():
    m, w, p, emap, rp = _setup("HEX8")
    qt = fa.compact_quadrature_table([p] * len(rp), [w] * len(rp), [[fa.LameParameters(*x) for x in r] for r in rp], emap)
    assert isinstance(qt, fa.CompactQuadratureTable)
    rules = _mixed_rules()
    qt2 = fa.compact_quadrature_table([r[1] for r in rules], [r[0] for r in rules], [r[2] for r in rules], emap % 2)
    assert hasattr(qt2, "rules") and len(qt2.rules) == 2
    with pytest.raises(ValueError):
        fa.compact_quadrature_table([p], [w], [[fa.LameParameters(1, 1)] * len(w)], np.array([0, 1]))


@pytest.mark.gpu
def test_rule_set_table_through_the_c_abi(engine, oracle):
    """fh_set_quadrature_rules called the way a Rust host would (raw arrays, include/fenris_hip.h): a GeneralQuadratureTable
    with one rule PER ELEMENT (quadrature_table.rs:57-210) whose rules use two different point sets and per-element data.
    The library groups the rules by (points, weights) -- two passes, not E -- and walks them for the matrix, the vector, the
    energy and the element matrices; an element mask restricts all of them; entry points that do not walk say so."""
    import ctypes as C

    from fenris_amd import _ffi

    m, _, _, _, _ = _setup("HEX8", seed=11)
    E = m.num_elements()
    w2, p2 = quadrature.tensor.hexahedron_gauss(2)
    w3, p3 = quadrature.tensor.hexahedron_gauss(3)
    rng = np.random.default_rng(12)
    use3 = np.arange(E) % 4 == 1
    lam = [(RULES[0][0] * (1.0 + 0.1 * rng.random()), RULES[0][1] * (1.0 + 0.1 * rng.random())) for _ in range(E)]
    offs, W, P, D, rules = [0], [], [], [], []
    for e in range(E):
        w, p = (w3, p3) if use3[e] else (w2, p2)
        d = np.tile(np.array(lam[e]), (len(w), 1))
        offs.append(offs[-1] + len(w)); W.append(w); P.append(p); D.append(d)
        rules.append((w, p, [fa.LameParameters(*lam[e])] * len(w)))
    offs = np.array(offs, dtype=np.uint64)
    W, P, D = (np.ascontiguousarray(np.concatenate(x), dtype=np.float64) for x in (W, P, D))
    u = 0.01 * rng.standard_normal(3 * m.num_nodes())
    engine.set_mesh(m)
    engine.set_operator(_ffi.NEO_HOOKEAN)
    lib, h = engine._lib, engine._h
    engine._check(lib.fh_set_quadrature_rules(h, E, _ffi.up(offs), _ffi.fp(W), _ffi.fp(P), _ffi.fp(D), None))
    ng = C.c_uint64()
    assert lib.fh_quadrature_rule_groups(h, C.byref(ng)) == 0 and ng.value == 2
    engine.set_u(u)
    nnz = engine.build_pattern()
    ro, ci, vals, fo, eo = _oracle_sum_over_rules(oracle, m, oracle.NEO_HOOKEAN, rules, np.arange(E), u)
    for flags in (fa.SCATTER_GATHER, fa.SCATTER_ATOMIC):
        buf = np.full(nnz, 7.0)
        engine.assemble_matrix(buf, flags | fa.ASSEMBLE_OVERWRITE)
        assert np.abs(buf - vals).max() <= 1e-12 * np.abs(vals).max()
        engine.assemble_matrix(buf, flags)  # accumulate on top
        assert np.abs(buf - 2.0 * vals).max() <= 2e-12 * np.abs(vals).max()
    f = np.zeros(3 * m.num_nodes())
    engine.assemble_vector(f)
    assert np.abs(f - fo).max() <= 1e-12 * np.abs(fo).max()
    assert abs(engine.assemble_scalar() - eo) <= 1e-12 * abs(eo)
    # element matrices: every element with ITS rule
    ke = engine.element_matrices(0, E)
    for e in (0, 1, 2, 5, E - 1):
        w, p, d = rules[e]
        sub = oracle.ElementAssembler(oracle.HEX8, oracle.NEO_HOOKEAN, m.vertices, m.connectivity, w, p,
                                      params=np.array([x.as_pair() for x in d]), u=u)
        st, oke = sub.element_matrix(e)
        assert st == 0
        assert np.abs(ke[e] - oke).max() <= 1e-12 * np.abs(oke).max()
    # element mask: both groups restricted
    mask = (np.arange(E) % 3 != 0).astype(np.uint8)
    engine.set_active_elements(mask)
    sel = mask.astype(bool)
    ro2, ci2, vals2, _, _ = _oracle_sum_over_rules(oracle, m, oracle.NEO_HOOKEAN, rules, np.where(sel, np.arange(E), -1), u)
    buf = np.zeros(nnz)
    engine.assemble_matrix(buf, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
    assert np.abs(buf - vals2).max() <= 1e-12 * np.abs(vals).max()
    engine.set_active_elements(None)
    # entry points that do not walk the groups refuse instead of using one rule for every element
    x = np.zeros((E * len(w2), 3))
    assert lib.fh_physical_quadrature_points(h, _ffi.fp(x)) == _ffi.FH_UNSUPPORTED
    # bad arguments: rule index out of bounds, empty rule
    bad = np.arange(E, dtype=np.uint64)
    bad[3] = E
    assert lib.fh_set_quadrature_rules(h, E, _ffi.up(offs), _ffi.fp(W), _ffi.fp(P), _ffi.fp(D), _ffi.up(bad)) == _ffi.FH_BAD_ARGUMENT
    offs_bad = offs.copy()
    offs_bad[2] = offs_bad[1]
    assert lib.fh_set_quadrature_rules(h, E, _ffi.up(offs_bad), _ffi.fp(W), _ffi.fp(P), _ffi.fp(D), None) == _ffi.FH_BAD_ARGUMENT
    # a uniform table replaces the rule set
    engine.set_quadrature_uniform(w2, p2, np.tile(np.array(RULES[0]), (len(w2), 1)))
    assert lib.fh_quadrature_rule_groups(h, C.byref(ng)) == 0 and ng.value == 0
