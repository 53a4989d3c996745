"""Residual and energy (element pass, element_pass.hpp) against the oracle on awkward inputs: long rules, inverted elements (negative
det J), affine and distorted elements mixed inside one wavefront, large displacements (NeoHookean with det F <= 0 somewhere: the NaN
entries must be the oracle's)."""
import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import quadrature

pytestmark = pytest.mark.gpu
rng = np.random.default_rng(3)
lame = fa.LameParameters(3.0e2, 5.0e2)
OPS = {"LAPLACE": lambda: fa.LaplaceOperator(), "LINEAR_ELASTIC": lambda: fa.MaterialEllipticOperator(fa.LinearElasticMaterial()),
       "NEO_HOOKEAN": lambda: fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()), "STVK": lambda: fa.MaterialEllipticOperator(fa.StVKMaterial())}


@pytest.fixture(scope="module")
def engine():
    eng = fa.Engine(0)
    yield eng
    eng.close()


def meshes(oracle):
    h8 = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, 6)
    yield "HEX8 affine", h8, oracle.HEX8, [quadrature.tensor.hexahedron_gauss(n) for n in (1, 2, 3, 5)]
    v = h8.vertices.copy()
    far = v[:, 0] > 0.5
    v[far] += rng.uniform(-0.02, 0.02, (int(far.sum()), 3))
    yield "HEX8 half distorted", fa.Mesh(v, h8.connectivity, h8.elem_kind), oracle.HEX8, [quadrature.tensor.hexahedron_gauss(2)]
    c = np.asarray(h8.connectivity).copy()
    flip = np.arange(len(c)) % 3 == 0
    c[flip] = c[flip][:, [4, 5, 6, 7, 0, 1, 2, 3]]      # bottom and top faces swapped: det J < 0
    yield "HEX8 every third inverted", fa.Mesh(h8.vertices, c, h8.elem_kind), oracle.HEX8, [quadrature.tensor.hexahedron_gauss(2)]
    t4 = fa.procedural.create_unit_box_uniform_tet_mesh_3d(3)
    c = np.asarray(t4.connectivity).copy()
    c[::2] = c[::2][:, [1, 0, 2, 3]]
    yield "TET4 every second inverted", fa.Mesh(t4.vertices + rng.uniform(-0.03, 0.03, t4.vertices.shape), c, t4.elem_kind), oracle.TET4, \
        [quadrature.total_order.tetrahedron(o) for o in (1, 2, 4)]
    q4 = fa.procedural.create_unit_square_uniform_quad_mesh_2d(9)
    yield "QUAD4 distorted", fa.Mesh(q4.vertices + rng.uniform(-0.02, 0.02, q4.vertices.shape), q4.connectivity, q4.elem_kind), oracle.QUAD4, \
        [quadrature.tensor.quadrilateral_gauss(n) for n in (1, 2, 4)]



def test_vector_and_energy_on_awkward_inputs(engine, oracle):
    cases = 0
    for name, mesh, okind, rules in meshes(oracle):
        d = mesh.vertices.shape[1]
        for (w, p) in rules:
            for opname in OPS:
                for scale in (1e-3, 0.4):
                    if opname in ("LAPLACE", "LINEAR_ELASTIC") and scale > 0.1:
                        continue
                    s = 1 if opname == "LAPLACE" else d
                    u = scale * rng.standard_normal(s * mesh.num_nodes())
                    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
                    if opname != "LAPLACE":
                        qt = qt.with_uniform_data(lame)
                    asm = (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh).with_operator(OPS[opname]())
                           .with_quadrature_table(qt).with_u(u).build())
                    ref = oracle.ElementAssembler(okind, getattr(oracle, opname), mesh.vertices, mesh.connectivity, w, p,
                                                  params=(lame.as_pair() if opname != "LAPLACE" else None), u=u)
                    st, _, of = oracle.assemble_vector(ref)
                    st2, _, oe = oracle.assemble_scalar(ref)
                    assert st == 0 and st2 == 0
                    f = fa.VectorAssembler().assemble_vector(asm)
                    e = fa.assemble_scalar(asm)
                    where = (name, len(w), opname, scale, engine.last_kernel_name())
                    assert np.array_equal(np.isnan(f), np.isnan(of)), where
                    ok = ~np.isnan(of)
                    if ok.any():
                        assert np.abs(f[ok] - of[ok]).max() <= 1e-11 * max(np.abs(of[ok]).max(), 1e-300), where
                    if np.isfinite(oe):
                        assert abs(e - oe) <= 1e-11 * max(abs(oe), 1e-300), where
                    else:
                        assert np.isnan(e) == np.isnan(oe), where
                    cases += 1
    assert cases > 60
