"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol the header
declares, and the host input generators (quadrature, meshes, Hex27 conversion, Lame) reproduce the
oracle / the reference's golden data bit-exactly.  No device calls here."""
import os
import re

import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import _ffi, quadrature
from conftest import ROOT, load_golden_mesh


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "fenris_hip.h")).read()
    declared = set(re.findall(r"\b(fh_[A-Za-z0-9_]+)\s*\(", header))
    declared.discard("fh_ctx")
    lib = _ffi.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in fenris_hip.h but not exported"
    assert declared == set(_ffi.exported_symbols())
    assert lib.fh_abi_version() == 1


def test_engine_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fa.FenrisError):
        fa.Engine()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 16])
def test_gauss_matches_oracle_bitwise(oracle, n):
    w, x = quadrature.univariate.gauss(n)
    ow, ox = oracle.gauss(n)
    assert np.array_equal(w, ow) and np.array_equal(x[:, 0], ox)


def test_tensor_and_simplex_rules_match_oracle_bitwise(oracle):
    for n in (1, 2, 3, 4):
        for mine, ref in ((quadrature.tensor.hexahedron_gauss(n), oracle.hexahedron_gauss(n)),
                          (quadrature.tensor.quadrilateral_gauss(n), oracle.quadrilateral_gauss(n))):
            assert np.array_equal(mine[0], ref[0]) and np.array_equal(mine[1], ref[1])
    for s in (0, 1, 2, 3):
        mine, ref = quadrature.total_order.tetrahedron(s), oracle.tetrahedron_rule(max(s, 1))
        assert np.array_equal(mine[0], ref[0]) and np.array_equal(mine[1], ref[1])
    for s in (1, 2):
        mine, ref = quadrature.total_order.triangle(s), oracle.triangle_rule(s)
        assert np.array_equal(mine[0], ref[0]) and np.array_equal(mine[1], ref[1])
    with pytest.raises(fa.FenrisError):
        quadrature.total_order.tetrahedron(9)


@pytest.mark.parametrize("res", [1, 2])
def test_tet_mesh_matches_reference_snapshot(res):
    gv, gc = load_golden_mesh(f"tet_mesh_res{res}")
    m = fa.procedural.create_rectangular_uniform_tet_mesh(1.0, 1, 1, 1, res)
    assert np.array_equal(m.connectivity, gc) and np.array_equal(m.vertices, gv)


def test_generators_match_oracle_bitwise(oracle):
    m = fa.procedural.create_rectangular_uniform_hex_mesh(2.0, 1, 2, 3, 3)
    v, c = oracle.hex_mesh(2.0, 1, 2, 3, 3)
    assert np.array_equal(m.vertices, v) and np.array_equal(m.connectivity, c)
    m = fa.procedural.create_unit_square_uniform_quad_mesh_2d(7)
    v, c = oracle.unit_square_quad_mesh(7)
    assert np.array_equal(m.vertices, v) and np.array_equal(m.connectivity, c)
    m = fa.procedural.create_rectangular_uniform_tet_mesh(1.5, 2, 1, 3, 2)
    v, c = oracle.tet_mesh(1.5, 2, 1, 3, 2)
    assert np.array_equal(m.vertices, v) and np.array_equal(m.connectivity, c)
    h8 = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 2, 1, 1, 3)
    m27 = fa.hex27_mesh_from_hex8(h8)
    v, c = oracle.hex8_to_hex27(h8.vertices, h8.connectivity)
    assert np.array_equal(m27.vertices, v) and np.array_equal(m27.connectivity, c)
    # degenerate arguments give empty meshes like the reference (procedural.rs:58-60, 236-238, 303-305)
    assert fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 0, 1, 1, 3).num_elements() == 0
    assert fa.procedural.create_rectangular_uniform_tet_mesh(1.0, 1, 1, 0, 3).num_nodes() == 0


def test_lame_matches_reference_kat():
    # fenris-solid/tests/unit_tests/materials.rs:74-85
    lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e3, 0.3))
    assert lame.mu == pytest.approx(384.6153846153846, rel=4e-16)
    assert lame.lambda_ == pytest.approx(576.9230769230769, rel=4e-16)


def test_builder_requires_all_parts():
    with pytest.raises(ValueError):
        fa.ElementEllipticAssemblerBuilder().with_operator(fa.LaplaceOperator()).build()
