"""Gmsh MSH 4.1 loader (src/io/msh.rs) against the reference's own assets (tests/golden/msh/*.msh, copied data files)
and insta snapshots (tests/golden/*.json, see make_fixtures.py).  Host code only."""
import os

import numpy as np
import pytest

import fenris_amd as fa
from fenris_amd import io
from conftest import GOLDEN, load_golden_mesh

MSH = os.path.join(GOLDEN, "msh")


@pytest.mark.parametrize("name,kind", [("sphere_tet4_593", fa.TET4), ("cube_hex8_8", fa.HEX8), ("cube_hex27_8", fa.HEX27),
                                       ("square_quad4_79", fa.QUAD4), ("cube_tet4_24", fa.TET4), ("cube_tet10_24", fa.TET10),
                                       ("rectangle_tri3_110", fa.TRI3), ("square_quad4_4", fa.QUAD4), ("square_quad9_4", fa.QUAD9),
                                       ("square_tri3_4", fa.TRI3), ("square_tri6_4", fa.TRI6)])
def test_load_msh_matches_reference_snapshots(name, kind):
    """tests/unit_tests/io/msh.rs: load_msh_* snapshot tests (vertices and connectivity, bit for bit)"""
    mesh = io.load_msh_from_file(os.path.join(MSH, name + ".msh"), kind)
    v, c = load_golden_mesh(name)
    assert np.array_equal(mesh.vertices, v)
    assert np.array_equal(mesh.connectivity, c)


@pytest.mark.parametrize("name,kind,nv,ne", [("square_quad4_4", fa.QUAD4, 9, 4), ("square_tri3_4", fa.TRI3, 5, 4),
                                             ("cube_tet4_24", fa.TET4, 14, 24), ("rectangle_tri3_110", fa.TRI3, None, 110)])
def test_load_msh_counts(name, kind, nv, ne):
    """element counts are in the asset names (and the module doc example: square_tri3_4 has 5 vertices, 4 elements)"""
    mesh = io.load_msh_from_file(os.path.join(MSH, name + ".msh"), kind)
    assert mesh.num_elements() == ne
    if nv is not None:
        assert mesh.num_nodes() == nv
    assert mesh.connectivity.max() < mesh.num_nodes()
    # positively oriented 2-D elements / positive volumes: the loader does not reorder nodes (msh.rs:262-268)
    if kind == fa.TRI3:
        p = mesh.vertices[mesh.connectivity.astype(int)]
        area = 0.5 * ((p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 2, 0] - p[:, 0, 0]) * (p[:, 1, 1] - p[:, 0, 1]))
        assert np.all(area > 0)


def test_load_msh_errors():
    data = open(os.path.join(MSH, "cube_hex8_8.msh"), "rb").read()
    with pytest.raises(io.MshError, match="does not contain an element block of the requested type"):
        io.load_msh_from_bytes(data, fa.TET4)  # msh.rs:71-81
    with pytest.raises(io.MshError, match="does not contain nodes"):
        io.load_msh_from_bytes(b"$MeshFormat\n4.1 0 8\n$EndMeshFormat\n", fa.HEX8)
    with pytest.raises(io.MshError, match="failed to parse"):
        io.load_msh_from_bytes(b"$MeshFormat\n2.2 0 8\n$EndMeshFormat\n", fa.HEX8)
    # a quad9 file holds no quad4 block
    with pytest.raises(io.MshError):
        io.load_msh_from_file(os.path.join(MSH, "square_quad9_4.msh"), fa.QUAD4)
    # sparse node tags are refused (msh.rs:117-122)
    text = data.decode().replace("\n1\n2\n3\n", "\n1\n2\n4\n", 1)
    with pytest.raises(io.MshError, match="not consecutive"):
        io.load_msh_from_bytes(text.encode(), fa.HEX8)
    with pytest.raises(io.MshError, match="failed to read file"):
        io.load_msh_from_file(os.path.join(MSH, "nope.msh"), fa.HEX8)


def test_loaded_mesh_feeds_the_oracle_pattern(oracle):
    """a loaded unstructured mesh goes through the same pattern path as generated ones"""
    mesh = io.load_msh_from_file(os.path.join(MSH, "sphere_tet4_593.msh"), fa.TET4)
    w, p = oracle.tetrahedron_rule(1)
    asm = oracle.ElementAssembler(oracle.TET4, oracle.LAPLACE, mesh.vertices, mesh.connectivity, w, p)
    ro, ci = oracle.pattern_for(asm)
    assert len(ro) == mesh.num_nodes() + 1 and ro[-1] == len(ci)
