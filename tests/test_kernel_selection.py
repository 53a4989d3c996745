"""Which stiffness kernel FH_SCATTER_GATHER runs, as ONE table: the rules of DESIGN.md ("Which stiffness kernel ...")
written as a function of (element / geometry, quadrature rule, operator, material data, element mask), checked against what the library
actually launches for every combination (scripts/kernel_selection_table.py enumerates them: 126 assemblies on small meshes).  A change of the
selection logic in engine_matrix.hip / engine_two_pass.hip has to show up here."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))

TWO_PASS_GENERIC = "k_assemble_matrix<dump> + k_rows_from_dense"
TWO_PASS_TRI = "k_assemble_matrix<dump> + k_rows_from_tri"   # the same first pass writing the lower node-block triangle: 3 x 3 blocks on the 3D elements
TWO_PASS_MFMA = "k_hex27_dense_blocks + k_rows_from_tri"


def expected(mesh, rule, op, data, mask):
    """the selection rules (mask: no influence on the choice -- every owner-computes kernel has its masked instantiation)"""
    if mesh == "hex27":
        # more than 8 nodes: always two passes; the first one on the matrix cores for LinearElastic / NeoHookean with a 27-point rule
        if op in ("LinearElastic", "NeoHookean") and rule == "gauss3":
            return TWO_PASS_MFMA
        return TWO_PASS_GENERIC if op == "Laplace" else TWO_PASS_TRI
    if op in ("NeoHookean", "StVK"):
        # nonlinear materials: recomputing the prologue per owner block costs more than the round trip; triangles where the blocks are 3 x 3
        return TWO_PASS_TRI if mesh in ("hex8 affine", "hex8 general", "tet4") else TWO_PASS_GENERIC
    if data == "per-point":
        if mesh == "tet4" and rule == "strength1":
            return "k_gather_rows"                   # one point: "per point" is "per rule", the row-owner kernel takes it
        return "k_assemble_matrix<gather>"           # parameters that vary over the points: the generic one-pass gather
    if mesh == "hex8 affine":
        return "k_affine_rows"                       # parallelepipeds: no quadrature loop, any rule
    if mesh == "hex8 general":
        return "k_hex8_rows" if rule == "gauss2" else "k_gather_pipelined"   # row-owner lanes for the eight-point rule
    if mesh == "tet4":
        return "k_gather_rows"                       # affine by construction: any rule collapses to one point
    return "k_gather_pipelined"                      # Quad4 / Tri3


@pytest.mark.gpu
def test_kernel_selection_table():
    import kernel_selection_table as kst

    rows = kst.table()
    assert len(rows) == 126
    wrong = [(r, expected(*r[:5])) for r in rows if r[5] != expected(*r[:5])]
    assert not wrong, wrong[:5]
    # every kernel family of the gather mode appears
    assert {r[5] for r in rows} == {"k_affine_rows", "k_hex8_rows", "k_gather_pipelined", "k_gather_rows", "k_assemble_matrix<gather>",
                                     TWO_PASS_GENERIC, TWO_PASS_TRI, TWO_PASS_MFMA}
