"""The assembled-matrix fixtures under tests/golden/csr/: what each case is (mesh, operator, rule, parameters, u).

One table for the generator (make_csr_fixtures.py), the oracle test (CPU) and the HIP test (GPU).  A fixture holds the CSR
`offsets`, `indices` and `values` of K exactly as the oracle produced them when the fixture was made (oracle pinned by the
reference's KATs, tests/test_oracle_kat.py) -- a drift of the oracle's compiler or flags, or of the HIP path, shows up against
the stored numbers, not against a checker rebuilt on the test box.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSR_DIR = os.path.join(HERE, "csr")
YOUNG, POISSON = 1e6, 0.2  # tests/integration_tests/assembly.rs:30-33


def _golden_mesh(name):
    m = json.load(open(os.path.join(HERE, name + ".json")))
    return np.array(m["vertices"], dtype=np.float64), np.array(m["connectivity"], dtype=np.uint64)


def _deform(v, eps=0.05):
    A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
    return (eps * v @ A.T).reshape(-1)


# name -> (element kind, operator, mesh builder(oracle) -> (vertices, connectivity), rule(oracle) -> (w, p), u builder or None)
def cases(o):
    def hex27(cells):
        v8, c8 = o.hex_mesh(1.0, 1, 1, 1, cells)
        return o.hex8_to_hex27(v8, c8)

    def deform_inverted(v):
        # the displacement reflects the corner nodes of element 3's cell through its centre: det F <= 0 at some of its
        # points => all-NaN blocks there, not an error (fenris-solid/src/materials.rs:298-300)
        u = _deform(v).reshape(-1, 3)
        _, c = hex27(2)
        nodes = c[3].astype(int)
        centre = v[nodes[:8]].mean(axis=0)
        for n in nodes:
            u[n] = -2.2 * (v[n] - centre)
        return u.reshape(-1)

    return {
        "quad4_4x4_laplace": ("QUAD4", "LAPLACE", lambda: o.unit_square_quad_mesh(4), lambda: o.quadrilateral_gauss(2), None),
        "hex8_2_laplace": ("HEX8", "LAPLACE", lambda: o.unit_box_hex_mesh(2), lambda: o.hexahedron_gauss(2), None),
        "hex8_3_laplace": ("HEX8", "LAPLACE", lambda: o.unit_box_hex_mesh(3), lambda: o.hexahedron_gauss(2), None),
        "hex8_2_elastic": ("HEX8", "LINEAR_ELASTIC", lambda: o.unit_box_hex_mesh(2), lambda: o.hexahedron_gauss(2), None),
        "hex8_3_elastic": ("HEX8", "LINEAR_ELASTIC", lambda: o.unit_box_hex_mesh(3), lambda: o.hexahedron_gauss(2), None),
        "tet4_bcc1_laplace": ("TET4", "LAPLACE", lambda: o.unit_box_tet_mesh(1), lambda: o.tetrahedron_rule(1), None),
        "tet4_bcc2_elastic": ("TET4", "LINEAR_ELASTIC", lambda: o.unit_box_tet_mesh(2), lambda: o.tetrahedron_rule(1), None),
        "tet4_sphere593_elastic": ("TET4", "LINEAR_ELASTIC", lambda: _golden_mesh("sphere_tet4_593"), lambda: o.tetrahedron_rule(1), None),
        "hex27_2_neohookean": ("HEX27", "NEO_HOOKEAN", lambda: hex27(2), lambda: o.hexahedron_gauss(3), _deform),
        "hex27_2_neohookean_inverted": ("HEX27", "NEO_HOOKEAN", lambda: hex27(2), lambda: o.hexahedron_gauss(3), deform_inverted),
    }


def oracle_assembler(o, name):
    kind, op, mesh_f, rule_f, u_f = cases(o)[name]
    v, c = mesh_f()
    w, p = rule_f()
    params = None if op == "LAPLACE" else o.lame_from_young_poisson(YOUNG, POISSON)
    u = u_f(v) if u_f else None
    asm = o.ElementAssembler(getattr(o, kind), getattr(o, op), v, c, w, p, params=params, u=u)
    return asm, (kind, op, v, c, w, p, params, u)


def load(name):
    z = np.load(os.path.join(CSR_DIR, name + ".npz"))
    return z["offsets"], z["indices"], z["values"]
