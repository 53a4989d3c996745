#!/usr/bin/env python3
"""Extract golden DATA fixtures from the reference checkout into tests/golden/*.json.

Run in the build container only (needs /root/reference):  python tests/golden/make_fixtures.py

What is extracted is data, not source:
  * insta debug snapshots of meshes (vertex coordinates + connectivity) that the reference's own
    tests hold -- tests/unit_tests/mesh/snapshots/*mesh_{1,2}.snap (BCC tet generator, res 1 and 2) and
    tests/unit_tests/io/snapshots/*{sphere_tet4_large,cube_hex8,cube_hex27,square_quad4d2_large}.snap
  * the MMS convergence reference values tests/convergence_tests/reference_values/*.json
  * the strength-6 tetrahedron rule table (error quadrature of the Tet4 MMS test)
The known-answer numbers of the reference's unit tests (pattern offsets/indices, Lame parameters,
material energies, element matrices) are small enough to be written directly in tests/test_oracle_kat.py,
each with its file:line citation.
"""
import json
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

SNAPS = {
    "tet_mesh_res1": "tests/unit_tests/mesh/snapshots/unit__unit_tests__mesh__procedural__mesh_1.snap",
    "tet_mesh_res2": "tests/unit_tests/mesh/snapshots/unit__unit_tests__mesh__procedural__mesh_2.snap",
    "sphere_tet4_593": "tests/unit_tests/io/snapshots/unit__unit_tests__io__msh__load_msh_sphere_tet4_large.snap",
    "cube_hex8_8": "tests/unit_tests/io/snapshots/unit__unit_tests__io__msh__load_msh_cube_hex8.snap",
    "cube_hex27_8": "tests/unit_tests/io/snapshots/unit__unit_tests__io__msh__load_msh_cube_hex27.snap",
    "square_quad4_79": "tests/unit_tests/io/snapshots/unit__unit_tests__io__msh__load_msh_square_quad4d2_large.snap",
    # msh loader snapshots of the remaining assets (tests/test_msh.py)
    "cube_tet4_24": "tests/unit_tests/io/snapshots/unit__unit_tests__io__msh__load_msh_cube_tet4.snap",
    "cube_tet10_24": "tests/unit_tests/io/snapshots/unit__unit_tests__io__msh__load_msh_cube_tet10.snap",
    "rectangle_tri3_110": "tests/unit_tests/io/snapshots/unit__unit_tests__io__msh__load_msh_rect_tri3d2_large.snap",
    "square_quad4_4": "tests/unit_tests/io/snapshots/unit__unit_tests__io__msh__load_msh_square_quad4d2.snap",
    "square_quad9_4": "tests/unit_tests/io/snapshots/unit__unit_tests__io__msh__load_msh_square_quad9d2.snap",
    "square_tri3_4": "tests/unit_tests/io/snapshots/unit__unit_tests__io__msh__load_msh_square_tri3d2.snap",
    "square_tri6_4": "tests/unit_tests/io/snapshots/unit__unit_tests__io__msh__load_msh_square_tri6d2.snap",
}

NUM = re.compile(r"-?(?:\d+\.\d*(?:e-?\d+)?|\d+(?:e-?\d+)?)")


def parse_snap(path):
    text = open(path).read()
    body = text.split("---", 2)[2]
    vpart, cpart = body.split("connectivity:", 1)
    # vertices: nested lists of floats; dimension = numbers per innermost list
    vert_lists = re.findall(r"\[\s*((?:-?[\d.e-]+,\s*)+)\]", vpart)
    vertices = [[float(x) for x in NUM.findall(v)] for v in vert_lists]
    conn_lists = re.findall(r"Connectivity\(\s*\[\s*((?:\d+,\s*)+)\]", cpart)
    connectivity = [[int(x) for x in re.findall(r"\d+", c)] for c in conn_lists]
    kind = re.search(r"(\w+Connectivity)\(", cpart).group(1)
    return {"connectivity_type": kind, "vertices": vertices, "connectivity": connectivity}


def main():
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present; fixtures are committed, nothing to do")
    for name, rel in SNAPS.items():
        mesh = parse_snap(os.path.join(REF, rel))
        mesh["source"] = rel
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(mesh, f, separators=(",", ":"))
        print(name, len(mesh["vertices"]), "vertices", len(mesh["connectivity"]), "cells", mesh["connectivity_type"])
    mms = {}
    d = os.path.join(REF, "tests/convergence_tests/reference_values")
    for fn in sorted(os.listdir(d)):
        if fn.endswith(".json"):
            mms[fn[:-5]] = json.load(open(os.path.join(d, fn)))
    with open(os.path.join(OUT, "mms_reference_values.json"), "w") as f:
        json.dump({"source": "tests/convergence_tests/reference_values/*.json", "summaries": mms}, f, indent=1)
    print("mms summaries:", list(mms))
    # quadrature table used only as the ERROR quadrature of the Tet4 MMS test (poisson_3d_mms.rs:123)
    rows = [[float(x) for x in line.split()]
            for line in open(os.path.join(REF, "fenris-quadrature/rules/polyquad/expanded/tet/6-24.txt")) if line.strip()]
    with open(os.path.join(OUT, "tet_rule_6_24.json"), "w") as f:
        json.dump({"source": "fenris-quadrature/rules/polyquad/expanded/tet/6-24.txt",
                   "points": [r[:3] for r in rows], "weights": [r[3] for r in rows]}, f)
    # ... and of the Tri3 / Tri6 MMS tests (poisson_2d_mms.rs:103, 112)
    rows = [[float(x) for x in line.split()]
            for line in open(os.path.join(REF, "fenris-quadrature/rules/polyquad/expanded/tri/6-12.txt")) if line.strip()]
    with open(os.path.join(OUT, "tri_rule_6_12.json"), "w") as f:
        json.dump({"source": "fenris-quadrature/rules/polyquad/expanded/tri/6-12.txt",
                   "points": [r[:2] for r in rows], "weights": [r[2] for r in rows]}, f)


if __name__ == "__main__":
    main()
