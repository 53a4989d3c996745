#!/usr/bin/env python3
"""Tet4 "bouquets": M hub nodes close together, each with T tetrahedra that share nothing but the hub (three private vertices each) -- a block
of nine hubs sees 9 T elements but 9 + 27 T distinct vertices: more than the 256 the row-owner kernel's per-position vertex table holds.
    python scripts/exp_bouquet.py 12 25"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402
from oracle import oracle  # noqa: E402

oracle.lib()
M, T = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
hubs = rng.uniform(0.0, 0.05, (M, 3))
verts, conn = [hubs], []
n = M
for h in range(M):
    for t in range(T):
        ctr = hubs[h] + 2.0 + rng.uniform(0.0, 5.0, 3)
        verts.append(ctr + rng.uniform(-0.3, 0.3, (3, 3)))
        conn.append([h, n, n + 1, n + 2])
        n += 3
mesh = fa.Mesh(np.concatenate(verts), np.asarray(conn, dtype=np.uint64), fa.TET4)
w, p = quadrature.total_order.tetrahedron(1)
eng = fa.Engine(0)
for opname, op in (("LAPLACE", fa.LaplaceOperator()), ("LINEAR_ELASTIC", fa.MaterialEllipticOperator(fa.LinearElasticMaterial()))):
    s = 1 if opname == "LAPLACE" else 3
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    lame = fa.LameParameters(3.0e2, 5.0e2)
    if opname != "LAPLACE":
        qt = qt.with_uniform_data(lame)
    asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(op).with_quadrature_table(qt)
           .with_u(np.zeros(s * mesh.num_nodes())).build())
    ref = oracle.ElementAssembler(oracle.TET4, getattr(oracle, opname), mesh.vertices, mesh.connectivity, w, p,
                                  params=(lame.as_pair() if opname != "LAPLACE" else None), u=np.zeros(s * mesh.num_nodes()))
    st, _, oro, oci, ovals = oracle.assemble(ref)
    k = fa.CsrAssembler(fa.SCATTER_GATHER).assemble(asm)
    kern = eng.last_kernel_name()
    ka = fa.CsrAssembler(fa.SCATTER_ATOMIC).assemble(asm)
    # conditioning of the elements: a needle between a hub and three far, clustered vertices
    v, c = mesh.vertices, np.asarray(mesh.connectivity).astype(np.int64)
    J = np.stack([v[c[:, 1]] - v[c[:, 0]], v[c[:, 2]] - v[c[:, 0]], v[c[:, 3]] - v[c[:, 0]]], axis=2)
    print(M, T, opname, kern, "pattern", bool(np.array_equal(k.col_indices, oci)), "gather vs oracle",
          float(np.abs(k.values - ovals).max() / np.abs(ovals).max()), "atomic vs oracle", float(np.abs(ka.values - ovals).max() / np.abs(ovals).max()),
          "gather vs atomic", float(np.abs(k.values - ka.values).max() / np.abs(ovals).max()), "max cond(J)", float(np.linalg.cond(J).max()), flush=True)
