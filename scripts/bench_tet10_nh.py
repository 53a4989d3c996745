#!/usr/bin/env python3
"""Tet10 NeoHookean / LinearElastic tangent stiffness (BCC res R refined to quadratic tetrahedra, strength-2 rule of 4 points and strength-3 of 6/8):
which kernel, ms per assembly.  python scripts/bench_tet10_nh.py [res]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa
from fenris_amd import quadrature

res = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mesh = fa.tet10_mesh_from_tet4(fa.procedural.create_unit_box_uniform_tet_mesh_3d(res))
lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
u = torch.from_numpy((0.05 * mesh.vertices @ A.T).reshape(-1)).cuda()
for strength in (2, 4):
    w, p = quadrature.total_order.tetrahedron(strength)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
    for name, mat in (("NeoHookean", fa.NeoHookeanMaterial()), ("LinearElastic", fa.LinearElasticMaterial())):
        asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh)
               .with_operator(fa.MaterialEllipticOperator(mat)).with_quadrature_table(qt).with_u(u).build())
        nnz = eng.build_pattern()
        values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
        flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
        eng.assemble_matrix(values, flags)
        ms = eng.time_assembly(values, flags, 5)
        print(json.dumps({"operator": name, "points": len(w), "elements": mesh.num_elements(), "nodes": mesh.num_nodes(), "nnz_GB": round(nnz * 8 / 1e9, 2),
                          "kernel": eng.last_kernel_name(), "ms": round(ms, 4), "elements_per_s": round(mesh.num_elements() / ms * 1e3)}), flush=True)
