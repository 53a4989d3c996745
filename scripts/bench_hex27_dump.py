#!/usr/bin/env python3
"""C4 experiment: time of the dense element-matrix pass (MODE_DUMP) for Hex27 NeoHookean"""
import ctypes as C, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa
from fenris_amd import quadrature
h8 = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 5, 5, 8, 10)
m = fa.hex27_mesh_from_hex8(h8)
A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
u = (0.05 * m.vertices @ A.T).reshape(-1)
lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
w, p = quadrature.tensor.hexahedron_gauss(3)
eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(m).with_operator(fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()))
       .with_quadrature_table(fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)).with_u(u).build())
E = m.num_elements()
ke = torch.zeros(E * 81 * 81, dtype=torch.float64, device="cuda")
for _ in range(2):
    eng._check(eng._lib.fh_assemble_element_matrices_dev(eng._h, 0, E, C.c_void_p(ke.data_ptr())))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng._check(eng._lib.fh_assemble_element_matrices_dev(eng._h, 0, E, C.c_void_p(ke.data_ptr())))
torch.cuda.synchronize()
print(json.dumps({"dump_ms": (time.perf_counter() - t0) / 3 * 1e3, "elements": E, "bytes": ke.numel() * 8}))
