import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fenris_amd as fa
from fenris_amd import quadrature
eng = fa.Engine(0)
rng = np.random.default_rng(7130)
kind = rng.choice(["HEX8", "HEX8", "TET4", "TET4", "QUAD4", "TRI3", "HEX27"])
dims = rng.integers(1, 26, 3)
m = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, int(dims[0]), int(dims[1]), int(dims[2]), 1)
nq1 = int(rng.integers(1, 4))
v, c = m.vertices.copy(), np.asarray(m.connectivity).astype(np.int64)
geo = rng.choice(["affine", "distorted", "mixed"])
keep = rng.random(len(c)) >= rng.choice([0.0, 0.1, 0.4])
c = c[keep]
perm_p = rng.random()
print(kind, dims, nq1, geo, len(c), "perm?", perm_p < 0.3)
if perm_p < 0.3:
    perm = rng.permutation(len(v)); inv = np.empty_like(perm); inv[perm] = np.arange(len(v))
    v, c = v[perm], inv[c][rng.permutation(len(c))]
mesh = fa.Mesh(v, c.astype(np.uint64), m.elem_kind)
lame = fa.LameParameters(3.0e2, 5.0e2)
w, p = quadrature.tensor.hexahedron_gauss(2)
qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(lame)
asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt)
       .with_u(np.zeros(3 * mesh.num_nodes())).build())
nnz = eng.build_pattern()
ro, ci = eng.pattern(want_cols=True)
want = torch.zeros(nnz, dtype=torch.float64, device="cuda"); eng.assemble_matrix(want, fa.SCATTER_ATOMIC)
wv = want.cpu().numpy()
rows_of = np.repeat(np.arange(len(ro) - 1), np.diff(ro).astype(np.int64))
opts = [{}, {"FENRIS_HIP_AFFINE_GRID": "600"}, {"FENRIS_HIP_AFFINE_NO_CARRY": "1"}, {"FENRIS_HIP_AFFINE_GRID": "600", "FENRIS_HIP_AFFINE_NO_CARRY": "1"}, {"FENRIS_HIP_AFFINE_GRID": "500", "FENRIS_HIP_AFFINE_NO_CARRY": "1"}, {"FENRIS_HIP_AFFINE_NT": "1"}]
for opt in opts:
    for k, v_ in opt.items():
        eng.set_option(k, v_)
    for rep in range(3):
        got = torch.full((nnz,), 4.5, dtype=torch.float64, device="cuda")
        eng.assemble_matrix(got, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
        gv = got.cpu().numpy()
        bad = np.where(np.abs(gv - wv) > 1e-10 * np.abs(wv).max())[0]
        nodes = np.unique(rows_of[bad] // 3)
        print(opt, "rep", rep, eng.last_kernel_name(), "bad entries", len(bad), "nodes", nodes[:12].tolist(), "rows%3", (rows_of[bad] % 3)[:10].tolist(), "cols", (ci[bad] // 3)[:10].tolist(), "got", gv[bad[:4]])
    for k in opt:
        eng.set_option(k, None)
