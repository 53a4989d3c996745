import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fenris_amd as fa
from fenris_amd import quadrature
eng = fa.Engine(0)
m = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 5, 4, 3, 1)
w, p = quadrature.tensor.hexahedron_gauss(2)
n = m.num_nodes()
rng = np.random.default_rng(0)
for opname in ("LAPLACE", "LINEAR_ELASTIC"):
    s = 1 if opname == "LAPLACE" else 3
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
    if opname != "LAPLACE":
        qt = qt.with_uniform_data(fa.LameParameters(3.0e2, 5.0e2))
    op = fa.LaplaceOperator() if opname == "LAPLACE" else fa.MaterialEllipticOperator(fa.LinearElasticMaterial())
    asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(m).with_operator(op).with_quadrature_table(qt).with_u(np.zeros(s * n)).build())
    nnz = eng.build_pattern()
    ro, ci = eng.pattern(want_cols=True)
    rows_of = np.repeat(np.arange(len(ro) - 1), np.diff(ro).astype(np.int64))
    for trial in range(3):
        active = rng.random(m.num_elements()) < 0.7
        eng.set_active_elements(active)
        want = torch.zeros(nnz, dtype=torch.float64, device="cuda"); eng.assemble_matrix(want, fa.SCATTER_ATOMIC)
        wv = want.cpu().numpy()
        for grid in (None, "1", "2", "3"):
            eng.set_option("FENRIS_HIP_AFFINE_GRID", grid)
            for fill in (0.0, 4.5):
                got = torch.full((nnz,), fill, dtype=torch.float64, device="cuda")
                eng.assemble_matrix(got, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
                gv = got.cpu().numpy()
                bad = np.where(np.abs(gv - wv) > 1e-10 * np.abs(wv).max())[0]
                print(opname, "trial", trial, "grid", grid, "fill", fill, eng.last_kernel_name(), "bad", len(bad), "nodes", np.unique(rows_of[bad] // s)[:8].tolist(),
                      "got", np.round(gv[bad[:5]], 3), "want", np.round(wv[bad[:5]], 3))
        eng.set_option("FENRIS_HIP_AFFINE_GRID", None)
        eng.set_active_elements(None)
