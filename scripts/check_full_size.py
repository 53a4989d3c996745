#!/usr/bin/env python3
"""Full-size cross-check of the owner-computes kernels bench.py times: every BASELINE configuration assembled once with
FH_SCATTER_GATHER (affine rows / pipelined / row-owner / two-pass matrix-core kernels) and once with FH_SCATTER_ATOMIC (the
generic element kernel + fp64 atomics), on the device, compared entry by entry; plus the properties the full-size tests use
(symmetry, zero row sums of a stiffness matrix).  The small-mesh parity tests compare with the oracle; this one guards
against anything that only shows at scale (offsets beyond 2^31 bytes, table strides, the locality order of 860 k nodes).
    python scripts/check_full_size.py [ns ns-perturbed c2 c3 c4 ns-slab c5-slab]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

LAME = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))


def problem(cfg):
    if cfg in ("ns", "ns-perturbed", "c2"):
        cells = 128 if cfg == "c2" else 216
        mesh = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, cells)
        if cfg == "ns-perturbed":
            rng = np.random.Generator(np.random.MT19937(2024))
            mesh = fa.Mesh(mesh.vertices + (0.1 / cells) * rng.uniform(-1.0, 1.0, mesh.vertices.shape), mesh.connectivity, fa.HEX8)
        w, p = quadrature.tensor.hexahedron_gauss(2)
        qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
        if cfg == "c2":
            return mesh, fa.LaplaceOperator(), qt, None, 1
        return mesh, fa.MaterialEllipticOperator(fa.LinearElasticMaterial()), qt.with_uniform_data(LAME), None, 3
    if cfg == "c3":
        m = fa.procedural.create_unit_box_uniform_tet_mesh_3d(75)
        rng = np.random.Generator(np.random.MT19937(12345))
        vp = rng.permutation(m.num_nodes())
        inv = np.empty_like(vp)
        inv[vp] = np.arange(len(vp))
        mesh = fa.Mesh(m.vertices[vp], inv[m.connectivity.astype(np.int64)][rng.permutation(m.num_elements())].astype(np.uint64), fa.TET4)
        w, p = quadrature.total_order.tetrahedron(1)
        qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(LAME)
        return mesh, fa.MaterialEllipticOperator(fa.LinearElasticMaterial()), qt, None, 3
    if cfg == "c4":
        mesh = fa.hex27_mesh_from_hex8(fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 5, 5, 8, 10))
        w, p = quadrature.tensor.hexahedron_gauss(3)
        qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(LAME)
        A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
        return mesh, fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()), qt, (0.05 * mesh.vertices @ A.T).reshape(-1), 3
    raise SystemExit(f"unknown configuration {cfg}")


def slab_check(cfg):
    """what ONE RANK of the multi-GPU runs assembles: rank 1 of 8 of `bench.py --gpus 8` (ns: 216 x 216 x 216 own cells; c5: 256 x 256 x 32), the
    extended mesh with its halo layers, the element mask set, the two launches of SlabAssembly into an array of garbage -- against the atomic
    scatter of the same context.  (Round 3: the masked affine kernel was wrong whenever a workgroup walked several positions.)"""
    from fenris_amd import distributed as fd

    w, p = quadrature.tensor.hexahedron_gauss(2)
    qt = fa.UniformQuadratureTable.from_points_and_weights(p, w).with_uniform_data(LAME)

    def configure(engine, mesh):
        return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh)
                .with_operator(fa.MaterialEllipticOperator(fa.LinearElasticMaterial())).with_quadrature_table(qt).with_u(None).build())

    slab = fd.make_slab(1.0, 1, 1, 8, 216, 1, 8) if cfg == "ns-slab" else fd.make_slab(1.0, 1, 1, 1, 256, 1, 8)
    sa = fd.SlabAssembly(slab, configure, device=0, overlap=True, stream=torch.cuda.current_stream().cuda_stream)
    flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
    sa.values.fill_(-11.5)
    sa.main.assemble_matrix_rows_async(sa.values, flags, 0, sa.split)
    sa.main.assemble_matrix_async(sa.values, flags)
    sa.poll_status()
    torch.cuda.synchronize()
    kernel = sa.main.last_kernel_name()
    sa.main.set_row_range(0, slab.mesh.num_nodes())
    want = torch.zeros_like(sa.values)
    sa.main.assemble_matrix(want, fa.SCATTER_ATOMIC)
    scale = want.abs().max().item()
    diff = (sa.values - want).abs().max().item() / scale
    good = diff <= 1e-11
    print(f"{cfg}: rank 1 of 8, {slab.num_own_elements()} own of {slab.mesh.num_elements()} elements, nnz {sa.values.numel()}, kernel {kernel}: "
          f"max |gather - atomic| / max |K| = {diff:.2e}  {'OK' if good else 'FAILED'}", flush=True)
    sa.close()
    del want
    torch.cuda.empty_cache()
    return good


def main():
    ok = True
    for cfg in (sys.argv[1:] or ["ns", "ns-perturbed", "c2", "c3", "c4", "ns-slab", "c5-slab"]):
        if cfg.endswith("-slab"):
            ok &= slab_check(cfg)
            continue
        mesh, op, qt, u, s = problem(cfg)
        eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
        (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(op).with_quadrature_table(qt)
         .with_u(u).build())
        nnz = eng.build_pattern()
        ro, ci = eng.pattern()
        vg = torch.zeros(nnz, dtype=torch.float64, device="cuda:0")
        va = torch.zeros(nnz, dtype=torch.float64, device="cuda:0")
        eng.assemble_matrix(vg, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
        kernel = eng.last_kernel_name()
        eng.assemble_matrix(va, fa.SCATTER_ATOMIC)          # accumulates into zeros
        scale = va.abs().max().item()
        diff = (vg - va).abs().max().item() / scale
        # properties: zero row sums (rigid translations / constants are in the kernel of K), symmetry through K x . y = x . K y
        rows = torch.repeat_interleave(torch.arange(len(ro) - 1, device="cuda:0"), torch.as_tensor(np.diff(ro).astype(np.int64), device="cuda:0"))
        rs = torch.zeros(len(ro) - 1, dtype=torch.float64, device="cuda:0").index_add_(0, rows, vg).abs().max().item() / scale
        cols = torch.as_tensor(ci.astype(np.int64), device="cuda:0")
        g = torch.Generator(device="cuda:0").manual_seed(1)
        x = torch.rand(len(ro) - 1, dtype=torch.float64, device="cuda:0", generator=g)
        y = torch.rand(len(ro) - 1, dtype=torch.float64, device="cuda:0", generator=g)
        kx = torch.zeros_like(x).index_add_(0, rows, vg * x[cols])
        ky = torch.zeros_like(x).index_add_(0, rows, vg * y[cols])
        sym = abs((kx @ y - x @ ky).item()) / max(abs((kx @ y).item()), 1e-300)
        good = diff <= 1e-11 and sym <= 1e-10 and (cfg == "c4" or rs <= 1e-10)
        ok &= good
        print(f"{cfg}: {mesh.num_elements()} elements, nnz {nnz}, kernel {kernel}: max |gather - atomic| / max |K| = {diff:.2e}, "
              f"max |row sum| / max |K| = {rs:.2e}, |Kx.y - x.Ky| / |Kx.y| = {sym:.2e}  {'OK' if good else 'FAILED'}", flush=True)
        eng.close()
        del vg, va, rows, cols, kx, ky
        torch.cuda.empty_cache()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
