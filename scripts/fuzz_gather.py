#!/usr/bin/env python3
"""Randomised cross-check of the owner-computes (gather) kernels against the generic atomic scatter on the device: random box sizes, random
holes, optionally permuted numbering, distorted / affine / mixed geometry, element masks, row ranges, FH_ASSEMBLE_OVERWRITE into garbage.
The atomic path is the reference-shaped one (pinned by the oracle in tests/); this looks for table-builder corner cases the fixed tests miss.
    python scripts/fuzz_gather.py [cases] [seed] [big]  (prints the case before it runs: a GPU fault names its culprit; `big`: boxes up to 25^3,
                                                         Hex27, NeoHookean / StVK as well)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

def run(cases=200, seed0=0, quiet=False, big=False):
    eng = fa.Engine(0)
    lame = fa.LameParameters(3.0e2, 5.0e2)
    bad = 0
    kernels = {}
    for it in range(cases):
        rng = np.random.default_rng(seed0 + it)
        kind = rng.choice(["HEX8", "HEX8", "TET4", "TET4", "QUAD4", "TRI3", "HEX27"] if big else ["HEX8", "HEX8", "TET4", "TET4", "QUAD4", "TRI3"])
        dims = rng.integers(1, 26 if (big and kind != "HEX27") else (5 if kind == "HEX27" else 10), 3)
        if kind == "HEX8":
            m = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, int(dims[0]), int(dims[1]), int(dims[2]), 1)
            w, p = quadrature.tensor.hexahedron_gauss(int(rng.integers(1, 4)))
        elif kind == "HEX27":
            m = fa.hex27_mesh_from_hex8(fa.procedural.create_rectangular_uniform_hex_mesh(1.0, int(dims[0]), int(dims[1]), int(dims[2]), 1))
            w, p = quadrature.tensor.hexahedron_gauss(3)
        elif kind == "TET4":
            m = fa.procedural.create_unit_box_uniform_tet_mesh_3d(int(dims[0] % (12 if big else 5) + 1))
            w, p = quadrature.total_order.tetrahedron(int(rng.integers(1, 4)))
        elif kind == "QUAD4":
            m = fa.procedural.create_unit_square_uniform_quad_mesh_2d(int(dims[0] + 1))
            w, p = quadrature.tensor.quadrilateral_gauss(int(rng.integers(1, 4)))
        else:
            q = fa.procedural.create_unit_square_uniform_quad_mesh_2d(int(dims[0] + 1))
            c = np.asarray(q.connectivity)
            m = fa.Mesh(q.vertices, np.concatenate([c[:, [0, 1, 2]], c[:, [0, 2, 3]]]), fa.TRI3)
            w, p = quadrature.total_order.triangle(int(rng.integers(1, 4)))
        v, c = m.vertices.copy(), np.asarray(m.connectivity).astype(np.int64)
        geo = rng.choice(["affine", "distorted", "mixed"])
        h = 1.0 / max(dims[0], 1) if kind != "HEX8" else 1.0
        if geo == "distorted":
            v += rng.uniform(-0.08 * h, 0.08 * h, v.shape)
        elif geo == "mixed":
            sel = rng.random(len(v)) < 0.3
            v[sel] += rng.uniform(-0.08 * h, 0.08 * h, (int(sel.sum()), v.shape[1]))
        keep = rng.random(len(c)) >= rng.choice([0.0, 0.1, 0.4])
        if not keep.any():
            keep[0] = True
        c = c[keep]
        if rng.random() < 0.3:      # permuted numbering (isolated vertices stay in: empty rows)
            perm = rng.permutation(len(v))
            inv = np.empty_like(perm)
            inv[perm] = np.arange(len(v))
            v, c = v[perm], inv[c][rng.permutation(len(c))]
        mesh = fa.Mesh(v, c.astype(np.uint64), m.elem_kind)
        opname = rng.choice(["LAPLACE", "LINEAR_ELASTIC", "NEO_HOOKEAN", "STVK"] if big else ["LAPLACE", "LINEAR_ELASTIC"])
        d = v.shape[1]
        s = 1 if opname == "LAPLACE" else d
        qt = fa.UniformQuadratureTable.from_points_and_weights(p, w)
        if opname != "LAPLACE":
            qt = qt.with_uniform_data(lame)
        op = {"LAPLACE": fa.LaplaceOperator, "LINEAR_ELASTIC": lambda: fa.MaterialEllipticOperator(fa.LinearElasticMaterial()),
              "NEO_HOOKEAN": lambda: fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()),
              "STVK": lambda: fa.MaterialEllipticOperator(fa.StVKMaterial())}[opname]()
        masked = rng.random() < 0.3
        ranged = rng.random() < 0.3
        n = mesh.num_nodes()
        lo_n, hi_n = (sorted(rng.integers(0, n + 1, 2)) if ranged else (0, n))
        if not quiet:
            print(f"case {seed0 + it}: {kind} dims {dims.tolist()} {geo} E={len(c)} N={n} {opname} nq={len(w)} mask={masked} rows=[{lo_n},{hi_n})", end=" ", flush=True)
        asm = (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(op).with_quadrature_table(qt)
               .with_u(1e-3 * rng.standard_normal(s * n) if opname in ("NEO_HOOKEAN", "STVK") else np.zeros(s * n)).build())
        nnz = eng.build_pattern()
        ro, _ = eng.pattern(want_cols=False)
        if masked:
            eng.set_active_elements(rng.random(len(c)) < 0.7)
        want = torch.zeros(nnz, dtype=torch.float64, device="cuda")
        eng.assemble_matrix(want, fa.SCATTER_ATOMIC)
        # few workgroups: every one walks many positions (carries between positions, chains, prefetch stages) even on a small mesh
        grid = rng.choice([0, 0, 1, 2, 3, 5, 17])
        for name in ("FENRIS_HIP_AFFINE_GRID", "FENRIS_HIP_PIPE_GRID", "FENRIS_HIP_TWO_PASS_GRID"):
            eng.set_option(name, str(int(grid)) if grid else None)
        eng.set_row_range(int(lo_n), int(hi_n))
        got = torch.full((nnz,), 4.5, dtype=torch.float64, device="cuda")
        eng.assemble_matrix(got, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
        kern = eng.last_kernel_name()
        if not quiet:
            print(f"grid={int(grid)} {kern}", flush=True)
        kernels[kern] = kernels.get(kern, 0) + 1
        eng.set_row_range(0, n)
        eng.set_active_elements(None)
        for name in ("FENRIS_HIP_AFFINE_GRID", "FENRIS_HIP_PIPE_GRID", "FENRIS_HIP_TWO_PASS_GRID"):
            eng.set_option(name, None)
        # FH_ASSEMBLE_REPRODUCIBLE on the whole range: the same matrix (to rounding: the atomic kernels take the two-pass form), twice the same bits
        rep_bad = False
        if it % 3 == 0:
            if masked:
                eng.set_active_elements(None)
            r1 = torch.full((nnz,), -7.25, dtype=torch.float64, device="cuda")
            eng.set_option("FENRIS_HIP_TWO_PASS_ROWS_GRID", "5" if it % 2 else None)
            eng.assemble_matrix(r1, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE | fa.ASSEMBLE_REPRODUCIBLE)
            kr = eng.last_kernel_name()
            eng.set_option("FENRIS_HIP_TWO_PASS_ROWS_GRID", None)
            r2 = torch.full((nnz,), 1.5, dtype=torch.float64, device="cuda")
            eng.assemble_matrix(r2, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE | fa.ASSEMBLE_REPRODUCIBLE)
            full = torch.zeros(nnz, dtype=torch.float64, device="cuda")
            eng.assemble_matrix(full, fa.SCATTER_ATOMIC)
            fv = full.cpu().numpy()
            sc = max(np.abs(fv).max(), 1e-300)
            same_nan = np.array_equal(np.isnan(r1.cpu().numpy()), np.isnan(fv))
            okv = ~np.isnan(fv)
            rep_bad = (not torch.equal(torch.nan_to_num(r1, nan=0.0), torch.nan_to_num(r2, nan=0.0))) or not same_nan or \
                (okv.any() and np.abs(r1.cpu().numpy()[okv] - fv[okv]).max() > 1e-12 * sc) or "pipelined" in kr or "<gather>" in kr
            if rep_bad:
                print(f"   REPRODUCIBLE MISMATCH ({kr})", flush=True)
        wv, gv = want.cpu().numpy(), got.cpu().numpy()
        lo, hi = int(ro[s * lo_n]), int(ro[s * hi_n])
        scale = max(np.abs(wv).max(), 1e-300)
        ok = np.all(gv[:lo] == 4.5) and np.all(gv[hi:] == 4.5) and (hi == lo or np.abs(gv[lo:hi] - wv[lo:hi]).max() <= 1e-12 * scale)
        if rep_bad:
            bad += 1
        if not ok:
            bad += 1
            err = np.abs(gv[lo:hi] - wv[lo:hi]).max() / scale if hi > lo else 0.0
            print(f"   MISMATCH ({kern}): inside {err:.3e}, outside touched {int((gv[:lo] != 4.5).sum() + (gv[hi:] != 4.5).sum())}", flush=True)
    print("kernels:", kernels)
    print("cases", cases, "mismatches", bad)
    return bad, kernels


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0, big=len(sys.argv) > 3)
