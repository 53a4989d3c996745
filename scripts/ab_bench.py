#!/usr/bin/env python3
"""In-process A/B timing of library variants (FENRIS_HIP_* switches are read when a context is created): one context per
variant on the same mesh, the variants timed in turn, round after round, so that clock / temperature drift of the device hits
all of them alike.

    python scripts/ab_bench.py --config ns --rounds 8 --reps 5 "label1:VAR=1,VAR2=3" "label2:" ...
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="ns", choices=["ns", "ns-perturbed", "c2", "c3", "c4", "c5"])
    ap.add_argument("--cells", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()

    import torch

    import fenris_amd as fa
    from fenris_amd import quadrature

    lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
    cfg = args.config
    u = None
    if cfg in ("ns", "ns-perturbed", "c5"):
        cells = args.cells or (256 if cfg == "c5" else 216)
        mesh = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, cells)
        if cfg == "ns-perturbed":
            rng = np.random.Generator(np.random.MT19937(2024))
            mesh = fa.Mesh(mesh.vertices + (0.1 / cells) * rng.uniform(-1.0, 1.0, mesh.vertices.shape), mesh.connectivity, fa.HEX8)
        rule = quadrature.tensor.hexahedron_gauss(2)
        op, params = fa.MaterialEllipticOperator(fa.LinearElasticMaterial()), lame
    elif cfg == "c2":
        mesh = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, args.cells or 128)
        rule = quadrature.tensor.hexahedron_gauss(2)
        op, params = fa.LaplaceOperator(), None
    elif cfg == "c3":
        m = fa.procedural.create_unit_box_uniform_tet_mesh_3d(args.cells or 75)
        rng = np.random.Generator(np.random.MT19937(12345))
        vp = rng.permutation(m.num_nodes())
        inv = np.empty_like(vp)
        inv[vp] = np.arange(len(vp))
        mesh = fa.Mesh(m.vertices[vp], inv[m.connectivity.astype(np.int64)][rng.permutation(m.num_elements())].astype(np.uint64), fa.TET4)
        rule = quadrature.total_order.tetrahedron(1)
        op, params = fa.MaterialEllipticOperator(fa.LinearElasticMaterial()), lame
    else:
        mesh = fa.hex27_mesh_from_hex8(fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 5, 5, 8, 10))
        rule = quadrature.tensor.hexahedron_gauss(3)
        op, params = fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()), lame
        A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
        u = (0.05 * mesh.vertices @ A.T).reshape(-1)
    weights, points = rule
    qtable = fa.UniformQuadratureTable.from_points_and_weights(points, weights)
    if params is not None:
        qtable = qtable.with_uniform_data(params)
    stream = torch.cuda.current_stream().cuda_stream

    engines, labels = [], []
    values = None
    for spec in args.variants:
        label, _, envs = spec.partition(":")
        added = []
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            os.environ[k] = v
            added.append(k)
        eng = fa.Engine(0, stream=stream)
        (fa.ElementEllipticAssemblerBuilder(eng).with_finite_element_space(mesh).with_operator(op).with_quadrature_table(qtable)
         .with_u(u).build())
        nnz = eng.build_pattern()
        for k in added:
            del os.environ[k]
        if values is None:
            values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
        eng.assemble_matrix_async(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)  # tables, warm-up
        eng.poll_status()
        engines.append(eng)
        labels.append(label)
    torch.cuda.synchronize()
    flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
    times = [[] for _ in engines]
    for _ in range(args.rounds):
        for k, eng in enumerate(engines):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                eng.assemble_matrix_async(values, flags)
            b.record()
            torch.cuda.synchronize()
            times[k].append(a.elapsed_time(b) / args.reps)
    for k, label in enumerate(labels):
        t = sorted(times[k])
        print(json.dumps({"variant": label, "config": cfg, "ms_median": round(t[len(t) // 2], 4), "ms_min": round(t[0], 4),
                          "ms_max": round(t[-1], 4), "kernel": engines[k].last_kernel_name()}), flush=True)


if __name__ == "__main__":
    main()
