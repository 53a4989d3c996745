#!/usr/bin/env python3
"""Headline benchmark: elements/sec assembling the global stiffness matrix K, 3-D Hex8 linear elasticity.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A step = one numeric assembly pass (CsrAssembler::assemble semantics: K written into a pre-built CSR
pattern; benches/assembly.rs:126-145 times the same call) over a synthetic structured mesh that is
already resident in HBM.  N = 1: BASELINE's north-star point, Hex8 elasticity on the 216^3 unit box
(10 077 696 elements).  N > 1: weak scaling, the global mesh is 216 x 216 x (216 N) cells cut into N
z-slabs, one per rank (see fenris_amd/distributed.py); value = all elements of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(E, N, s, n, d, nnz, uses_u=False):
    """connectivity as i32 + each vertex once + (u) + each value written once + each column
    index read once (i32) + row offsets."""
    return E * n * 4 + N * d * 8 + (s * N * 8 if uses_u else 0) + nnz * 8 + nnz * 4 + (s * N + 1) * 8


def cpu_baseline(cells, threads):
    """The oracle (restatement of fenris's CPU path, NOT the Rust binary) timed on the host cores."""
    import numpy as np

    from oracle import oracle

    v, c = oracle.unit_box_hex_mesh(cells)
    w, p = oracle.hexahedron_gauss(2)
    asm = oracle.ElementAssembler(oracle.HEX8, oracle.LINEAR_ELASTIC, v, c, w, p,
                                  params=oracle.lame_from_young_poisson(1e6, 0.2))
    ro, ci = oracle.pattern_for(asm)
    colors = oracle.color_nodes(asm)
    vals = np.zeros(len(ci))
    t0 = time.perf_counter()
    st, _ = oracle.par_assemble_into_csr(asm, colors, ro, ci, vals, num_threads=threads)
    t_par = time.perf_counter() - t0
    assert st == 0
    vals[:] = 0
    t0 = time.perf_counter()
    st, _ = oracle.assemble_into_csr(asm, ro, ci, vals)
    t_ser = time.perf_counter() - t0
    assert st == 0
    E = len(c)
    return {"value": E / t_par, "unit": "elements/s", "cores": threads, "kind": "port",
            "serial_value": E / t_ser,
            "sample": f"Hex8 linear elasticity {cells}^3 = {E} elements, coloured parallel assembly "
                      f"(CsrParAssembler restatement, {threads} OpenMP threads, {t_par:.1f} s) and serial "
                      f"({t_ser:.1f} s); restatement of fenris CPU path, not the Rust binary"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells", type=int, default=216, help="cells per unit box edge (216 = north-star point)")
    ap.add_argument("--scatter", default="gather", choices=["gather", "atomic", "colored"])
    ap.add_argument("--operator", default="elasticity", choices=["elasticity", "poisson"])
    ap.add_argument("--partition", default="exchange", choices=["exchange", "halo"],
                    help="N > 1: interface rows sent to their owner over RCCL (headline), or the halo element layer "
                         "recomputed locally with no communication")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: exchange after the kernel instead of beside it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cells", type=int, default=88)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import fenris_amd as fa
    from fenris_amd import quadrature

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # FENRIS_BENCH_SHARE_DEVICE=1 (validation only): all ranks on cuda:0 over gloo -- lets the N > 1 code path be
    # exercised on a single-GPU box; never used for reported numbers
    share = os.environ.get("FENRIS_BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # ---- synthetic input: this rank's slab of the 216 x 216 x (216 world) box
    cells = args.cells
    stream = torch.cuda.current_stream().cuda_stream
    weights, points = quadrature.tensor.hexahedron_gauss(2)
    qtable = fa.UniformQuadratureTable.from_points_and_weights(points, weights)
    if args.operator == "elasticity":
        lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
        qtable = qtable.with_uniform_data(lame)
        op, s = fa.MaterialEllipticOperator(fa.LinearElasticMaterial()), 3
    else:
        op, s = fa.LaplaceOperator(), 1

    def configure(engine, mesh_):
        return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh_).with_operator(op)
                .with_quadrature_table(qtable).with_u(None).build())

    flags = {"gather": fa.SCATTER_GATHER, "atomic": fa.SCATTER_ATOMIC, "colored": fa.SCATTER_COLORED}[args.scatter]
    slab_asm = None
    if world == 1:
        mesh = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, cells)
        eng = fa.Engine(local_rank, stream=stream)
        configure(eng, mesh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nnz = eng.build_pattern()  # assemble_pattern on the device (secondary metric)
        E = mesh.num_elements()
        values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    else:
        from fenris_amd import distributed as fd

        slab = fd.make_slab(1.0, 1, 1, world, cells, rank, world, args.partition)
        mesh = slab.mesh
        t0 = time.perf_counter()  # N > 1: engines, masks and patterns of this rank
        # interface rows first, their RCCL transfer overlapped with the rest (owner-computes only)
        slab_asm = fd.SlabAssembly(slab, configure, device=local_rank, overlap=(args.scatter == "gather" and not args.no_overlap),
                                   stream=stream)
        eng, values, nnz = slab_asm.main, slab_asm.values, slab_asm.values.numel()
        E = slab.num_own_elements()  # numerics over own (+ halo in "halo" mode) elements, pattern over own + halo
    t_pattern = time.perf_counter() - t0
    N = mesh.num_nodes()
    if args.scatter == "colored":
        eng.color()
    flags |= fa.ASSEMBLE_OVERWRITE

    def step():
        if slab_asm is not None:
            slab_asm.enqueue(flags)
        else:
            eng.assemble_matrix_async(values, flags)

    for _ in range(args.warmup):
        step()
    (slab_asm or eng).poll_status()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        if slab_asm is None or slab_asm.comm is not None:
            step()      # N > 1 overlapped: the events bracket both launches and the wait for the transfer
            b.record()
        else:
            eng.assemble_matrix_async(values, flags)
            b.record()
            slab_asm.exchange.run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    (slab_asm or eng).poll_status()
    kernel_ms = sorted(a.elapsed_time(b) for a, b in ev)
    kernel_avg_ms = sum(kernel_ms) / len(kernel_ms)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        e_t = torch.tensor([float(E)], dtype=torch.float64, device="cuda")
        dist.all_reduce(e_t, op=dist.ReduceOp.SUM)
        total_elements = float(e_t.item())
    else:
        total_elements = float(E)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_elements * args.steps / elapsed
        abytes = algorithmic_bytes(E, N, s, 8, 3, nnz)
        achieved = abytes / (kernel_avg_ms * 1e-3) / 1e9
        out = {
            "metric": "elements/sec assembling global stiffness K, 3D Hex8 elasticity" if s == 3 else
                      "elements/sec assembling global stiffness K, 3D Hex8 Poisson",
            "value": value, "unit": "elements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Hex8 {'linear elasticity' if s == 3 else 'Poisson'} stiffness assembly, "
                                   f"structured {cells}x{cells}x{cells * world} unit-cell box "
                                   f"({int(total_elements)} elements), hexahedron_gauss(2), "
                                   f"YoungPoisson(1e6, 0.2), u = 0, CSR pattern pre-built, values overwritten",
                       "elements_per_gpu": E, "nodes_per_gpu": N, "nnz_per_gpu": nnz, "scatter": args.scatter,
                       "partition": "single" if world == 1 else (f"{world} z-slabs, interface rows exchanged" if args.partition == "exchange"
                                                                   else f"{world} z-slabs, halo element layer recomputed, no communication"),
                       "pattern_build_s": t_pattern},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": achieved / PEAK_HBM_GBS, "traffic": None,
                         "kernel": eng.last_kernel_name(), "kernel_avg_ms": kernel_avg_ms,
                         "kernel_min_ms": kernel_ms[0], "algorithmic_bytes_per_launch": abytes,
                         "bytes_per_element": abytes / E},
        }
        # what the device sustains on the dominant traffic of this kernel (writing the values once): a plain fill of
        # the same array, timed the same way -- the practical ceiling next to the nominal 8 TB/s
        try:
            a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            scratch = torch.empty_like(values)
            scratch.fill_(1.0)
            torch.cuda.synchronize()
            a0.record()
            for _ in range(5):
                scratch.fill_(1.0)
            b0.record()
            torch.cuda.synchronize()
            out["roofline"]["measured_write_GBps"] = 5 * nnz * 8 / (a0.elapsed_time(b0) * 1e-3) / 1e9
            del scratch
        except RuntimeError:
            pass
        # HBM traffic of the dominant kernel from committed rocprofv3 PMC passes (bench.py cannot collect PMC
        # itself): FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B on wide
        # coalesced reads -- an upper bound for our mixed-width reads), WRITE_SIZE as reported; KB -> bytes.
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "traffic_r01.json")))
            if t["cells"] == cells and t["scatter"] == args.scatter and t["operator"] == args.operator and world == 1:
                out["roofline"]["traffic"] = (2.0 * t["fetch_size_kb"] + t["write_size_kb"]) * 1024.0
                out["roofline"]["traffic_source"] = "profiles/traffic_r01.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE)"
        except (OSError, KeyError, ValueError):
            pass
        if not args.no_cpu_baseline and world == 1:
            threads = os.cpu_count() or 1
            out["cpu_baseline"] = cpu_baseline(args.cpu_cells, threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
