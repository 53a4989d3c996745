#!/usr/bin/env python3
"""Benchmark of the hot path: numeric assembly of the global stiffness matrix K into a pre-built CSR pattern.

    python bench.py --gpus N --steps K --warmup W [--config ns|ns-perturbed|c2|c3|c4|c5]     (N > 1: ns, c5 as z-slabs; c3, ns-perturbed as element partitions)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
    (``python bench.py --gpus N`` with N > 1 and no launcher around it starts that launcher itself -- one rank per GPU as a child
    process group, before this process touches a GPU -- relays rank 0's line and fails loudly when it cannot.)

A step = one numeric assembly pass (CsrAssembler::assemble semantics: K written into a pre-built CSR pattern;
benches/assembly.rs:126-145 times the same call) over a synthetic mesh that is already resident in HBM.

Configurations (BASELINE.json `configs`):
  ns            north-star point, the default: Hex8 linear elasticity, structured 216^3 unit box (10 077 696 elements).
                N > 1: weak scaling, 216 x 216 x (216 N) cells cut into N z-slabs, one per rank.
  ns-perturbed  the same mesh with every vertex moved by up to +-0.1 h per coordinate: nothing is affine, the general
                owner-computes kernel runs (reported next to the headline, VERDICT r01 task 1).
  c2            Hex8 Poisson 128^3.
  c3            Tet4 linear elasticity, BCC res 75, vertices and elements randomly permuted (seed 12345).
  c4            Hex27 NeoHookean 50 x 50 x 80, hexahedron_gauss(3); roofline = fp64 matrix-core flops.
  c5            Hex8 linear elasticity 256^3 cut into N z-slabs (strong scaling; 32 layers per rank at N = 8).

Prints ONE JSON line on rank 0.  The default run (--config ns on one GPU) appends `secondary_vectors` (residual vector and energy of the
headline mesh) and `secondary`: every other configuration
(c2, c3, c4, c5 at N = 1, ns-perturbed) timed in the same process after the headline -- {ms, frac of its roofline, kernel}.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PEAK_FP64_TFLOPS = 78.6   # MI355X fp64 vector = fp64 matrix-core peak


def algorithmic_bytes(E, N, s, n, d, nnz, uses_u=False):
    """connectivity as i32 + each vertex once + (u) + each value written once + each column
    index read once (i32) + row offsets."""
    return E * n * 4 + N * d * 8 + (s * N * 8 if uses_u else 0) + nnz * 8 + nnz * 4 + (s * N + 1) * 8


def host_threads():
    """threads this process may actually use: CPU affinity, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, quota) if quota else n), n, quota


def cpu_baseline(kind, budget_s=25.0):
    """The oracle (restatement of fenris's CPU path, NOT the Rust binary) timed on the host cores: the serial assembler
    (CsrAssembler, global.rs:133-182) and the coloured parallel one (CsrParAssembler, global.rs:314-376, OpenMP over the
    elements of a colour) at several thread counts; the best parallel figure is the baseline."""
    import numpy as np

    from oracle import oracle

    lame = oracle.lame_from_young_poisson(1e6, 0.2)
    if kind == "hex8_elasticity":
        cells = 88
        v, c = oracle.unit_box_hex_mesh(cells)
        w, p = oracle.hexahedron_gauss(2)
        asm = oracle.ElementAssembler(oracle.HEX8, oracle.LINEAR_ELASTIC, v, c, w, p, params=lame)
        what = f"Hex8 linear elasticity {cells}^3"
    elif kind == "hex8_poisson":
        cells = 100
        v, c = oracle.unit_box_hex_mesh(cells)
        w, p = oracle.hexahedron_gauss(2)
        asm = oracle.ElementAssembler(oracle.HEX8, oracle.LAPLACE, v, c, w, p)
        what = f"Hex8 Poisson {cells}^3"
    elif kind == "tet4_elasticity":
        cells = 40
        v, c = oracle.unit_box_tet_mesh(cells)
        w, p = oracle.tetrahedron_rule(1)
        asm = oracle.ElementAssembler(oracle.TET4, oracle.LINEAR_ELASTIC, v, c, w, p, params=lame)
        what = f"Tet4 linear elasticity BCC res {cells}"
    else:  # hex27_neohookean
        v8, c8 = oracle.hex_mesh(1.0, 1, 1, 1, 10)
        v, c = oracle.hex8_to_hex27(v8, c8)
        w, p = oracle.hexahedron_gauss(3)
        A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
        u = (0.05 * v @ A.T).reshape(-1)
        asm = oracle.ElementAssembler(oracle.HEX27, oracle.NEO_HOOKEAN, v, c, w, p, params=lame, u=u)
        what = "Hex27 NeoHookean 10^3"
    E = len(c)
    ro, ci = oracle.pattern_for(asm)
    colors = oracle.color_nodes(asm)
    vals = np.zeros(len(ci))
    t_start = time.perf_counter()
    t0 = time.perf_counter()
    st, _ = oracle.assemble_into_csr(asm, ro, ci, vals)
    t_ser = time.perf_counter() - t0
    assert st == 0
    usable, affinity, quota = host_threads()
    sweep = sorted({t for t in (1, 8, 32, usable) if t <= usable})
    par = {}
    for t in sweep:
        if t == 1:
            continue
        if time.perf_counter() - t_start > budget_s:
            break
        vals[:] = 0
        t0 = time.perf_counter()
        st, _ = oracle.par_assemble_into_csr(asm, colors, ro, ci, vals, num_threads=t)
        par[t] = time.perf_counter() - t0
        assert st == 0
    best_t, best = (min(par.items(), key=lambda kv: kv[1]) if par else (1, t_ser))
    if t_ser < best:
        best_t, best = 1, t_ser
    return {"value": E / best, "unit": "elements/s", "cores": best_t, "kind": "port",
            "serial_value": E / t_ser,
            "threads_swept": {str(t): E / s for t, s in par.items()},
            "host": {"os_cpu_count": os.cpu_count(), "affinity": affinity, "cgroup_quota": quota, "usable": usable},
            "sample": f"{what} = {E} elements: serial CsrAssembler restatement {t_ser:.2f} s; coloured parallel "
                      f"CsrParAssembler restatement (OpenMP over the elements of a colour) at "
                      + ", ".join(f"{t} threads {s:.2f} s" for t, s in par.items())
                      + f"; best = {best_t} thread(s).  Restatement of fenris's CPU path, not the Rust binary"}


def measure_traffic(argv_child, kernel_substrs, timeout_s=240):
    """HBM bytes per launch of the dominant kernel from two rocprofv3 PMC passes of this very command (separate passes:
    FETCH_SIZE and WRITE_SIZE do not fit one; MI355X_MICROARCH.md, HBM section).  Runs bench.py as a CHILD process under
    rocprofv3 (never exec-replaces a process that touched the GPU).  FETCH_SIZE is doubled (gfx950 tallies 128-byte
    requests at 64 bytes on wide coalesced reads: an upper bound for mixed-width reads), WRITE_SIZE taken as reported;
    both are in KiB.  Returns (bytes or None, detail dict)."""
    import csv
    import glob
    import shutil
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, {"error": "rocprofv3 not found"}
    out = {}
    tmp = tempfile.mkdtemp(prefix="fenris_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "run", "--",
                   sys.executable, os.path.join(ROOT, "bench.py")] + argv_child
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=timeout_s, check=False)
            except subprocess.TimeoutExpired:
                return None, {"error": f"rocprofv3 pass {counter} timed out"}
            rows = []  # (kernel name, value) of this counter, one per dispatch
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter:
                            rows.append((row.get("Kernel_Name", ""), float(row["Counter_Value"])))
            if not rows:  # ROCm 7 default output: a rocpd SQLite database
                import sqlite3

                for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
                    try:
                        db = sqlite3.connect(f)
                        rows += [(k, float(v)) for k, v in db.execute(
                            "select kernel_name, value from counters_collection where counter_name = ?", (counter,))]
                    except sqlite3.Error:
                        pass
            total = 0.0
            for kname in kernel_substrs:
                vals = [v for k, v in rows if kname in k]
                if not vals:
                    return None, {"error": f"no {counter} rows for kernel {kname}"}
                total += sum(vals) / len(vals)
                out[f"{counter}_{kname}_KiB"] = sum(vals) / len(vals)
                out[f"{counter}_{kname}_dispatches"] = len(vals)
            out[counter] = total
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0, out


def start_traffic_helper(argv_child, kernel_substrs):
    """A child process started BEFORE this one touches a GPU; it waits on its stdin and, when told to, runs the two PMC passes
    (measure_traffic) and prints their result.  This way the passes come AFTER the timed region -- with them in front the device
    entered the timed region in a slower state (five runs on one box type: 4.80 - 4.88 ms after the passes, 4.62 - 4.68 without them;
    profiles/r03_affine_experiments.txt 11) -- and still no process that has initialised the GPU ever spawns another."""
    env = dict(os.environ, FENRIS_BENCH_CHILD="1")
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--traffic-helper",
                             json.dumps({"child": argv_child, "kernels": list(kernel_substrs)})],
                            stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)


def traffic_helper_main(spec):
    """body of the helper process: never imports torch, never touches the GPU itself"""
    spec = json.loads(spec)
    line = sys.stdin.readline().strip()
    if line != "go":
        return
    t0 = time.perf_counter()
    try:
        traffic, detail = measure_traffic(spec["child"], spec["kernels"])
    except Exception as exc:  # the profiler must never take the benchmark down
        traffic, detail = None, {"error": repr(exc)}
    if detail is not None:
        detail["kernels"] = spec["kernels"]
        detail["seconds"] = time.perf_counter() - t0
    print(json.dumps({"traffic": traffic, "detail": detail}), flush=True)


def finish_traffic_helper(helper, go, timeout_s=600):
    """tell the helper to run (or to leave), collect (traffic, detail)"""
    if helper is None:
        return None, None
    try:
        out, _ = helper.communicate("go\n" if go else "skip\n", timeout=timeout_s)
        for line in out.splitlines():
            if line.startswith("{"):
                d = json.loads(line)
                return d["traffic"], d["detail"]
        return None, ({"error": "traffic helper printed nothing"} if go else None)
    except Exception as exc:
        try:
            helper.kill()
        except OSError:
            pass
        return None, {"error": repr(exc)}


def self_launch(args):
    """--gpus N > 1 without a launcher (WORLD_SIZE unset): start `torch.distributed.run` with N ranks as a CHILD process (this
    process has not touched a GPU), relay rank 0's JSON line, exit with the child's code.  Refuses when the box has fewer GPUs."""
    import socket

    import torch  # device_count() does not initialise the GPU

    share = os.environ.get("FENRIS_BENCH_SHARE_DEVICE") == "1"
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not share:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but {ndev} GPU(s) visible: refusing to report an N = {args.gpus} figure "
                         f"from fewer devices\n")
        raise SystemExit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # every rank's stdout / stderr also goes to a file of its own (torchrun --tee): when the launch fails or hangs, the parent prints the tail of
    # EVERY rank's stderr, so that a stuck RCCL bootstrap can be told from a kernel failure
    import shutil
    import signal
    import tempfile

    log_dir = tempfile.mkdtemp(prefix="fenris_bench_ranks_")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--log-dir", log_dir, "--tee", "3", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("NCCL_DEBUG", "WARN")   # RCCL reports what it could not do (a failed bootstrap is otherwise silent)
    # a fresh child in a process group of its own (never a re-exec of this process): on expiry exactly that group is killed
    pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    timed_out = False
    try:
        out, err = pr.communicate(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        timed_out = True
        try:
            os.killpg(pr.pid, signal.SIGKILL)   # start_new_session: the child leads the group pr.pid
        except OSError:
            pass
        out, err = pr.communicate()

    def rank_tails():
        tails = []
        for root, _, files in sorted(os.walk(log_dir)):
            for fn in sorted(files):
                if fn == "stderr.log":
                    try:
                        with open(os.path.join(root, fn), errors="replace") as fh:
                            tails.append((os.path.basename(root), fh.read()[-6000:]))
                    except OSError:
                        pass
        return tails

    lines = [ln for ln in (out or "").splitlines() if ln.startswith("{") and '"metric"' in ln]
    # (--tee prefixes the ranks' lines with "[default0]:" on some torch versions)
    if not lines:
        lines = [ln[ln.index("{"):] for ln in (out or "").splitlines() if '"metric"' in ln and "{" in ln]
    if timed_out or pr.returncode != 0 or not lines:
        for rk, tail in rank_tails():
            sys.stderr.write(f"---- rank {rk}: last of its stderr ----\n{tail}\n")
        sys.stderr.write("---- launcher stderr (tail) ----\n" + (err or "")[-6000:])
        if timed_out:
            sys.stderr.write(f"\nbench.py: the {args.gpus}-rank launch did not finish within --launch-timeout {args.launch_timeout} s: its process "
                             f"group was killed\n")
            shutil.rmtree(log_dir, ignore_errors=True)
            raise SystemExit(124)
        sys.stderr.write(f"\nbench.py: the {args.gpus}-rank launch failed (exit code {pr.returncode})\n")
        shutil.rmtree(log_dir, ignore_errors=True)
        raise SystemExit(pr.returncode or 1)
    shutil.rmtree(log_dir, ignore_errors=True)
    line = json.loads(lines[-1])
    if line.get("n_gpus") != args.gpus:
        sys.stderr.write(f"bench.py: asked for {args.gpus} GPUs, the ranks report {line.get('n_gpus')}\n")
        raise SystemExit(1)
    print(lines[-1])
    raise SystemExit(0)


def config_problem(cfg, cells, fa, quadrature, np):
    """mesh builder, operator, quadrature table and bookkeeping of one configuration (one GPU)"""
    lame = fa.LameParameters.from_young_poisson(fa.YoungPoisson(1e6, 0.2))
    c = {"n_el": 8, "d": 3, "s": 3, "uses_u": False, "cpu_kind": "hex8_elasticity", "params": lame,
         "metric": "elements/sec assembling global stiffness K, 3D Hex8 elasticity"}
    if cfg in ("ns", "ns-perturbed", "c5"):
        cells = cells or (256 if cfg == "c5" else 216)
        rule = quadrature.tensor.hexahedron_gauss(2)
        c["op"] = fa.MaterialEllipticOperator(fa.LinearElasticMaterial())
        c["desc"] = f"Hex8 linear elasticity stiffness assembly, structured {cells}^3 unit-cell box"
    elif cfg == "c2":
        cells = cells or 128
        rule = quadrature.tensor.hexahedron_gauss(2)
        c.update(op=fa.LaplaceOperator(), params=None, s=1, cpu_kind="hex8_poisson",
                 metric="elements/sec assembling global stiffness K, 3D Hex8 Poisson",
                 desc=f"Hex8 Poisson stiffness assembly, structured {cells}^3 unit-cell box")
    elif cfg == "c3":
        cells = cells or 75
        rule = quadrature.total_order.tetrahedron(1)
        c.update(op=fa.MaterialEllipticOperator(fa.LinearElasticMaterial()), n_el=4, cpu_kind="tet4_elasticity",
                 metric="elements/sec assembling global stiffness K, 3D Tet4 elasticity",
                 desc=f"Tet4 linear elasticity stiffness assembly, BCC unit box res {cells}, vertices and elements permuted (MT19937 seed 12345)")
    else:  # c4
        cells = 0
        rule = quadrature.tensor.hexahedron_gauss(3)
        c.update(op=fa.MaterialEllipticOperator(fa.NeoHookeanMaterial()), n_el=27, uses_u=True, cpu_kind="hex27_neohookean",
                 metric="elements/sec assembling global stiffness K, 3D Hex27 NeoHookean",
                 desc="Hex27 NeoHookean stiffness assembly, 50x50x80 box, hexahedron_gauss(3), u = 0.05 A X")
    c["cells"], c["weights"], c["points"] = cells, rule[0], rule[1]
    qtable = fa.UniformQuadratureTable.from_points_and_weights(rule[1], rule[0])
    c["qtable"] = qtable.with_uniform_data(c["params"]) if c["params"] is not None else qtable

    def mesh():
        if cfg == "c3":
            m = fa.procedural.create_unit_box_uniform_tet_mesh_3d(cells)
            rng = np.random.Generator(np.random.MT19937(12345))
            vp = rng.permutation(m.num_nodes())
            inv = np.empty_like(vp)
            inv[vp] = np.arange(len(vp))
            return fa.Mesh(m.vertices[vp], inv[m.connectivity.astype(np.int64)][rng.permutation(m.num_elements())].astype(np.uint64), fa.TET4)
        if cfg == "c4":
            return fa.hex27_mesh_from_hex8(fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 5, 5, 8, 10))
        m = fa.procedural.create_rectangular_uniform_hex_mesh(1.0, 1, 1, 1, cells)
        if cfg == "ns-perturbed":
            rng = np.random.Generator(np.random.MT19937(2024))
            m = fa.Mesh(m.vertices + (0.1 / cells) * rng.uniform(-1.0, 1.0, m.vertices.shape), m.connectivity, fa.HEX8)
        return m

    def u_for(mesh_):
        if not c["uses_u"]:
            return None
        A = np.array([[1, .2, 0], [0, 1, .3], [.1, 0, 1]])
        return (0.05 * mesh_.vertices @ A.T).reshape(-1)

    def configure(engine, mesh_):
        return (fa.ElementEllipticAssemblerBuilder(engine).with_finite_element_space(mesh_).with_operator(c["op"])
                .with_quadrature_table(c["qtable"]).with_u(u_for(mesh_)).build())

    c["mesh"], c["configure"] = mesh, configure
    return c


def roofline_of(cfg, c, E, N, nnz, kernel_ms):
    """(bound, achieved, peak, unit, frac) of one assembly of kernel_ms milliseconds"""
    if cfg == "c4":
        nq, n_el = len(c["weights"]), c["n_el"]
        flops = E * (2 * (3 * n_el) ** 2 * nq * 2 + 6 * n_el * n_el * nq)
        tf = flops / (kernel_ms * 1e-3) / 1e12
        return "mfma", tf, PEAK_FP64_TFLOPS, "TFLOP/s", tf / PEAK_FP64_TFLOPS
    ab = algorithmic_bytes(E, N, c["s"], c["n_el"], c["d"], nnz, c["uses_u"])
    gbs = ab / (kernel_ms * 1e-3) / 1e9
    return "hbm", gbs, PEAK_HBM_GBS, "GB/s", gbs / PEAK_HBM_GBS


def probe_placement(eng, values, flags, torch, tries):
    """fenris_amd/placement.py: the better of several allocations of the values array and of the library's record buffer"""
    from fenris_amd.placement import probe_placement as probe

    return probe(eng, values, flags, tries)


def settle_device(eng, values, flags, max_s=2.5, group=20):
    """fenris_amd/placement.py: untimed assemblies until a fresh device has reached its steady rate (part of the set-up, like the probe)"""
    from fenris_amd.placement import settle_device as settle

    return settle(eng, values, flags, max_s, group)


def time_secondary(cfg, fa, quadrature, np, torch, stream, steps=5, warmup=2, tries=2, settle=False):
    """one secondary configuration on this GPU: {ms, frac, kernel, ...}; everything it allocates is released before it returns"""
    c = config_problem(cfg, 0, fa, quadrature, np)
    mesh = c["mesh"]()
    eng = fa.Engine(0, stream=stream)
    try:
        c["configure"](eng, mesh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nnz = eng.build_pattern()
        torch.cuda.synchronize()
        t_pattern = time.perf_counter() - t0
        values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
        flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.assemble_matrix_async(values, flags)     # the first assembly of the context: owner / lane tables, once per pattern
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t0
        settled = settle_device(eng, values, flags) if settle else None   # (the headline's set-up: a device that just released 30 GB is not at its steady rate)
        values, placement = probe_placement(eng, values, flags, torch, tries)
        for _ in range(warmup):
            eng.assemble_matrix_async(values, flags)
        eng.poll_status()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in ev:
            a.record()
            eng.assemble_matrix_async(values, flags)
            b.record()
        torch.cuda.synchronize()
        eng.poll_status()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        avg = sum(ms) / len(ms)
        E, N = mesh.num_elements(), mesh.num_nodes()
        bound, ach, peak, unit, frac = roofline_of(cfg, c, E, N, nnz, avg)
        return {"workload": c["desc"], "elements": E, "nodes": N, "nnz": nnz,
                "algorithmic_bytes": algorithmic_bytes(E, N, c["s"], c["n_el"], c["d"], nnz, c["uses_u"]),
                "ms": avg, "ms_min": ms[0], "elements_per_s": E / (avg * 1e-3),
                "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": frac, "kernel": eng.last_kernel_name(), "steps": steps,
                "pattern_build_s": t_pattern, "first_assembly_s": t_first, "placement_probe": placement, "device_settle": settled}
    finally:
        eng.close()
        values = None
        torch.cuda.empty_cache()



def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--traffic-helper":
        return traffic_helper_main(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="ns", choices=["ns", "ns-perturbed", "c2", "c3", "c4", "c5"])
    ap.add_argument("--cells", type=int, default=0, help="override the cells per box edge of ns / ns-perturbed / c2 / c5")
    ap.add_argument("--scatter", default="gather", choices=["gather", "atomic", "colored"])
    ap.add_argument("--operator", default=None, choices=["elasticity", "poisson"], help="(compatibility) poisson = --config c2 sizes")
    ap.add_argument("--partition", default="exchange", choices=["exchange", "halo"],
                    help="N > 1: interface rows sent to their owner over RCCL (headline), or the halo element layer "
                         "recomputed locally with no communication")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: exchange after the kernel instead of beside it")
    ap.add_argument("--exchange", default="torch", choices=["torch", "abi"],
                    help="N > 1: interface rows moved by torch.distributed point-to-point (default) or by the library's own RCCL "
                         "calls behind the C ABI (fh_group_*)")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="N > 1 without a launcher: seconds the torch.distributed.run child may take before its process group is killed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 PMC child passes (roofline.traffic = null)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other configurations timed after the headline (N = 1, --config ns)")
    ap.add_argument("--placement-tries", type=int, default=6,
                    help="N = 1: allocations of the values array (and of the library's record buffer) tried before the timed region, the "
                         "fastest kept; 0 = take the first (see probe_placement)")
    ap.add_argument("--no-settle", action="store_true", help="skip the untimed assemblies that bring a fresh device to its steady rate (settle_device)")
    ap.add_argument("--no-module-warmup", action="store_true",
                    help="skip the tiny assembly that loads the code objects before anything is timed (profiling runs: its dispatches would "
                         "enter the per-kernel averages)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if launched and world > 1 and os.environ.get("FENRIS_BENCH_TEST_STALL_RANK") in (str(rank), "all"):
        # test only (tests/test_bench_launch.py): this rank never arrives at the rendezvous -- the parent's --launch-timeout has to end the launch
        sys.stderr.write(f"[bench rank {rank}] FENRIS_BENCH_TEST_STALL_RANK: stalling before the first barrier\n")
        sys.stderr.flush()
        time.sleep(3600)
    cfg = args.config
    if args.operator == "poisson" and cfg == "ns":  # old spelling of the Poisson run
        cfg = "c2"
        args.cells = args.cells or 216
    if args.gpus > 1 and cfg not in ("ns", "c5", "c3", "ns-perturbed"):
        raise SystemExit(f"--config {cfg} is a single-GPU configuration")
    if args.gpus > 1 and not launched:
        self_launch(args)  # does not return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # ---- HBM traffic of the dominant kernel: two PMC passes of this command as child processes of a helper that is started HERE,
    # before this process makes any torch.cuda call, and told to run after the timed region (N = 1, rank 0 only)
    traffic, traffic_detail, helper = None, None, None
    if world == 1 and not args.no_traffic and os.environ.get("FENRIS_BENCH_CHILD") != "1":
        child = ["--config", cfg, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-traffic", "--no-secondary", "--no-module-warmup",
                 "--scatter", args.scatter]
        if args.cells:
            child += ["--cells", str(args.cells)]
        aff = ("k_affine_records", "k_affine_rows<")  # element records, then the rows: both run in every assembly
        knames = {"ns": aff, "c5": aff, "c2": aff, "ns-perturbed": ("k_hex8_rows",),
                  "c3": ("k_gather_rows_tet4",), "c4": ("k_hex27_dense_blocks", "k_rows_from_tri")}[cfg]
        try:
            helper = start_traffic_helper(child, knames)
        except OSError as exc:
            traffic_detail = {"error": repr(exc)}

    import numpy as np
    import torch
    import torch.distributed as dist

    import fenris_amd as fa
    from fenris_amd import quadrature

    if torch.cuda.device_count() == 0:
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # FENRIS_BENCH_SHARE_DEVICE=1 (validation only): all ranks on cuda:0 over gloo -- lets the N > 1 code path be
    # exercised on a single-GPU box; never used for reported numbers
    share = os.environ.get("FENRIS_BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but {torch.cuda.device_count()} GPU(s) visible")

    c = config_problem(cfg, args.cells, fa, quadrature, np)
    cells, desc, metric, s, n_el, d, uses_u = c["cells"], c["desc"], c["metric"], c["s"], c["n_el"], c["d"], c["uses_u"]
    configure = c["configure"]

    torch.cuda.set_device(local_rank)
    rccl = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        sys.stderr.write(f"[bench rank {rank}/{world}] device {local_rank}: joining the process group\n")
        sys.stderr.flush()
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            t = torch.ones(1, device="cuda")
            dist.all_reduce(t)  # the communicator exists from here on; every rank answered
            rccl = {"backend": dist.get_backend(), "rccl_ranks": int(t.item()), "world_size": dist.get_world_size()}
        sys.stderr.write(f"[bench rank {rank}/{world}] process group up: backend {dist.get_backend()}, "
                         f"rccl_ranks {rccl['rccl_ranks'] if rccl else 0} (all-reduce of ones)\n")
        sys.stderr.flush()
    stream = torch.cuda.current_stream().cuda_stream

    # code objects of the library loaded and the device warm before anything is timed: one tiny assembly
    t0 = time.perf_counter()
    if not args.no_module_warmup:
        wc = config_problem("ns", 4, fa, quadrature, np)
        weng = fa.Engine(local_rank, stream=stream)
        wc["configure"](weng, wc["mesh"]())
        wv = torch.zeros(weng.build_pattern(), dtype=torch.float64, device="cuda")
        weng.assemble_matrix_async(wv, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
        weng.poll_status()
        torch.cuda.synchronize()
        weng.close()
        del wv, weng
    t_warm = time.perf_counter() - t0

    flags = {"gather": fa.SCATTER_GATHER, "atomic": fa.SCATTER_ATOMIC, "colored": fa.SCATTER_COLORED}[args.scatter]
    slab_asm = None
    scaling = "weak"
    layers = None
    if world == 1:
        mesh = c["mesh"]()
        if cfg == "ns-perturbed":
            desc += ", every vertex moved by up to +-0.1 h per coordinate (MT19937 seed 2024): no affine element"
        eng = fa.Engine(local_rank, stream=stream)
        configure(eng, mesh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nnz = eng.build_pattern()  # assemble_pattern on the device (secondary metric)
        torch.cuda.synchronize()
        t_pattern_only = time.perf_counter() - t0   # (the allocation of the values below is not part of the pattern; on a fresh box a first
        E = mesh.num_elements()                     #  20 GB allocation has taken up to a second: BENCH_r02)
        values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
    elif cfg in ("c3", "ns-perturbed"):
        # a mesh without planes: element partition elem_to_part[] (Morton order of the centroids), packed interface-row exchange with any
        # number of neighbours (fenris_amd/partition.py); the fixed mesh cut into `world` parts: strong scaling
        from fenris_amd import partition as fp

        mesh_g = c["mesh"]()
        if cfg == "ns-perturbed":
            desc += ", every vertex moved by up to +-0.1 h per coordinate (MT19937 seed 2024): no affine element"
        part_of = fp.morton_partition(mesh_g, world)
        t0 = time.perf_counter()
        prob = fp.make_part(mesh_g, part_of, rank, world, args.partition)
        slab_asm = fp.PartAssembly(prob, configure, device=local_rank, stream=stream, exchange=("torch" if share else args.exchange))
        mesh, eng, values, nnz = prob.mesh, slab_asm.main, slab_asm.values, slab_asm.values.numel()
        E = prob.num_own_elements()
        scaling = "strong"
        part_counts = np.bincount(part_of, minlength=world).tolist()
        del mesh_g
    else:
        from fenris_amd import distributed as fd

        if cfg == "c5":
            slab = fd.make_slab(1.0, 1, 1, 1, cells, rank, world, args.partition)   # 256^3 cut into `world` z-slabs
            scaling = "strong"
            layers = [fd.slab_layers(cells, r, world) for r in range(world)]
        else:
            slab = fd.make_slab(1.0, 1, 1, world, cells, rank, world, args.partition)  # 216 x 216 x (216 world)
            layers = [fd.slab_layers(cells * world, r, world) for r in range(world)]
        mesh = slab.mesh
        t0 = time.perf_counter()  # N > 1: engines, masks and patterns of this rank
        # interface rows first, their RCCL transfer overlapped with the rest (owner-computes only)
        slab_asm = fd.SlabAssembly(slab, configure, device=local_rank, overlap=(args.scatter == "gather" and not args.no_overlap),
                                   stream=stream, exchange=("torch" if share else args.exchange),
                                   placement_tries=(args.placement_tries if args.scatter == "gather" else 0))
        eng, values, nnz = slab_asm.main, slab_asm.values, slab_asm.values.numel()
        E = slab.num_own_elements()  # numerics over own (+ halo in "halo" mode) elements, pattern over own + halo
        if rccl is not None and hasattr(slab_asm.exchange, "size"):
            rccl["fh_group_ranks"] = slab_asm.exchange.size()   # ncclCommCount of the library's own communicator
        sys.stderr.write(f"[bench rank {rank}/{world}] engines and patterns built, fh_group_ranks "
                         f"{rccl.get('fh_group_ranks') if rccl else None}: first assembly next\n")
        sys.stderr.flush()
    t_pattern = time.perf_counter() - t0
    t_values_alloc = None
    if world == 1:
        t_values_alloc = t_pattern - t_pattern_only
        t_pattern = t_pattern_only
    if slab_asm is not None and slab_asm.placement is not None:
        t_pattern -= slab_asm.placement.get("seconds", 0.0)   # the settle / placement probe inside SlabAssembly is reported on its own
    N = mesh.num_nodes()
    if args.scatter == "colored":
        eng.color()
    flags |= fa.ASSEMBLE_OVERWRITE
    placement = slab_asm.placement if slab_asm is not None else None   # N > 1: rank 0's (every rank probes its own buffers)
    settle = None
    t_first = None
    if world == 1:
        # the first assembly of the context: owner / lane tables of the pattern are built inside it (once per pattern).  What a caller of
        # the reference's one-shot `assemble` (global.rs:122-131) pays on top of pattern_build_s.
        torch.cuda.synchronize()
        t0f = time.perf_counter()
        eng.assemble_matrix_async(values, flags)
        eng.poll_status()
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t0f
    if world == 1 and not args.no_settle and os.environ.get("FENRIS_BENCH_CHILD") != "1":
        settle = settle_device(eng, values, flags)     # first: the probe below compares allocations, not a cold device with a warm one
    if world == 1 and args.placement_tries > 0 and args.scatter == "gather":
        values, placement = probe_placement(eng, values, flags, torch, args.placement_tries)

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
    per_rank_ms = None
    exchange_ms = None
    if world > 1:
        # every rank's own time per step (the line's ms_per_step is their maximum) and, after the timed region, what the interface exchange
        # costs when nothing hides it: a scaling run can then tell load imbalance from communication
        mine = torch.tensor([1e3 * elapsed / args.steps], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [float(x.item()) for x in allr]
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ex = getattr(slab_asm, "exchange", None)
        if ex is not None and hasattr(ex, "run") and args.partition == "exchange":
            try:
                torch.cuda.synchronize()
                dist.barrier()
                ex.run()                       # (untimed: buffers, communicator warm)
                torch.cuda.synchronize()
                dist.barrier()
                t0x = time.perf_counter()
                for _ in range(3):
                    ex.run()
                torch.cuda.synchronize()
                exchange_ms = 1e3 * (time.perf_counter() - t0x) / 3
                # (the rows now hold four extra copies of the neighbours' contributions: restore them with one more assembly)
                step()
                torch.cuda.synchronize()
                dist.barrier()
            except Exception as exc:   # never take the line down
                exchange_ms = None
                print(f"[bench] exchange timing failed: {exc!r}", file=sys.stderr)
        e_t = torch.tensor([float(E)], dtype=torch.float64, device="cuda")
        dist.all_reduce(e_t, op=dist.ReduceOp.SUM)
        total_elements = float(e_t.item())
    else:
        total_elements = float(E)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_elements * args.steps / elapsed
        abytes = algorithmic_bytes(E, N, s, n_el, d, nnz, uses_u)
        achieved = abytes / (kernel_avg_ms * 1e-3) / 1e9
        if world > 1 and cfg == "ns":
            desc = desc.replace(f"{cells}^3", f"{cells}x{cells}x{cells * world}")
        out = {
            "metric": metric, "value": value, "unit": "elements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{cfg}: {desc} ({int(total_elements)} elements), "
                                   + ("YoungPoisson(1e6, 0.2), " if c["params"] is not None else "")
                                   + ("u = 0, " if not uses_u else "") + "CSR pattern pre-built, values overwritten",
                       "name": cfg, "elements_per_gpu": E, "nodes_per_gpu": N, "nnz_per_gpu": nnz, "scatter": args.scatter,
                       "partition": "single" if world == 1 else (f"{world} z-slabs, interface rows exchanged" if args.partition == "exchange"
                                                                   else f"{world} z-slabs, halo element layer recomputed, no communication"),
                       "pattern_build_s": t_pattern, "first_assembly_s": t_first, "values_alloc_s": t_values_alloc, "module_warmup_s": t_warm,
                       "placement_probe": placement, "device_settle": settle,
                       # the placement a caller of fh_assemble_matrix_dev gets without shopping for allocations (the first one of this process,
                       # after the device settled): time and roofline fraction next to the probed figures of the line
                       "first_placement_ms": (placement or {}).get("values_ms_seen", [None])[0] if world == 1 else None,
                       "first_placement_frac": ((abytes / ((placement or {}).get("values_ms_seen", [None])[0] * 1e-3) / 1e9 / PEAK_HBM_GBS)
                                                if world == 1 and (placement or {}).get("values_ms_seen") else None)},
        }
        if world > 1:
            out["config"]["per_rank_ms_per_step"] = [round(x, 4) for x in per_rank_ms] if per_rank_ms else None
            out["config"]["exchange_ms"] = round(exchange_ms, 4) if exchange_ms is not None else None
            out["config"]["exchange_ms_note"] = ("the interface exchange alone, three times back to back after the timed region (host clock, rank 0); inside a "
                                                 "step it runs beside the main launch")
        if world > 1 and layers is None:
            # general partition: who holds what, what rank 0 exchanges, and the same check as for the slabs -- the rows of rank 0's owned
            # nodes that other parts contribute to are complete only after the exchange, and a stiffness row sums to zero
            ex = slab_asm.exchange
            out["config"]["partition"] = (f"{world} parts: Morton order of the element centroids cut into equal runs, node owner = lowest part, "
                                          + ("interface rows exchanged through packed index lists" if args.partition == "exchange"
                                             else "halo elements recomputed, no communication"))
            out["config"]["elements_per_rank"] = part_counts
            out["config"]["exchange"] = ("torch.distributed point-to-point (batch_isend_irecv)" if (share or args.exchange == "torch")
                                         else "fh_group_set_exchange_nodes (pack kernel, one RCCL group of ncclSend / ncclRecv, unpack-add kernel; C ABI)")
            out["config"]["neighbours_of_rank0"] = sorted(set(slab_asm.prob.send) | set(slab_asm.prob.recv))
            out["config"]["interface_bytes_per_step"] = ex.bytes_sent()
            try:
                ro_h, _ = eng.pattern(want_cols=False)
                nodes = np.unique(np.concatenate([np.asarray(v) for v in slab_asm.prob.recv.values()])) if slab_asm.prob.recv else np.zeros(0, dtype=np.int64)
                if len(nodes):
                    rows = (s * nodes[:, None] + np.arange(s)[None, :]).reshape(-1)
                    a0 = torch.as_tensor(np.asarray(ro_h)[rows].astype(np.int64), device=values.device)
                    a1 = torch.as_tensor(np.asarray(ro_h)[rows + 1].astype(np.int64), device=values.device)
                    cs = torch.cat([torch.zeros(1, dtype=values.dtype, device=values.device), torch.cumsum(values, 0)])
                    rowsum = (cs[a1] - cs[a0]).abs().max().item()
                    out["config"]["interface_row_sum_over_max"] = rowsum / max(values.abs().max().item(), 1e-300)
            except Exception as exc:  # never take the line down
                out["config"]["interface_row_sum_over_max"] = repr(exc)
            out["rccl"] = rccl if rccl is not None else {"backend": "gloo (FENRIS_BENCH_SHARE_DEVICE validation mode)", "rccl_ranks": 0}
            out["rccl_ranks"] = out["rccl"]["rccl_ranks"]
        elif world > 1:
            out["config"]["element_layers_per_rank"] = [l1 - l0 for l0, l1 in layers]
            out["config"]["exchange"] = "torch.distributed point-to-point" if (share or args.exchange == "torch") else "fh_group_* (RCCL behind the C ABI)"
            ex = slab_asm.exchange
            rb = getattr(ex, "recv_buf", None)
            if rb is not None:   # what rank 0 receives from rank 1 per step (every interior interface carries the same amount)
                out["config"]["interface_bytes_per_step"] = int(rb.numel() * rb.element_size())
                out["config"]["interface_rows_packed"] = getattr(ex, "recv_idx", None) is not None
            # a check the line carries with it (the timed steps are done): the rows of rank 0's owned interface plane are complete only
            # after rank 1's contribution has arrived and been added, and a stiffness row sums to zero (constants / translations lie in
            # the kernel of K) -- a wrong or missing exchange, or a wrong masked assembly, shows here
            try:
                seg = getattr(ex, "recv_seg", None)
                if seg is not None:
                    ro_h, _ = eng.pattern(want_cols=False)
                    n0, n1 = slab.recv_nodes
                    ro_seg = torch.as_tensor(np.asarray(ro_h[s * n0:s * n1 + 1]).astype(np.int64) - int(ro_h[s * n0]), device=values.device)
                    vseg = values[seg[0]:seg[1]]
                    cs = torch.cat([torch.zeros(1, dtype=vseg.dtype, device=vseg.device), torch.cumsum(vseg, 0)])
                    rowsum = (cs[ro_seg[1:]] - cs[ro_seg[:-1]]).abs().max().item()
                    out["config"]["interface_row_sum_over_max"] = rowsum / max(vseg.abs().max().item(), 1e-300)
            except Exception as exc:  # never take the line down
                out["config"]["interface_row_sum_over_max"] = repr(exc)
            out["rccl"] = rccl if rccl is not None else {"backend": "gloo (FENRIS_BENCH_SHARE_DEVICE validation mode)", "rccl_ranks": 0}
            out["rccl_ranks"] = out["rccl"]["rccl_ranks"]
        hbm = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "frac": achieved / PEAK_HBM_GBS, "traffic": traffic,
               "kernel": eng.last_kernel_name(), "kernel_avg_ms": kernel_avg_ms,
               "kernel_min_ms": kernel_ms[0], "algorithmic_bytes_per_launch": abytes,
               "bytes_per_element": abytes / E}
        if cfg == "c4":
            # compute-bound configuration: the two weighted Gram products of the dense element matrix,
            # 2 (3n)^2 nq 2 flop, plus the trace term 6 n^2 nq flop, on the fp64 matrix cores; the HBM figures go along
            _, tf, _, _, fr = roofline_of(cfg, c, E, N, nnz, kernel_avg_ms)
            flops = tf * 1e12 * kernel_avg_ms * 1e-3
            out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": fr,
                               "traffic": traffic, "kernel": eng.last_kernel_name(), "kernel_avg_ms": kernel_avg_ms,
                               "kernel_min_ms": kernel_ms[0], "algorithmic_flops_per_launch": flops, "flops_per_element": flops / E,
                               "note": "kernel_avg_ms covers both passes of the two-pass assembly (dense element matrices on the "
                                       "matrix cores, then the row gather)", "hbm": hbm}
        else:
            out["roofline"] = hbm
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
            hbm["measured_write_GBps"] = 5 * nnz * 8 / (a0.elapsed_time(b0) * 1e-3) / 1e9
            del scratch
        except RuntimeError:
            pass
        # ---- the other configurations, same process, after the headline (driver-verifiable figures for the rows of DESIGN 3.4)
        if world == 1 and cfg == "ns" and not args.cells and not args.no_secondary and os.environ.get("FENRIS_BENCH_CHILD") != "1":
            # residual vector and energy of the same mesh and operator ("Secondary: ... residual assembly"), the same context
            try:
                eng.set_u(1e-3 * np.sin(np.arange(s * mesh.num_nodes())))
                fvec = torch.zeros(s * mesh.num_nodes(), dtype=torch.float64, device="cuda")
                for _ in range(3):
                    eng.assemble_vector(fvec)
                torch.cuda.synchronize()
                a1, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a1.record()
                for _ in range(10):
                    eng.assemble_vector(fvec)
                b1.record()
                torch.cuda.synchronize()
                vec_kernel = eng.last_kernel_name()
                eng.assemble_scalar()      # untimed: the first call allocates its partial sums (round 5's first line timed it: 1.7 instead of 0.56 ms)
                t_e = time.perf_counter()
                for _ in range(5):
                    eng.assemble_scalar()
                t_e = (time.perf_counter() - t_e) / 5
                out["secondary_vectors"] = {"workload": "residual vector and energy of the headline mesh and operator, u = 1e-3 sin(i)",
                                            "residual_ms": a1.elapsed_time(b1) / 10, "residual_kernel": vec_kernel,
                                            "energy_ms_blocking_call": 1e3 * t_e, "energy_kernel": eng.last_kernel_name()}
                del fvec
            except Exception as exc:  # never take the headline down
                out["secondary_vectors"] = {"error": repr(exc)}
            eng.close()
            del values
            torch.cuda.empty_cache()
            sec = {}
            for name in ("c5", "ns-perturbed", "c4", "c3", "c2"):   # the short ones last: the driver is still busy with the 20 GB just freed
                time.sleep(0.3)
                t0 = time.perf_counter()
                try:
                    # c5 is BASELINE's largest configuration: the same set-up as the headline (settle, six more allocations tried -- on one box a standalone run found 7.83 ms on its fourth allocation after 8.10, 8.80, 8.34); the others: two
                    sec[name] = time_secondary(name, fa, quadrature, np, torch, stream, tries=6 if name == "c5" else 2, settle=(name == "c5"))
                    sec[name]["seconds_total"] = time.perf_counter() - t0
                except Exception as exc:  # a secondary line must never take the headline down
                    sec[name] = {"error": repr(exc)}
            out["secondary"] = sec
        # ---- now the PMC passes (the helper started at the top runs them; this process only waits, its GPU work is done)
        if helper is not None:
            try:
                eng.close()
            except Exception:
                pass
            values = None
            torch.cuda.empty_cache()
            traffic, traffic_detail = finish_traffic_helper(helper, True)
        hbm["traffic"] = traffic
        if "hbm" in out["roofline"]:
            out["roofline"]["traffic"] = traffic
        if traffic_detail is not None:
            hbm["traffic_source"] = ("two rocprofv3 --pmc child passes of this command, run after the timed region (FETCH_SIZE x 2 + "
                                     "WRITE_SIZE, KiB -> bytes, averaged over the dispatches of the dominant kernel)")
            hbm["traffic_detail"] = traffic_detail
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(c["cpu_kind"])
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        if slab_asm is not None:
            slab_asm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
