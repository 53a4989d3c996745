#!/usr/bin/env python3
"""Residual vector of any bench configuration's mesh / operator (u = small sine field), tiled path against the round-3 two-pass path.
    python scripts/time_residual_cfg.py [ns|c2|c3|ns-perturbed] [cells]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
cells = int(sys.argv[2]) if len(sys.argv) > 2 else None
c = bench.config_problem(cfg, cells, fa, quadrature, np)
mesh = c["mesh"]()
for tiles in (True, False):
    eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
    c["configure"](eng, mesh)
    s = eng.solution_dim()
    eng.set_u(1e-3 * np.sin(np.arange(s * mesh.num_nodes())))
    if not tiles:
        eng.set_option("FENRIS_HIP_NO_VECTOR_TILES", "1")
    out = torch.zeros(s * mesh.num_nodes(), dtype=torch.float64, device="cuda")
    for _ in range(3):
        eng.assemble_vector(out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        eng.assemble_vector(out)
    b.record()
    torch.cuda.synchronize()
    print("%s residual, %d elements: %s  %.4f ms" % (cfg, mesh.num_elements(), eng.last_kernel_name(), a.elapsed_time(b) / 10), flush=True)
    eng.close()
