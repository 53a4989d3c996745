#!/usr/bin/env python3
"""How long a blocking fh_assemble_scalar / fh_assemble_vector call of the headline mesh takes from Python, call by call (the bench line's
`energy_ms_blocking_call` was 0.62 ms in round 4 and 1.7 ms in one round-5 run).    python scripts/probe_energy_call.py [cells]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else None
c = bench.config_problem("ns", cells, fa, quadrature, np)
mesh = c["mesh"]()
eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
c["configure"](eng, mesh)
eng.set_u(1e-3 * np.sin(np.arange(3 * mesh.num_nodes())))
eng.build_pattern()
out = torch.zeros(3 * mesh.num_nodes(), dtype=torch.float64, device="cuda")
for name, call in (("scalar", lambda: eng.assemble_scalar()), ("vector", lambda: eng.assemble_vector(out))):
    ts = []
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(name, eng.last_kernel_name(), " ".join("%.3f" % t for t in ts), flush=True)
