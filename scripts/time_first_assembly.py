#!/usr/bin/env python3
"""First assembly of a context (owner tables, lane tables, ... built once per pattern) against the steady state, for any configuration of
bench.py.  FENRIS_HIP_VERBOSE=1 prints the wall time of the set-up's stages; under rocprofv3 --kernel-trace --stats: its device part by
kernel.    python scripts/time_first_assembly.py [ns|ns-perturbed|c2|c3|c4|c5] [cells]"""
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

cfg = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "ns"
if "perturbed" in sys.argv[1:]:
    cfg = "ns-perturbed"
cells = next((int(a) for a in sys.argv[1:] if a.isdigit()), None)
c = bench.config_problem(cfg, cells, fa, quadrature, np)
mesh = c["mesh"]()
for it in range(2):
    eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
    c["configure"](eng, mesh)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nnz = eng.build_pattern()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    eng.assemble_matrix(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    eng.assemble_matrix(values, fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("%s context %d: %d elements, pattern %.1f ms, values alloc %.1f ms, first assembly %.1f ms (%s), second %.2f ms" %
          (cfg, it, mesh.num_elements(), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), eng.last_kernel_name(), 1e3 * (t4 - t3)), flush=True)
    del values
    eng.close()
    if os.environ.get("FENRIS_SLEEP_BETWEEN"):
        time.sleep(float(os.environ["FENRIS_SLEEP_BETWEEN"]))
