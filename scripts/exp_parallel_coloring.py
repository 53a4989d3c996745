#!/usr/bin/env python3
"""fh_color_parallel against fh_color on the north-star mesh (Hex8 216^3) and on C3 (Tet4 BCC res 75): colours, rounds (FENRIS_HIP_VERBOSE),
seconds, and the coloured scatter's assembly time driven by each.   python scripts/exp_parallel_coloring.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

for cfg in ("ns", "c3"):
    c = bench.config_problem(cfg, 0, fa, quadrature, np)
    mesh = c["mesh"]()
    eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
    c["configure"](eng, mesh)
    nnz = eng.build_pattern()
    values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    out = {"config": cfg, "elements": mesh.num_elements()}
    for name, fn in (("parallel", eng.color_parallel), ("sequential_host", eng.color)):
        t0 = time.perf_counter()
        colors = fn()
        torch.cuda.synchronize()
        out[name] = {"colors": len(colors), "seconds": round(time.perf_counter() - t0, 3)}
        flags = fa.SCATTER_COLORED | fa.ASSEMBLE_OVERWRITE
        out[name]["colored_assembly_ms"] = round(eng.time_assembly(values, flags, 3), 3)
    print(json.dumps(out), flush=True)
    eng.close()
    del values
    torch.cuda.empty_cache()
