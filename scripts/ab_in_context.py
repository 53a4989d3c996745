#!/usr/bin/env python3
"""A/B timing of launch variants INSIDE one context, on the same buffers (fh_set_option): the spread is ~0.5 %, against up to 10 % between
contexts / allocations.    python scripts/ab_in_context.py --config ns "base:" "depth1:FENRIS_HIP_AFFINE_DEPTH=1" ..."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import fenris_amd as fa  # noqa: E402
from fenris_amd import quadrature  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="ns")
    ap.add_argument("--cells", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    c = bench.config_problem(args.config, args.cells, fa, quadrature, np)
    mesh = c["mesh"]()
    eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
    c["configure"](eng, mesh)
    nnz = eng.build_pattern()
    values = torch.zeros(nnz, dtype=torch.float64, device="cuda")
    flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
    values, rep = bench.probe_placement(eng, values, flags, torch, 3)
    print(json.dumps({"placement": rep}), flush=True)
    specs = []
    for spec in args.variants:
        label, _, envs = spec.partition(":")
        specs.append((label, [kv.partition("=")[::2] for kv in filter(None, envs.split(","))]))
    times = {label: [] for label, _ in specs}
    for _ in range(args.rounds):
        for label, kvs in specs:
            for k, v in kvs:
                eng.set_option(k, v)
            times[label].append(eng.time_assembly(values, flags, args.reps))
            kern = eng.last_kernel_name()
            for k, _ in kvs:
                eng.set_option(k, None)
    for label, _ in specs:
        t = sorted(times[label])
        print(json.dumps({"variant": label, "ms_median": round(t[len(t) // 2], 4), "ms_min": round(t[0], 4), "ms_max": round(t[-1], 4)}), flush=True)


if __name__ == "__main__":
    main()
