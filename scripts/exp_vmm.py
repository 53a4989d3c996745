#!/usr/bin/env python3
"""Round 5: does the physical chunk size of the `values` allocation choose the level the headline kernel runs at?
One fresh process per call: builds the configuration, allocates `values` as asked, times assemblies (fh_time_assembly_dev).
    python scripts/exp_vmm.py --config ns --alloc torch | vmm:<chunk MiB, 0 = one chunk>"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fenris_amd as fa  # noqa: E402
from fenris_amd import _ffi, quadrature  # noqa: E402

import bench  # noqa: E402


class RawValues:
    def __init__(self, ptr):
        self._p = ptr

    def data_ptr(self):
        return self._p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="ns")
    ap.add_argument("--alloc", default="torch")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    c = bench.config_problem(args.config, 0, fa, quadrature, np)
    eng = fa.Engine(0, stream=torch.cuda.current_stream().cuda_stream)
    c["configure"](eng, c["mesh"]())
    nnz = eng.build_pattern()
    flags = fa.SCATTER_GATHER | fa.ASSEMBLE_OVERWRITE
    lib = _ffi.lib()
    gran = C.c_uint64(0)
    if args.alloc == "torch":
        keep = torch.zeros(nnz, dtype=torch.float64, device="cuda")
        values = keep
    else:
        chunk_mib = int(args.alloc.split(":")[1])
        p = C.c_void_p()
        rc = lib.fh_vmm_alloc(0, C.c_uint64(8 * nnz), C.c_uint64(chunk_mib << 20), C.byref(p), C.byref(gran))
        assert rc == 0 and p.value, rc
        values = RawValues(p.value)
    times = [eng.time_assembly(values, flags, args.reps) for _ in range(3)]
    print(json.dumps({"config": args.config, "alloc": args.alloc, "granularity": int(gran.value), "ms": [round(t, 4) for t in times]}), flush=True)


if __name__ == "__main__":
    main()
