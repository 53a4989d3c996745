#!/usr/bin/env python3
"""How long do first allocations of device memory take in a fresh process (hipMalloc through torch, cache emptied after each)?"""
import time

import torch

torch.cuda.init()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
for gb in (0.25, 0.5, 1.0, 2.6, 2.6, 5.0, 19.7, 19.7, 2.6, 40.0, 40.0):
    n = int(gb * (1 << 30) / 8)
    t0 = time.perf_counter()
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    x.fill_(1.0)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    del x
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("%5.2f GB: alloc %8.2f ms, first fill %7.2f ms, free %7.2f ms" % (gb, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)), flush=True)
