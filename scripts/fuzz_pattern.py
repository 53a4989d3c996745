#!/usr/bin/env python3
"""Randomised check of assemble_pattern (global.rs:65-120; index arrays must be bit-exact): random RAGGED connectivities (elements of 0 .. 40
nodes, repeated nodes inside an element, isolated nodes, hubs shared by hundreds of elements), solution dimensions 1 - 3, against the pattern
of A^T A computed by scipy -- and the colouring (host and device) valid on the same input.
    python scripts/fuzz_pattern.py [cases] [seed]"""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fenris_amd as fa  # noqa: E402


def run(cases=300, seed0=0, quiet=False):
    bad = 0
    eng = fa.Engine(0)
    for it in range(cases):
        rng = np.random.default_rng(seed0 + it)
        n = int(rng.integers(1, 400))
        ne = int(rng.integers(0, 300))
        s = int(rng.integers(1, 4))
        style = rng.choice(["uniform", "hubs", "local"])
        elems = []
        for _ in range(ne):
            k = int(rng.integers(0, 9)) if rng.random() < 0.9 else int(rng.integers(9, 41))
            if style == "uniform":
                e = rng.integers(0, n, k)
            elif style == "hubs":
                e = np.where(rng.random(k) < 0.3, rng.integers(0, min(n, 3), k), rng.integers(0, n, k))
            else:
                c0 = int(rng.integers(0, n))
                e = np.clip(c0 + rng.integers(-6, 7, k), 0, n - 1)
            elems.append([int(x) for x in e])
        if not quiet:
            print(f"case {seed0 + it}: N={n} E={ne} s={s} {style}", flush=True)
        mock = fa.MockElementAssembler(s, n, elems, eng)
        ro, ci = fa.CsrAssembler().assemble_pattern(mock)
        rows, cols = [], []
        for j, e in enumerate(elems):
            rows += [j] * len(e)
            cols += e
        inc = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(max(ne, 1), n))
        pat = (inc.T @ inc).tocsr()
        pat.data[:] = 1.0
        full = sp.kron(pat, np.ones((s, s))).tocsr()
        full.sort_indices()
        ok = np.array_equal(np.asarray(ro, dtype=np.int64), full.indptr.astype(np.int64)) and np.array_equal(np.asarray(ci, dtype=np.int64), full.indices.astype(np.int64))
        # colourings of the same connectivity: valid, complete
        for colors in (mock.engine.color(),):
            lab = np.asarray(colors.labels).astype(np.int64)
            okc = sorted(lab.tolist()) == list(range(ne))
            for c in range(len(colors)):
                seen = set()
                for e in colors.color(c):
                    nodes = set(elems[int(e)])
                    if seen & nodes:
                        okc = False
                    seen |= nodes
            ok = ok and okc
        if not ok:
            bad += 1
            print(f"   MISMATCH case {seed0 + it}: N={n} E={ne} s={s} {style}", flush=True)
    print("cases", cases, "mismatches", bad)
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
