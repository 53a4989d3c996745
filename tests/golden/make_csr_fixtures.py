#!/usr/bin/env python3
"""Write the assembled-matrix fixtures tests/golden/csr/*.npz with the oracle (run after the oracle passes
tests/test_oracle_kat.py):   python tests/golden/make_csr_fixtures.py

Each file: offsets (u64), indices (u64), values (f64, NaN where the reference produces NaN) of K = CsrAssembler::assemble
(global.rs:122-182) for the case of the same name in csr_cases.py.  Data only; nothing of the reference is stored."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import csr_cases  # noqa: E402
from oracle import oracle as o  # noqa: E402


def main():
    os.makedirs(csr_cases.CSR_DIR, exist_ok=True)
    for name in csr_cases.cases(o):
        asm, _ = csr_cases.oracle_assembler(o, name)
        st, failed, ro, ci, vals = o.assemble(asm)
        assert st == 0, (name, st, failed)
        path = os.path.join(csr_cases.CSR_DIR, name + ".npz")
        np.savez_compressed(path, offsets=ro.astype(np.uint64), indices=ci.astype(np.uint64), values=vals)
        print(f"{name}: rows {len(ro) - 1} nnz {len(vals)} NaN {int(np.isnan(vals).sum())} -> {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
