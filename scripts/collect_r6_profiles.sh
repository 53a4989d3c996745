#!/bin/bash
# copies what scripts/gpu_r6_final.sh left under gpurun_out/ into profiles/r06_* and regenerates the table of DESIGN.md 3.4 from it
set -e
cd "$(dirname "$0")/.."
cp gpurun_out/r6p/bench_default.json profiles/r06_bench_default_n1.json
for c in ns ns-perturbed c2 c3 c4 c5; do cp gpurun_out/r6p/${c}_rocprofv3_summary.txt profiles/r06_${c}_rocprofv3_summary.txt; done
cp gpurun_out/r6p/other_kernels.jsonl profiles/r06_other_kernels.jsonl
cp gpurun_out/r6p/full_size_check.txt profiles/r06_full_size_check.txt
(echo "round 6: pytest -m gpu, smoke(), fuzzers on one MI355X (scripts/gpu_r6_final.sh)"; echo "== pytest -m gpu"; cat gpurun_out/r6f/pytest_gpu.txt; echo "== smoke()"; tail -5 gpurun_out/r6f/smoke.txt
 echo "== fuzz_gather 1500 / fuzz_vector 600 / fuzz_pattern 300"; tail -1 gpurun_out/r6f/fuzz_gather.txt; tail -1 gpurun_out/r6f/fuzz_vector.txt; tail -1 gpurun_out/r6f/fuzz_pattern.txt) > profiles/r06_fuzz_and_suite.txt
python3 scripts/gen_design_table.py --round 06
