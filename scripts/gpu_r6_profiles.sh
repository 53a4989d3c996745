#!/bin/bash
# round 6: the default bench line and the rocprofv3 summaries of every configuration (copied to profiles/r06_*)
mkdir -p gpurun_out/r6p
python bench.py > gpurun_out/r6p/bench_default.json 2> gpurun_out/r6p/bench_default.err; echo "bench rc=$?"
for cfg in ns ns-perturbed c4 c2 c3 c5; do
  bash scripts/gpu_profile_config.sh $cfg > gpurun_out/r6p/prof_$cfg.log 2>&1
  cp gpurun_out/prof_$cfg/summary.txt gpurun_out/r6p/${cfg}_rocprofv3_summary.txt
done
python scripts/bench_other_kernels.py > gpurun_out/r6p/other_kernels.jsonl 2> gpurun_out/r6p/other.err
python scripts/check_full_size.py > gpurun_out/r6p/full_size_check.txt 2>&1; tail -3 gpurun_out/r6p/full_size_check.txt
