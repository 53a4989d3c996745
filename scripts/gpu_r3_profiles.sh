#!/bin/bash
# round 3: the default bench line and the rocprofv3 summaries of every configuration (copied to profiles/r03_*)
mkdir -p gpurun_out/r3p
python bench.py > gpurun_out/r3p/bench_default.json 2> gpurun_out/r3p/bench_default.err; echo "bench rc=$?"
for cfg in ns c2 c3 c4 c5 ns-perturbed; do
  bash scripts/gpu_profile_config.sh $cfg > gpurun_out/r3p/prof_$cfg.log 2>&1
  cp gpurun_out/prof_$cfg/summary.txt gpurun_out/r3p/${cfg}_rocprofv3_summary.txt
  python bench.py --config $cfg --no-secondary > gpurun_out/r3p/bench_${cfg}_n1.json 2> gpurun_out/r3p/bench_${cfg}.err
done
python scripts/bench_other_kernels.py > gpurun_out/r3p/other_kernels.jsonl 2> gpurun_out/r3p/other.err
python scripts/check_full_size.py > gpurun_out/r3p/full_size_check.txt 2>&1; tail -3 gpurun_out/r3p/full_size_check.txt
