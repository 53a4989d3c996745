# full-size headline bench + rocprofv3 evidence (run on the GPU box from the repo root)
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 3000 gpurun_out/bench_full.json
bash scripts/gpu_profile.sh 216 > gpurun_out/profile_full.log 2>&1
tail -40 gpurun_out/prof/summary.txt
