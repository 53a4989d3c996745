# round 4, final state of the library: the three randomised cross-checks at many more cases than the 200 each that run under pytest
mkdir -p gpurun_out/r4
{
  echo "== fuzz_gather (owner-computes kernels, forced small grids, masks) =="; timeout 900 python scripts/fuzz_gather.py 1500 50000 2>&1 | tail -3
  echo "== fuzz_vector (tiled residual / energy / source against the staged kernels) =="; timeout 600 python scripts/fuzz_vector.py 1500 70000 2>&1 | tail -3
  echo "== fuzz_pattern (assemble_pattern against scipy) =="; timeout 600 python scripts/fuzz_pattern.py 1500 90000 2>&1 | tail -3
} > gpurun_out/r4/fuzz.txt 2>&1
cat gpurun_out/r4/fuzz.txt
