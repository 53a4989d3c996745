# A/B of two BUILDS of the library on one box, alternating processes (old, new, old, new): put the other build at scripts/bin/libfenris_hip_old.so
#   git worktree add /tmp/wt <commit> && make -C /tmp/wt/fenris_amd/csrc && cp /tmp/wt/fenris_amd/lib/libfenris_hip.so scripts/bin/libfenris_hip_old.so
# (scripts/bin is git-ignored but travels with gpurun).  profiles/r03_affine_experiments.txt 15.
cd $GRAFT_REPO_ROOT
cp fenris_amd/lib/libfenris_hip.so /tmp/new.so
for i in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp scripts/bin/libfenris_hip_old.so fenris_amd/lib/libfenris_hip.so; else cp /tmp/new.so fenris_amd/lib/libfenris_hip.so; fi
    echo "$v c2 $(timeout 300 python scripts/ab_in_context.py --config c2 --reps 20 "base:" 2>&1 | grep variant | cut -c1-90)"
    echo "$v ns $(timeout 300 python scripts/ab_in_context.py --config ns --reps 5 "base:" 2>&1 | grep variant | cut -c1-90)"
  done
done
cp /tmp/new.so fenris_amd/lib/libfenris_hip.so
