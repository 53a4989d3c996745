#!/bin/bash
# five fresh processes per way of allocating `values` (headline configuration, no settle, no probe)
for rep in 1 2 3 4 5; do
  for a in torch vmm:0 vmm:2 vmm:64 vmm:1024 vmm:4096; do
    python scripts/exp_vmm.py --config ${1:-ns} --alloc $a 2>/dev/null | tail -1
  done
done
