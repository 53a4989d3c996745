#!/bin/bash
# Resource usage + instruction histogram of one kernel of libfenris_hip.so (no GPU needed).
#   [OBJ=unit.o] scripts/kernel_info.sh <mangled-name-substring> [instruction-regex]
# e.g. scripts/kernel_info.sh k_gather_pipelinedILi1ELi1ELi8ELi2ELb0ELb1E 'ds_|global_|s_waitcnt|s_barrier'
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
# (the library holds one bundle per translation unit since the engine was split: OBJ=<file>.o names the unit the kernel lives in)
SRC="${OBJ:+$ROOT/fenris_amd/csrc/$OBJ}"; SRC="${SRC:-$ROOT/fenris_amd/lib/libfenris_hip.so}"
objcopy -O binary --only-section=.hip_fatbin "$SRC" "$TMP/fat.bin"
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$TMP/fat.bin" --output="$TMP/dev.co" --unbundle
SYM=$($LLVM/llvm-readelf --notes "$TMP/dev.co" | grep '\.name:' | awk '{print $2}' | grep "$1" | head -1)
echo "kernel: $SYM"
$LLVM/llvm-readelf --notes "$TMP/dev.co" | awk -v s="$SYM" '
  /^  - \.agpr_count|^  - \.args/ {blk=""} {blk=blk"\n"$0}
  /\.name:/ && index($0, s) {hit=1}
  /\.wavefront_size/ {if (hit) {print blk; exit} blk=""}' | grep -E "agpr_count|vgpr_count|sgpr_count|group_segment_fixed|spill|private_segment_fixed" || true
$LLVM/llvm-objdump -d --disassemble-symbols="$SYM" "$TMP/dev.co" > "$TMP/k.s"
echo "instructions: $(grep -c '^\s' "$TMP/k.s")"
grep -E "${2:-ds_|global_|buffer_|s_waitcnt|s_barrier|v_fma_f64|scratch_}" "$TMP/k.s" | awk '{print $1}' | sort | uniq -c | sort -rn | head -40
[ -n "$KEEP" ] && cp "$TMP/k.s" "$KEEP"
rm -rf "$TMP"
