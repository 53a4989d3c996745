#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (gcc; the GPU pool refuses sanitizer runs: CPU build only): builds a sanitized
# libfenris_oracle.so in place, runs the CPU tests that drive the oracle, restores the regular build.
set -e
cd "$(dirname "$0")/.."
cp oracle/libfenris_oracle.so /tmp/libfenris_oracle.keep.so
trap 'cp /tmp/libfenris_oracle.keep.so oracle/libfenris_oracle.so' EXIT
gcc -O1 -g -march=x86-64-v3 -ffp-contract=off -fno-fast-math -fPIC -fopenmp -std=c11 -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o oracle/libfenris_oracle.so oracle/fenris_oracle.c -lm
nm -D oracle/libfenris_oracle.so | grep -c "__asan_\|__ubsan_" | sed 's/^/sanitizer symbols referenced: /'
G=/usr/lib/gcc/x86_64-linux-gnu/11
LD_PRELOAD="$G/libasan.so $G/libubsan.so" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4 \
    python -m pytest tests -x -q -m "not gpu" -k "oracle or kat or independent or golden or tensor or restatement or patch" -p no:cacheprovider 2>&1 | tail -4
