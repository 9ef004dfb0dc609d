#!/bin/bash
# the wide-tile candidates (and the best 4-wave one) on the K-heavy bf16 shapes at B=256
out=${1:-gpurun_out/bf16_wide.txt}
: > $out
run() { echo "== $*" >> $out; for c in 5 9 10 11 12 13; do python tools/conv_bench.py "$@" --dtype bf16 --relu --reps 30 --cand $c 2>&1 | grep -v amdgpu.ids >> $out; done; }
run 256 28 28 128 128 3 1 1
run 256 14 14 256 256 3 1 1
run 256 7 7 512 512 3 1 1
run 256 14 14 1024 256 1 1 0
run 256 14 14 256 1024 1 1 0 --residual
run 256 7 7 2048 512 1 1 0
run 256 28 28 512 128 1 1 0
run 256 28 28 128 512 1 1 0 --residual
