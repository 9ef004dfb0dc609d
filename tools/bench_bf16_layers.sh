#!/bin/bash
# every tile candidate on the bf16 ResNet-50 layer shapes at B=256 (tools/conv_bench.py)
set -e
out=${1:-gpurun_out/bf16_layers.txt}
: > $out
run() { echo "== $*" >> $out; python tools/conv_bench.py "$@" --dtype bf16 --relu --reps 20 >> $out 2>&1; }
run 256 56 56 64 64 3 1 1
run 256 28 28 128 128 3 1 1
run 256 14 14 256 256 3 1 1
run 256 7 7 512 512 3 1 1
run 256 56 56 128 128 3 2 1
run 256 14 14 1024 256 1 1 0
run 256 14 14 256 1024 1 1 0 --residual
run 256 7 7 2048 512 1 1 0
run 256 7 7 512 2048 1 1 0 --residual
run 256 28 28 512 128 1 1 0
run 256 28 28 128 512 1 1 0 --residual
run 256 56 56 256 64 1 1 0
run 256 56 56 64 256 1 1 0 --residual
