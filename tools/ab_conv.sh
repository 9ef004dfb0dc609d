#!/bin/bash
# A/B of two builds of the library (.ab/librn_hip_head.so against the tree's) over the network's convolution shapes:
# the 64x64 tile candidates (plain and persistent) of tools/conv_bench.py, fp32
for shape in "256 56 56 64 64 3 1 1 --relu" "256 28 28 128 512 1 1 0 --residual --relu" "256 28 28 512 128 1 1 0 --relu" "256 28 28 128 128 3 1 1 --relu" "256 14 14 256 1024 1 1 0 --residual --relu" "256 14 14 1024 256 1 1 0 --relu" "256 14 14 256 256 3 1 1 --relu" "256 7 7 512 512 3 1 1 --relu" "256 7 7 512 2048 1 1 0 --residual --relu" "256 7 7 2048 512 1 1 0 --relu"; do
  echo "== $shape"
  for which in head tree head tree; do
    if [ $which = head ]; then export RN_HIP_LIB=$PWD/.ab/librn_hip_head.so; else unset RN_HIP_LIB; fi
    echo "$which $(python tools/conv_bench.py $shape --reps 20 2>&1 | grep -v amdgpu.ids | grep -E '^(64x64|P64x64) ' | awk '{printf "%s %s us  ", $1, $2}')"
  done
done
