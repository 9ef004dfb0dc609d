#!/bin/bash
# A/B of two builds of the library on the bench line: .ab/librn_hip_head.so against the tree's, alternating.
#   bash tools/ab_bench.sh [rounds] [bench.py flags ...]
n=${1:-3}; shift
for i in $(seq $n); do
  for which in head tree; do
    if [ $which = head ]; then export RN_HIP_LIB=$PWD/.ab/librn_hip_head.so; else unset RN_HIP_LIB; fi
    python bench.py --no-cpu-baseline --no-pipeline --no-dropin --no-ops-leg "$@" 2> /dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); ro=r['roofline']
print('$which', r['value'], 'img/s', r['ms_per_step'], 'ms; one stream', r.get('ms_per_step_one_stream'), 'frac', ro['frac'])"
  done
done
