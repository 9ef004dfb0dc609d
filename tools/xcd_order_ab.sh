#!/bin/bash
# Tile order over the XCDs (rn_ctx_set_xcd_groups / RN_XCD_NGROUPS): time and FETCH_SIZE of the stage 3-4 shapes
# for 1 (M panel major), 2, 4, 8 groups of N tiles and for the per-launch choice (0), on fixed tile candidates.
#   bash tools/xcd_order_ab.sh            -> gpurun_out/xcd/all.txt
export TMPDIR=/tmp; R=$PWD; o=$R/gpurun_out/xcd; rm -rf $o; mkdir -p $o
i=0
for shape in "256 7 7 512 512 3 1 1" "256 7 7 512 2048 1 1 0" "256 7 7 2048 512 1 1 0" "256 14 14 256 256 3 1 1" "256 14 14 1024 256 1 1 0" "256 14 14 256 1024 1 1 0" "256 14 14 1024 512 1 1 0"; do
  i=$((i+1))
  echo "== $shape" >> $o/all.txt
  for g in 0 1 2 4 8; do
    export RN_XCD_NGROUPS=$g
    for cand in 4 3 2; do
      t=$(python3 tools/conv_bench.py $shape --reps 20 --cand $cand --relu 2>/dev/null | tail -1)
      (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/p${i}_${g}_$cand -- python3 $R/tools/conv_bench.py $shape --reps 3 --cand $cand --relu > /dev/null 2>&1) || exit 1
      f=$(python3 tools/pmc_per_dispatch.py $o/p${i}_${g}_$cand | grep conv_gemm | sort -k3 -n -r | head -1)
      echo "groups $g  $t   | fetch: $f" >> $o/all.txt
      rm -rf $o/p${i}_${g}_$cand
    done
  done
done
unset RN_XCD_NGROUPS
cat $o/all.txt
