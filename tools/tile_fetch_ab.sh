export TMPDIR=/tmp; R=$PWD; o=$R/gpurun_out/tf; mkdir -p $o
i=0
for shape in "256 7 7 512 512 3 1 1" "256 7 7 512 2048 1 1 0" "256 14 14 256 256 3 1 1" "256 7 7 2048 512 1 1 0" "256 14 14 1024 256 1 1 0"; do
  i=$((i+1))
  python3 tools/conv_bench.py $shape --reps 20 > $o/t$i.txt 2>&1 || exit 1
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/p$i -- python3 $R/tools/conv_bench.py $shape --reps 3 > /dev/null 2>&1) || exit 1
  echo "== $shape" >> $o/all.txt; cat $o/t$i.txt >> $o/all.txt; python3 tools/pmc_per_dispatch.py $o/p$i >> $o/all.txt
done
cat $o/all.txt
