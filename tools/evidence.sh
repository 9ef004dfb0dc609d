#!/bin/bash
# Round evidence on one MI355X: bench lines, per-layer tables, rocprofv3 kernel traces, PMC traffic.
# Everything lands under gpurun_out/ev/ ; copy what is to be judged into profiles/roundN/.
#   tools/evidence.sh [bench|trace|pmc|all]
what=${1:-all}
o=gpurun_out/ev
mkdir -p $o   # (one phase per gpurun call fits the call's time limit; the phases do not share files)
export TMPDIR=/tmp
if [ $what = bench ] || [ $what = all ]; then
  python bench.py > $o/bench_f32.json 2> $o/bench_f32.err
  python bench.py --streams 1 --no-cpu-baseline --no-pipeline > $o/bench_f32_one_stream.json 2> $o/bench_f32_1s.err
  python bench.py --dtype bf16 --no-cpu-baseline --no-pipeline > $o/bench_bf16.json 2> $o/bench_bf16.err
  python bench.py --dtype bf16 --streams 1 --no-cpu-baseline --no-pipeline > $o/bench_bf16_one_stream.json 2> $o/bench_bf16_1s.err
  python bench.py --arch resnet152 --batch 128 --no-cpu-baseline --no-pipeline > $o/bench_resnet152_b128.json 2> $o/bench_resnet152.err
  python bench.py --mode ops --no-cpu-baseline --no-pipeline > $o/bench_ops_mode.json 2> $o/bench_ops.err
  python tools/layer_report.py --tune > $o/layers_f32.txt 2>&1
  python tools/layer_report.py --tune --dtype bf16 > $o/layers_bf16.txt 2>&1
  python tools/veneer_rate.py --batch 256 --steps 3 > $o/dropin_route_rates.txt 2>&1
  python tools/shard_rate.py > $o/shard_upload_rates.txt 2>&1
  python tools/stem_bench.py > $o/stem_pool_alone.txt 2>&1
  python tools/stem_stamps.py --dtype bf16 > $o/stem_pool_stamps_bf16.txt 2>&1
  python tools/stem_stamps.py --dtype f32 > $o/stem_pool_stamps_f32.txt 2>&1
fi
if [ $what = trace ] || [ $what = all ]; then
  for dt in f32 bf16; do
    # one stream: with the batch as two parts on two streams the kernels of the parts overlap and a
    # kernel's duration in the trace is no longer its own
    rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_$dt -- python3 bench.py --dtype $dt --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline --no-ops-leg > $o/bench_under_rocprof_$dt.json 2> $o/trace_$dt.err
    python tools/ktrace_summary.py $o/trace_$dt 13 $o/timed_region_kernels_$dt.json > /dev/null
    f=$(find $o/trace_$dt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/kernel_stats_$dt.csv
    find $o/trace_$dt -name '*kernel_trace.csv' -size +8M -delete
  done
  # BASELINE.json configs[4]: ResNet-152 fp32 B=128
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_rn152 -- python3 bench.py --arch resnet152 --batch 128 --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline --no-ops-leg > $o/bench_under_rocprof_resnet152_b128.json 2> $o/trace_rn152.err
  python tools/ktrace_summary.py $o/trace_rn152 13 $o/timed_region_kernels_resnet152_b128.json > /dev/null
  f=$(find $o/trace_rn152 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/kernel_stats_resnet152_b128.csv
  find $o/trace_rn152 -name '*kernel_trace.csv' -size +8M -delete
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_ops -- python3 bench.py --mode ops --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline > $o/bench_under_rocprof_ops.json 2> $o/trace_ops.err
  python tools/ktrace_summary.py $o/trace_ops 13 $o/timed_region_kernels_ops_mode.json > /dev/null
  f=$(find $o/trace_ops -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/kernel_stats_ops_mode.csv
  find $o/trace_ops -name '*kernel_trace.csv' -size +8M -delete
fi
if [ $what = pmc ] || [ $what = all ]; then
  for dt in f32 bf16; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --output-format csv -d $o/pmc_${dt}_$c -- python3 bench.py --dtype $dt --streams 1 --steps 2 --warmup 1 --profile-forwards 1 --no-cpu-baseline --no-pipeline --no-ops-leg > /dev/null 2> $o/pmc_${dt}_$c.err
    done
  done
  # launches per forward and algorithmic bytes of the contraction family: what a bench line of this build says
  for dt in f32 bf16; do
    python3 bench.py --dtype $dt --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-ops-leg > $o/pmc_ref_$dt.json 2> /dev/null
    read n bytes <<< $(python3 -c "import json;r=json.loads(open('$o/pmc_ref_$dt.json').read().strip().splitlines()[-1])['roofline'];print(r['launches_per_forward'], r['bytes_per_forward'])")
    python tools/pmc_traffic.py $o/pmc_${dt}_FETCH_SIZE $o/pmc_${dt}_WRITE_SIZE $o/hbm_traffic_pmc_$dt.json $n $bytes "ResNet-50 $dt B=256 fused; separate --pmc passes; FETCH_SIZE x2" > /dev/null
    # the same bytes launch by launch next to each layer's algorithmic bytes (where an excess sits)
    python tools/layer_report.py --tune --dtype $dt > $o/pmc_layers_$dt.txt 2> /dev/null
    python tools/pmc_per_layer.py $o/pmc_${dt}_FETCH_SIZE $o/pmc_${dt}_WRITE_SIZE $o/pmc_layers_$dt.txt $n > $o/hbm_traffic_pmc_$dt.per_layer.txt
  done
  # the same two passes for ResNet-152 fp32 B=128 (configs[4])
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $o/pmc_rn152_$c -- python3 bench.py --arch resnet152 --batch 128 --streams 1 --steps 2 --warmup 1 --profile-forwards 1 --no-cpu-baseline --no-pipeline --no-ops-leg > /dev/null 2> $o/pmc_rn152_$c.err
  done
  python3 bench.py --arch resnet152 --batch 128 --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-ops-leg > $o/pmc_ref_rn152.json 2> /dev/null
  read n bytes <<< $(python3 -c "import json;r=json.loads(open('$o/pmc_ref_rn152.json').read().strip().splitlines()[-1])['roofline'];print(r['launches_per_forward'], r['bytes_per_forward'])")
  python tools/pmc_traffic.py $o/pmc_rn152_FETCH_SIZE $o/pmc_rn152_WRITE_SIZE $o/hbm_traffic_pmc_resnet152_b128.json $n $bytes "ResNet-152 f32 B=128 fused; separate --pmc passes; FETCH_SIZE x2" > /dev/null
  find $o -name '*counter_collection.csv' -size +8M -delete
fi
ls -la $o | head -60
