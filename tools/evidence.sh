#!/bin/bash
# Round evidence on one MI355X: bench lines, per-layer tables, rocprofv3 kernel traces, PMC passes.
# Everything lands under gpurun_out/ev/ ; tools/copy_evidence.sh takes what is to be judged into profiles/roundN/.
#   tools/evidence.sh [bench|trace|pmc|mfma|all]
# Every phase deletes the files it is about to write first: what lies in gpurun_out/ev afterwards is this
# build's or nothing (a failed step leaves a hole, not an older build's file), and every phase writes
# $o/<phase>.stamp = digest of the kernel sources it ran on (tools/copy_evidence.sh refuses another build's).
what=${1:-all}
o=gpurun_out/ev
mkdir -p $o
export TMPDIR=/tmp
digest=$(python3 -c "import resnet_c_amd as R; print(R._lib.source_digest())")
stamp() { echo "$digest" > $o/$1.stamp; }
if [ $what = bench ] || [ $what = all ]; then
  rm -f $o/bench.stamp $o/bench_*.json $o/bench_*.err $o/layers_*.txt $o/dropin_route_rates.txt $o/shard_upload_rates.txt \
        $o/stem_pool_alone.txt $o/stem_pool_stamps_*.txt
  python bench.py > $o/bench_f32.json 2> $o/bench_f32.err
  python bench.py --streams 1 --no-cpu-baseline --no-pipeline --no-dropin > $o/bench_f32_one_stream.json 2> $o/bench_f32_1s.err
  python bench.py --dtype bf16 --no-cpu-baseline --no-pipeline --no-dropin > $o/bench_bf16.json 2> $o/bench_bf16.err
  python bench.py --dtype bf16 --streams 1 --no-cpu-baseline --no-pipeline --no-dropin > $o/bench_bf16_one_stream.json 2> $o/bench_bf16_1s.err
  python bench.py --arch resnet152 --batch 128 --no-cpu-baseline --no-pipeline --no-dropin > $o/bench_resnet152_b128.json 2> $o/bench_resnet152.err
  python bench.py --mode ops --no-cpu-baseline --no-pipeline --no-dropin > $o/bench_ops_mode.json 2> $o/bench_ops.err
  python tools/layer_report.py --tune > $o/layers_f32.txt 2>&1
  python tools/layer_report.py --tune --dtype bf16 > $o/layers_bf16.txt 2>&1
  python tools/veneer_rate.py --batch 256 --steps 3 > $o/dropin_route_rates.txt 2>&1
  python tools/shard_rate.py > $o/shard_upload_rates.txt 2>&1
  python tools/stem_bench.py > $o/stem_pool_alone.txt 2>&1
  python tools/stem_stamps.py --dtype bf16 > $o/stem_pool_stamps_bf16.txt 2>&1
  python tools/stem_stamps.py --dtype f32 > $o/stem_pool_stamps_f32.txt 2>&1
  stamp bench
fi
trace_one() {  # name, bench.py arguments...
  local n=$1; shift
  rm -rf $o/trace_$n $o/bench_under_rocprof_$n.json $o/timed_region_kernels_$n.json $o/kernel_stats_$n.csv
  # one stream: with the batch as two parts on two streams the kernels of the parts overlap and a
  # kernel's duration in the trace is no longer its own
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_$n -- python3 bench.py "$@" --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline --no-dropin --no-ops-leg > $o/bench_under_rocprof_$n.json 2> $o/trace_$n.err
  python tools/ktrace_summary.py $o/trace_$n 13 $o/timed_region_kernels_$n.json > /dev/null
  local f=$(find $o/trace_$n -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/kernel_stats_$n.csv
  find $o/trace_$n -name '*kernel_trace.csv' -size +8M -delete
}
if [ $what = trace ] || [ $what = all ]; then
  rm -f $o/trace.stamp
  trace_one f32 --dtype f32
  trace_one bf16 --dtype bf16
  trace_one resnet152_b128 --arch resnet152 --batch 128    # BASELINE.json configs[4]
  # (the ops-mode trace keeps its ops leg: it IS that mode)
  rm -rf $o/trace_ops_mode $o/bench_under_rocprof_ops_mode.json $o/timed_region_kernels_ops_mode.json $o/kernel_stats_ops_mode.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace_ops_mode -- python3 bench.py --mode ops --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline --no-dropin > $o/bench_under_rocprof_ops_mode.json 2> $o/trace_ops_mode.err
  python tools/ktrace_summary.py $o/trace_ops_mode 13 $o/timed_region_kernels_ops_mode.json > /dev/null
  f=$(find $o/trace_ops_mode -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/kernel_stats_ops_mode.csv
  find $o/trace_ops_mode -name '*kernel_trace.csv' -size +8M -delete
  stamp trace
fi
# launches per forward and algorithmic bytes of the contraction family: what a bench line of this build says
ref_line() {  # tag, bench.py arguments... -> sets n, bytes
  local t=$1; shift
  python3 bench.py "$@" --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-dropin --no-ops-leg > $o/pmc_ref_$t.json 2> /dev/null
  read n bytes <<< $(python3 -c "import json;r=json.loads(open('$o/pmc_ref_$t.json').read().strip().splitlines()[-1])['roofline'];print(r['launches_per_forward'], r['bytes_per_forward'])")
}
if [ $what = pmc ] || [ $what = all ]; then
  rm -rf $o/pmc.stamp $o/pmc_f32_* $o/pmc_bf16_* $o/pmc_rn152_* $o/pmc_ref_*.json $o/hbm_traffic_pmc_* $o/pmc_layers_*.txt
  for dt in f32 bf16; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --output-format csv -d $o/pmc_${dt}_$c -- python3 bench.py --dtype $dt --streams 1 --steps 2 --warmup 1 --profile-forwards 1 --no-cpu-baseline --no-pipeline --no-dropin --no-ops-leg > /dev/null 2> $o/pmc_${dt}_$c.err
    done
  done
  for dt in f32 bf16; do
    ref_line $dt --dtype $dt
    python tools/pmc_traffic.py $o/pmc_${dt}_FETCH_SIZE $o/pmc_${dt}_WRITE_SIZE $o/hbm_traffic_pmc_$dt.json $n $bytes "ResNet-50 $dt B=256 fused; separate --pmc passes; FETCH_SIZE x2" > /dev/null
    # the same bytes launch by launch next to each layer's algorithmic bytes (where an excess sits)
    python tools/layer_report.py --tune --dtype $dt > $o/pmc_layers_$dt.txt 2> /dev/null
    python tools/pmc_per_layer.py $o/pmc_${dt}_FETCH_SIZE $o/pmc_${dt}_WRITE_SIZE $o/pmc_layers_$dt.txt $n > $o/hbm_traffic_pmc_$dt.per_layer.txt
  done
  # the same two passes for ResNet-152 fp32 B=128 (configs[4])
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $o/pmc_rn152_$c -- python3 bench.py --arch resnet152 --batch 128 --streams 1 --steps 2 --warmup 1 --profile-forwards 1 --no-cpu-baseline --no-pipeline --no-dropin --no-ops-leg > /dev/null 2> $o/pmc_rn152_$c.err
  done
  ref_line rn152 --arch resnet152 --batch 128
  python tools/pmc_traffic.py $o/pmc_rn152_FETCH_SIZE $o/pmc_rn152_WRITE_SIZE $o/hbm_traffic_pmc_resnet152_b128.json $n $bytes "ResNet-152 f32 B=128 fused; separate --pmc passes; FETCH_SIZE x2" > /dev/null
  find $o -name '*counter_collection.csv' -size +8M -delete
  stamp pmc
fi
if [ $what = mfma ] || [ $what = all ]; then
  # matrix-pipe utilisation of every contraction instantiation: one SQ pass (8 SQ slots + GRBM), program right after --
  rm -rf $o/mfma.stamp $o/mfma_f32 $o/mfma_bf16 $o/pmc_mfma_utilisation_* $o/mfma_layers_*
  for dt in f32 bf16; do
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
      --output-format csv -d $o/mfma_$dt -- python3 bench.py --dtype $dt --streams 1 --steps 2 --warmup 1 --profile-forwards 1 --no-cpu-baseline --no-pipeline --no-dropin --no-ops-leg > /dev/null 2> $o/mfma_$dt.err
    ref_line mfma_$dt --dtype $dt
    python tools/layer_report.py --tune --dtype $dt > $o/mfma_layers_$dt.txt 2> /dev/null
    python tools/pmc_mfma.py $o/mfma_$dt $o/pmc_mfma_utilisation_$dt.json $n "ResNet-50 $dt B=256 fused, one stream; one rocprofv3 --pmc pass" $o/mfma_layers_$dt.txt > $o/mfma_$dt.summary 2>&1
  done
  find $o -name '*counter_collection.csv' -size +8M -delete
  stamp mfma
fi
ls -la $o | head -80
