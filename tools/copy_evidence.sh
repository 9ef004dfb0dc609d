#!/bin/bash
# gpurun_out/ev (scratch, what tools/evidence.sh wrote on the box) -> profiles/roundN/final_* (tracked)
#   bash tools/copy_evidence.sh pmc|mfma|bench|trace [round dir]
# Refuses a phase whose stamp (digest of the kernel sources it ran on) is not the tree's: evidence of
# another build must not be committed as HEAD's.
what=${1:-pmc}; d=${2:-profiles/round4}; o=gpurun_out/ev
mkdir -p $d
tree=$(python3 -c "import resnet_c_amd as R; print(R._lib.source_digest())")
if [ ! -f $o/$what.stamp ] || [ "$(cat $o/$what.stamp)" != "$tree" ]; then
  echo "copy_evidence: $o/$what.stamp = $(cat $o/$what.stamp 2>/dev/null) but the tree is $tree: not copying" >&2
  exit 1
fi
if [ $what = pmc ]; then
  for pair in "f32:" "bf16:_bf16" "resnet152_b128:_resnet152_b128"; do
    src=$o/hbm_traffic_pmc_${pair%%:*}; dst=$d/final_hbm_traffic_pmc${pair##*:}
    for ext in json per_kernel.txt per_layer.txt; do [ -f $src.$ext ] && cp $src.$ext $dst.$ext; done
  done
fi
if [ $what = mfma ]; then
  for pair in "f32:" "bf16:_bf16"; do
    src=$o/pmc_mfma_utilisation_${pair%%:*}
    [ -f $src.json ] && cp $src.json $d/final_pmc_mfma_utilisation${pair##*:}.json
    [ -f $src.per_kernel.txt ] && cp $src.per_kernel.txt $d/pmc_mfma_utilisation_per_kernel_${pair%%:*}.txt
    [ -f $src.per_layer.txt ] && cp $src.per_layer.txt $d/pmc_mfma_utilisation_per_layer_${pair%%:*}.txt
  done
fi
if [ $what = bench ]; then
  for n in f32 f32_one_stream bf16 bf16_one_stream resnet152_b128 ops_mode; do cp $o/bench_$n.json $d/final_bench_$n.json; done
  cp $o/layers_f32.txt $d/final_layers_f32.txt; cp $o/layers_bf16.txt $d/final_layers_bf16.txt
  for f in dropin_route_rates.txt shard_upload_rates.txt stem_pool_alone.txt stem_pool_stamps_bf16.txt stem_pool_stamps_f32.txt; do cp $o/$f $d/$f; done
fi
if [ $what = trace ]; then
  for n in f32 bf16 ops_mode resnet152_b128; do
    for k in kernel_stats_$n.csv timed_region_kernels_$n.json bench_under_rocprof_$n.json; do [ -f $o/$k ] && cp $o/$k $d/final_$k; done
  done
fi
git status --short $d | head -40
