for cfg in "--arch resnet152 --batch 128" "--batch 128" "--batch 128 --dtype bf16" "--batch 192" "--batch 192 --dtype bf16"; do
  for s in 1 2 1 2; do
    python bench.py $cfg --streams $s --no-cpu-baseline --no-pipeline --no-dropin --no-ops-leg 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'streams', $s, r['value'], r['ms_per_step'])"
  done
done
