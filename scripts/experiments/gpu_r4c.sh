#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c; mkdir -p $O; cd $R
for e in 0 1 2; do
  export MMTTA_EXP_S2=$e
  timeout -k 10 300 python -m pytest tests/test_hip_conv.py -x -q -k "bf16_operands and 32-64" > $O/t$e.log 2>&1; tail -1 $O/t$e.log
  timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers_$e.txt 2>&1; echo "exp $e"; head -2 $O/layers_$e.txt | tail -1; grep -E "32->64 k3s2" $O/layers_$e.txt
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
