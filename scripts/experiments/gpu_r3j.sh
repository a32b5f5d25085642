#!/bin/bash
# round 3: kernel-level softmax DiceCE, per-layer times of the grouped arrangement, bench lines with 3 lanes x 8
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3j; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_hip_tta.py -q -k "dicece" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout -k 10 300 python scripts/layer_times.py > $O/layers.txt 2>&1; head -60 $O/layers.txt
for a in "" "--steps 20 --warmup 5"; do
  timeout -k 10 300 python bench.py $a --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet [$a] %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
