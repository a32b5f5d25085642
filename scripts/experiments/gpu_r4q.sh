#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_conv.py -x -q -k "thin or matrix_tiles or bf16_operands" > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers.txt 2>&1; head -2 $O/layers.txt | tail -1; grep -E "3->3" $O/layers.txt
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
