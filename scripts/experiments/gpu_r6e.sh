#!/bin/bash
# round 3: paired dense column blocks in the stride-2 transposed-read weight gradient: parity + layer times (pair on / off)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_dispatch.py tests/test_hip_unet.py tests/test_hip_groups.py tests/test_hip_tta.py tests/test_hip_golden.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers.txt 2>&1; grep -E "wgrad_tr_kernel|conv time" $O/layers.txt
MMTTA_WGRAD_PAIR=0 timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers_nopair.txt 2>&1; grep -E "wgrad_tr_kernel|conv time" $O/layers_nopair.txt
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 48 --warmup 24 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
