#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4f; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_conv.py tests/test_hip_unet.py tests/test_hip_groups.py tests/test_hip_tta.py tests/test_hip_pointwise.py tests/test_hip_golden.py tests/test_dispatch.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout -k 10 300 python -m pytest tests/test_hip_fullsize.py -x -q -s -k "full_width_adaptation or ten_step or grouped_lanes" > $O/full.log 2>&1; tail -2 $O/full.log; grep fullsize $O/full.log
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers.txt 2>&1; head -2 $O/layers.txt; grep -E "4->32" $O/layers.txt | head -8
