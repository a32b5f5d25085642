#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4j; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_pointwise.py tests/test_hip_tta.py tests/test_hip_groups.py tests/test_hip_golden.py tests/test_hip_deepfusion.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
