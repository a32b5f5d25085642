#!/bin/bash
# round 3: bf16-stored activation gradients: kernel tests, network tests, same-box A/B of the bench
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_pointwise.py tests/test_hip_conv.py tests/test_hip_groups.py tests/test_hip_unet.py tests/test_hip_tta.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout -k 10 500 python -m pytest tests/test_hip_fullsize.py -x -q -s > $O/full.log 2>&1; tail -3 $O/full.log; grep fullsize $O/full.log
for gs in bf16 fp32 bf16 fp32; do
  timeout -k 10 300 python bench.py --grad-storage $gs --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grad storage $gs: %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
