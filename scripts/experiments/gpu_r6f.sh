#!/bin/bash
# round 3: transposed-read weight gradients at one workgroup per CU (accumulators in AGPRs, no spills): layer times + bench
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r6f}; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_dispatch.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers.txt 2>&1; grep -E "wgrad_tr_kernel|conv time" $O/layers.txt
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 48 --warmup 24 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
