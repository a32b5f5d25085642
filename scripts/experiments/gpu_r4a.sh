#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_deepfusion.py tests/test_hip_pointwise.py tests/test_hip_groups.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout -k 10 600 python -m pytest tests/test_hip_fullsize.py -x -q -s -k "deepfusion" > $O/full.log 2>&1; tail -3 $O/full.log; grep fullsize $O/full.log
timeout -k 10 400 python bench.py --model unet_multimodal_deepfusion --steps 24 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>$O/df.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('deepfusion %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
