#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_conv.py tests/test_hip_deepfusion.py tests/test_dispatch.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout -k 10 600 python scripts/layer_times.py --model unet_multimodal_deepfusion --tune-volumes 24 > $O/layers_df.txt 2>&1; head -2 $O/layers_df.txt | tail -1; grep -E "k1s1" $O/layers_df.txt | head
timeout -k 10 400 python bench.py --model unet_multimodal_deepfusion --steps 24 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('deepfusion %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
