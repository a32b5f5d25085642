#!/bin/bash
# round 3: thin transposed-read weight gradient: its parity cases + dispatch coverage, deep-fusion bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_dispatch.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout -k 10 400 python bench.py --model unet_multimodal_deepfusion --no-cpu-baseline --no-profile-pass --no-variants 2>$O/df.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('deepfusion %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']), d['config'].get('lanes'), d['config'].get('group'))"
