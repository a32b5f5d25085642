#!/bin/bash
# round 3: full GPU suite with bf16 deep fusion, deep-fusion bench, unet bench
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/all.log 2>&1; tail -5 $O/all.log; grep "fullsize" $O/all.log
timeout -k 10 400 python bench.py --model unet_multimodal_deepfusion --steps 24 --warmup 2 --no-cpu-baseline --no-profile-pass --no-variants 2>$O/df.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('deepfusion %.2f vol/s %.2f ms lanes %s group %s' % (d['value'], d['ms_per_step'], d['config']['lanes'], d['config']['group']))" || tail -5 $O/df.err
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
