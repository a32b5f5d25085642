#!/bin/bash
# Octet norm-backward apply, 9-deep pack loads, one-channel slices on the thin-K kernel: tests, then unet and deep-fusion lines.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2w; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "rc=$rc" >> $O/tests.log; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
  python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('unet %.2f' % json.loads(sys.stdin.read())['value'])"
done
python bench.py --model unet_multimodal_deepfusion --steps 4 --warmup 1 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('deepfusion %.2f' % d['value'], d['ms_per_step'])"
