#!/bin/bash
# Quick check of a kernel change: conv / dispatch / unet tests, then two unet lines (and optionally the deep-fusion line).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2x; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_hip_pointwise.py tests/test_dispatch.py tests/test_hip_unet.py tests/test_hip_deepfusion.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "rc=$rc" >> $O/tests.log; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
  python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('unet %.2f' % json.loads(sys.stdin.read())['value'])"
done
if [ "$1" = "df" ]; then
python bench.py --model unet_multimodal_deepfusion --steps 4 --warmup 1 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('deepfusion %.2f' % d['value'], d['ms_per_step'])"
fi
