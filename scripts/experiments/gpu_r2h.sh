#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_hip_conv.py -m gpu -x -q > $O/tests_conv.log 2>&1; echo "rc=$?" >> $O/tests_conv.log; tail -4 $O/tests_conv.log
grep -q "rc=0" $O/tests_conv.log || exit 1
timeout -k 10 600 python -m pytest tests/test_hip_unet.py tests/test_hip_tta.py tests/test_hip_fullsize.py tests/test_hip_deepfusion.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
python scripts/layer_times.py > $O/layers_ws.txt 2>$O/layers.err
MMTTA_NO_WS=1 python scripts/layer_times.py > $O/layers_nows.txt 2>>$O/layers.err
for w in 0 1; do
  v=$(MMTTA_NO_WS=$w python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])")
  echo "no_ws=$w -> $v vol/s" | tee -a $O/ws.txt
done
