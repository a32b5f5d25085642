#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_hip_conv.py tests/test_hip_unet.py -m gpu -x -q > $O/tests_conv.log 2>&1; echo "rc=$?" >> $O/tests_conv.log; tail -3 $O/tests_conv.log
for w in 1 0 1 0; do
  v=$(MMTTA_WS=$w python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])")
  echo "ws=$w -> $v vol/s" | tee -a $O/ws.txt
done
