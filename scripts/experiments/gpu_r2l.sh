#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2l; mkdir -p $O; cd $R
MMTTA_LEAN=1 timeout -k 10 300 python -m pytest tests/test_hip_conv.py tests/test_hip_unet.py -m gpu -x -q > $O/tests_lean.log 2>&1; echo "rc=$?" >> $O/tests_lean.log; tail -3 $O/tests_lean.log
timeout -k 10 300 python -m pytest tests/test_hip_conv.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -3 $O/tests.log
MMTTA_LEAN=1 python scripts/layer_times.py 2>/dev/null | grep "32->32\|128->32\|32->64" > $O/layers_lean.txt; cat $O/layers_lean.txt
python scripts/layer_times.py 2>/dev/null | grep "32->32\|128->32\|32->64" > $O/layers_base.txt; cat $O/layers_base.txt
for w in 1 0 1 0; do
  v=$(MMTTA_LEAN=$w python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])")
  echo "lean=$w -> $v vol/s" | tee -a $O/lean.txt
done
