#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2m; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_hip_conv.py tests/test_hip_pointwise.py -m gpu -x -q > $O/tests_conv.log 2>&1; echo "rc=$?" >> $O/tests_conv.log; tail -5 $O/tests_conv.log
grep -q "rc=0" $O/tests_conv.log || exit 1
timeout -k 10 900 python -m pytest tests/test_hip_unet.py tests/test_hip_tta.py tests/test_hip_fullsize.py tests/test_hip_golden.py tests/test_hip_deepfusion.py -m gpu -q -s > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -12 $O/tests.log; grep "fullsize" $O/tests.log
for st in bf16 fp32; do
  v=$(MMTTA_STORAGE=$st python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>$O/bench_$st.err | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])")
  echo "storage=$st -> $v vol/s" | tee -a $O/storage.txt
done
