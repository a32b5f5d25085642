#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; mkdir -p $O; cd $R
python -m pytest tests/test_hip_conv.py tests/test_hip_unet.py tests/test_hip_tta.py tests/test_hip_fullsize.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
python scripts/layer_times.py > $O/layers_pipe.txt 2>$O/layers.err
MMTTA_NO_PIPE=1 python scripts/layer_times.py > $O/layers_nopipe.txt 2>>$O/layers.err
for p in 1 0; do
  v=$(MMTTA_NO_PIPE=$((1-p)) python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])")
  echo "pipeline=$p -> $v vol/s" | tee -a $O/pipe.txt
done
