#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_hip_conv.py -m gpu -x -q > $O/tests_conv.log 2>&1; echo "rc=$?" >> $O/tests_conv.log; tail -3 $O/tests_conv.log
grep -q "rc=0" $O/tests_conv.log || exit 1
python scripts/layer_times.py > $O/layers_ws.txt 2>$O/layers.err
grep "igemm" $O/layers_ws.txt | head -30
