#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2o; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_hip_conv.py -m gpu -x -q > $O/tests_conv.log 2>&1; echo "rc=$?" >> $O/tests_conv.log; tail -3 $O/tests_conv.log
grep -q "rc=0" $O/tests_conv.log || exit 1
python scripts/layer_times.py > $O/layers.txt 2>/dev/null; head -1 $O/layers.txt
grep "igemm" $O/layers.txt | sort -k1 -n -r | head -24
python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f vol/s' % json.loads(sys.stdin.read())['value'])"
