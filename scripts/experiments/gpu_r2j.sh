#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2j; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_hip_conv.py tests/test_hip_unet.py -m gpu -x -q > $O/tests_conv.log 2>&1; echo "rc=$?" >> $O/tests_conv.log; tail -3 $O/tests_conv.log
grep -q "rc=0" $O/tests_conv.log || exit 1
MMTTA_NO_WS=1 timeout -k 10 300 python -m pytest tests/test_hip_conv.py tests/test_hip_unet.py -m gpu -x -q > $O/tests_conv2.log 2>&1; echo "rc=$?" >> $O/tests_conv2.log; tail -3 $O/tests_conv2.log
grep -q "rc=0" $O/tests_conv2.log || exit 1
python scripts/ws_phases.py --cin 32 --cout 32 --size 64 > $O/ws_phases.txt 2>&1; cat $O/ws_phases.txt
python scripts/layer_times.py > $O/layers_ws.txt 2>$O/layers.err
MMTTA_NO_WS=1 python scripts/layer_times.py > $O/layers_nows.txt 2>>$O/layers.err
MMTTA_NO_WS=1 MMTTA_NO_EPIVEC=1 python scripts/layer_times.py > $O/layers_old.txt 2>>$O/layers.err
for cfg in "0 0" "1 0" "1 1"; do set -- $cfg
  v=$(MMTTA_NO_WS=$1 MMTTA_NO_EPIVEC=$2 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])")
  echo "no_ws=$1 no_epivec=$2 -> $v vol/s" | tee -a $O/ws.txt
done
