#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2n; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_hip_conv.py -m gpu -x -q > $O/tests_conv.log 2>&1; echo "rc=$?" >> $O/tests_conv.log; tail -3 $O/tests_conv.log
grep -q "rc=0" $O/tests_conv.log || exit 1
for st in fp32 bf16; do for v in 0 1; do
  MMTTA_STORAGE=$st MMTTA_WGVEC=$v python scripts/layer_times.py 2>/dev/null | grep "wgrad_bf16" > $O/wg_${st}_$v.txt
  echo "== storage $st vec $v: $(awk '{s+=$1*$2} END {print s}' $O/wg_${st}_$v.txt) us of wgrad_bf16 per step"
done; done
for cfg in "fp32 0" "fp32 1" "bf16 0" "bf16 1"; do set -- $cfg
  v=$(MMTTA_STORAGE=$1 MMTTA_WGVEC=$2 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])")
  echo "storage=$1 wgvec=$2 -> $v vol/s" | tee -a $O/res.txt
done
