#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "transposed_read" > $O/tests_tr.log 2>&1; echo "rc=$?" >> $O/tests_tr.log; tail -15 $O/tests_tr.log
grep -q "rc=0" $O/tests_tr.log || exit 1
for st in bf16; do for v in 2 3; do
  MMTTA_STORAGE=$st MMTTA_WGVEC=$v python scripts/layer_times.py 2>/dev/null > $O/layers_${st}_$v.txt
  echo "== storage $st wgvec $v: $(grep 'wgrad_' $O/layers_${st}_$v.txt | grep -v small | awk '{s+=$1*$2} END {print s}') us of wgrad per step; $(head -1 $O/layers_${st}_$v.txt)"
done; done
grep "wgrad" $O/layers_bf16_3.txt | sort -k1 -n -r | head -20
for cfg in "fp32 1" "fp32 3" "bf16 2" "bf16 3"; do set -- $cfg
  v=$(MMTTA_STORAGE=$1 MMTTA_WGVEC=$2 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])")
  echo "storage=$1 wgvec=$2 -> $v vol/s" | tee -a $O/res.txt
done
