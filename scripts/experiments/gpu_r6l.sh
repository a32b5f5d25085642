#!/bin/bash
# round 3: pipelined class-fused stride-2 kernel: two workgroups per CU with spills (occ2) against one with none (occ1)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6l; mkdir -p $O; cd $R
for v in occ2 occ1; do
  cp scripts/experiments/ab_libs/libmmtta_$v.so multimodal_tta_amd/csrc/libmmtta.so
  timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_hip_groups.py -x -q > $O/tests_$v.log 2>&1; tail -1 $O/tests_$v.log
  timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers_$v.txt 2>&1; grep -E "cls8|conv time" $O/layers_$v.txt
  for i in 1 2; do
  timeout -k 10 300 python bench.py --steps 48 --warmup 24 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
  done
done
