#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O; cd $R
for m in "4 8" "8 8" "16 4" "32 2"; do
  set -- $m
  export MMTTA_THIN_SLAB_MULT=$1 MMTTA_THIN_MIN_TILES=$2
  timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers_$1.txt 2>&1; echo "mult $1 min tiles $2"; grep -E "wgrad_thin" $O/layers_$1.txt | head -8
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
