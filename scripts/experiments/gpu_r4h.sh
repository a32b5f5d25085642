#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4h; mkdir -p $O; cd $R
for rep in 1 2; do for v in 0 1; do for a in "--steps 20 --warmup 5" ""; do
MMTTA_THIN_GRAD_FP32=$v timeout -k 10 300 python bench.py $a --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('thin fp32=$v [$a] %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done; done; done
