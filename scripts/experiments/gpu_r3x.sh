#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3x; mkdir -p $O; cd $R
timeout -k 10 600 python scripts/sweep_tuning.py --combos 4x8 4x10 4x12 --knobs scaled --volumes 96 --repeat 1 > $O/sweep.txt 2>&1; tail -3 $O/sweep.txt
for l in 3 4; do for a in "--steps 20 --warmup 5" ""; do
timeout -k 10 300 python bench.py --lanes $l $a --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes $l [$a] %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done; done
