#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4t; mkdir -p $O; cd $R
for rep in 1 2; do for v in "96 128" "192 256"; do
set -- $v
MMTTA_SPLITK_AT4="$1,$2" timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('splitk $v: %.2f vol/s; one_volume %.2f; one_lane %.2f; dominant %s %.1f us frac %.3f mfma %.3f' % (d['value'], d['variants']['one_volume']['value'], d['variants']['one_lane']['value'], r['kernel'], r['avg_launch_us'], r['frac'], r['mfma']['frac']))"
done; done
