#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4x; mkdir -p $O; cd $R
for rep in 1 2; do for f in "" "--from-host"; do
timeout -k 10 400 python bench.py $f --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$f] %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done; done
