#!/bin/bash
# round 3: lanes x group on the final build (whole rounds of volumes)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c; mkdir -p $O; cd $R
run() { # lanes group
  local v=$(( $1 * $2 ))
  timeout -k 10 300 python bench.py --lanes $1 --group $2 --steps $(( 2 * v )) --warmup $v --no-cpu-baseline --no-variants --no-profile-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lanes $1 group $2: %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))" | tee -a $O/out.txt
}
run 3 8 && run 4 8 && run 4 6 && run 6 4 && run 5 8 && run 3 12 && run 3 8
