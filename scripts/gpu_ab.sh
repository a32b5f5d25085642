#!/bin/bash
# usage: gpu_ab.sh "<ENV_A>" "<ENV_B>" [reps]   - same-box A/B of bench.py under two environments
R=$GRAFT_REPO_ROOT; cd $R
A="$1"; B="$2"; N=${3:-3}
for i in $(seq 1 $N); do
  for cfg in "$A" "$B"; do
    v=$(env $cfg python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])")
    echo "rep $i [$cfg] -> $v vol/s"
  done
done
