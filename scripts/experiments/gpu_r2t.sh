#!/bin/bash
# lanes x hardware queues
R=$GRAFT_REPO_ROOT; cd $R
for q in 5 6 8 10 12; do for l in 5 6; do
  v=$(GPU_MAX_HW_QUEUES=$q python bench.py --lanes $l --steps 24 --warmup 6 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])")
  echo "queues=$q lanes=$l -> $v vol/s"
done; done
