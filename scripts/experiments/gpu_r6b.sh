#!/bin/bash
# round 3: one volume alone: do the side streams of the weight gradients overlap anything?
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6b; mkdir -p $O; cd $R
for ss in 0 1 2 3; do
timeout -k 10 200 python bench.py --lanes 1 --group 1 --steps 8 --warmup 3 --side-streams $ss --no-cpu-baseline --no-variants --no-profile-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side streams $ss: %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))" | tee -a $O/out.txt
done
timeout -k 10 200 python bench.py --lanes 1 --group 1 --steps 8 --warmup 3 --no-graph --no-cpu-baseline --no-variants --no-profile-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no graph: %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))" | tee -a $O/out.txt
