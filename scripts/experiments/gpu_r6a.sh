#!/bin/bash
# round 3: one volume alone (lanes 1, group 1): kernel time against launch gaps
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6a; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof -o run -- python3 $R/bench.py --lanes 1 --group 1 --steps 6 --warmup 3 --no-cpu-baseline --no-variants --no-profile-pass > $O/bench.json 2> $O/trace.err
cd $R; cat $O/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
python scripts/experiments/one_volume_gaps.py $O/prof/run_results.db --ms 90 > $O/gaps.md 2>> $O/trace.err; head -50 $O/gaps.md; rm -rf $O/prof
