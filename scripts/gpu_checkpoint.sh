#!/bin/bash
# usage: bash scripts/gpu_checkpoint.sh <tag> [notests]   -> gpurun_out/<tag>/{tests.log,bench.json,kernels.md,stats.csv,timeline.md}
set -o pipefail
TAG=${1:-ckpt}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
if [ "$2" != "notests" ]; then
  python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -3 $O/tests.log
fi
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
# the headline arrangement (lanes x group volumes per round) needs a whole round of volumes: 24 warm-up + 24 dry-run + 24 timed
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/prof -o run -- python3 $R/bench.py --steps 24 --warmup 24 --no-cpu-baseline --no-variants > $O/bench_traced.json 2> $O/trace.err
cd $R
python scripts/trace_summary.py $O/prof/run_results.db > $O/kernels.md 2>> $O/trace.err
python scripts/trace_summary.py $O/prof/run_results.db --json > $O/trace_avg_us.json 2>> $O/trace.err
python scripts/trace_timeline.py $O/prof/run_results.db > $O/timeline.md 2>> $O/trace.err
ls $O/prof | head; cp $O/prof/*stats*.csv $O/ 2>/dev/null; rm -rf $O/prof
python -c "
import json; d=json.load(open('$O/bench.json'))
print('value',d['value'],'parity',d.get('parity_full_size',{}).get('within_tolerance'),'variants',d.get('variants'))
r=d['roofline']; print({k:r[k] for k in ('bound','kernel','achieved','frac','avg_launch_us')}, r['whole_volume'])"
head -12 $O/kernels.md
