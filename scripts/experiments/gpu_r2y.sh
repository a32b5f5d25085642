#!/bin/bash
# Kernel-level check of a change: per-kernel average durations of a traced bench run (kernels matching $1), then two bench lines.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2y; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/prof -o run -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-variants > $O/bench_traced.json 2> $O/trace.err
cd $R
python scripts/trace_summary.py $O/prof/run_results.db > $O/kernels.md 2>> $O/trace.err
rm -rf $O/prof
grep -E "$1" $O/kernels.md
for i in 1 2; do
  python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('unet %.2f' % json.loads(sys.stdin.read())['value'])"
done
