#!/bin/bash
# round 3: deep-fusion family runtime + grouped tests, bench lines, kernel trace of the default arrangement
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_hip_deepfusion.py tests/test_hip_groups.py tests/test_hip_golden.py tests/test_hip_tta.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
for a in "" "--steps 20 --warmup 5"; do
  timeout -k 10 300 python bench.py $a --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet [$a] %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
timeout -k 10 400 python bench.py --model unet_multimodal_deepfusion --steps 8 --warmup 2 --no-cpu-baseline --no-profile-pass --no-variants 2>$O/df.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('deepfusion %.2f vol/s %.2f ms lanes %s group %s' % (d['value'], d['ms_per_step'], d['config']['lanes'], d['config']['group']))" || tail -5 $O/df.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python3 $R/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-variants --no-profile-pass > $O/bench_traced.json 2> $O/trace.err
cd $R
python scripts/trace_summary.py $O/prof/run_results.db > $O/kernels.md 2>> $O/trace.err
rm -rf $O/prof
head -45 $O/kernels.md
