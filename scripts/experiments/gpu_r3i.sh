#!/bin/bash
# round 3: softmax DiceCE kernel test, wgrad_tiny A/B (bench), deep-fusion kernel trace, lanes x group re-sweep with bf16 gradients
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_hip_tta.py tests/test_hip_conv.py -x -q -k "dicece or supervised or tiny or 3-3-3 or conv_fwd" > $O/tests.log 2>&1; tail -4 $O/tests.log
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet: %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
timeout -k 10 500 python scripts/sweep_tuning.py --combos 2x8 3x8 4x4 2x12 3x6 --knobs scaled > $O/sweep.txt 2>&1; grep -v amdgpu.ids $O/sweep.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python3 $R/bench.py --model unet_multimodal_deepfusion --steps 16 --warmup 2 --no-cpu-baseline --no-variants --no-profile-pass > $O/df_traced.json 2> $O/trace.err
cd $R
python scripts/trace_summary.py $O/prof/run_results.db > $O/df_kernels.md 2>> $O/trace.err
rm -rf $O/prof
head -40 $O/df_kernels.md
