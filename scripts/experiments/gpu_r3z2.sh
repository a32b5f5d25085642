#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3z; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format rocpd csv -d $O/prof -o run -- python3 $R/bench.py --model unet_multimodal_deepfusion --steps 24 --warmup 8 --no-cpu-baseline --no-variants --no-profile-pass > $O/bench_df.json 2> $O/trace.err
cd $R
python scripts/trace_summary.py $O/prof/run_results.db > $O/kernels_df.md 2>> $O/trace.err; rm -rf $O/prof
head -45 $O/kernels_df.md
