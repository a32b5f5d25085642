#!/bin/bash
# Deep-fusion kernel trace (profile the 128^3 decoder stage) of the current build.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/prof -o run -- python3 $R/bench.py --model unet_multimodal_deepfusion --steps 4 --warmup 1 --no-cpu-baseline --no-variants > $O/bench_traced.json 2> $O/trace.err
echo "trace rc=$?"
cd $R
python scripts/trace_summary.py $O/prof/run_results.db > $O/kernels.md 2>> $O/trace.err
python scripts/trace_timeline.py $O/prof/run_results.db > $O/timeline.md 2>> $O/trace.err
python scripts/chip_equivalent.py $O/timeline.md --volumes 5 > $O/chip_equivalent.md 2>> $O/trace.err
cp $O/prof/*stats*.csv $O/ 2>/dev/null; rm -rf $O/prof
cat $O/bench_traced.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'])"
head -40 $O/kernels.md
