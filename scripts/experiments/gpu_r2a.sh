#!/bin/bash
# round-2 checkpoint A: fixed tests, new bench line, one-lane + two-lane kernel traces
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
python -m pytest tests/test_hip_tta.py -m gpu -x -q -k "workspace_growth or factory_optimizers or dicece" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python -m pytest tests/test_hip_pointwise.py tests/test_hip_conv.py -m gpu -x -q > $O/tests2.log 2>&1; echo "tests2 rc=$?" >> $O/tests2.log
python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_]*MFMA[A-Z_0-9]*\|SQ_BUSY_CU_CYCLES\|SQ_LDS_BANK_CONFLICT\|SQ_INSTS_MFMA" | sort -u > $O/mfma_counters.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof1 -o run -- python3 $R/bench.py --steps 4 --warmup 2 --lanes 1 --no-cpu-baseline --no-profile-pass --no-variants > $O/trace1.json 2> $O/trace1.err
rocprofv3 --kernel-trace -d $O/prof2 -o run -- python3 $R/bench.py --steps 4 --warmup 2 --lanes 2 --no-cpu-baseline --no-profile-pass --no-variants > $O/trace2.json 2> $O/trace2.err
cd $R
python scripts/trace_timeline.py $O/prof1/run_results.db > $O/timeline_1lane.md 2>> $O/trace1.err
python scripts/trace_timeline.py $O/prof2/run_results.db > $O/timeline_2lanes.md 2>> $O/trace2.err
rm -rf $O/prof1 $O/prof2
tail -3 $O/tests.log $O/tests2.log; cat $O/bench.err | tail -3; head -c 600 $O/bench.json
