#!/bin/bash
# One GPU-box pass: parity tests, the default bench line, and the rocprofv3 kernel trace of the same bench command.
# Usage (from the repo root, on the GPU box): bash scripts/gpu_round.sh <tag>
set -e -o pipefail
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1 || { tail -30 $out/pytest_gpu.log; exit 1; }
tail -2 $out/pytest_gpu.log
timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.err
cat $out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_prof.json 2> $GRAFT_REPO_ROOT/$out/prof.log
cd $GRAFT_REPO_ROOT
cat $out/bench_prof.json
python scripts/trace_summary.py $out/prof/run_results.db > $out/kernels.md
head -12 $out/kernels.md
