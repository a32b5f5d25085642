#!/bin/bash
# SQ / memory counter passes over WHOLE adapted volumes: bench.py, one lane, ONE GROUP of 8 volumes under the headline's launch
# geometry (--tune-volumes 24: 3 lanes x 8), eager launches: 8 warm-up + 8 dry-run + 8 timed volumes = 24 volumes of 10 steps + final forward
# each (scripts/traffic_rank.py turns the table into bytes per volume: 24 volumes); one counter group per pass, kernel trace
# only, the program directly after `--`.
set -e -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --lanes 1 --group 8 --tune-volumes 24 --steps 8 --warmup 8 --no-cpu-baseline --no-variants --no-profile-pass --no-graph"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT -d $out/q1 -o run -- $B > $out/q1.json 2> $out/q1.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/q2 -o run -- $B > $out/q2.json 2> $out/q2.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/q3 -o run -- $B > $out/q3.json 2> $out/q3.err
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $out/q1/run_results.db > $out/sq_per_kernel.txt
python scripts/pmc_summary.py $out/q2/run_results.db $out/q3/run_results.db --json $out/traffic.json > $out/mem_per_kernel.txt
rm -rf $out/q1 $out/q2 $out/q3
ls -la $out
