#!/bin/bash
# PMC passes over one adaptation step (scripts/layer_times.py); one counter group per pass, kernel trace only.
set -e -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_lt
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $out/p1 -o run -- python3 $GRAFT_REPO_ROOT/scripts/layer_times.py --reps 1 > /dev/null 2> $out/p1.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD -d $out/p2 -o run -- python3 $GRAFT_REPO_ROOT/scripts/layer_times.py --reps 1 > /dev/null 2> $out/p2.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/p3 -o run -- python3 $GRAFT_REPO_ROOT/scripts/layer_times.py --reps 1 > /dev/null 2> $out/p3.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/p4 -o run -- python3 $GRAFT_REPO_ROOT/scripts/layer_times.py --reps 1 > /dev/null 2> $out/p4.err
ls -la $out/*
