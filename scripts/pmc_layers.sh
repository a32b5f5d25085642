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
# matrix-core utilisation (north_star: "MFMA utilisation shown via rocprof"): SQ_VALU_MFMA_BUSY_CYCLES counts cycles
# (= 32 x the number of v_mfma_f32_32x32x16_bf16 per SIMD), SQ_BUSY_CYCLES / SQ_WAVE_CYCLES give the denominator;
# its own pass, kernel trace only, the program directly after `--`
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -d $out/p5 -o run -- python3 $GRAFT_REPO_ROOT/scripts/layer_times.py --reps 1 > /dev/null 2> $out/p5.err || echo "MFMA counter pass failed (see p5.err)"
ls -la $out/*
