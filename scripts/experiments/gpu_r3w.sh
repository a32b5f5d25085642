#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w; mkdir -p $O; cd $R
timeout -k 10 900 python scripts/sweep_tuning.py --combos 3x8 4x8 4x6 3x10 2x12 5x6 --knobs scaled --volumes 48 --repeat 2 > $O/sweep.txt 2>&1; tail -12 $O/sweep.txt
