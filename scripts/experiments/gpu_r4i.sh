#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4i; mkdir -p $O; cd $R
timeout -k 10 1000 python scripts/sweep_tuning.py --combos 3x8 --knobs v24 --volumes 48 --repeat 2 > $O/sweep.txt 2>&1; grep pass $O/sweep.txt
