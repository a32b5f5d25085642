#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4n; mkdir -p $O; cd $R
timeout -k 10 1000 python scripts/sweep_tuning.py --combos 3x8 --knobs v24w --volumes 48 --repeat 3 > $O/sweep.txt 2>&1; grep pass $O/sweep.txt | sed -e "s/'splitk_below': 16, 'splitk_target': 22, //" -e "s/, 'wgrad_thin_slabs': 43, 'cls_fused_min': 22//"
