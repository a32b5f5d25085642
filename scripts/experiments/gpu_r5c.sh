#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
timeout -k 10 600 python scripts/layer_times.py --task hecktor21 --tune-volumes 24 > $O/layers_hecktor.txt 2>&1; head -30 $O/layers_hecktor.txt
