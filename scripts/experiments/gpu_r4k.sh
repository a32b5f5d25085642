#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; mkdir -p $O; cd $R
timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers.txt 2>&1; head -3 $O/layers.txt
timeout -k 10 600 python scripts/layer_times.py --model unet_multimodal_deepfusion --tune-volumes 24 > $O/layers_df.txt 2>&1; head -3 $O/layers_df.txt
