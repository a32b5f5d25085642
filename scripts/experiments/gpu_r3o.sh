#!/bin/bash
# round 3: where do the conv kernels' microseconds go: layer table with the norm-on-load / statistics dropped from the timed repeats
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; mkdir -p $O; cd $R
for w in "" no_norm_on_load no_stats no_norm_on_load,no_stats; do
  timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 --what-if "$w" > $O/layers_${w:-base}.txt 2>&1; head -1 $O/layers_${w:-base}.txt
done
