#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2q; mkdir -p $O; cd $R
for st in bf16 fp32; do for v in 1 3; do
  MMTTA_STORAGE=$st MMTTA_WGVEC=$v python scripts/layer_times.py > $O/layers_${st}_$v.txt 2> $O/err_${st}_$v.txt
  echo "== storage $st wgvec $v: $(grep 'wgrad_' $O/layers_${st}_$v.txt | grep -v small | awk '{s+=$1*$2} END {print s}') us of wgrad per step; $(head -1 $O/layers_${st}_$v.txt)"
done; done
grep "wgrad" $O/layers_bf16_3.txt | sort -k1 -n -r | head -24
echo; grep "wgrad" $O/layers_bf16_1.txt | sort -k1 -n -r | head -24
