#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4y; mkdir -p $O; cd $R
for w in 0 12; do for nv in 24 120; do
t0=$(date +%s.%N)
timeout -k 10 500 python main.py task=brats dataset=brats model=unet method=tta_entmin method.precision=bf16 dataset.synthetic.num_volumes=$nv training.num_workers=$w > $O/main_${w}_$nv.log 2> $O/main_${w}_$nv.err
t1=$(date +%s.%N); echo "workers $w num_volumes $nv wall $(python -c "print(round($t1 - $t0, 2))") s"; tail -2 $O/main_${w}_$nv.err | cut -c1-200
done; done
