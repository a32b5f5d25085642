#!/bin/bash
# bimodal multi-lane throughput: is it the stream -> hardware-queue mapping?  (GPU_MAX_HW_QUEUES, default 4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2e; mkdir -p $O; cd $R
for q in default 8 16; do
  for rep in 1 2 3; do
    for lanes in 3 4; do
      if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
      v=$(python bench.py --steps 16 --warmup 4 --lanes $lanes --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])")
      echo "queues=$q rep=$rep lanes=$lanes -> $v vol/s" | tee -a $O/queues.txt
    done
  done
done
