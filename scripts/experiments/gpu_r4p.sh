#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; mkdir -p $O; cd $R
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/dist1.json 2> $O/dist1.err; echo "rc=$?"; python -c "
import json; d=json.loads(open('$O/dist1.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['ranks'], d['lanes_equal'], d['config']['parallelism'][:80])"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
