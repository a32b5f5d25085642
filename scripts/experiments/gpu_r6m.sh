#!/bin/bash
# round 3: final verification of HEAD: whole GPU suite, smoke, the driver's bench command
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6m; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['lanes_equal'], d['parity_full_size']['within_tolerance'], {k: round(v['value'],1) for k,v in d['variants'].items()})"
