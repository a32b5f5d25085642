#!/bin/bash
# round 3: whole GPU suite + smoke on the build with the paired weight-gradient blocks
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6i; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
