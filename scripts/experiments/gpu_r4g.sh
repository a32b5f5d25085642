#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_conv.py tests/test_hip_pointwise.py -x -q -k "thin or entropy" > $O/tests.log 2>&1; tail -15 $O/tests.log
