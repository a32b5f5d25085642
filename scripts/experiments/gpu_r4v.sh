#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_tta.py tests/test_hip_unet.py tests/test_hip_groups.py tests/test_hip_deepfusion.py tests/test_hip_golden.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout -k 10 600 python -m pytest tests/test_hip_fullsize.py -x -q -s -k "config5" > $O/full.log 2>&1; tail -2 $O/full.log; grep fullsize $O/full.log
timeout -k 10 500 python bench.py --method tta_moddrop --steps 48 --warmup 8 --no-variants --no-profile-pass > $O/bench_moddrop_unet.json 2> $O/m1.err; python -c "
import json; d=json.load(open('$O/bench_moddrop_unet.json')); print('moddrop unet', round(d['value'],2))"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
