#!/bin/bash
# round 3: canonical 27-tap stage of the lean 64^3 tile (immediate tap offsets, scalar-based weight requests): parity + layer times
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6d; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_dispatch.py tests/test_hip_unet.py tests/test_hip_groups.py tests/test_hip_tta.py tests/test_hip_golden.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers.txt 2>&1; head -12 $O/layers.txt
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 48 --warmup 24 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
