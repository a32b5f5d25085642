#!/bin/bash
# round 3: parity classes interleaved over the XCDs: conv parity, bench lines, layer times
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_dispatch.py tests/test_hip_unet.py tests/test_hip_groups.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
timeout -k 10 300 python scripts/layer_times.py > $O/layers.txt 2>&1; head -3 $O/layers.txt; grep -E "convT 768|convT 256|s2 read 16x16x16|s2 read 8x8x8|dgrad conv 64->128|dgrad conv 128->256" $O/layers.txt
