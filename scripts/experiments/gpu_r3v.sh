#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3v; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_conv.py -x -q -k "upconvolution or bf16_operands or stored" > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers.txt 2>&1; head -2 $O/layers.txt; grep -E "upconv8|chan_mfma|direct_conv" $O/layers.txt | head -8
