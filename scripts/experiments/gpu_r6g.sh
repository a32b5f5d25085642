#!/bin/bash
# round 3: paired column blocks: stride 2 only (default) against stride 1 too; parity; bench
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6g; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_dispatch.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
MMTTA_WGRAD_PAIR=2 timeout -k 10 600 python -m pytest tests/test_hip_conv.py -x -q -k "wgrad or weight_grad or gradient" > $O/tests2.log 2>&1; tail -2 $O/tests2.log
timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers.txt 2>&1; grep -E "wgrad_tr_kernel|conv time" $O/layers.txt
MMTTA_WGRAD_PAIR=2 timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers_pair2.txt 2>&1; grep -E "wgrad_tr_kernel<4|conv time" $O/layers_pair2.txt
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 48 --warmup 24 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
done
MMTTA_WGRAD_PAIR=2 timeout -k 10 300 python bench.py --steps 48 --warmup 24 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pair2 unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))"
