#!/bin/bash
# round 3: paired column blocks: twice the slabs (same workgroup count) against the same slabs (half the workgroups)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6h; mkdir -p $O; cd $R
b() { timeout -k 10 300 python bench.py --steps 48 --warmup 24 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 unet %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))" | tee -a $O/out.txt; }
b slabs2x && MMTTA_WGRAD_PAIR_S=1 b sameslabs && b slabs2x && MMTTA_WGRAD_PAIR_S=1 b sameslabs && MMTTA_WGRAD_PAIR=0 b nopair && b slabs2x && MMTTA_WGRAD_PAIR_S=1 b sameslabs && MMTTA_WGRAD_PAIR=0 b nopair
