#!/bin/bash
# round 3: paired weight-gradient blocks on the HECKTOR-shaped volume (same-box A/B) + layer times of the final build
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6k; mkdir -p $O; cd $R
b() { timeout -k 10 300 python bench.py --task hecktor21 --steps 48 --warmup 24 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 hecktor %.2f vol/s %.2f ms' % (d['value'], d['ms_per_step']))" | tee -a $O/out.txt; }
b pair && MMTTA_WGRAD_PAIR=0 b nopair && b pair && MMTTA_WGRAD_PAIR=0 b nopair
timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 > $O/layers_unet.txt 2>&1; head -5 $O/layers_unet.txt
timeout -k 10 400 python scripts/layer_times.py --tune-volumes 24 --model unet_multimodal_deepfusion > $O/layers_df.txt 2>&1; head -5 $O/layers_df.txt
timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 --task hecktor21 > $O/layers_hecktor.txt 2>&1; grep -E "wgrad_tr|conv time" $O/layers_hecktor.txt | head -12
MMTTA_WGRAD_PAIR=0 timeout -k 10 300 python scripts/layer_times.py --tune-volumes 24 --task hecktor21 > $O/layers_hecktor_nopair.txt 2>&1; grep -E "wgrad_tr|conv time" $O/layers_hecktor_nopair.txt | head -12
