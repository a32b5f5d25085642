#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout -k 10 500 python bench.py --model unet_multimodal_deepfusion --steps 24 --warmup 4 --no-variants > $O/bench_deepfusion_brats.json 2> $O/df.err; python -c "
import json; d=json.load(open('$O/bench_deepfusion_brats.json')); print('deepfusion', round(d['value'],2), d['parity_full_size']['within_tolerance'], d['parity_full_size']['logits_err_over_max'])"
timeout -k 10 500 python bench.py --method tta_moddrop --model unet_multimodal_deepfusion --steps 24 --warmup 4 --no-variants > $O/bench_moddrop_deepfusion.json 2> $O/m2.err; python -c "
import json; d=json.load(open('$O/bench_moddrop_deepfusion.json')); print('moddrop deepfusion', round(d['value'],2))"
timeout -k 10 600 python scripts/layer_times.py --model unet_multimodal_deepfusion --tune-volumes 24 > $O/layers_df.txt 2>&1; head -2 $O/layers_df.txt | tail -1
