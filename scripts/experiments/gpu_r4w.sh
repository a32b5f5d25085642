#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4w; mkdir -p $O; cd $R
timeout -k 10 500 python bench.py --method tta_moddrop --steps 48 --warmup 8 --no-variants > $O/bench_moddrop_unet.json 2> $O/m1.err; python -c "
import json; d=json.load(open('$O/bench_moddrop_unet.json')); print('moddrop unet', round(d['value'],2))"
timeout -k 10 500 python bench.py --method tta_moddrop --model unet_multimodal_deepfusion --steps 24 --warmup 4 --no-variants > $O/bench_moddrop_deepfusion.json 2> $O/m2.err; python -c "
import json; d=json.load(open('$O/bench_moddrop_deepfusion.json')); print('moddrop deepfusion', round(d['value'],2))"
