#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 300 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "transposed_read" 2>&1 | tail -2
b() { env $1 python bench.py $2 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])"; }
for i in 1 2; do
  echo "deepfusion WGVEC=1: $(b MMTTA_WGVEC=1 '--model unet_multimodal_deepfusion --steps 4 --warmup 1')"
  echo "deepfusion WGVEC=3: $(b MMTTA_WGVEC=3 '--model unet_multimodal_deepfusion --steps 4 --warmup 1')"
  echo "unet storage fp32 WGVEC=1: $(b MMTTA_WGVEC=1 '--storage fp32 --steps 16 --warmup 4')"
  echo "unet storage fp32 WGVEC=3: $(b MMTTA_WGVEC=3 '--storage fp32 --steps 16 --warmup 4')"
  echo "unet default: $(b MMTTA_X=1 '--steps 24 --warmup 4')"
done
